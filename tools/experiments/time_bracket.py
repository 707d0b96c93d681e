"""The reference's "CUDA time" bracket (Upload + Denoise + Download + Synchronize) through tools/bin/statmc_denoise on a
1080p dump, one stream against the band pipeline.  python tools/experiments/time_bracket.py"""
import os
import re
import shutil
import subprocess
import sys
import tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, build, film, pfm, synthetic

W, H, spp = 1920, 1080, 32
dev = torch.device("cuda:0")
api.setup(0)
exe = build.build_tools()
scene = synthetic.Scene(W, H, seed=1, device=dev)
fs = film.FilmStats(W, H, dev)
fs.accumulate(scene.samples(spp, seed=2, features=("radiance", "normal", "albedo")))
torch.cuda.synchronize()
rad = fs.state["radiance"]
d = tempfile.mkdtemp(prefix="statmc_bracket_", dir="/dev/shm")
try:
    stem = os.path.join(d, "scene")
    dump = {"film": rad["film_mean"], "t0-b0-n": rad["n"], "t0-b0-mean": rad["mean"], "t0-b0-m2": rad["m2"],
            "t0-b0-m3": rad["m3"], "t1-b0-film-mean": fs.g_buffer("normal"), "t2-b0-film-mean": fs.g_buffer("albedo")}
    for name, img in dump.items():
        pfm.write_pfm("%s-%d-%s.pfm" % (stem, spp, name), img.cpu().numpy())
    config = sys.argv[1] if len(sys.argv) > 1 else "denoise"
    if config == "acrr":   # five luminance buffers: 5 x (n, mean = film-mean, m2, m3) up, 5 x film-mean-f down
        for b in range(5):
            k = 1.0 / (1 + b)
            lum = (rad["film_mean"].mean(dim=2, keepdim=True) * k).contiguous()
            for name, img in (("n", rad["n"]), ("film-mean", lum), ("mean", lum), ("m2", (rad["m2"].mean(dim=2, keepdim=True) * k * k).contiguous()),
                              ("m3", (rad["m3"].mean(dim=2, keepdim=True) * k ** 3).contiguous())):
                pfm.write_pfm("%s-%d-t0-b%d-%s.pfm" % (stem, spp, b, name), img.cpu().numpy())
    del fs, scene
    torch.cuda.empty_cache()
    for bands in (1, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6):
        out = subprocess.run([exe, "--stem", stem, "--spp", ",".join([str(spp)] * 6), "--warmup", "--bands", str(bands), "--config", config,
                              "--output", "film-f" if config == "denoise" else "t0-b0-film-mean-f"],
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        ns = sorted(int(v) for v in re.findall(r"HIP time \[ns\]: (\d+)", out.stdout)[1:])
        used = re.search(r"pipeline bands: (\d+)", out.stdout).group(1)
        print("bands %d (%s used): best %.3f ms, median %.3f ms" % (bands, used, ns[0] / 1e6, ns[len(ns) // 2] / 1e6), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
