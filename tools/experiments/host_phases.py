import os, re, subprocess, sys, tempfile, shutil
sys.path.insert(0, os.getcwd())
import torch
from statmc_amd import api, build, film, pfm, synthetic
W, H, spp = 1920, 1080, 32
dev = torch.device("cuda:0"); api.setup(0); build.build_tools()
scene = synthetic.Scene(W, H, seed=1, device=dev)
fs = film.FilmStats(W, H, dev)
fs.accumulate(scene.samples(spp, seed=2, features=("radiance", "normal", "albedo"))); torch.cuda.synchronize()
rad = fs.state["radiance"]
d = tempfile.mkdtemp(prefix="statmc_q_", dir="/dev/shm")
stem = os.path.join(d, "scene")
for name, img in {"film": rad["film_mean"], "t0-b0-n": rad["n"], "t0-b0-mean": rad["mean"], "t0-b0-m2": rad["m2"], "t0-b0-m3": rad["m3"],
                  "t1-b0-film-mean": fs.g_buffer("normal"), "t2-b0-film-mean": fs.g_buffer("albedo")}.items():
    pfm.write_pfm("%s-%d-%s.pfm" % (stem, spp, name), img.cpu().numpy())
del fs, scene; torch.cuda.empty_cache()
for mode in ("1", "2", "3"):
    out = subprocess.run([build.DENOISE_BIN, "--stem", stem, "--spp", ",".join([str(spp)] * 8), "--filtersd", "10", "--filterradius", "20",
                          "--warmup", "--bands", "6", "--output", "film-f"], capture_output=True, text=True, timeout=300, env=dict(os.environ, STATMC_UPLOAD_QUEUES=mode))
    ns = [int(v) / 1e6 for v in re.findall(r"HIP time \[ns\]: (\d+)", out.stdout)]
    ph = [tuple(int(v) / 1e6 for v in m) for m in re.findall(r"host phases \[ns\]: upload (\d+) denoise (\d+) download (\d+) synchronize (\d+)", out.stdout)]
    for t, p4 in list(zip(ns, ph))[1:]:
        print("mode %s: %.2f ms: host enqueue upload %.3f denoise %.3f download %.3f, synchronize %.3f" % ((mode, t) + p4), flush=True)
shutil.rmtree(d, ignore_errors=True)
