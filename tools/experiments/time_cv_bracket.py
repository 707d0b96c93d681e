"""The reference's call order through include/statmc_cv.hpp (tools/bin/test_cv_adaptor) on a 1080p dump: the bracket
Upload x 7 + filter<float3> + download + synchronize, one stream against the adaptor's band pipeline.
python tools/experiments/time_cv_bracket.py"""
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
build.build_tools()
scene = synthetic.Scene(W, H, seed=1, device=dev)
fs = film.FilmStats(W, H, dev)
fs.accumulate(scene.samples(spp, seed=2, features=("radiance", "normal", "albedo")))
torch.cuda.synchronize()
rad = fs.state["radiance"]
d = tempfile.mkdtemp(prefix="statmc_cv_", dir="/dev/shm")
try:
    stem = os.path.join(d, "scene")
    for name, img in {"film": rad["film_mean"], "t0-b0-n": rad["n"], "t0-b0-mean": rad["mean"], "t0-b0-m2": rad["m2"], "t0-b0-m3": rad["m3"],
                      "t1-b0-film-mean": fs.g_buffer("normal"), "t2-b0-film-mean": fs.g_buffer("albedo")}.items():
        pfm.write_pfm("%s-%d-%s.pfm" % (stem, spp, name), img.cpu().numpy())
    del fs, scene
    torch.cuda.empty_cache()
    queues = sys.argv[1:] or ["2"]          # numbers of upload queues to try
    for q in queues:
        for bands in ("1", "6", "6", "6", "6", "6", "6", "6", "6", "6", "6", "6", "6"):
            env = dict(os.environ, STATMC_CV_BANDS=bands, STATMC_CV_UPLOAD_QUEUES=q)
            out = subprocess.run([build.CV_ADAPTOR_BIN, stem, str(spp), os.path.join(d, "f.pfm")], capture_output=True, text=True,
                                 env=env, timeout=300)
            assert out.returncode == 0, out.stderr
            m = re.search(r"bracket_ns (\d+) bands (\d+)", out.stdout)
            print("upload queues %s STATMC_CV_BANDS=%s: %.3f ms (bands used %s)" % (q, bands, int(m.group(1)) / 1e6, m.group(2)), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
