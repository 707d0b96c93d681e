"""The reference's "CUDA time" bracket (tools/bin/statmc_denoise, 1080p, 6 bands) under the three transports of the copies
in: one copy-engine queue (1), two copy-engine queues (2), one copy-engine queue + a pulling kernel (3) -- several
processes of 16 iterations each; per mode every iteration's time, so that tails show.
python tools/experiments/upload_modes.py [processes]"""
import os, re, subprocess, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, build, film, pfm, synthetic
W, H, spp = 1920, 1080, 32
procs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
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
try:
    for mode in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("1", "2", "3")):
        allt = []
        for p in range(procs):
            out = subprocess.run([build.DENOISE_BIN, "--stem", stem, "--spp", ",".join([str(spp)] * 17), "--filtersd", "10", "--filterradius", "20",
                                  "--warmup", "--bands", os.environ.get("BANDS", "6"), "--output", "film-f"], capture_output=True, text=True, timeout=300,
                                 env=dict(os.environ, STATMC_UPLOAD_QUEUES=mode))
            assert out.returncode == 0, out.stderr
            ns = [int(v) / 1e6 for v in re.findall(r"HIP time \[ns\]: (\d+)", out.stdout)][1:]
            allt += ns
            print("mode %s process %d: %s" % (mode, p, " ".join("%.2f" % t for t in ns)), flush=True)
        allt.sort()
        print("mode %s: %d iterations, best %.3f median %.3f mean %.3f worst %.3f ms; over 4.5 ms: %d" %
              (mode, len(allt), allt[0], allt[len(allt) // 2], sum(allt) / len(allt), allt[-1], sum(t > 4.5 for t in allt)), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
