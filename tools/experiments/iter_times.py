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
for bands, env in (("6", {}), ("6", {"HSA_ENABLE_INTERRUPT": "0"}), ("6", {"ROC_ACTIVE_WAIT_TIMEOUT": "20000"}), ("6", {"GPU_MAX_HW_QUEUES": "8"}),
                   ("6", {"HIP_FORCE_DEV_KERNARG": "1"}), ("6", {"ROC_SIGNAL_POOL_SIZE": "256"}), ("6", {"HSA_ENABLE_SDMA": "0"})):
    for p in range(6):
        out = subprocess.run([build.DENOISE_BIN, "--stem", stem, "--spp", ",".join([str(spp)] * 16), "--filtersd", "10", "--filterradius", "20",
                              "--warmup", "--bands", bands, "--output", "film-f"], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        ns = [int(v) / 1e6 for v in re.findall(r"HIP time \[ns\]: (\d+)", out.stdout)]
        ph = [tuple(int(v) / 1e6 for v in m) for m in re.findall(r"host phases \[ns\]: upload (\d+) denoise (\d+) download (\d+) synchronize (\d+)", out.stdout)]
        print("bands %s %s: HIP %s" % (bands, env, " ".join("%.1f" % t for t in ns)), flush=True)
        for t, p4 in list(zip(ns, ph))[1:]:
            if t > 5.0:
                print("      slow iteration %.1f ms: host enqueue upload %.2f denoise %.2f download %.2f, synchronize %.2f" % ((t,) + p4), flush=True)
        fast = [p4 for t, p4 in list(zip(ns, ph))[1:] if t <= 5.0]
        if fast:
            print("      fast iterations (median): host enqueue upload %.2f denoise %.2f download %.2f, synchronize %.2f" % tuple(sorted(f[i] for f in fast)[len(fast) // 2] for i in range(4)), flush=True)
shutil.rmtree(d, ignore_errors=True)
