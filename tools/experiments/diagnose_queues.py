"""Why does the adaptor's two-queue upload pipeline run at ~4.0 ms in most processes and at 5.8 - 7 ms in some?
(include/statmc_cv.hpp, STATMC_CV_UPLOAD_QUEUES=2)  Hypothesis: the HIP runtime multiplexes streams
over GPU_MAX_HW_QUEUES (default 4) hardware queues; the adaptor's streams (cv::cuda::Stream + up + up2 + down, + the null
stream) are one too many, and when a copy stream shares a hardware queue with the kernel stream its event-wait barrier
packets hold the other stream's packets back (an AQL queue is processed in order).
Measures N processes per configuration, then traces a few with rocprofv3 and keeps a slow and a fast trace.
python tools/experiments/diagnose_queues.py [n_processes]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, build, film, pfm, synthetic

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
W, H, spp = 1920, 1080, 32
dev = torch.device("cuda:0")
api.setup(0)
build.build_tools()
scene = synthetic.Scene(W, H, seed=1, device=dev)
fs = film.FilmStats(W, H, dev)
fs.accumulate(scene.samples(spp, seed=2, features=("radiance", "normal", "albedo")))
torch.cuda.synchronize()
rad = fs.state["radiance"]
d = tempfile.mkdtemp(prefix="statmc_q_", dir="/dev/shm")
out_dir = os.path.join("gpurun_out", "queues")
os.makedirs(out_dir, exist_ok=True)
try:
    stem = os.path.join(d, "scene")
    for name, img in {"film": rad["film_mean"], "t0-b0-n": rad["n"], "t0-b0-mean": rad["mean"], "t0-b0-m2": rad["m2"], "t0-b0-m3": rad["m3"],
                      "t1-b0-film-mean": fs.g_buffer("normal"), "t2-b0-film-mean": fs.g_buffer("albedo")}.items():
        pfm.write_pfm("%s-%d-%s.pfm" % (stem, spp, name), img.cpu().numpy())
    del fs, scene
    torch.cuda.empty_cache()

    def run(env_extra, prefix=()):
        env = dict(os.environ, **dict({"STATMC_CV_BANDS": "6"}, **env_extra))
        out = subprocess.run(list(prefix) + [build.CV_ADAPTOR_BIN, stem, str(spp), os.path.join(d, "f.pfm")], capture_output=True,
                             text=True, env=env, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        return int(re.search(r"bracket_ns (\d+) bands (\d+)", out.stdout).group(1)) / 1e6, out

    def show(label, env):
        rows = []
        for _ in range(N):
            t, out = run(dict(env, STATMC_CV_DIAG="1"))
            m = re.search(r"diag upload_1q_ns (\d+) upload_2q_ns (\d+) download_ns (\d+) upload_2q_plus_download_ns (\d+)", out.stdout)
            rows.append((t,) + tuple(int(v) / 1e6 for v in m.groups()))
        print("%-72s: %s  (%d of %d above 5 ms)" % (label, " ".join("%.2f" % r[0] for r in rows), sum(r[0] > 5.0 for r in rows), N), flush=True)
        slow = [r for r in rows if r[0] > 5.0]
        fast = [r for r in rows if r[0] <= 5.0]
        for name, grp in (("fast", fast), ("slow", slow)):
            if grp:
                med = lambda i: sorted(g[i] for g in grp)[len(grp) // 2]
                print("      %s processes (median): bracket %.2f | raw copies: upload 1 queue %.2f, 2 queues %.2f, download %.2f, 2 queues + download %.2f ms"
                      % (name, med(0), med(1), med(2), med(3), med(4)), flush=True)

    show("adaptor, 2 upload queues, 6 bands", {"STATMC_CV_UPLOAD_QUEUES": "2"})
    show("adaptor, 2 upload queues, 3 bands", {"STATMC_CV_UPLOAD_QUEUES": "2", "STATMC_CV_BANDS": "3"})
    show("adaptor, 2 upload queues, 4 bands", {"STATMC_CV_UPLOAD_QUEUES": "2", "STATMC_CV_BANDS": "4"})
    # the C++ Estimator (tools/bin/statmc_denoise): 8 timed iterations per process after a warm-up
    for bands in ("6", "3", "4"):
        firsts, worsts = [], []
        for _ in range(N):
            out = subprocess.run([build.DENOISE_BIN, "--stem", stem, "--spp", ",".join([str(spp)] * 8), "--filtersd", "10", "--filterradius", "20",
                                  "--warmup", "--bands", bands, "--output", "film-f"], capture_output=True, text=True, timeout=300)
            assert out.returncode == 0, out.stderr[-500:]
            ns = [int(v) / 1e6 for v in re.findall(r"HIP time \[ns\]: (\d+)", out.stdout)][1:]
            firsts.append(min(ns))
            worsts.append(max(ns))
        print("Estimator --bands %s: best of 8 per process: %s | worst of 8: %s" % (bands, " ".join("%.2f" % t for t in firsts), " ".join("%.2f" % t for t in worsts)), flush=True)
    # runtime log of one process: which hardware queue does each stream get?
    _, out = run({"STATMC_CV_UPLOAD_QUEUES": "2", "AMD_LOG_LEVEL": "4", "AMD_LOG_MASK": "0x8000"})
    lines = [l for l in out.stderr.splitlines() if re.search(r"[Qq]ueue", l)]
    open(os.path.join(out_dir, "amd_log_queues.txt"), "w").write("\n".join(lines[:400]))
    print("runtime log: %d lines mentioning queues; first 12:" % len(lines))
    for l in lines[:12]:
        print("   ", l[:200])
    # traces: the program directly after `--` (no env / shell hop under the profiler)
    os.environ["STATMC_CV_UPLOAD_QUEUES"] = "2"
    os.environ["STATMC_CV_BANDS"] = "6"
    kept = {}
    for i in range(0):
        td = os.path.join(d, "trace%d" % i)
        t, out = run({}, prefix=("rocprofv3", "--hip-trace", "--memory-copy-trace", "--kernel-trace", "-d", td, "-o", "t", "--"))
        kind = "slow" if t > 5.0 else "fast"
        print("traced process %d: %.2f ms (%s)" % (i, t, kind), flush=True)
        if kind not in kept:
            kept[kind] = t
            for root, _, files in os.walk(td):
                for f in files:
                    if f.endswith(".csv"):
                        shutil.copy(os.path.join(root, f), os.path.join(out_dir, "%s_%s" % (kind, f)))
        if len(kept) == 2:
            break
    print("kept traces:", kept)
finally:
    shutil.rmtree(d, ignore_errors=True)
