"""Round 5 (VERDICT r4 "What's weak" 3): what CPU share does a GPU box give this process, and how do the oracle's OpenMP legs scale
on it?  Touches no GPU.  Prints the cgroup CPU quota, the affinity mask, the load, and seconds per Mpixel of the two legs at
1 .. N threads (accumulate: 128 rows x 1920 px x 64 spp x 11 ch, film-major planes; filter: 1920 px x `threads` rows at least).
python tools/experiments/cpu_probe.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle


def read(p):
    try:
        return open(p).read().strip()
    except OSError as e:
        return "<%s>" % e.__class__.__name__


print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "omp max", oracle.num_threads())
print("cgroup cpu.max:", read("/sys/fs/cgroup/cpu.max"), "| v1 quota:", read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"),
      read("/sys/fs/cgroup/cpu/cpu.cfs_period_us"))
print("cpu.stat:", read("/sys/fs/cgroup/cpu.stat").replace("\n", " "))
print("cpuset:", read("/sys/fs/cgroup/cpuset.cpus.effective"))
print("loadavg:", read("/proc/loadavg"), "| OMP env:", {k: v for k, v in os.environ.items() if k.startswith(("OMP_", "GOMP_"))})
W, rows, S = 1920, 128, 64
rng = np.random.default_rng(1)
types = (("radiance", 3, True, 3), ("normal", 3, False, 1), ("albedo", 3, False, 1), ("depth", 1, False, 1), ("materialid", 1, False, 1))
smp = {t: rng.random((S, rows, W, c), dtype=np.float32) for t, c, _, _ in types}
H = 256
mc = rng.random((H, W, 3), dtype=np.float32)
dc = (rng.random((H, W, 3), dtype=np.float32) * 0.05).astype(np.float32)
col = rng.random((H, W, 3), dtype=np.float32)
gb = [rng.random((H, W, 3), dtype=np.float32) for _ in range(2)]
nmax = len(os.sched_getaffinity(0))
for th in [t for t in (1, 2, 4, 8, 16, 32, 64, 128, 256) if t <= nmax]:
    t0 = time.perf_counter()
    for t, c, tr, mm in types:
        oracle.accumulate(oracle.new_state(rows, W, c), smp[t], tr, mm, threads=th)
    ta = time.perf_counter() - t0
    fr = max(2, min(H - 40, th))
    t0 = time.perf_counter()
    oracle.filter_image(mc, dc, col, gb, [-50.0, -1250.0], -0.005, 20, roi=(0, 20, W, 20 + fr), threads=th)
    tf = time.perf_counter() - t0
    print("threads %3d: accumulate %.3f s (%.3f s/Mpx at 64 spp)   filter %d rows %.3f s (%.3f s/Mpx)   throttled: %s"
          % (th, ta, ta / (rows * W) * 1e6, fr, tf, tf / (fr * W) * 1e6, read("/sys/fs/cgroup/cpu.stat").split("nr_throttled")[-1].split()[0:1]), flush=True)
