"""Scratch: does a low-register HBM-bound kernel (torch elementwise) run beside the LDS window filter?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from statmc_amd import build
if len(sys.argv) > 1: os.environ.setdefault("STATMC_ALLOW_DIAGNOSTIC_BUILD", "1"); build.SO = os.path.abspath(sys.argv[1])
import torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
W, H, S = 1920, 1080, 8
sc = synthetic.Scene(W, H, seed=1, device=dev)
smp = sc.samples(S, seed=3, features=synthetic.FEATURES)
fs = film.FilmStats(W, H, dev, types=synthetic.FEATURES)
fs.accumulate(smp); fs.prepass()
big = torch.ones(1 << 30, dtype=torch.float32, device=dev)  # 4 GiB: mul_ moves 8 GiB
def run_filter(): fs.window_filter()
def run_stream(): big.mul_(1.0001)
def wall(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t_f, t_s = wall(run_filter), wall(run_stream)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both(first):
    def f():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        order = [(s2, run_filter), (s1, run_stream)] if first == "filter" else [(s1, run_stream), (s2, run_filter)]
        for st, fn in order:
            with torch.cuda.stream(st): fn()
        cur.wait_stream(s1); cur.wait_stream(s2)
    return f
print("filter %.3f ms | torch mul_ 4 GiB %.3f ms (%.0f GB/s) | sum %.3f | concurrent filter-first %.3f, stream-first %.3f"
      % (t_f, t_s, 8 * 2**30 / t_s / 1e6, t_f + t_s, wall(both("filter")), wall(both("stream"))))
