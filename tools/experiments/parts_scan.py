"""Scratch: window-sweep parts per tile -- automatic choice against a scan, at several film sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
def t(fs, n=6):
    fs.window_filter(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fs.window_filter()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for W, H in ((1280, 720), (1920, 1080), (3840, 2160), (960, 540), (1920, 270), (3840, 270)):
    sc = synthetic.Scene(W, H, seed=1, device=dev)
    fs = film.FilmStats(W, H, dev)
    fs.accumulate(sc.samples(16, seed=2)); fs.prepass()
    api.force_filter_parts(0)
    auto = min(t(fs) for _ in range(2))
    res = []
    for k in (1, 2, 3, 4, 5, 6, 7, 8, 10, 14):
        api.force_filter_parts(k)
        res.append((min(t(fs) for _ in range(2)), k))
    api.force_filter_parts(0)
    auto = min(auto, min(t(fs) for _ in range(2)))   # again, now that the clocks have settled
    best = min(res)
    print("%dx%d: auto %.3f ms | best k=%d %.3f ms | " % (W, H, auto, best[1], best[0]) + " ".join("k%d:%.3f" % (k, v) for v, k in res))
    del fs, sc
