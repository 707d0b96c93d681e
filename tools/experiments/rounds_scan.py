"""Window-filter time against the number of 128 x 8 tiles of a 1280-wide film at parts = 1 (ten tile columns: the tile count is
10 x H / 8), to read off what a round of 256 workgroups costs and what is fixed: 1 .. 5 rounds, the 720p film among them.
python tools/experiments/rounds_scan.py [width]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
cols = (W + 127) // 128


def t(fs, n=10):
    fs.window_filter(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fs.window_filter()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


for H in (96, 200, 208, 304, 408, 416, 512, 608, 616, 720, 816, 824, 1016, 1024, 1080):
    sc = synthetic.Scene(W, H, seed=1, device=dev)
    fs = film.FilmStats(W, H, dev)
    fs.accumulate(sc.samples(8, seed=2)); fs.prepass()
    line = "%dx%d: %4d tiles = %.2f rounds:" % (W, H, cols * ((H + 7) // 8), cols * ((H + 7) // 8) / 256.0)
    for k in (1, 0):
        api.force_filter_parts(k)
        line += "  %s %.3f ms" % ("parts 1" if k else "auto   ", t(fs))
    api.force_filter_parts(0)
    print(line, flush=True)
    del fs, sc
