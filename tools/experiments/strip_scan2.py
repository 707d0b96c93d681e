"""How long does ONE round of the pair-symmetric filter take by itself?  1920-wide strips of k tile rows at parts = 1.  The launch
holds 15 (k + 3) items: the ROI's tiles and the three tile rows above it whose window rows reach into it; up to 256 items every
workgroup has a CU of its own (tools/microbench/wg_capacity.hip: the chip holds exactly 256 workgroups of this footprint).  python tools/experiments/strip_scan2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
W = 1920


def t(fs, roi, n=20):
    fs.window_filter(roi=roi); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fs.window_filter(roi=roi)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


H = 8 * 40 + 40
sc = synthetic.Scene(W, H, seed=1, device=dev)
fs = film.FilmStats(W, H, dev)
fs.accumulate(sc.samples(8, seed=2)); fs.prepass()
for k in (1, 2, 4, 8, 12, 16, 17, 18, 20, 24, 32, 34, 35, 40):
    roi = (0, 24, W, 24 + 8 * k)
    api.force_filter_parts(1)
    ms1 = t(fs, roi)
    api.force_filter_parts(0)
    ms0 = t(fs, roi)
    print("%2d tile rows = %3d tiles, %3d items: parts 1 %.3f ms   auto %.3f ms (%d parts)" % (k, 15 * k, 15 * (k + 3), ms1, ms0, api.load().statmc_debug_last_filter_parts()), flush=True)
api.force_filter_parts(0)
