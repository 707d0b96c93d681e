"""Where does the pair-symmetric filter's launch go from one tile's time to two?  ROIs of c x k tiles (128 c columns, 8 k rows) of a
1920 x 360 film at parts = 1; the launch holds min(15, c + 1) x (k + 3) items (the tile column right of the ROI and the three tile
rows above it take part as the other end of pairs): two rounds from 257 items.  python tools/experiments/strip_scan3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
W, H = 1920, 360


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


sc = synthetic.Scene(W, H, seed=1, device=dev)
fs = film.FilmStats(W, H, dev)
fs.accumulate(sc.samples(8, seed=2)); fs.prepass()
api.force_filter_parts(1)
for c, k in ((15, 12), (13, 14), (12, 16), (13, 15), (14, 14), (11, 18), (10, 20), (15, 14), (13, 16), (14, 15), (12, 18), (15, 15), (14, 16), (15, 16)):
    roi = (0, 24, 128 * c, 24 + 8 * k)
    print("%2d x %2d = %3d tiles, %3d items: %.3f ms" % (c, k, c * k, min(15, c + 1) * (k + 3), t(fs, roi)), flush=True)
api.force_filter_parts(0)
