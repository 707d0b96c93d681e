"""Scratch: how long does one round of regular tiles take against one round of DUAL tiles? (ROI launches
of a single tile column: 135 regular workgroups or 72 DUAL ones, each on its own CU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
W, H = 1920, 1080
sc = synthetic.Scene(W, H, seed=1, device=dev)
fs = film.FilmStats(W, H, dev)
fs.accumulate(sc.samples(16, seed=2)); fs.prepass()
def t(roi, n=10):
    fs.window_filter(roi=roi); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fs.window_filter(roi=roi)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
api.force_filter_parts(1)
for name, roi in (("regular column (135 WGs)", (256, 0, 512, H)), ("DUAL column (72 WGs)", (1792, 0, 1920, H)),
                  ("regular column at the left edge", (0, 0, 256, H)), ("two regular columns (270 WGs = 2 rounds)", (256, 0, 768, H)),
                  ("whole film", None)):
    print("%-44s %.3f ms" % (name, min(t(roi) for _ in range(3))))
api.force_filter_parts(0)
