"""Window-filter time at 1080p with all four feature types as G-buffers (eight feature planes) over the radius: the
pair-symmetric kernel's eight-plane build, compile-time radius 20 and runtime radius below (round 4).
python tools/experiments/time_g8_radii.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

W, H = 1920, 1080
dev = torch.device("cuda:0")
api.setup(0)
scene = synthetic.Scene(W, H, seed=1, device=dev)
for names in (("normal", "albedo"), ("materialid", "depth", "normal", "albedo")):
    for r, sd in ((3, 2.0), (6, 3.0), (10, 5.0), (13, 6.0), (20, 10.0)):
        fs = film.FilmStats(W, H, dev, types=synthetic.FEATURES, filter_sd=sd, radius=r, g_buffers=names)
        fs.accumulate(scene.samples(16, seed=2))
        fs.prepass()
        for _ in range(3):
            fs.window_filter()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fs.window_filter()
        e1.record()
        torch.cuda.synchronize()
        print("%d feature planes  r = %2d  %-12s %.3f ms" % (8 if len(names) == 4 else 6, r, api.last_filter_variant(), e0.elapsed_time(e1) / 20), flush=True)
        del fs
