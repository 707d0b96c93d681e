"""Welch degrees of freedom on the pair-symmetric kernel at 1080p (round 4): against the default spec on the same film, with
LDS-DMA staging (width 1920) and with register staging (width 1922: what the Welch build always uses).
python tools/experiments/time_welch.py [variant.so]     (a variant library: the timing-only ablations STATMC_SYM_WELCH_ABLATE;
                                                          QUICK=1: the first shape only -- what tools/profile_variants.sh counts)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from statmc_amd import build
if len(sys.argv) > 1:
    os.environ.setdefault("STATMC_ALLOW_DIAGNOSTIC_BUILD", "1")
    build.SO = os.path.abspath(sys.argv[1])
import torch
from statmc_amd import api, film, synthetic

dev = torch.device("cuda:0")
api.setup(0)
for W, spp in (((1920, 32),) if os.environ.get("QUICK") else ((1920, 32), (1922, 32), (1922, 256), (1922, 8))):
    H = 1080
    scene = synthetic.Scene(W, H, seed=1, device=dev)
    for r, sd in ((20, 10.0), (6, 3.0)):
        fs = film.FilmStats(W, H, dev, filter_sd=sd, radius=r)
        for b in range(max(spp // 32, 1)):
            fs.accumulate(scene.samples(min(spp, 32), seed=2 + b, features=("radiance", "normal", "albedo")))
        for kw in (dict(), dict(dof=1), dict(dof=1, channel_rule=1)):
            api.set_filter_spec(**kw)
            fs.prepass()
            a, keep = fs.filter_args()
            for _ in range(2):
                api.window_filter(a, 3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                api.window_filter(a, 3)
            e1.record()
            torch.cuda.synchronize()
            print("%dx%d %3d spp r = %2d %-32s %-18s %.3f ms" % (W, H, spp, r, kw or "default", api.last_filter_variant(), e0.elapsed_time(e1) / 10), flush=True)
        api.set_filter_spec()
        del fs
