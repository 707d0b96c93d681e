"""filter<float> with 5 (ACRR) and 12 (SMIS) 1-channel buffers at 1080p, r = 20: pair-symmetric kernel (two buffers per
launch) against the one-sided LDS kernel (three per launch).  python tools/experiments/time_float.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

W, H = 1920, 1080
RADIUS = int(sys.argv[1]) if len(sys.argv) > 1 else 20      # (STATMC_FLOAT_MIX=0: odd counts stay on the pair-symmetric kernel)
dev = torch.device("cuda:0")
api.setup(0)
scene = synthetic.Scene(W, H, seed=1, device=dev)
fs = film.FilmStats(W, H, dev)
fs.accumulate(scene.samples(32, seed=2, features=("radiance", "normal", "albedo")))
fs.prepass()
torch.cuda.synchronize()
gbs = [fs.g_buffer("normal"), fs.g_buffer("albedo")]
for nb in (1, 2, 3, 5, 7, 12):
    mc = [(fs.mean_corr[..., b % 3:b % 3 + 1] * (1.0 / (1 + b))).contiguous() for b in range(nb)]
    dc = [(fs.disc[..., b % 3:b % 3 + 1] * (1.0 / (1 + b)) ** 2).contiguous() for b in range(nb)]
    col = [(fs.state["radiance"]["film_mean"][..., b % 3:b % 3 + 1] * (1.0 / (1 + b))).contiguous() for b in range(nb)]
    out = [torch.zeros(H, W, 1, device=dev) for _ in range(nb)]
    a, keep = api.make_filter_args(n=[], mean=[], m2=[], m3=[], film=col, mean_corr=mc, disc=dc, film_filtered=out,
                                   g_buffers=gbs, g_sds=[0.1, 0.02], filter_sd=10.0, radius=RADIUS)
    res = {}
    for force in (0, 3 if RADIUS == 20 else 2):
        api.force_filter_variant(force)
        for _ in range(2):
            api.window_filter(a, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            api.window_filter(a, 1)
        e1.record()
        torch.cuda.synchronize()
        res[force] = (api.last_filter_variant(), e0.elapsed_time(e1) / 10, [o.clone() for o in out])
        api.force_filter_variant(0)
    f2 = 3 if RADIUS == 20 else 2
    err = max(float(((x - y).double().pow(2).sum() / y.double().pow(2).sum()).sqrt()) for x, y in zip(res[0][2], res[f2][2]))
    print("r = %2d, %2d buffers: %-20s %.3f ms   %-10s %.3f ms   max rel L2 between them %.2e" % (RADIUS, nb, res[0][0], res[0][1], res[f2][0], res[f2][1], err), flush=True)
