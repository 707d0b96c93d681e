"""Back-to-back timing of the window filter at several radii (pair-symmetric runtime-radius build against the one-sided
runtime-radius kernel) and with eight feature channels.  python tools/experiments/time_radii.py [W H]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
dev = torch.device("cuda:0")
api.setup(0)
scene = synthetic.Scene(W, H, seed=1, device=dev)
fs = film.FilmStats(W, H, dev, types=synthetic.FEATURES)
fs.accumulate(scene.samples(32, seed=2))
fs.prepass()
torch.cuda.synchronize()
rad = fs.state["radiance"]


def run(radius, sd, force, gnames=("normal", "albedo"), sds=(0.1, 0.02), parts=0, reps=20):
    out = torch.zeros_like(fs.film_f)
    a, keep = api.make_filter_args(n=[rad["n"]], mean=[rad["mean"]], m2=[rad["m2"]], m3=[rad["m3"]], film=[rad["film_mean"]],
                                   mean_corr=[fs.mean_corr], disc=[fs.disc], film_filtered=[out],
                                   g_buffers=[fs.g_buffer(g) for g in gnames], g_sds=list(sds), filter_sd=sd, radius=radius)
    api.force_filter_variant(force)
    api.force_filter_parts(parts)
    try:
        for _ in range(3):
            api.window_filter(a, 3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            api.window_filter(a, 3)
        e1.record()
        torch.cuda.synchronize()
        v, p = api.last_filter_variant(), api.load().statmc_debug_last_filter_parts()
    finally:
        api.force_filter_variant(0)
        api.force_filter_parts(0)
    return e0.elapsed_time(e1) / reps, v, p, out


for radius, sd in ((6, 3.0), (3, 2.0), (10, 5.0), (13, 6.0), (16, 8.0), (19, 9.5), (20, 10.0)):
    base = None
    for force, parts in ((0, 0), (2, 0), (0, 1), (0, 2), (0, 3)):
        ms, v, p, out = run(radius, sd, force, parts=parts)
        if base is None:
            base = out
        err = float(((out - base).double().pow(2).sum() / base.double().pow(2).sum()).sqrt())
        print("%dx%d r=%2d %-10s parts %d : %.3f ms  (%.0f Mpx/s)  rel L2 vs first %.1e" % (W, H, radius, v, p, ms, W * H / ms / 1e3, err), flush=True)
for gn, sds in ((("normal", "albedo", "depth", "materialid"), (0.1, 0.02, 1.0, 0.5)), (("normal", "albedo", "depth"), (0.1, 0.02, 1.0)),
                (("normal", "albedo"), (0.1, 0.02))):
    for force in (0, 3):
        if force == 3 and len(gn) == 4:
            continue
        ms, v, p, out = run(20, 10.0, force, gn, sds)
        print("%dx%d r=20 %-12s %-34s: %.3f ms" % (W, H, v, "+".join(gn), ms), flush=True)
