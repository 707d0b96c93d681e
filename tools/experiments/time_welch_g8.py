"""Welch degrees of freedom x 1-channel G-buffers at 1080p: the eight-plane Welch builds of the pair-symmetric kernel (one
plane n - 1 per staged row, E = v * v / (n - 1) formed per tap) against the six-plane Welch build (no 1-channel feature), the
eight-plane build without Welch, and the general kernel that served this product before.  RGB buffer and two float buffers.
python tools/experiments/time_welch_g8.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

W, H = 1920, 1080
dev = torch.device("cuda:0")
api.setup(0)
scene = synthetic.Scene(W, H, seed=1, device=dev)
types = ("radiance", "normal", "albedo", "depth", "materialid")
fs = film.FilmStats(W, H, dev, types=types)
fs.accumulate(scene.samples(32, seed=2, features=types))
n = fs.state["radiance"]["n"]
g6 = [fs.g_buffer("normal"), fs.g_buffer("albedo")]
g8 = g6 + [fs.g_buffer("depth"), fs.g_buffer("materialid")]


def timed(a, channels, force, reps):
    api.force_filter_variant(force)
    try:
        api.window_filter(a, channels)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            api.window_filter(a, channels)
        e1.record()
        torch.cuda.synchronize()
        return api.last_filter_variant(), e0.elapsed_time(e1) / reps
    finally:
        api.force_filter_variant(0)


for radius in (20, 6):
    for dof in (0, 1):
        api.set_filter_spec(dof=dof)
        fs.prepass()
        torch.cuda.synchronize()
        for name, gbs, sds in (("two RGB", g6, [0.1, 0.02]), ("two RGB + depth + material id", g8, [0.1, 0.02, 2.0, 0.5])):
            out = torch.zeros(H, W, 3, device=dev)
            a, keep = api.make_filter_args(n=[n], mean=[], m2=[], m3=[], film=[fs.state["radiance"]["film_mean"]], mean_corr=[fs.mean_corr],
                                           disc=[fs.disc], film_filtered=[out], g_buffers=gbs, g_sds=sds, filter_sd=radius / 2.0, radius=radius)
            v, ms = timed(a, 3, 0, 10)
            line = "r=%2d dof=%d RGB   %-30s %-20s %7.3f ms" % (radius, dof, name, v, ms)
            if dof and len(gbs) == 4:
                vg, msg = timed(a, 3, 1, 1)
                line += "   (%s %.1f ms)" % (vg, msg)
            print(line, flush=True)
            mc = [fs.mean_corr[..., b:b + 1].contiguous() for b in range(2)]
            dc = [fs.disc[..., b:b + 1].contiguous() for b in range(2)]
            col = [fs.state["radiance"]["film_mean"][..., b:b + 1].contiguous() for b in range(2)]
            outs = [torch.zeros(H, W, 1, device=dev) for _ in range(2)]
            a, keep = api.make_filter_args(n=[n, n], mean=[], m2=[], m3=[], film=col, mean_corr=mc, disc=dc, film_filtered=outs,
                                           g_buffers=gbs, g_sds=sds, filter_sd=radius / 2.0, radius=radius)
            v, ms = timed(a, 1, 0, 10)
            print("r=%2d dof=%d float %-30s %-20s %7.3f ms  (two buffers)" % (radius, dof, name, v, ms), flush=True)
api.set_filter_spec()
