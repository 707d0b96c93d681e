"""Round 6: every build of the pair-symmetric filter that a split of the window between the two waves of a row affects, in one
process, 1080p, r = 20, back to back: RGB default / pooled channels / one-sided gate / Moon gate (time_specs' modes), eight feature
planes, two float buffers per launch (filter<float>), and the runtime-radius build at r = 19.  One line per mode.
python tools/experiments/with_variant.py NAME tools/experiments/time_modes.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

W, H = 1920, 1080
dev = torch.device("cuda:0")
api.setup(0)
scene = synthetic.Scene(W, H, seed=1, device=dev)
fs = film.FilmStats(W, H, dev, types=synthetic.FEATURES)
fs.accumulate(scene.samples(32, seed=2))
torch.cuda.synchronize()
rad = fs.state["radiance"]


def timed(a, ch, reps=20):
    for _ in range(3):
        api.window_filter(a, ch)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            api.window_filter(a, ch)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


res = []
for name, kw in (("rgb", dict()), ("joint", dict(channel_rule=1)), ("asym", dict(gate=1)), ("centre", dict(gate=2))):
    api.set_filter_spec(**kw)
    fs.prepass()
    a, keep = fs.filter_args()
    res.append((name, api.last_filter_variant() if False else None, timed(a, 3)))
    res[-1] = (name, api.last_filter_variant(), res[-1][2])
api.set_filter_spec()
fs.prepass()
out = torch.zeros_like(fs.film_f)
names, sds = ["normal", "albedo", "depth", "materialid"], [0.1, 0.02, 1.0, 0.5]
a, keep = api.make_filter_args(n=[rad["n"]], mean=[rad["mean"]], m2=[rad["m2"]], m3=[rad["m3"]], film=[rad["film_mean"]], mean_corr=[fs.mean_corr],
                               disc=[fs.disc], film_filtered=[out], g_buffers=[fs.g_buffer(g) for g in names], g_sds=sds, filter_sd=10.0, radius=20)
t = timed(a, 3)
res.append(("g8", api.last_filter_variant(), t))
a, keep = api.make_filter_args(n=[rad["n"]], mean=[rad["mean"]], m2=[rad["m2"]], m3=[rad["m3"]], film=[rad["film_mean"]], mean_corr=[fs.mean_corr],
                               disc=[fs.disc], film_filtered=[out], g_buffers=[fs.g_buffer(g) for g in names[:2]], g_sds=sds[:2], filter_sd=9.5, radius=19)
t = timed(a, 3)
res.append(("r19", api.last_filter_variant(), t))
gbs = [fs.g_buffer("normal"), fs.g_buffer("albedo")]
mc = [(fs.mean_corr[..., b:b + 1] * (1.0 / (1 + b))).contiguous() for b in range(2)]
dc = [(fs.disc[..., b:b + 1] * (1.0 / (1 + b)) ** 2).contiguous() for b in range(2)]
col = [(rad["film_mean"][..., b:b + 1] * (1.0 / (1 + b))).contiguous() for b in range(2)]
outs = [torch.zeros(H, W, 1, device=dev) for _ in range(2)]
a, keep = api.make_filter_args(n=[], mean=[], m2=[], m3=[], film=col, mean_corr=mc, disc=dc, film_filtered=outs, g_buffers=gbs, g_sds=[0.1, 0.02],
                               filter_sd=10.0, radius=20)
t = timed(a, 1, reps=10)
res.append(("pair", api.last_filter_variant(), t))
print("  ".join("%s(%s) %.3f" % r for r in res), flush=True)
