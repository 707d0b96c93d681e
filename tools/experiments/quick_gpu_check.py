"""First-contact GPU check: small parity vs the oracle + 1080p timings. Scratch tool."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from statmc_amd import api, film, synthetic
from oracle import oracle

dev = torch.device("cuda:0")
api.setup(0)

def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))

# ---- small parity
W, H, S = 300, 41, 16
sc = synthetic.Scene(W, H, seed=3)
smp = sc.samples(S, seed=5, features=("radiance", "normal", "albedo"))
fs = film.FilmStats(W, H, dev)
fs.accumulate({k: v.to(dev) for k, v in smp.items()})
torch.cuda.synchronize()
ost = {}
for t in ("radiance", "normal", "albedo"):
    st = oracle.new_state(H, W, 3)
    oracle.accumulate(st, smp[t].numpy(), film.STAT_TYPES[t]["transform"], film.STAT_TYPES[t]["max_moment"])
    ost[t] = st
    for k in ("n", "mean", "m2", "m3", "film_mean", "film_m2"):
        g = fs.state[t][k]
        if g is None: continue
        g = g.cpu().numpy()
        print(t, k, "exact" if np.array_equal(g, st[k]) else "relL2=%.3g" % rel_l2(g, st[k]))
for variant, force in (("auto", 0), ("lds_rt", 2), ("generic", 1)):
    api.force_filter_variant(force)
    out = fs.denoise().cpu().numpy()
    torch.cuda.synchronize()
    v = api.last_filter_variant()
    # oracle on the GPU's own accumulated state (isolates the filter)
    rad = {k: (x.cpu().numpy() if x is not None else None) for k, x in fs.state["radiance"].items()}
    mc, dc = oracle.prepass(rad["n"], rad["mean"], rad["m2"], rad["m3"])
    print(variant, v, "prepass mc exact:", np.array_equal(mc, fs.mean_corr.cpu().numpy(), equal_nan=True),
          "disc exact:", np.array_equal(dc, fs.disc.cpu().numpy(), equal_nan=True))
    gb = [fs.g_buffer("normal").cpu().numpy(), fs.g_buffer("albedo").cpu().numpy()]
    ref = oracle.filter_image(mc, dc, rad["film_mean"], gb, [-0.5 / 0.1 ** 2, -0.5 / 0.02 ** 2], -0.5 / 100.0, 20)
    print("   filter relL2 per channel:", [rel_l2(out[..., c], ref[..., c]) for c in range(3)],
          "max abs", float(np.abs(out - ref).max()))
api.force_filter_variant(0)

# ---- 1080p timings
W, H, S = 1920, 1080, 64
sc = synthetic.Scene(W, H, seed=1, device=dev)
smp = sc.samples(S, seed=2)
fs = film.FilmStats(W, H, dev, types=synthetic.FEATURES)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t_acc = timeit(lambda: fs.accumulate(smp))
bytes_acc = W * H * (44 * S + 224)
print("accumulate 11ch S=%d: %.3f ms  %.1f GB/s" % (S, t_acc, bytes_acc / t_acc / 1e6))
t_pre = timeit(fs.prepass)
print("prepass: %.3f ms  %.1f GB/s" % (t_pre, W * H * 64 / t_pre / 1e6))
for force in (0, 2):
    api.force_filter_variant(force)
    t_f = timeit(fs.window_filter, 3)
    print("filter %s: %.3f ms  %.1f Mpx/s  %.1f GB/s" % (api.last_filter_variant(), t_f, W * H / t_f / 1e3, W * H * 72 / t_f / 1e6))
api.force_filter_variant(0)
for parts in (1, 2, 3, 4, 5, 6, 8):
    api.force_filter_parts(parts)
    t_f = timeit(fs.window_filter, 5)
    print("filter lds_r20 parts=%d: %.3f ms  %.1f Mpx/s" % (parts, t_f, W * H / t_f / 1e3))
api.force_filter_parts(0)
api.force_filter_variant(1)
t_g = timeit(lambda: fs.window_filter(roi=(0, 0, 1920, 64)), 1)
print("generic (64 rows): %.3f ms -> full %.1f ms" % (t_g, t_g * 1080 / 64))
