#!/usr/bin/env python3
"""Secondary measurements on one MI355X: the other BASELINE.json configs'
shapes, the 9-channel variant, the r = 6 / sd = 3 glass-caustics filter, and the multi-buffer
filter<float> calls of ACRR (5 buffers) and SMIS (12).  Writes one JSON object to stdout."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from statmc_amd import api, film, synthetic  # noqa: E402

dev = torch.device("cuda:0")
api.setup(0)
PLACED = os.environ.get("STATMC_SWEEP_PLACED", "1") != "0"      # moments and arenas from statmc_malloc_placed; 0: torch's allocator


def arena_of(chunks, t):
    """the type's samples of all chunks as one arena (placed: a statmc_malloc_placed stream block)"""
    if not PLACED:
        return torch.cat([c[t] for c in chunks])
    S = sum(c[t].shape[0] for c in chunks)
    a = api.empty_placed((S,) + tuple(chunks[0][t].shape[1:]), torch.float32, dev, api.MEM_STREAM)
    pos = 0
    for c in chunks:
        a[pos:pos + c[t].shape[0]] = c[t]
        pos += c[t].shape[0]
    return a


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def acc_bpp(spp, types):
    t = 0
    for x in types:
        c = film.STAT_TYPES[x]
        planes = c["max_moment"] + (2 if c["transform"] else 0)
        t += 4 * c["channels"] * spp + 2 * (4 + 4 * c["channels"] * planes)
    return t


out = {"placed_buffers": PLACED}
for name, W, H, spp, types, radius, sd in (
        ("C2_1280x720_64spp_11ch_r20", 1280, 720, 64, synthetic.FEATURES, 20, 10.0),
        ("C3_1920x1080_256spp_9ch_r20", 1920, 1080, 256, ("radiance", "normal", "albedo"), 20, 10.0),
        ("C3_1920x1080_64spp_11ch_r6_sd3", 1920, 1080, 64, synthetic.FEATURES, 6, 3.0),
        ("C5_3840x2160_64spp_11ch_r20", 3840, 2160, 64, synthetic.FEATURES, 20, 10.0)):
    sc = synthetic.Scene(W, H, seed=1, device=dev)
    chunks = [sc.samples(min(32, spp - s0), seed=10 + s0, features=types) for s0 in range(0, spp, 32)]
    smp = {t: arena_of(chunks, t) for t in types}
    del chunks
    fs = film.FilmStats(W, H, dev, types=types, filter_sd=sd, radius=radius, placed=PLACED)
    t_acc = timeit(lambda: fs.accumulate(smp), 3)
    t_pre = timeit(fs.prepass)
    t_flt = timeit(fs.window_filter, 3)
    px = W * H
    out[name] = {
        "accumulate_ms": round(t_acc, 4), "accumulate_GBs": round(acc_bpp(spp, types) * px / t_acc / 1e6, 1),
        "prepass_ms": round(t_pre, 4), "filter_ms": round(t_flt, 4), "filter_variant": api.last_filter_variant(),
        "filter_mpix_s": round(px / t_flt / 1e3, 1),
        "step_mpix_s": round(px / (t_acc + t_pre + t_flt) / 1e3, 1),
    }
    del smp, fs, sc
    torch.cuda.empty_cache()

# filter<float> with nBuffers = 5 (ACRR) and 12 (SMIS) at 1080p: statistics of scaled luminance
W, H = 1920, 1080
sc = synthetic.Scene(W, H, seed=2, device=dev)
smp = sc.samples(16, seed=3, features=("radiance", "normal", "albedo"))
fs = film.FilmStats(W, H, dev)
fs.accumulate(smp)
lum = smp["radiance"].mean(dim=3, keepdim=True).contiguous()
for nb, label in ((5, "acrr_filter_f32_5_buffers"), (12, "smis_filter_f32_12_buffers")):
    sts = []
    for b in range(nb):
        st = film.new_state(H, W, 1, dev, transform=True)
        api.accumulate(W, H, [api.make_stat_type((lum / (1 + b)).contiguous(), st, True, 3)])
        sts.append(st)
    z = lambda: [torch.zeros(H, W, 1, device=dev) for _ in range(nb)]
    mc, dc, ff = z(), z(), z()
    a, keep = api.make_filter_args([s["n"] for s in sts], [s["mean"] for s in sts], [s["m2"] for s in sts],
                                   [s["m3"] for s in sts], [s["film_mean"] for s in sts], mc, dc, ff,
                                   [fs.g_buffer("normal"), fs.g_buffer("albedo")], g_sds=[0.1, 0.02])
    t = timeit(lambda: api.filter_f32(a), 3)
    out[label] = {"ms": round(t, 4), "variant": api.last_filter_variant(), "mpix_s_per_buffer": round(nb * W * H / t / 1e3, 1)}

# The accumulation at the batch sizes the reference's progressive schedule launches (statpath.cpp:272-279: 4, 4, 8, 16, ...
# samples per iteration): film-major and tile-fed, 1080p and 4K (bench.py's `accumulate_by_batch` leg on two films)
import bench  # noqa: E402
bench.torch = torch
bench.PLACED["on"] = PLACED
for W, H in ((1920, 1080), (3840, 2160)):
    sc = synthetic.Scene(W, H, seed=1, device=dev)
    chunks = [sc.samples(32, seed=10 + s0, features=synthetic.FEATURES) for s0 in range(0, 64, 32)]
    smp = {t: arena_of(chunks, t) for t in synthetic.FEATURES}
    del chunks
    fs = film.FilmStats(W, H, dev, types=synthetic.FEATURES, placed=PLACED)
    out["accumulate_by_batch_%dx%d" % (W, H)] = bench.accumulate_by_batch(fs, smp, list(synthetic.FEATURES))
    del smp, fs, sc
    torch.cuda.empty_cache()
if PLACED:
    out["placement"] = api.placement_info()
print(json.dumps(out, indent=1))
