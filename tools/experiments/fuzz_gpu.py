"""Scratch: long randomised differential run of the HIP entry points against the CPU oracle
(the seeded short versions live in tests/test_gpu_parity.py).
usage: fuzz_gpu.py [seconds] [first_case]      run cases first_case, first_case+1, ... for `seconds`
       fuzz_gpu.py --case N                    re-run one case and print where it differs"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import oracle
from statmc_amd import api as gpu
import test_gpu_parity as T
oracle.build(); gpu.setup(0)
R20 = "--r20" in sys.argv
if R20:
    sys.argv.remove("--r20")
WELCH = "--welch" in sys.argv      # round 4: Welch degrees of freedom half of the time (the pair-symmetric kernel's Welch build)
if WELCH:
    sys.argv.remove("--welch")
G8 = "--g8" in sys.argv            # round 5: filter<float> with 1-channel G-buffers six times out of ten (the eight-plane builds; with --welch
if G8:                             # the eight-plane Welch builds, two buffers per launch with their own n - 1 planes)
    sys.argv.remove("--g8")


def random_spec(rng):
    """Mostly the default spec; otherwise any combination of the options the window filter sees (the Welch lookup,
    which runs the general kernel only, rarely)."""
    if rng.random() < (0.25 if R20 else 0.5):
        return {}
    kw = dict(gate=int(rng.integers(0, 2)), channel_rule=int(rng.integers(0, 2)), border=int(rng.integers(0, 2)))
    if rng.random() < (0.7 if WELCH else 0.1):
        kw["dof"] = 1
    return kw


def filter_case(case, verbose=False):
    rng = np.random.default_rng(1000003 * 17 + case)
    W = int(rng.choice([rng.integers(1, 12), rng.integers(12, 300), rng.integers(250, 800)]))
    H = int(rng.choice([rng.integers(1, 6), rng.integers(6, 70)]))
    radius = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 13, 19, 20, 20, 20, 21, 25]))
    if R20:       # --r20: the shipped radius only, non-default specs three times out of four (the pair-symmetric kernel's modes)
        radius = 20
    sd = float(rng.uniform(0.7, 15.0))
    g_sds = [float(10 ** rng.uniform(-2, 0)), float(10 ** rng.uniform(-2.3, 0))]
    g_dr = [-0.5 / s ** 2 for s in g_sds]
    scale = float(10 ** rng.uniform(-3, 3))
    mc = (rng.standard_normal((H, W, 3)) * scale).astype(np.float32)
    disc = ((rng.random((H, W, 3)) ** 3) * 2.0 * scale * scale).astype(np.float32)
    inj = []
    for _ in range(int(rng.integers(0, 4))):
        y, x = int(rng.integers(0, H)), int(rng.integers(0, W))
        kind = int(rng.integers(0, 5))
        inj.append((kind, x, y))
        if kind == 0: disc[y, x] = np.inf
        elif kind == 1: mc[y, x, rng.integers(0, 3)] = np.nan
        elif kind == 2: disc[y, x] = 0.0
        elif kind == 3: mc[y, x] = np.inf
        else: disc[y, x, rng.integers(0, 3)] = np.nan
    colour = (rng.random((H, W, 3), dtype=np.float32) * 3 * float(10 ** rng.uniform(-2, 3))).astype(np.float32)
    layouts = [[3, 3]] * 6 + [[3], [], [3, 1, 1], [1, 3], [1, 1, 1, 1, 1, 1], [3, 3, 1], [1], [3, 3, 1, 1], [1, 3, 1, 3], [3, 1, 3]]
    layout = layouts[int(rng.integers(0, len(layouts)))]
    gbs = [(rng.random((H, W, c), dtype=np.float32) * 2 - (i % 2)).astype(np.float32) for i, c in enumerate(layout)]
    g_sds = (g_sds + [float(10 ** rng.uniform(-1.5, 0)) for _ in layout])[:len(layout)]
    g_dr = [-0.5 / s ** 2 for s in g_sds]
    if rng.random() < 0.3:   # piecewise-constant features: many exactly equal taps
        gbs = [(np.round(g * 2) / 2).astype(np.float32) for g in gbs]
    roi = None
    if rng.random() < 0.4 and W > 2 and H > 2:
        x0, y0 = int(rng.integers(0, W - 1)), int(rng.integers(0, H - 1))
        roi = (x0, y0, int(rng.integers(x0 + 1, W + 1)), int(rng.integers(y0 + 1, H + 1)))
    if rng.random() < 0.3:   # a non-finite colour: the pixel takes no part, its own output passes it through
        y, x = int(rng.integers(0, H)), int(rng.integers(0, W))
        colour[y, x, rng.integers(0, 3)] = [np.nan, np.inf, -np.inf][int(rng.integers(0, 3))]
        inj.append((9, x, y))
    spec_kw = random_spec(rng)
    n = None
    if spec_kw.get("dof"):      # Welch: sample counts -- ragged, uniform (the usual film), or with pixels of fewer than two samples
        kind = int(rng.integers(0, 3))
        n = rng.integers(2, 400, size=(H, W)).astype(np.int32) if kind == 0 else np.full((H, W), int(rng.choice([2, 3, 4, 16, 64, 256, 1024, 5000])), np.int32)
        if kind == 2:
            n[rng.random((H, W)) < 0.05] = int(rng.integers(0, 2))
    ref = oracle.filter_image(mc, disc, colour, gbs, g_dr, -0.5 / sd ** 2, radius, roi=roi, spec=oracle.FilterSpec(**spec_kw), n=n)
    force = int(rng.choice([0, 0, 0, 2, 1, 3]))
    parts = int(rng.choice([0, 0, 1, 2, 3, 5, 41]))
    gpu.force_filter_parts(parts)
    gpu.set_filter_spec(**spec_kw)
    try:
        out, v = T.run_filter(gpu, mc, disc, colour, gbs, g_dr, sd, radius, roi=roi, force=force, n=n)
    finally:
        gpu.force_filter_parts(0)
        gpu.set_filter_spec()
    mask = np.isfinite(ref)
    ok = np.array_equal(np.isfinite(out), mask)
    err = 0.0
    if ok and mask.any():
        err = max(T.rel_l2(np.where(mask[..., c], out[..., c], 0), np.where(mask[..., c], ref[..., c], 0)) for c in range(3))
    desc = dict(case=case, W=W, H=H, radius=radius, sd=round(sd, 3), g_sds=[round(g, 4) for g in g_sds], scale=scale, roi=roi,
                force=force, parts=parts, variant=v, inj=inj, err=err, finite_ok=ok, layout=layout, spec=spec_kw)
    if verbose:
        print(desc)
        d = np.abs(out.astype(np.float64) - ref) / (np.abs(ref) + 1e-30)
        d[~mask] = 0
        idx = np.argsort(d.max(axis=2).ravel())[::-1][:8]
        for i in idx:
            y, x = divmod(int(i), W)
            print("  (x=%d,y=%d) gpu %s ref %s colour %s" % (x, y, out[y, x], ref[y, x], colour[y, x]))
    return (ok and err <= 1e-5), desc


def accumulate_case(case):
    rng = np.random.default_rng(1000003 * 31 + case)
    W2, H2 = int(rng.integers(1, 90)), int(rng.integers(1, 40))
    C = int(rng.choice([1, 3])); tr = bool(rng.integers(0, 2)); mm = int(rng.integers(1, 4)); S = int(rng.choice([0, 1, 2, 3, 5, 7, 16, 33]))
    st = oracle.new_state(H2, W2, C)
    if rng.random() < 0.5:
        st["n"][...] = rng.integers(0, 5000, size=(H2, W2)).astype(np.int32)
        for k in ("mean", "film_mean"): st[k][...] = rng.standard_normal((H2, W2, C)).astype(np.float32)
        for k in ("m2", "film_m2", "m3"): st[k][...] = rng.random((H2, W2, C)).astype(np.float32)
    smp = (rng.lognormal(0, 2.0, size=(S, H2, W2, C)) * float(10 ** rng.uniform(-3, 3))).astype(np.float32)
    smp[rng.random(smp.shape) < 0.2] = 0.0
    dst = T.dev_state(st)
    oracle.accumulate(st, smp, tr, mm)
    gpu.accumulate(W2, H2, [gpu.make_stat_type(T.to_dev(smp), dst, tr, mm)])
    torch.cuda.synchronize()
    bad = []
    for k in ("n", "mean", "m2", "m3", "film_mean", "film_m2"):
        if k == "m2" and mm < 2: continue
        if k == "m3" and mm < 3: continue
        if k in ("film_mean", "film_m2") and not tr: continue
        got = dst[k].cpu().numpy()
        if (not tr) or k in ("n", "film_mean", "film_m2"):
            good = np.array_equal(got, st[k])
        elif k == "m3":   # sums of cubes cancel (exactly 0 after two samples): error against the size of the terms
            den = np.sqrt((st["m2"].astype(np.float64) ** 3).sum()) + 1e-300
            good = np.sqrt(((got.astype(np.float64) - st[k]) ** 2).sum()) / den <= 1e-5
        else:
            good = T.rel_l2(got, st[k]) <= 1e-5
        if not good: bad.append(k)
    return not bad, dict(case=case, W=W2, H=H2, C=C, transform=tr, max_moment=mm, S=S, bad=bad)


def prepass_case(case):
    rng = np.random.default_rng(1000003 * 43 + case)
    W, H, C = int(rng.integers(1, 70)), int(rng.integers(1, 20)), int(rng.choice([1, 3]))
    n = rng.choice([0, 1, 2, 3, 7, 64, 4095, 4096, 4097, 5000, 100000], size=(H, W)).astype(np.int32)
    scale = float(10 ** rng.uniform(-4, 4))
    shape = (H, W, C)
    mean = (rng.standard_normal(shape) * scale).astype(np.float32)
    m2 = (rng.random(shape) ** 2 * scale * scale * 50).astype(np.float32)
    m3 = (rng.standard_normal(shape) * scale ** 3 * 100).astype(np.float32)
    m2[rng.random(shape) < 0.1] = 0.0
    m2[rng.random(shape) < 0.02] = -1.0
    m3[rng.random(shape) < 0.02] = np.nan
    mean[rng.random(shape) < 0.02] = np.inf
    alpha = int(rng.integers(0, 3))
    mc_ref, dc_ref = oracle.prepass(n, mean, m2, m3, alpha_index=alpha)
    mc, dc = torch.zeros(shape, device=T.DEV), torch.zeros(shape, device=T.DEV)
    dummy = torch.zeros(shape, device=T.DEV)
    a, keep = gpu.make_filter_args([T.to_dev(n)], [T.to_dev(mean)], [T.to_dev(m2)], [T.to_dev(m3)], [dummy], [mc], [dc], [dummy],
                                   [], g_dr=[], filter_sd=10.0, radius=1)
    gpu.check(gpu.load().statmc_set_significance(alpha))
    try:
        gpu.prepass(a, C)
        torch.cuda.synchronize()
    finally:
        gpu.load().statmc_set_significance(0)
    eq = lambda x, y: np.array_equal(x, y, equal_nan=True)
    ok = eq(mc.cpu().numpy(), mc_ref) and eq(dc.cpu().numpy(), dc_ref)
    return ok, dict(case=case, W=W, H=H, C=C, alpha=alpha, scale=scale)


def float_filter_case(case):
    rng = np.random.default_rng(1000003 * 59 + case)
    W, H = int(rng.integers(1, 330)), int(rng.integers(1, 30))
    nb = int(rng.integers(1, 8)); radius = int(rng.choice([1, 3, 6, 8, 13, 20, 20, 22]))
    sd = float(rng.uniform(1, 12)); g_sds = [float(rng.uniform(0.05, 0.6)), float(rng.uniform(0.02, 0.5))]
    g_dr = [-0.5 / x ** 2 for x in g_sds]
    gbs = [rng.random((H, W, 3), dtype=np.float32) * 2 - 1, rng.random((H, W, 3), dtype=np.float32)]
    if G8:      # (a generator of its own: the cases without --g8 stay what they were)
        rng8 = np.random.default_rng(1000003 * 61 + case)
        if rng8.random() < 0.6:
            layout = [[3, 3, 1], [3, 3, 1, 1], [1, 3], [1], [3, 1, 1], [1, 3, 1, 3]][int(rng8.integers(0, 6))]
            gbs = [(rng8.random((H, W, c), dtype=np.float32) * 2 - (i % 2)).astype(np.float32) for i, c in enumerate(layout)]
            if rng8.random() < 0.3:
                gbs = [(np.round(g * 2) / 2).astype(np.float32) for g in gbs]
            g_dr = [-0.5 / float(10 ** rng8.uniform(-1.5, 0)) ** 2 for _ in layout]
    mcs = [rng.standard_normal((H, W, 1)).astype(np.float32) for _ in range(nb)]
    dcs = [((rng.random((H, W, 1)) ** 3) * 2).astype(np.float32) for _ in range(nb)]
    cols = [rng.random((H, W, 1), dtype=np.float32) * 5 for _ in range(nb)]
    for b in range(nb):
        for _ in range(int(rng.integers(0, 3))):
            y, x = int(rng.integers(0, H)), int(rng.integers(0, W))
            k = int(rng.integers(0, 4))
            if k == 0: dcs[b][y, x] = np.inf
            elif k == 1: mcs[b][y, x] = np.nan
            elif k == 2: mcs[b][y, x] = np.inf
            else: dcs[b][y, x] = np.nan
        if rng.random() < 0.2:
            y, x = int(rng.integers(0, H)), int(rng.integers(0, W))
            cols[b][y, x] = [np.nan, np.inf][int(rng.integers(0, 2))]
    outs = [torch.zeros(H, W, 1, device=T.DEV) for _ in range(nb)]
    force = int(rng.choice([0, 0, 0, 2, 1, 3]))
    parts = int(rng.choice([0, 0, 1, 2, 3, 7]))
    spec_kw = random_spec(rng)
    if not WELCH:
        spec_kw.pop("dof", None)
    ns = None
    if spec_kw.get("dof"):      # Welch: every buffer its own sample counts -- uniform, ragged, or with pixels of fewer than two samples
        ns = []
        for b in range(nb):
            kind = int(rng.integers(0, 3))
            n = rng.integers(2, 400, size=(H, W)).astype(np.int32) if kind == 0 else np.full((H, W), int(rng.choice([2, 3, 4, 16, 64, 256, 1024, 3000, 5000])), np.int32)
            if kind == 2:
                n[rng.random((H, W)) < 0.05] = int(rng.integers(0, 2))
            ns.append(n)
    a, keep = gpu.make_filter_args([T.to_dev(n) for n in ns] if ns else [], [], [], [], [T.to_dev(c) for c in cols], [T.to_dev(m) for m in mcs],
                                   [T.to_dev(d) for d in dcs], outs, [T.to_dev(g) for g in gbs], g_dr=g_dr, filter_sd=sd, radius=radius)
    gpu.force_filter_variant(force)
    gpu.force_filter_parts(parts)
    gpu.set_filter_spec(**spec_kw)
    try:
        gpu.window_filter(a, 1)
        torch.cuda.synchronize()
    finally:
        gpu.force_filter_variant(0)
        gpu.force_filter_parts(0)
        gpu.set_filter_spec()
    worst, finite_ok = 0.0, True
    for b in range(nb):
        ref = oracle.filter_image(mcs[b], dcs[b], cols[b], gbs, g_dr, -0.5 / sd ** 2, radius, spec=oracle.FilterSpec(**spec_kw),
                                  n=ns[b] if ns else None)
        out = outs[b].cpu().numpy()
        mask = np.isfinite(ref)
        finite_ok = finite_ok and np.array_equal(np.isfinite(out), mask)
        if mask.any():
            worst = max(worst, T.rel_l2(np.where(mask, out, 0), np.where(mask, ref, 0)))
    return finite_ok and worst <= 1e-5, dict(case=case, W=W, H=H, nb=nb, radius=radius, force=force, parts=parts, spec=spec_kw,
                                             variant=gpu.last_filter_variant(), err=worst, finite_ok=finite_ok)


def tiles_case(case):
    rng = np.random.default_rng(1000003 * 71 + case)
    W, H = int(rng.integers(1, 130)), int(rng.integers(1, 60))
    ts = int(rng.choice([4, 8, 12, 16, 16, 16, 20, 32]))
    C = int(rng.choice([1, 3])); tr = bool(rng.integers(0, 2)); mm = int(rng.integers(1, 4))
    ref = oracle.new_state(H, W, C)
    if rng.random() < 0.5:
        ref["n"][...] = rng.integers(0, 300, size=(H, W)).astype(np.int32)
        for k in ("mean", "film_mean"): ref[k][...] = rng.standard_normal((H, W, C)).astype(np.float32)
        for k in ("m2", "film_m2", "m3"): ref[k][...] = rng.random((H, W, C)).astype(np.float32)
    dev = T.dev_state(ref)
    tiles = [(x, y, min(x + ts, W), min(y + ts, H)) for y in range(0, H, ts) for x in range(0, W, ts)]
    order = rng.permutation(len(tiles))
    bounds, offsets, counts, blocks, off = [], [], [], [], 0
    pad4 = bool(rng.integers(0, 2))
    for k in order:
        x0, y0, x1, y1 = tiles[k]
        S = int(rng.choice([0, 1, 2, 3, 5, 9, 17])); npx = (x1 - x0) * (y1 - y0)
        smp = rng.lognormal(0, 1.5, size=(S, y1 - y0, x1 - x0, C)).astype(np.float32)
        size = (S * npx + 3) // 4 * 4 if pad4 else S * npx
        blk = np.zeros(size * C, np.float32); blk[:smp.size] = smp.ravel()
        bounds.append((x0, y0, x1, y1)); offsets.append(off); counts.append(S); blocks.append(blk); off += size
        if S:
            sub = {key: np.ascontiguousarray(v[y0:y1, x0:x1]) for key, v in ref.items()}
            oracle.accumulate(sub, smp, tr, mm)
            for key, v in sub.items(): ref[key][y0:y1, x0:x1] = v
    arena = T.to_dev(np.concatenate(blocks + [np.zeros(4, np.float32)]))
    st = gpu.make_stat_type_arena(arena, C, dev, tr, mm)
    gpu.accumulate_tiles(W, H, [st], T.to_dev(np.array(bounds, np.int32).reshape(-1, 4)), T.to_dev(np.array(offsets, np.int64)),
                         T.to_dev(np.array(counts, np.int32)))
    torch.cuda.synchronize()
    bad = []
    for k in ("n", "mean", "m2", "m3", "film_mean", "film_m2"):
        if k == "m2" and mm < 2: continue
        if k == "m3" and mm < 3: continue
        if k in ("film_mean", "film_m2") and not tr: continue
        got = dev[k].cpu().numpy()
        if (not tr) or k in ("n", "film_mean", "film_m2"): good = np.array_equal(got, ref[k])
        elif k == "m3":
            den = np.sqrt((ref["m2"].astype(np.float64) ** 3).sum()) + 1e-300
            good = np.sqrt(((got.astype(np.float64) - ref[k]) ** 2).sum()) / den <= 1e-5
        else: good = T.rel_l2(got, ref[k]) <= 1e-5
        if not good: bad.append(k)
    return not bad, dict(case=case, W=W, H=H, ts=ts, C=C, transform=tr, max_moment=mm, pad4=pad4, bad=bad)


if len(sys.argv) > 2 and sys.argv[1] == "--case":
    print(filter_case(int(sys.argv[2]), verbose=True)[0])
    for fn in (accumulate_case, prepass_case, float_filter_case, tiles_case):
        print(fn.__name__, fn(int(sys.argv[2])))
    sys.exit(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
case = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t_end = time.time() + budget
n = fails = 0
worst = 0.0
import collections
seen = collections.Counter()
last_print = time.time()
while time.time() < t_end:
    ok, d = filter_case(case)
    worst = max(worst, d["err"])
    seen[d["variant"]] += 1
    if time.time() - last_print > 60:
        print("... %d cases, %d failures" % (n, fails), flush=True)
        last_print = time.time()
    if not ok:
        fails += 1
        print("FAIL filter", d, flush=True)
    for name, fn in (("accumulate", accumulate_case), ("prepass", prepass_case), ("float_filter", float_filter_case),
                     ("tiles", tiles_case)):
        ok, d = fn(case)
        if name == "float_filter":
            seen[d["variant"]] += 1
        if not ok:
            fails += 1
            print("FAIL", name, d, flush=True)
    case += 1
    n += 1
print("cases %d (next %d), worst filter rel L2 %.3g, failures %d" % (n, case, worst, fails))
print("filter variants exercised:", dict(sorted(seen.items())))
