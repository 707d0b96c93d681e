"""CPU tests of the oracle itself: the one known-answer vector that exists for this path, and
reference-free properties of the restated algorithms."""
import json
import os

import numpy as np
import pytest

from conftest import rel_l2

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_known_answer_survey_appendix_a(oracle):
    """float pixel, samples {0.25, 1.5, 0, 7.25, 0.5} through
    AddTransformSampleM3 of the compiled reference header (estimator.h:188-226)."""
    kat = json.load(open(os.path.join(GOLDEN, "kat_survey_appendix_a.json")))
    px = oracle.add_samples_to_pixel(kat["samples"], 1, True, 3)
    assert int(px["n"]) == kat["n"]
    for k in ("mean", "m2", "m3", "film_mean", "film_m2"):
        # the survey printed 9 significant digits: identical after rounding to float32
        assert np.float32(px[k]) == np.float32(kat[k]), (k, px[k], kat[k])


def test_box_cox(oracle):
    assert oracle.box_cox(0.0) == -2.0              # zero-radiance paths (hard parts)
    assert oracle.box_cox(1.0) == 0.0
    assert oracle.box_cox(4.0) == 2.0
    assert np.isnan(oracle.box_cox(-1e-6))          # tiny negative samples give NaN in the reference


@pytest.mark.parametrize("channels", [1, 3])
@pytest.mark.parametrize("max_moment", [1, 2, 3])
@pytest.mark.parametrize("transform", [False, True])
def test_accumulate_matches_float64_moments(oracle, channels, max_moment, transform):
    rng = np.random.default_rng(5)
    S, H, W = 24, 9, 21   # ragged vs the 16x16 tiles
    smp = rng.lognormal(0, 1, size=(S, H, W, channels)).astype(np.float32)
    st = oracle.new_state(H, W, channels)
    oracle.accumulate(st, smp[:10], transform, max_moment)
    oracle.accumulate(st, smp[10:], transform, max_moment)   # state persists across batches
    assert (st["n"] == S).all()
    v = smp.astype(np.float64)
    tv = (np.sqrt(v) - 1) / 0.5 if transform else v
    mean = tv.mean(0)
    assert rel_l2(st["mean"], mean) < 1e-5
    if max_moment >= 2:
        assert rel_l2(st["m2"], ((tv - mean) ** 2).sum(0)) < 1e-4
    else:
        assert not st["m2"].any()
    if max_moment >= 3:
        assert rel_l2(st["m3"], ((tv - mean) ** 3).sum(0)) < 2e-3   # ill-conditioned in fp32
    else:
        assert not st["m3"].any()
    if transform:
        assert rel_l2(st["film_mean"], v.mean(0)) < 1e-5
        assert rel_l2(st["film_m2"], ((v - v.mean(0)) ** 2).sum(0)) < 1e-4
    else:   # AddSample copies mean/m2 (estimator.h:209-210)
        assert np.array_equal(st["film_mean"], st["mean"])
        assert np.array_equal(st["film_m2"], st["m2"])


def test_accumulate_edge_cases(oracle):
    # constant samples: m2 = m3 = 0 exactly; single sample: n = 1, mean = sample
    st = oracle.new_state(4, 5, 3)
    oracle.accumulate(st, np.full((6, 4, 5, 3), 0.75, np.float32), False, 3)
    assert (st["mean"] == 0.75).all() and not st["m2"].any() and not st["m3"].any()
    st = oracle.new_state(2, 2, 1)
    oracle.accumulate(st, np.full((1, 2, 2, 1), 3.0, np.float32), True, 3)
    assert (st["n"] == 1).all() and (st["mean"] == oracle.box_cox(3.0)).all() and (st["film_mean"] == 3.0).all()
    # empty batch leaves the state alone
    before = {k: v.copy() for k, v in st.items()}
    oracle.accumulate(st, np.zeros((0, 2, 2, 1), np.float32), True, 3)
    assert all(np.array_equal(before[k], st[k]) for k in st)


def test_tile_order_is_irrelevant(oracle):
    """Tiles are disjoint (statpath.cpp:132-190): any tile size gives the same images."""
    rng = np.random.default_rng(2)
    smp = rng.random((8, 33, 47, 3), dtype=np.float32)
    a, b = oracle.new_state(33, 47, 3), oracle.new_state(33, 47, 3)
    oracle.accumulate(a, smp, True, 3, tile_size=16, threads=1)
    oracle.accumulate(b, smp, True, 3, tile_size=5, threads=0)
    assert all(np.array_equal(a[k], b[k]) for k in a)


@pytest.mark.parametrize("channels,transform,max_moment", [(3, True, 3), (1, False, 1), (3, False, 2), (1, True, 3)])
def test_tile_stream_gives_the_bits_of_the_film_major_entry(oracle, channels, transform, max_moment):
    """oracle_accumulate_tile_stream (samples laid out [tile][pixel][S][C], the order Render<T> produces them in: what bench.py's
    cpu_baseline times) == oracle_accumulate_image (film-major planes), bit for bit, ragged edge tiles and a second batch included."""
    rng = np.random.default_rng(11 + channels + max_moment)
    for h, w in ((40, 52), (32, 64), (5, 7)):
        S = 7
        a, b = oracle.new_state(h, w, channels), oracle.new_state(h, w, channels)
        for batch in range(2):
            smp = rng.random((S, h, w, channels), dtype=np.float32) * (1 + batch)
            oracle.accumulate(a, smp, transform, max_moment)
            oracle.accumulate_tile_stream(b, oracle.to_tile_major(smp), S, transform, max_moment, threads=1 + batch)
        for k in a:
            assert np.array_equal(a[k].view(np.int32), b[k].view(np.int32)), (k, h, w)


def test_bench_cpu_share_reads_the_quota():
    """bench.py's cpu_share(): the affinity mask and, where the cgroup sets one, the CPU quota -- what the CPU baseline sizes its
    thread count by (a GPU box of the pool: 256 CPUs in the mask, a quota of 16)."""
    import bench
    sh = bench.cpu_share()
    assert sh["affinity_cpus"] == len(os.sched_getaffinity(0)) >= 1
    assert 1 <= sh["usable_cpus"] <= sh["affinity_cpus"]
    assert sh["cgroup_quota_cpus"] is None or sh["cgroup_quota_cpus"] > 0


def test_merge_tile_layout(oracle):
    """StatTilePixel<Vec3> is 128 B, <float> 64 B (estimator.h:104-124); MergeTile does not
    touch the film images, MergeTransformTile does (estimator.cpp:341-388)."""
    assert oracle.TILE_PIXEL_DTYPE[1].itemsize == 64 and oracle.TILE_PIXEL_DTYPE[3].itemsize == 128
    tile = np.zeros(6, dtype=oracle.TILE_PIXEL_DTYPE[3])   # 3 x 2 tile at (4, 1)
    tile["n"] = np.arange(6) + (1 << 33)                    # uint64 -> int32 cast (estimator.cpp:347)
    for i, k in enumerate(("mean", "m2", "m3", "film_mean", "film_m2")):
        tile[k] = (np.arange(18).reshape(6, 3) + 100 * i).astype(np.float32)
    st = oracle.new_state(5, 9, 3)
    st["film_mean"][:] = -1
    oracle.merge_tile(tile, 3, 4, 1, 7, 3, st, transform=False)
    assert np.array_equal(st["n"][1:3, 4:7].ravel(), np.arange(6))
    assert np.array_equal(st["m3"][2, 5], [212, 213, 214])
    assert (st["film_mean"] == -1).all()
    oracle.merge_tile(tile, 3, 4, 1, 7, 3, st, transform=True)
    assert np.array_equal(st["film_m2"][1, 4], [400, 401, 402])
    assert st["n"].sum() == 15 and st["mean"][0].sum() == 0


def test_mean_vars_row_quirk(oracle):
    n = np.array([[4, 9, 9], [2, 2, 5]], np.int32)
    m2 = np.ones((2, 3, 3), np.float32)
    exact = oracle.mean_vars(n, m2, row_n_quirk=False)
    quirk = oracle.mean_vars(n, m2, row_n_quirk=True)      # estimator.cpp:540,558
    assert np.allclose(exact[0, 1], 1 / 72) and np.allclose(quirk[0, 1], 1 / 12)
    assert np.array_equal(exact[:, 0], quirk[:, 0])


def test_prepass_spec(oracle):
    n = np.array([[0, 1, 2, 16, 16]], np.int32)
    mean = np.full((1, 5, 1), 0.5, np.float32)
    m2 = np.array([0, 0, 2.0, 0.0, 30.0], np.float32).reshape(1, 5, 1)
    m3 = np.array([0, 0, 1.0, 0.0, -12.0], np.float32).reshape(1, 5, 1)
    mc, d = oracle.prepass(n, mean, m2, m3)
    assert np.isinf(d[0, 0, 0]) and np.isinf(d[0, 1, 0]) and mc[0, 0, 0] == 0.5   # n < 2: cannot reject
    assert d[0, 3, 0] == 0 and mc[0, 3, 0] == 0.5                                  # zero variance
    var, t = 30.0 / 15, oracle.t_quantile(0, 15)
    assert np.isclose(d[0, 4, 0], t * t * var / 16, rtol=1e-6)
    assert np.isclose(mc[0, 4, 0], 0.5 + (-12.0 / 16) / (6 * var * 16), rtol=1e-6)  # Johnson


def _joint_bilateral(colour, gbs, drs, ds, r):
    """Independent numpy cross-bilateral filter (float64), window clipped at the border."""
    h, w, c = colour.shape
    out = np.zeros((h, w, c))
    for y in range(h):
        for x in range(w):
            y0, y1, x0, x1 = max(0, y - r), min(h, y + r + 1), max(0, x - r), min(w, x + r + 1)
            yy, xx = np.mgrid[y0:y1, x0:x1]
            e = ds * ((yy - y) ** 2 + (xx - x) ** 2)
            for g, dr in zip(gbs, drs):
                e = e + dr * ((g[y0:y1, x0:x1].astype(np.float64) - g[y, x]) ** 2).sum(-1)
            wgt = np.exp(e)
            out[y, x] = (wgt[..., None] * colour[y0:y1, x0:x1]).sum((0, 1)) / wgt.sum()
    return out


def test_filter_without_gate_is_joint_bilateral(oracle):
    rng = np.random.default_rng(9)
    h, w, r = 13, 17, 4
    colour = rng.random((h, w, 3), dtype=np.float32)
    gb = [rng.random((h, w, 3), dtype=np.float32), rng.random((h, w, 1), dtype=np.float32)]
    drs = [-0.5 / 0.4 ** 2, -0.5 / 0.7 ** 2]
    mc = rng.random((h, w, 3), dtype=np.float32)
    disc = np.full((h, w, 3), np.inf, np.float32)             # every pair passes
    out = oracle.filter_image(mc, disc, colour, gb, drs, -0.5 / 3.0 ** 2, r)
    assert rel_l2(out, _joint_bilateral(colour, gb, drs, -0.5 / 9.0, r)) < 1e-6


def test_filter_properties(oracle):
    rng = np.random.default_rng(4)
    h, w = 12, 15
    mc = rng.random((h, w, 3), dtype=np.float32)
    disc = (0.05 * rng.random((h, w, 3))).astype(np.float32)
    colour = rng.random((h, w, 3), dtype=np.float32)
    gb = [rng.random((h, w, 3), dtype=np.float32)]
    args = dict(g_buffers=gb, g_dr=[-2.0], ds=-0.02, radius=5)
    # radius 0 -> identity; constant colour is a fixed point; zero discriminator keeps only equal means
    assert np.array_equal(oracle.filter_image(mc, disc, colour, gb, [-2.0], -0.02, 0), colour)
    const = np.full_like(colour, 0.25)
    assert np.allclose(oracle.filter_image(mc, disc, const, **args), 0.25, rtol=1e-6)
    assert np.array_equal(oracle.filter_image(mc, np.zeros_like(disc), colour, **args), colour)
    # linear in the colour image (weights do not depend on it)
    c2 = rng.random((h, w, 3), dtype=np.float32)
    f1, f2 = oracle.filter_image(mc, disc, colour, **args), oracle.filter_image(mc, disc, c2, **args)
    f12 = oracle.filter_image(mc, disc, (colour + 2 * c2).astype(np.float32), **args)
    assert rel_l2(f12, f1 + 2 * f2) < 1e-6
    # a NaN pixel (negative radiance sample) is excluded from its neighbours and passes through itself
    mcn = mc.copy()
    mcn[5, 7] = np.nan
    fn = oracle.filter_image(mcn, disc, colour, **args)
    assert np.isfinite(fn).all() and np.array_equal(fn[5, 7], colour[5, 7])
    # ROI computes exactly the same values inside, leaves the rest untouched (zero)
    roi = oracle.filter_image(mc, disc, colour, roi=(3, 2, 9, 8), **args)
    assert np.array_equal(roi[2:8, 3:9], f1[2:8, 3:9]) and not roi[:2].any() and not roi[:, 9:].any()


def test_filter_float_variant(oracle):
    rng = np.random.default_rng(6)
    h, w = 10, 11
    mc = rng.random((h, w, 1), dtype=np.float32)
    disc = (0.1 * rng.random((h, w, 1))).astype(np.float32)
    colour = rng.random((h, w, 1), dtype=np.float32)
    out = oracle.filter_image(mc, disc, colour, [], [], -0.05, 3)
    assert out.shape == colour.shape and np.isfinite(out).all()
    assert out.min() >= colour.min() - 1e-6 and out.max() <= colour.max() + 1e-6   # convex combination


def test_film_update(oracle):
    """Film::UpdateImage (film.cpp:188-222): known values + the clamp and the zero-weight case."""
    px = np.zeros(4, dtype=oracle.FILM_PIXEL_DTYPE)
    px["xyz"] = [[0.9505, 1.0, 1.089], [0.2, 0.1, 0.05], [0.0, 0.5, 0.0], [1, 1, 1]]   # D65 white -> RGB (1,1,1)
    px["filter_weight_sum"] = [1.0, 2.0, 1.0, 0.0]
    px["splat_xyz"][1] = [0.1, 0.1, 0.1]
    rgb = oracle.film_update(px, splat_scale=0.5, scale=2.0)
    assert np.allclose(rgb[0], [2.0, 2.0, 2.0], atol=2e-3)
    assert (rgb >= 0).all() or rgb[3].min() < 0          # only the zero-weight pixel skips the clamp
    assert rgb[2, 0] == 0.0 and rgb[2, 2] == 0.0         # negative R and B of a pure-Y colour are clamped
    assert np.allclose(rgb[3], 2.0 * np.array([3.240479 - 1.537150 - 0.498535, -0.969256 + 1.875991 + 0.041556,
                                               0.055648 - 0.204043 + 1.057311]), rtol=1e-6)   # weight 0: no division


# ------------------------------------------------------------------ filter spec v2: the open choices
def _spec_case(oracle, seed=3, W=40, H=22, spp=6):
    from conftest import make_case
    _, smp, st = make_case(W, H, spp, seed=seed)
    return st


SPEC_FIELDS = ("gate", "channel_rule", "sides", "dof", "border", "small_n")


def test_default_spec_is_symmetric_in_the_pair(oracle):
    """Spec v2: member(p, q) and the range weight have the same bits for (p, q) and (q, p), so the weight
    matrix of a window is symmetric -- what lets a kernel evaluate every pair once.  Read off the filter
    itself: with the indicator image of pixel a as colour, out[b] = w(b, a) / sum_w(b), and a constant image
    filtered WITHOUT normalisation is not available, so the check is on membership (w > 0) and on the ratio
    w(b, a) / w(a, b) = sum_w(a) / sum_w(b), with sum_w taken from a second indicator pair."""
    st = _spec_case(oracle)
    rad = st["radiance"]
    mc, dc = oracle.prepass(rad["n"], rad["mean"], rad["m2"], rad["m3"])
    H, W = rad["n"].shape
    gbs, g_dr, ds, r = [st["normal"]["mean"], st["albedo"]["mean"]], [-50.0, -1250.0], -0.005, 6
    rng = np.random.default_rng(0)

    def column(a):   # w(b, a) / sum_w(b) for every b
        ind = np.zeros((H, W, 3), np.float32)
        ind[a] = 1.0
        return oracle.filter_image(mc, dc, ind, gbs, g_dr, ds, r)[..., 0]

    cols = {}
    pts = [(int(rng.integers(0, H)), int(rng.integers(0, W))) for _ in range(10)]
    pts += [(p[0] + int(rng.integers(-r, r + 1)), p[1] + int(rng.integers(-r, r + 1))) for p in pts]
    pts = [p for p in pts if 0 <= p[0] < H and 0 <= p[1] < W]
    for a in pts:
        cols[a] = column(a)
    n_pairs = 0
    for a in pts:
        for b in pts:
            if a == b or abs(a[0] - b[0]) > r or abs(a[1] - b[1]) > r:
                continue
            n_pairs += 1
            wba, wab = cols[a][b], cols[b][a]          # normalised by sum_w(b) and sum_w(a)
            assert (wba > 0) == (wab > 0)              # a is a member of b's window  <=>  b of a's
            if wba > 0:                                # same weight: w(b,a) sum_w... ratios agree with the self weights
                sa, sb = 1.0 / cols[a][a], 1.0 / cols[b][b]   # w(a, a) = 1, so the diagonal entry is 1 / sum_w
                assert abs(wba * sb - wab * sa) <= 1e-5 * wba * sb
    assert n_pairs > 10


@pytest.mark.parametrize("field", SPEC_FIELDS)
def test_each_spec_field_changes_the_result_and_keeps_the_invariants(oracle, field):
    """Every option is live (flipping it changes some pixel of a noisy low-spp case) and every variant keeps
    what all of them must keep: a constant colour image is a fixed point, output within the window's range."""
    st = _spec_case(oracle, seed=11, spp=3)
    rad = st["radiance"]
    n = rad["n"].copy()
    n[2, 3] = 1                                   # one pixel with fewer than two samples
    gbs, g_dr, ds, r = [st["normal"]["mean"], st["albedo"]["mean"]], [-50.0, -1250.0], -0.005, 5
    base = oracle.FilterSpec()
    var = oracle.FilterSpec(**{field: 1})
    outs = []
    for spec in (base, var):
        mc, dc = oracle.prepass(n, rad["mean"], rad["m2"], rad["m3"], spec=spec)
        out = oracle.filter_image(mc, dc, rad["film_mean"], gbs, g_dr, ds, r, spec=spec, n=n)
        const = oracle.filter_image(mc, dc, np.full_like(rad["film_mean"], 0.25), gbs, g_dr, ds, r, spec=spec, n=n)
        assert np.abs(const - 0.25).max() < 1e-6
        assert np.isfinite(out).all()
        assert out.min() >= rad["film_mean"].min() - 1e-6 and out.max() <= rad["film_mean"].max() + 1e-4
        outs.append(out)
    if field != "gate":   # the two gate forms differ by the rounding of one fma: decisions rarely flip
        assert not np.array_equal(outs[0], outs[1]), field


def test_spec_semantics(oracle):
    """What each option means, on hand-built statistics."""
    H, W = 1, 3
    mc = np.zeros((H, W, 3), np.float32)
    dc = np.zeros((H, W, 3), np.float32)
    col = np.zeros((H, W, 3), np.float32)
    col[0, :, 0] = (1.0, 2.0, 4.0)
    col[..., 1:] = 1.0
    # pixel 1 differs from pixel 0 by d = (1, 0, 0); D_0 = 0.6, D_1 = 0.6 in channel 0: d^2 = 1 <= 1.2 passes
    # symmetrically, fails asymmetrically the way fma(d, d, -D_q) <= D_p reads: 1 - 0.6 = 0.4 <= 0.6 passes too;
    # make it bite: D_0 = 0.9, D_1 = 0.05 -> sym: 1 <= 0.95 fails; asym (p=0,q=1): 1 - 0.05 <= 0.9 fails; equal.
    mc[0, 1, 0] = 1.0
    dc[0, 0, :] = 0.9
    dc[0, 1, :] = 0.2
    dc[0, 2, :] = 5.0
    mc[0, 2, 0] = -1.5
    f = lambda **kw: oracle.filter_image(mc, dc, col, [], [], -0.5, 2, spec=oracle.FilterSpec(**kw))
    a = f()
    # pair (0,1): d^2 = 1 <= 0.9 + 0.2 = 1.1 -> members of each other under every per-channel rule
    assert a[0, 0, 0] != col[0, 0, 0] and a[0, 1, 0] != col[0, 1, 0]
    # joint rule: channels 1 and 2 have d = 0, their slack 2 * (D_p + D_q) lets the pair (1, 2) in, which the AND rule
    # rejects: d = 2.5, d^2 = 6.25 > 0.2 + 5.0
    j = f(channel_rule=oracle.CHANNELS_JOINT)
    w12_and = oracle.filter_image(mc, dc, np.eye(3, dtype=np.float32)[None, :, :].copy(), [], [], -0.5, 2)[0, 1, 2]
    w12_joint = oracle.filter_image(mc, dc, np.eye(3, dtype=np.float32)[None, :, :].copy(), [], [], -0.5, 2,
                                    spec=oracle.FilterSpec(channel_rule=oracle.CHANNELS_JOINT))[0, 1, 2]
    assert w12_and == 0.0 and w12_joint > 0.0
    assert not np.array_equal(a, j)
    # clamp: the edge pixels repeat, so pixel 0's own colour weighs more than under clipping
    c = f(border=oracle.BORDER_CLAMP)
    assert abs(c[0, 0, 0] - col[0, 0, 0]) < abs(a[0, 0, 0] - col[0, 0, 0])
    # small_n exclude: a pixel with one sample drops out of every window and keeps its colour
    n = np.array([[5, 1, 5]], np.int32)
    mean = mc.copy()
    m2 = np.ones_like(mc)
    m3 = np.zeros_like(mc)
    spec = oracle.FilterSpec(small_n=oracle.SMALL_N_EXCLUDE)
    mcx, dcx = oracle.prepass(n, mean, m2, m3, spec=spec)
    assert np.isnan(mcx[0, 1]).all() and np.isnan(dcx[0, 1]).all()
    out = oracle.filter_image(mcx, dcx, col, [], [], -0.5, 2, spec=spec)
    assert np.array_equal(out[0, 1], col[0, 1])
    mca, dca = oracle.prepass(n, mean, m2, m3)
    assert np.isinf(dca[0, 1]).all() and np.array_equal(mca[0, 1], mean[0, 1])
    # one-sided quantiles are smaller: narrower intervals
    d1 = oracle.prepass(n, mean, m2, m3, spec=oracle.FilterSpec(sides=oracle.SIDES_ONE))[1]
    assert (d1[0, 0] < dca[0, 0]).all()
    # Welch: the discriminator image is s^2 / n and equal-n, equal-variance pairs get dof = 2(n - 1)
    sw = oracle.FilterSpec(dof=oracle.DOF_WELCH)
    mcw, dcw = oracle.prepass(n, mean, m2, m3, spec=sw)
    assert np.allclose(dcw[0, 0], (1.0 / 4) / 5)
    t8 = oracle.t_quantile(0, 8)
    thr = np.sqrt(t8 * t8 * 2 * dcw[0, 0, 0])          # |d| at which the pair (0, 2) flips
    for d, member in ((thr * 0.999, True), (thr * 1.001, False)):
        mcw2 = np.zeros_like(mcw)
        mcw2[0, 2, 0] = d
        ind = np.zeros((1, 3, 3), np.float32)
        ind[0, 2] = 1
        w = oracle.filter_image(mcw2, dcw, ind, [], [], -0.5, 2, spec=sw, n=n)[0, 0, 0]
        assert (w > 0) == member


def test_non_finite_colour_takes_the_pixel_out(oracle):
    """Spec v2: a pixel whose colour is NaN / inf takes no part -- it poisons no window and keeps its own colour."""
    st = _spec_case(oracle, seed=5)
    rad = st["radiance"]
    mc, dc = oracle.prepass(rad["n"], rad["mean"], rad["m2"], rad["m3"])
    col = rad["film_mean"].copy()
    col[4, 7, 1] = np.nan
    col[10, 20] = np.inf
    out = oracle.filter_image(mc, dc, col, [st["normal"]["mean"], st["albedo"]["mean"]], [-50.0, -1250.0], -0.005, 8)
    bad = ~np.isfinite(out)
    assert bad.sum() == 4 and bad[4, 7, 1] and bad[10, 20].all()
    mc2 = mc.copy()
    mc2[4, 7] = np.nan
    mc2[10, 20] = np.nan
    ref = oracle.filter_image(mc2, dc, np.nan_to_num(col, nan=0.0, posinf=0.0), [st["normal"]["mean"], st["albedo"]["mean"]],
                              [-50.0, -1250.0], -0.005, 8)
    mask = np.ones(out.shape, bool)
    mask[4, 7] = False
    mask[10, 20] = False
    assert np.array_equal(out[mask], ref[mask])


# ------------------------------------------------------------------ FP contraction of the reference's own build
def test_fp_contract_mode(oracle):
    """clang -O3 -march=native (the reference's recipe) fuses m2 / m3 / filmM2 updates of StatTile<Float>; the Vec3
    tiles are not contracted.  The mode changes scalar-tile results a little and RGB-tile results not at all."""
    rng = np.random.default_rng(9)
    S, H, W = 256, 4, 16
    smp1 = (rng.lognormal(0, 1.2, size=(S, H, W, 1)) * (rng.random((S, H, W, 1)) > 0.2)).astype(np.float32)
    smp3 = np.repeat(smp1, 3, axis=3).copy()
    res = {}
    try:
        for mode in (False, True):
            oracle.set_fp_contract(mode)
            s1, s3 = oracle.new_state(H, W, 1), oracle.new_state(H, W, 3)
            oracle.accumulate(s1, smp1, True, 3)
            oracle.accumulate(s3, smp3, True, 3)
            res[mode] = (s1, s3)
    finally:
        oracle.set_fp_contract(False)
    for k in ("mean", "m2", "m3", "film_mean", "film_m2"):
        assert np.array_equal(res[False][1][k], res[True][1][k]), k          # Vec3 tiles: identical
        assert np.array_equal(res[False][0][k][..., 0], res[False][1][k][..., 0]), k   # un-contracted: scalar == vector lanes
    assert np.array_equal(res[False][0]["mean"], res[True][0]["mean"])          # the mean update has nothing to fuse
    assert not np.array_equal(res[False][0]["m2"], res[True][0]["m2"])
    for k, tol in (("m2", 1e-6), ("film_m2", 1e-6), ("m3", 1e-4)):
        assert rel_l2(res[True][0][k], res[False][0][k]) < tol, k
    # the survey's known-answer vector came from a g++ -O2 probe build (no FMA): the contracted mode may differ from it
    kat = json.load(open(os.path.join(GOLDEN, "kat_survey_appendix_a.json")))
    try:
        oracle.set_fp_contract(True)
        px = oracle.add_samples_to_pixel(kat["samples"], 1, True, 3)
    finally:
        oracle.set_fp_contract(False)
    assert abs(float(px["m3"]) - kat["m3"]) <= 1e-5 * abs(kat["m3"])


def test_filter_pixel_with_a_non_finite_feature_takes_no_part(oracle):
    """Spec v2.1 (round 4): a NaN / infinite G-buffer value no longer spreads through the range weight -- the pixel takes no
    part (it keeps its colour), its neighbours are filtered as if it were not there."""
    rng = np.random.default_rng(5)
    H, W, r = 16, 24, 4
    mc = rng.standard_normal((H, W, 3)).astype(np.float32) * 0.1
    disc = np.full((H, W, 3), 0.5, np.float32)
    colour = rng.random((H, W, 3), dtype=np.float32)
    g = [rng.random((H, W, 3), dtype=np.float32), rng.random((H, W, 1), dtype=np.float32)]
    g[0][5, 7, 2] = np.nan
    g[1][10, 20, 0] = np.inf
    out = oracle.filter_image(mc, disc, colour, g, [-2.0, -3.0], -0.5 / 9.0, r)
    assert np.isfinite(out).all()
    assert np.array_equal(out[5, 7], colour[5, 7]) and np.array_equal(out[10, 20], colour[10, 20])
    # the same film with the two pixels made invalid by their statistics instead: every other pixel gets the same bits
    mc2 = mc.copy()
    mc2[5, 7] = np.nan
    mc2[10, 20] = np.nan
    g2 = [np.nan_to_num(x, nan=0.25, posinf=0.25) for x in g]
    out2 = oracle.filter_image(mc2, disc, colour, g2, [-2.0, -3.0], -0.5 / 9.0, r)
    mask = np.ones((H, W), bool)
    mask[5, 7] = mask[10, 20] = False
    assert np.array_equal(out[mask], out2[mask])


def test_centre_gate_is_the_confidence_interval_of_the_centre_pixel(oracle):
    """STATMC_GATE_CENTRE (round 4): q is a member of p's window iff d^2 <= D_p in every channel -- q's own interval plays no
    part (Moon et al. 2013; what the reference's CUDA source does under -DMEMFNC=1, README.md:147-150).  Hand-built: a tight
    pixel next to a wide one is averaged INTO the wide one but does not take the wide one in."""
    H, W = 1, 2
    mc = np.zeros((H, W, 3), np.float32)
    mc[0, 1, :] = 1.0                       # d = 1 in every channel
    dc = np.zeros((H, W, 3), np.float32)
    dc[0, 0, :] = 4.0                       # pixel 0: wide interval (d^2 = 1 <= 4)
    dc[0, 1, :] = 0.25                      # pixel 1: tight interval (1 > 0.25)
    col = np.zeros((H, W, 3), np.float32)
    col[0, 0, :] = 10.0
    col[0, 1, :] = 20.0
    f = lambda **kw: oracle.filter_image(mc, dc, col, [], [], -0.5, 1, spec=oracle.FilterSpec(**kw))
    c = f(gate=oracle.GATE_CENTRE)
    assert 10.0 < c[0, 0, 0] < 20.0         # pixel 1 lies inside pixel 0's interval
    assert np.all(c[0, 1] == 20.0)          # pixel 0 lies outside pixel 1's
    s = f()                                  # the symmetric gate takes both: 1 <= 4.25
    assert 10.0 < s[0, 0, 0] < 20.0 and 10.0 < s[0, 1, 0] < 20.0
    j = f(gate=oracle.GATE_CENTRE, channel_rule=oracle.CHANNELS_JOINT)    # pooled: 3 <= 12 / 3 > 0.75
    assert 10.0 < j[0, 0, 0] < 20.0 and np.all(j[0, 1] == 20.0)
