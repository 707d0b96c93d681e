"""GPU parity tests proper: every HIP entry point, called through the C ABI, against the CPU
oracle on the same seeded inputs.  Bar: bit-exact for integer work and for every floating-point
stage whose operations are IEEE-identical on both sides (raw-sample moments, pre-pass, mean-vars,
tile scatter); <= 1e-5 relative L2 per channel (the tolerance BASELINE.json states) where the
GPU uses sqrt for pow(x, 0.5) or v_exp_f32 for expf."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import FILTER_SD, RADIUS, SD_ALBEDO, SD_NORMAL, make_case, rel_l2

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
G_DR = [-0.5 / SD_NORMAL ** 2, -0.5 / SD_ALBEDO ** 2]
TOL = 1e-5


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def dev_state(st):
    return {k: to_dev(v) for k, v in st.items()}


# ------------------------------------------------------------------ accumulate
@pytest.mark.parametrize("channels", [1, 3])
@pytest.mark.parametrize("transform,max_moment", [(False, 1), (False, 2), (False, 3), (True, 1), (True, 2), (True, 3)])
def test_accumulate_matches_oracle(gpu, oracle, channels, transform, max_moment):
    rng = np.random.default_rng(10 * channels + max_moment)
    H, W = 23, 37                                           # ragged: not a multiple of 4 pixels
    batches = [4, 4, 8, 1]                                   # the reference's 4,4,8,... schedule + a single sample
    smp = rng.lognormal(0, 1, size=(sum(batches), H, W, channels)).astype(np.float32)
    smp[rng.random(smp.shape) < 0.2] = 0.0                   # zero-radiance paths -> Box-Cox -2
    smp[3, 5, 7] *= 1000.0                                   # a firefly
    ref = oracle.new_state(H, W, channels)
    st = dev_state(oracle.new_state(H, W, channels))
    s0 = 0
    for b in batches:
        part = np.ascontiguousarray(smp[s0:s0 + b])
        oracle.accumulate(ref, part, transform, max_moment)
        gpu.accumulate(W, H, [gpu.make_stat_type(to_dev(part), st, transform, max_moment)])
        s0 += b
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in st.items()}
    assert np.array_equal(got["n"], ref["n"])
    if transform:
        assert np.array_equal(got["film_mean"], ref["film_mean"])       # raw-sample moments: exact
        assert np.array_equal(got["film_m2"], ref["film_m2"])
        for k in ("mean", "m2", "m3"):
            assert rel_l2(got[k], ref[k]) <= TOL, k
    else:
        for k in ("mean", "m2", "m3"):
            assert np.array_equal(got[k], ref[k]), k
    for k, lim in (("m2", 2), ("m3", 3)):                                # moments above max_moment stay untouched
        if max_moment < lim:
            assert not got[k].any()


def test_accumulate_randomised_configurations(gpu, oracle):
    """30 seeded random launches: film size (down to 1 x 1), 0..19 samples, 1..6 stat types of mixed
    channel count / transform / max moment in ONE launch, starting from a state with non-uniform
    per-pixel counts (pixels of one film need not have seen the same number of samples)."""
    rng = np.random.default_rng(777)
    for case in range(30):
        W, H = int(rng.integers(1, 300)), int(rng.integers(1, 24))
        n_types = int(rng.integers(1, 7))
        refs, devs, stypes, cfgs, keep = [], [], [], [], []
        for t in range(n_types):
            ch = int(rng.choice([1, 3]))
            transform, mm = bool(rng.integers(0, 2)), int(rng.integers(1, 4))
            S = int(rng.integers(0, 20))
            smp = rng.lognormal(0, 1.5, size=(S, H, W, ch)).astype(np.float32)
            smp[rng.random(smp.shape) < 0.2] = 0.0
            ref = oracle.new_state(H, W, ch)
            if rng.random() < 0.5:                          # continue from an earlier, ragged state
                n0 = rng.integers(0, 40, size=(H, W)).astype(np.int32)
                ref["n"][...] = n0
                for k in ("mean", "m2", "m3", "film_mean", "film_m2"):
                    ref[k][...] = (rng.random(ref[k].shape) * (n0[..., None] > 0)).astype(np.float32).reshape(ref[k].shape)
                if not transform:
                    ref["film_mean"][...] = ref["mean"]
                    ref["film_m2"][...] = ref["m2"]
            st = dev_state(ref)
            oracle.accumulate(ref, smp, transform, mm)
            d_smp = to_dev(smp)
            keep.append(d_smp)                              # the descriptors hold raw pointers
            stypes.append(gpu.make_stat_type(d_smp, st, transform, mm))
            refs.append(ref); devs.append(st); cfgs.append((ch, transform, mm, S))
        gpu.accumulate(W, H, stypes)
        torch.cuda.synchronize()
        for ref, st, cfg in zip(refs, devs, cfgs):
            ch, transform, mm, S = cfg
            got = {k: v.cpu().numpy() for k, v in st.items()}
            assert np.array_equal(got["n"], ref["n"]), (case, cfg)
            keys = ("mean", "m2", "m3")[:mm]
            if transform:
                assert np.array_equal(got["film_mean"], ref["film_mean"]), (case, cfg)
                assert np.array_equal(got["film_m2"], ref["film_m2"]), (case, cfg)
                for k in keys:
                    assert rel_l2(got[k], ref[k]) <= TOL, (case, cfg, k)
            else:
                for k in keys:
                    assert np.array_equal(got[k], ref[k]), (case, cfg, k)


def test_exact_division_by_count(gpu, oracle):
    """The kernel divides by the sample count with a refined reciprocal + residual correction
    instead of the IEEE sequence; for this path's operands the quotient must be bit-identical:
    counts 1..4200 (beyond the t-table), heavy-tailed samples, plain M1 and M3 chains."""
    rng = np.random.default_rng(99)
    H, W = 8, 8
    for max_moment, n0, S in ((1, 0, 1400), (3, 0, 300), (1, 2999, 1201)):
        smp = rng.lognormal(0, 2.5, size=(S, H, W, 3)).astype(np.float32)
        smp[rng.random(smp.shape) < 0.1] = 0.0
        smp[5] *= 1e6
        smp[6] *= 1e-6
        ref = oracle.new_state(H, W, 3)
        if n0:                                   # continue from an existing count
            ref["n"][...] = n0
            ref["mean"][...] = rng.standard_normal((H, W, 3)).astype(np.float32)
            ref["film_mean"][...] = ref["mean"]
        st = dev_state(ref)
        oracle.accumulate(ref, smp, False, max_moment)
        gpu.accumulate(W, H, [gpu.make_stat_type(to_dev(smp), st, False, max_moment)])
        torch.cuda.synchronize()
        assert int(st["n"].max()) == n0 + S
        for k in ("mean", "m2", "m3"):
            assert np.array_equal(st[k].cpu().numpy(), ref[k]), (max_moment, n0, k)


def test_exact_division_any_count(gpu, oracle):
    """Same property over the whole range a count can take before float(n) stops being exact:
    every pixel starts from its own count n0 in [0, 2^24 - 2] (so the update divides by n0 + 1) with
    a random running mean, and receives two samples.  Ragged counts take the general walk, a second
    film with one count per 4-pixel group takes the shared-reciprocal walk."""
    rng = np.random.default_rng(123)
    H, W = 256, 1024
    for shared in (False, True):
        n0 = rng.integers(0, 2 ** 24 - 2, size=(H, W), dtype=np.int64)
        n0[0, :64] = np.arange(64)                                  # the small counts, too
        n0[1, :64] = 2 ** 24 - 3 - np.arange(64)
        n0[2, :25] = 2 ** np.arange(25) - 1                          # divisors that are powers of two
        n0[2, 24] = 2 ** 24 - 3
        if shared:
            n0 = np.repeat(n0[:, ::4], 4, axis=1)
        smp = rng.lognormal(0, 2.5, size=(2, H, W, 3)).astype(np.float32)
        smp[rng.random(smp.shape) < 0.1] = 0.0
        ref = oracle.new_state(H, W, 3)
        ref["n"][...] = n0.astype(np.int32)
        for k in ("mean", "film_mean"):
            ref[k][...] = (rng.standard_normal((H, W, 3)) * 3).astype(np.float32)
        for k in ("m2", "film_m2"):
            ref[k][...] = rng.lognormal(0, 2, size=(H, W, 3)).astype(np.float32)
        ref["m3"][...] = rng.standard_normal((H, W, 3)).astype(np.float32)
        st = dev_state(ref)
        oracle.accumulate(ref, smp, True, 3)
        gpu.accumulate(W, H, [gpu.make_stat_type(to_dev(smp), st, True, 3)])
        torch.cuda.synchronize()
        assert np.array_equal(st["n"].cpu().numpy(), ref["n"])
        # the raw-sample Welford chain has no sqrt in it: every bit must agree
        for k in ("film_mean", "film_m2"):
            assert np.array_equal(st[k].cpu().numpy(), ref[k]), (shared, k)
        for k in ("mean", "m2", "m3"):                              # Box-Cox side: sqrt vs pow
            assert rel_l2(st[k].cpu().numpy(), ref[k]) <= TOL, (shared, k)


@pytest.mark.parametrize("W,H", [(64, 40), (50, 37), (16, 16)], ids=["vector-tiles", "scalar-tiles", "one-tile"])
def test_accumulate_tiles_matches_oracle(gpu, oracle, W, H):
    """statmc_accumulate_tiles: samples handed over tile by tile (16 x 16 tiles in shuffled order, a
    different sample count per tile, some tiles empty), two iterations on the same state; the
    reference result is the oracle run on every tile's sub-image.  (50, 37): image width and edge
    tiles that do not split into aligned 4-pixel groups take the scalar path."""
    rng = np.random.default_rng(W * 1000 + H)
    cfgs = [("radiance", 3, True, 3), ("normal", 3, False, 1), ("depth", 1, False, 2)]
    ref = {name: oracle.new_state(H, W, c) for name, c, _, _ in cfgs}
    dev = {name: dev_state(ref[name]) for name, _, _, _ in cfgs}
    tiles = [(x, y, min(x + 16, W), min(y + 16, H)) for y in range(0, H, 16) for x in range(0, W, 16)]
    for it in range(2):
        order = rng.permutation(len(tiles))
        counts = rng.choice([0, 1, 3, 7, 12], size=len(tiles))
        bounds, offsets, off = [], [], 0
        blocks = {name: [] for name, _, _, _ in cfgs}
        for k in order:
            x0, y0, x1, y1 = tiles[k]
            S, npx = int(counts[k]), (x1 - x0) * (y1 - y0)
            bounds.append((x0, y0, x1, y1))
            offsets.append(off)
            size = (S * npx + 3) // 4 * 4                      # blocks start on 4-pixel-sample boundaries
            for name, c, transform, mm in cfgs:
                smp = rng.lognormal(0, 1.5, size=(S, y1 - y0, x1 - x0, c)).astype(np.float32)
                smp[rng.random(smp.shape) < 0.15] = 0.0
                blk = np.zeros(size * c, np.float32)
                blk[:smp.size] = smp.ravel()
                blocks[name].append(blk)
                if S:
                    sub = {key: np.ascontiguousarray(v[y0:y1, x0:x1]) for key, v in ref[name].items()}
                    oracle.accumulate(sub, smp, transform, mm)
                    for key, v in sub.items():
                        ref[name][key][y0:y1, x0:x1] = v
            off += size
        arenas = {name: to_dev(np.concatenate(blocks[name])) for name, _, _, _ in cfgs}
        sts = [gpu.make_stat_type_arena(arenas[name], c, dev[name], transform, mm) for name, c, transform, mm in cfgs]
        gpu.accumulate_tiles(W, H, sts, to_dev(np.array(bounds, np.int32)), to_dev(np.array(offsets, np.int64)),
                             to_dev(counts[order].astype(np.int32)))
        torch.cuda.synchronize()
    for name, c, transform, mm in cfgs:
        got = {k: v.cpu().numpy() for k, v in dev[name].items()}
        assert np.array_equal(got["n"], ref[name]["n"]), name
        exact = ("mean", "m2", "m3")[:mm] if not transform else ("film_mean", "film_m2")
        for k in exact:
            assert np.array_equal(got[k], ref[name][k]), (name, k)
        if transform:
            for k in ("mean", "m2", "m3"):
                assert rel_l2(got[k], ref[name][k]) <= TOL, (name, k)


def test_accumulate_tiles_errors_and_empty(gpu):
    st = dev_state({"n": np.zeros((16, 16), np.int32), "mean": np.zeros((16, 16, 3), np.float32)})
    arena = torch.zeros(16, device=DEV)
    t = gpu.make_stat_type_arena(arena, 3, st, False, 1)
    z = torch.zeros(0, 4, dtype=torch.int32, device=DEV)
    gpu.accumulate_tiles(16, 16, [t], z, torch.zeros(0, dtype=torch.int64, device=DEV), torch.zeros(0, dtype=torch.int32, device=DEV))
    t.max_moment = 2                                               # m2 missing
    b = torch.tensor([[0, 0, 16, 16]], dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError):
        gpu.accumulate_tiles(16, 16, [t], b, torch.zeros(1, dtype=torch.int64, device=DEV), torch.ones(1, dtype=torch.int32, device=DEV))


def test_accumulate_all_types_one_launch(gpu, oracle):
    """The 11-channel sample vector: radiance, normal, albedo, depth, material id in one launch."""
    from statmc_amd import film, synthetic
    W, H, S = 52, 19, 6
    scene, smp, ref = make_case(W, H, S, seed=4, features=synthetic.FEATURES)
    fs = film.FilmStats(W, H, DEV, types=synthetic.FEATURES)
    fs.accumulate({k: to_dev(v) for k, v in smp.items()})
    torch.cuda.synchronize()
    for t in synthetic.FEATURES:
        assert np.array_equal(fs.state[t]["n"].cpu().numpy(), ref[t]["n"])
        if t == "radiance":
            assert rel_l2(fs.state[t]["mean"].cpu().numpy(), ref[t]["mean"]) <= TOL
            assert np.array_equal(fs.state[t]["film_mean"].cpu().numpy(), ref[t]["film_mean"])
        else:
            assert np.array_equal(fs.state[t]["mean"].cpu().numpy(), ref[t]["mean"]), t


def test_accumulate_resident_grid(gpu, oracle):
    """The resident-workgroup launch shape (few workgroups walking every stat type) gives the same
    images as the default interleaved grid."""
    from statmc_amd import film, synthetic
    W, H, S = 61, 33, 7                                               # odd sample count: pipelined loop + remainder
    scene, smp, ref = make_case(W, H, S, seed=9, features=synthetic.FEATURES)
    gpu.accumulate_resident_blocks(3)
    try:
        fs = film.FilmStats(W, H, DEV, types=synthetic.FEATURES)
        fs.accumulate({k: to_dev(v) for k, v in smp.items()})
        torch.cuda.synchronize()
    finally:
        gpu.accumulate_resident_blocks(0)
    for t in synthetic.FEATURES:
        assert np.array_equal(fs.state[t]["n"].cpu().numpy(), ref[t]["n"])
        if t == "radiance":
            assert rel_l2(fs.state[t]["m3"].cpu().numpy(), ref[t]["m3"]) <= TOL
            assert np.array_equal(fs.state[t]["film_m2"].cpu().numpy(), ref[t]["film_m2"])
        else:
            assert np.array_equal(fs.state[t]["mean"].cpu().numpy(), ref[t]["mean"]), t


@pytest.mark.parametrize("transform,max_moment", [(False, 1), (True, 3), (False, 3)])
def test_accumulate_lds_dma_walk(gpu, oracle, transform, max_moment):
    """The LDS-DMA walk of the RGB types (rows of a wave's samples land in a ring, counted waits) at its edges: batches
    shorter than, equal to and longer than the ring, waves whose last lanes have no pixel group (the transfers of a row
    are counted, so every row issues all of them), a ragged last group, uniform and ragged counts -- against the oracle
    and, bit for bit, against the walk with loads into registers."""
    rng = np.random.default_rng(31 + max_moment)
    for W, H in ((4, 1), (36, 5), (260, 3), (1028, 2), (254, 7)):         # 1, 45, 195, 514 and 444.5 four-pixel groups
        for S, ragged in ((0, False), (1, False), (2, True), (3, False), (4, False), (5, True), (7, False), (16, True)):
            smp = rng.lognormal(0, 1, size=(S, H, W, 3)).astype(np.float32)
            smp[rng.random(smp.shape) < 0.2] = 0.0
            ref = oracle.new_state(H, W, 3)
            if ragged:
                n0 = rng.integers(0, 9, size=(H, W)).astype(np.int32)
                ref["n"][...] = n0
                for k in ("mean", "m2", "m3", "film_mean", "film_m2"):
                    ref[k][...] = (rng.random(ref[k].shape) * (n0[..., None] > 0)).astype(np.float32)
                if not transform:
                    ref["film_mean"][...] = ref["mean"]
                    ref["film_m2"][...] = ref["m2"]
            st_dma, st_reg = dev_state(ref), dev_state(ref)
            oracle.accumulate(ref, smp, transform, max_moment)
            d_smp = to_dev(smp)
            try:
                gpu.accumulate_dma(1)
                gpu.accumulate(W, H, [gpu.make_stat_type(d_smp, st_dma, transform, max_moment)])
                gpu.accumulate_dma(0)
                gpu.accumulate(W, H, [gpu.make_stat_type(d_smp, st_reg, transform, max_moment)])
                torch.cuda.synchronize()
            finally:
                gpu.accumulate_dma(1)
            what = (W, H, S, ragged)
            for k in st_dma:
                assert torch.equal(st_dma[k].view(torch.int32), st_reg[k].view(torch.int32)), (what, k)
            got = {k: v.cpu().numpy() for k, v in st_dma.items()}
            assert np.array_equal(got["n"], ref["n"]), what
            keys = ("mean", "m2", "m3")[:max_moment]
            if transform:
                assert np.array_equal(got["film_mean"], ref["film_mean"]), what
                assert np.array_equal(got["film_m2"], ref["film_m2"]), what
                for k in keys:
                    assert rel_l2(got[k], ref[k]) <= TOL, (what, k)
            else:
                for k in keys:
                    assert np.array_equal(got[k], ref[k]), (what, k)


@pytest.mark.parametrize("W,H,cuts", [(36, 13, (0, 5, 6, 13)), (260, 9, (0, 2, 9)), (37, 6, (0, 3, 6))], ids=["36x13", "260x9", "37x6-scalar"])
def test_accumulate_rows_same_bits(gpu, oracle, W, H, cuts):
    """statmc_accumulate_rows: a batch folded in over any split of the film into row ranges (the multi-GPU step does the
    rows next to a neighbour first) leaves the bits of the one-launch call -- every stat type of the shipped set in one
    launch, row offsets that keep and that break the 16-byte alignment of the vector / LDS-DMA paths."""
    from statmc_amd import film, synthetic
    S = 5
    scene, smp, ref = make_case(W, H, S, seed=19, features=synthetic.FEATURES)
    dsmp = {k: to_dev(v) for k, v in smp.items()}
    whole = film.FilmStats(W, H, DEV, types=synthetic.FEATURES)
    whole.accumulate(dsmp)
    split = film.FilmStats(W, H, DEV, types=synthetic.FEATURES)
    ranges = list(zip(cuts[:-1], cuts[1:]))
    if len(ranges) >= 3:                                                 # first and last range in ONE launch, then the middle
        split.accumulate(dsmp, rows=[ranges[-1], ranges[0]])
        ranges = ranges[1:-1]
    for y0, y1 in reversed(ranges):                                      # any order
        split.accumulate(dsmp, rows=(y0, y1))
    torch.cuda.synchronize()
    for t in synthetic.FEATURES:
        for k, v in whole.state[t].items():
            if v is not None:
                assert torch.equal(v.view(torch.int32), split.state[t][k].view(torch.int32)), (t, k)
        assert np.array_equal(split.state[t]["n"].cpu().numpy(), ref[t]["n"]), t
    with pytest.raises(gpu.StatmcError):
        split.accumulate(dsmp, rows=(2, H + 1))


def test_accumulate_empty_and_errors(gpu, oracle):
    st = dev_state(oracle.new_state(4, 4, 3))
    empty = torch.zeros(0, 4, 4, 3, device=DEV)
    gpu.accumulate(4, 4, [gpu.make_stat_type(empty, st, True, 3)])      # zero samples: state untouched
    torch.cuda.synchronize()
    assert not st["n"].any() and not st["mean"].any()
    gpu.accumulate(4, 4, [])                                            # no stat types: no-op
    bad = gpu.make_stat_type(torch.zeros(1, 4, 4, 3, device=DEV), st, True, 3)
    bad.max_moment = 4
    with pytest.raises(gpu.StatmcError) as e:
        gpu.accumulate(4, 4, [bad])
    assert e.value.code == gpu.ERR_INVALID
    bad = gpu.make_stat_type(torch.zeros(1, 4, 4, 3, device=DEV), st, True, 3)
    bad.film_mean = None                                                # transform types need the film images
    with pytest.raises(gpu.StatmcError):
        gpu.accumulate(4, 4, [bad])


# ------------------------------------------------------------------ merge tiles / mean vars / tile moments
@pytest.mark.parametrize("channels,transform", [(1, False), (3, True), (3, False)])
def test_merge_tiles_matches_oracle(gpu, oracle, channels, transform):
    rng = np.random.default_rng(3)
    W, H, ts = 41, 27, 16                                               # 16x16 tiles, ragged right/bottom
    dt = oracle.TILE_PIXEL_DTYPE[channels]
    bounds, offsets, tiles = [], [], []
    off = 0
    for y0 in range(0, H, ts):
        for x0 in range(0, W, ts):
            x1, y1 = min(x0 + ts, W), min(y0 + ts, H)
            npx = (x1 - x0) * (y1 - y0)
            t = np.zeros(npx, dtype=dt)
            t["n"] = rng.integers(1, 1000, npx)
            for k in ("mean", "m2", "m3", "film_mean", "film_m2"):
                t[k] = rng.standard_normal(t[k].shape).astype(np.float32)
            tiles.append(t)
            bounds.append((x0, y0, x1, y1))
            offsets.append(off)
            off += npx
    ref = oracle.new_state(H, W, channels)
    for k in ref:
        ref[k][...] = -7                                               # MergeTile leaves film images alone
    st = dev_state(ref)
    for t, b in zip(tiles, bounds):
        oracle.merge_tile(t, channels, *b, ref, transform=transform)
    allpx = np.zeros(off, dtype=dt)          # np.concatenate would repack the padded struct dtype
    for t, o in zip(tiles, offsets):
        allpx[o:o + len(t)] = t
    assert allpx.strides[0] == (64 if channels == 1 else 128)
    raw = torch.from_numpy(allpx.view(np.uint8)).to(DEV)
    gpu.merge_tiles(W, H, channels, transform, raw, to_dev(np.array(bounds, np.int32)),
                    to_dev(np.array(offsets, np.int64)), ts * ts, st)
    torch.cuda.synchronize()
    for k in ref:
        assert np.array_equal(st[k].cpu().numpy(), ref[k]), k


@pytest.mark.parametrize("channels", [1, 3])
@pytest.mark.parametrize("quirk", [True, False])
def test_mean_vars_matches_oracle(gpu, oracle, channels, quirk):
    rng = np.random.default_rng(8)
    H, W = 9, 31
    n = rng.integers(2, 50, (H, W)).astype(np.int32)                   # non-uniform n pins the per-row quirk
    m2 = rng.random((H, W, channels), dtype=np.float32)
    out = torch.zeros(H, W, channels, device=DEV)
    gpu.calculate_mean_vars([to_dev(n)], [to_dev(m2)], [out], row_n_quirk=quirk)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), oracle.mean_vars(n, m2, row_n_quirk=quirk))


@pytest.mark.parametrize("tile_size", [8, 16])
def test_tile_moments(gpu, oracle, tile_size):
    """Wave-level Welford + Chan merges against the oracle's lane-by-lane restatement of the same tree (bit for bit),
    and both against float64 NumPy."""
    rng = np.random.default_rng(1)
    H, W, Cn = 37, 50, 3
    v = rng.lognormal(0, 1, (H, W, Cn)).astype(np.float32)
    ty, tx = -(-H // tile_size), -(-W // tile_size)
    out = torch.zeros(ty, tx, Cn, 3, device=DEV)
    gpu.tile_moments(to_dev(v), tile_size, out)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    assert np.array_equal(out, oracle.tile_moments(v, tile_size))
    for j in range(ty):
        for i in range(tx):
            blk = v[j * tile_size:(j + 1) * tile_size, i * tile_size:(i + 1) * tile_size].astype(np.float64)
            cnt = blk.shape[0] * blk.shape[1]
            assert np.allclose(out[j, i, :, 0], cnt)
            assert np.allclose(out[j, i, :, 1], blk.mean((0, 1)), rtol=1e-5)
            assert np.allclose(out[j, i, :, 2], ((blk - blk.mean((0, 1))) ** 2).sum((0, 1)), rtol=1e-4)


# ------------------------------------------------------------------ pre-pass
@pytest.mark.parametrize("channels", [1, 3])
def test_prepass_bit_exact(gpu, oracle, channels):
    rng = np.random.default_rng(12)
    H, W = 17, 29
    n = rng.integers(0, 40, (H, W)).astype(np.int32)                   # includes n = 0, 1 (discriminator = inf)
    n[0, :5] = [0, 1, 2, 4097, 100000]                                  # table edge and beyond
    mean = rng.standard_normal((H, W, channels)).astype(np.float32)
    m2 = rng.random((H, W, channels), dtype=np.float32)
    m2[1, :4] = 0.0                                                     # zero variance
    m3 = rng.standard_normal((H, W, channels)).astype(np.float32)
    mean[2, 3] = np.nan                                                 # negative sample upstream
    mc_ref, d_ref = oracle.prepass(n, mean, m2, m3)
    mc, d = torch.zeros(H, W, channels, device=DEV), torch.zeros(H, W, channels, device=DEV)
    dummy = torch.zeros(H, W, channels, device=DEV)
    a, keep = gpu.make_filter_args([to_dev(n)], [to_dev(mean)], [to_dev(m2)], [to_dev(m3)], [dummy], [mc], [d],
                                   [dummy.clone()], [], g_sds=[], radius=1)
    gpu.prepass(a, channels)
    torch.cuda.synchronize()
    assert np.array_equal(mc.cpu().numpy(), mc_ref, equal_nan=True)
    assert np.array_equal(d.cpu().numpy(), d_ref, equal_nan=True)


@pytest.mark.parametrize("W,H,S,spec", [(64, 20, 5, {}), (272, 24, 1, {}), (50, 37, 3, {}), (128, 32, 9, dict(sides=1, small_n=1)), (128, 16, 4, dict(dof=1))],
                         ids=["vector", "one-sample", "scalar-path", "two-sided-exclude", "welch"])
def test_prepass_fused_into_the_accumulation_is_bit_exact(gpu, oracle, W, H, S, spec):
    """statmc_stat_type::mean_corr / discriminator (round 6): the accumulation's epilogue writes the pre-pass of the moments it has just
    updated -- from the registers that hold them, the same prepass_elem -- so the images must equal, bit for bit, what statmc_prepass
    computes afterwards from the stored moments, and the oracle's pre-pass; film-major launches, a second batch on top, row ranges,
    the tile-fed entry, under the spec options that change the pre-pass (quantile sides, n < 2, Welch)."""
    from statmc_amd import film, synthetic
    types = ("radiance", "normal", "albedo")
    scene = synthetic.Scene(W, H, seed=21)
    gpu.set_filter_spec(**spec)
    try:
        fs_f = film.FilmStats(W, H, DEV, types=types, radius=3, fused_prepass=True)
        fs_u = film.FilmStats(W, H, DEV, types=types, radius=3)
        for b, seed in enumerate((5, 6)):
            smp = {k: v.to(DEV) for k, v in scene.samples(S, seed=seed, features=types).items()}
            if b == 1:                      # a pixel with a negative sample (NaN through the Box-Cox root), one far above the rest
                smp["radiance"][0, 1, 2, 0] = -1.0
                smp["radiance"][0, 2, 3, 1] = 1e6
            fs_f.accumulate(smp)
            fs_u.accumulate(smp)
            assert fs_f._prepass_current is not None
            mc_f, d_f = fs_f.mean_corr.clone(), fs_f.disc.clone()
            fs_u.prepass()
            torch.cuda.synchronize()
            for k in ("n", "mean", "m2", "m3", "film_mean", "film_m2"):      # the epilogue leaves the moments alone
                assert torch.equal(fs_f.state["radiance"][k].view(torch.int32), fs_u.state["radiance"][k].view(torch.int32)), k
            assert np.array_equal(mc_f.cpu().numpy().view(np.int32), fs_u.mean_corr.cpu().numpy().view(np.int32))
            assert np.array_equal(d_f.cpu().numpy().view(np.int32), fs_u.disc.cpu().numpy().view(np.int32))
            rad = {k: v.cpu().numpy() for k, v in fs_u.state["radiance"].items() if v is not None}
            ospec = oracle.default_spec()
            for k, v in spec.items():
                setattr(ospec, k, v)
            mc_ref, d_ref = oracle.prepass(rad["n"], rad["mean"], rad["m2"], rad["m3"], spec=ospec)
            assert np.array_equal(mc_f.cpu().numpy(), mc_ref, equal_nan=True) and np.array_equal(d_f.cpu().numpy(), d_ref, equal_nan=True)
        # prepass() has nothing to launch while the epilogue's result is current ... and launches again once the spec changes
        before = fs_f.mean_corr.clone()
        fs_f.mean_corr.fill_(7.0)
        fs_f.prepass()
        assert float(fs_f.mean_corr[0, 0, 0].item()) == 7.0
        gpu.set_filter_spec(**dict(spec, small_n=0 if spec.get("small_n") else 1))
        fs_f.prepass()
        torch.cuda.synchronize()
        assert float(fs_f.mean_corr[0, 0, 0].item()) != 7.0 and fs_f._prepass_current is None
        gpu.set_filter_spec(**spec)
        # row ranges through the C ABI: the epilogue writes exactly the rows of the launch
        if H >= 16:
            rad = fs_u.state["radiance"]
            mc2, d2 = torch.full_like(fs_u.mean_corr, 3.0), torch.full_like(fs_u.disc, 3.0)
            smp = {k: v.to(DEV) for k, v in scene.samples(2, seed=9, features=types).items()}
            st = gpu.make_stat_type(smp["radiance"], rad, True, 3, prepass_into=(mc2, d2))
            gpu.accumulate(W, H, [st], rows=[(2, 5), (H - 4, H - 1)])
            fs_u.prepass()
            torch.cuda.synchronize()
            for y0, y1 in ((2, 5), (H - 4, H - 1)):
                assert torch.equal(mc2[y0:y1].view(torch.int32), fs_u.mean_corr[y0:y1].view(torch.int32))
                assert torch.equal(d2[y0:y1].view(torch.int32), fs_u.disc[y0:y1].view(torch.int32))
            assert float(mc2[0].min().item()) == 3.0 and float(mc2[6:H - 4].max().item()) == 3.0 and float(d2[H - 1].min().item()) == 3.0
        # the tile-fed entry (16 x 16 tiles, ragged at the film's edges)
        fs_t = film.FilmStats(W, H, DEV, types=("radiance",), radius=3, g_buffers=())
        fs_v = film.FilmStats(W, H, DEV, types=("radiance",), radius=3, g_buffers=())
        smp = scene.samples(S, seed=11, features=("radiance",))["radiance"]
        tiles = [(x, y, min(x + 16, W), min(y + 16, H)) for y in range(0, H, 16) for x in range(0, W, 16)]
        blocks, offs, pos = [], [], 0
        for x0, y0, x1, y1 in tiles:
            blocks.append(smp[:, y0:y1, x0:x1].contiguous().reshape(-1))
            offs.append(pos)
            pos += (x1 - x0) * (y1 - y0) * S
            pos = (pos + 3) // 4 * 4
        arena = torch.zeros(pos * 3, dtype=torch.float32)
        for blk, o in zip(blocks, offs):
            arena[o * 3:o * 3 + blk.numel()] = blk
        arena = arena.to(DEV)
        st = gpu.make_stat_type_arena(arena, 3, fs_t.state["radiance"], True, 3, prepass_into=(fs_t.mean_corr, fs_t.disc))
        gpu.accumulate_tiles(W, H, [st], torch.tensor(tiles, dtype=torch.int32, device=DEV), torch.tensor(offs, dtype=torch.int64, device=DEV),
                             torch.full((len(tiles),), S, dtype=torch.int32, device=DEV))
        fs_v.accumulate({"radiance": smp.to(DEV)})
        fs_v.prepass()
        torch.cuda.synchronize()
        assert torch.equal(fs_t.state["radiance"]["m3"].view(torch.int32), fs_v.state["radiance"]["m3"].view(torch.int32))
        assert torch.equal(fs_t.mean_corr.view(torch.int32), fs_v.mean_corr.view(torch.int32)) and torch.equal(fs_t.disc.view(torch.int32), fs_v.disc.view(torch.int32))
        # both or neither, and only with three moments
        lib = gpu.load()
        bad = gpu.make_stat_type(smp.to(DEV), fs_v.state["radiance"], True, 3)
        bad.mean_corr = fs_v.mean_corr.data_ptr()
        assert lib.statmc_accumulate(W, H, C.byref(bad), 1, None) == gpu.ERR_INVALID
        bad = gpu.make_stat_type(smp.to(DEV), fs_v.state["radiance"], True, 2, prepass_into=(fs_v.mean_corr, fs_v.disc))
        assert lib.statmc_accumulate(W, H, C.byref(bad), 1, None) == gpu.ERR_INVALID
    finally:
        gpu.set_filter_spec()


def test_significance_levels(gpu, oracle):
    n = np.full((4, 8), 12, np.int32)
    rng = np.random.default_rng(0)
    mean, m2, m3 = (rng.random((4, 8, 3), dtype=np.float32) for _ in range(3))
    outs = []
    try:
        for idx in (0, 1, 2):
            gpu.check(gpu.load().statmc_set_significance(idx))
            mc, d = torch.zeros(4, 8, 3, device=DEV), torch.zeros(4, 8, 3, device=DEV)
            dummy = torch.zeros(4, 8, 3, device=DEV)
            a, keep = gpu.make_filter_args([to_dev(n)], [to_dev(mean)], [to_dev(m2)], [to_dev(m3)], [dummy], [mc],
                                           [d], [dummy.clone()], [], g_sds=[], radius=1)
            gpu.prepass(a, 3)
            torch.cuda.synchronize()
            assert np.array_equal(d.cpu().numpy(), oracle.prepass(n, mean, m2, m3, alpha_index=idx)[1])
            outs.append(d.cpu().numpy())
        assert (outs[1] > outs[0]).all() and (outs[0] > outs[2]).all()   # 0.002 > 0.005 > 0.05 quantiles
        assert gpu.load().statmc_set_significance(3) == gpu.ERR_INVALID
    finally:
        gpu.load().statmc_set_significance(0)


def test_custom_t_quantile_table(gpu, oracle):
    """A user-supplied quantile table (e.g. the reference's own array) replaces a built-in one."""
    from scipy import stats
    rng = np.random.default_rng(5)
    n = rng.integers(2, 300, (6, 9)).astype(np.int32)
    mean, m2, m3 = (rng.random((6, 9, 3), dtype=np.float32) for _ in range(3))
    table = stats.t.ppf(1 - 0.01, np.arange(1, 201)).astype(np.float32)       # one-sided 1 %, only 200 dof
    lib = gpu.load()
    try:
        gpu.check(lib.statmc_set_t_quantiles(2, table.ctypes.data_as(C.POINTER(C.c_float)), len(table)))
        gpu.check(lib.statmc_set_significance(2))
        oracle.set_t_quantiles(2, table)
        mc, d = torch.zeros(6, 9, 3, device=DEV), torch.zeros(6, 9, 3, device=DEV)
        dummy = torch.zeros(6, 9, 3, device=DEV)
        a, keep = gpu.make_filter_args([to_dev(n)], [to_dev(mean)], [to_dev(m2)], [to_dev(m3)], [dummy], [mc], [d],
                                       [dummy.clone()], [], g_sds=[], radius=1)
        gpu.prepass(a, 3)
        torch.cuda.synchronize()
        want = oracle.prepass(n, mean, m2, m3, alpha_index=2)[1]
        assert np.array_equal(d.cpu().numpy(), want)
        assert oracle.t_quantile(2, 250) == table[-1]                           # beyond the table: last entry
        assert lib.statmc_set_t_quantiles(6, table.ctypes.data_as(C.POINTER(C.c_float)), 10) == gpu.ERR_INVALID
        # a repeated setup of the device keeps its loaded table and its significance level (per-device state)
        gpu.check(lib.statmc_setup(0))
        assert lib.statmc_get_significance() == 2
        d.zero_()
        gpu.prepass(a, 3)
        torch.cuda.synchronize()
        assert np.array_equal(d.cpu().numpy(), want)
        # ... and the same table serves a pre-pass on a second stream
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            d2 = torch.zeros(6, 9, 3, device=DEV)
            a2, keep2 = gpu.make_filter_args([to_dev(n)], [to_dev(mean)], [to_dev(m2)], [to_dev(m3)], [dummy], [mc.clone()], [d2],
                                             [dummy.clone()], [], g_sds=[], radius=1)
            gpu.prepass(a2, 3)
        torch.cuda.synchronize()
        assert np.array_equal(d2.cpu().numpy(), want)
    finally:
        oracle.set_t_quantiles(2, None)
        gpu.check(lib.statmc_set_t_quantiles(2, None, 0))   # back to the built-in table
        gpu.check(lib.statmc_set_significance(0))
    d.zero_()
    gpu.prepass(a, 3)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy(), oracle.prepass(n, mean, m2, m3, alpha_index=0)[1])
    assert oracle.t_quantile(2, 30) < oracle.t_quantile(0, 30)


# ------------------------------------------------------------------ window filter
def run_filter(gpu, mc, disc, colour, gbs, g_dr, filter_sd, radius, roi=None, channels=3, force=0, n=None):
    out = torch.zeros_like(to_dev(colour))
    a, keep = gpu.make_filter_args([to_dev(n)] if n is not None else [], [], [], [], [to_dev(colour)], [to_dev(mc)], [to_dev(disc)], [out],
                                   [to_dev(g) for g in gbs], g_dr=g_dr, filter_sd=filter_sd, radius=radius, roi=roi)
    gpu.force_filter_variant(force)
    try:
        gpu.window_filter(a, channels)
        torch.cuda.synchronize()
    finally:
        gpu.force_filter_variant(0)
    return out.cpu().numpy(), gpu.last_filter_variant()


def stats_case(oracle, W, H, spp, seed):
    _, smp, st = make_case(W, H, spp, seed=seed)
    rad = st["radiance"]
    mc, disc = oracle.prepass(rad["n"], rad["mean"], rad["m2"], rad["m3"])
    return mc, disc, rad["film_mean"], [st["normal"]["mean"], st["albedo"]["mean"]]


@pytest.mark.parametrize("W,H,radius,sd,force,variant", [
    (300, 41, 20, 10.0, 0, "sym_r20"),     # shipped default; 2 tile columns, ragged right edge, 11 tile rows
    (300, 41, 20, 10.0, 3, "lds_r20"),     # the one-sided r = 20 kernel
    (300, 41, 20, 10.0, 2, "lds_rt"),
    (300, 41, 20, 10.0, 1, "generic"),
    (37, 21, 6, 3.0, 0, "sym_rt"),         # glass-caustics config (filterradius 6); image smaller than one tile
    (37, 21, 6, 3.0, 2, "lds_rt"),         # ... on the one-sided runtime-radius kernel
    (420, 70, 6, 3.0, 0, "sym_rt"),        # ... four tile columns, nine tile rows, LDS-DMA staging
    (259, 9, 3, 2.0, 0, "sym_rt"),         # radius not a multiple of 4, width 4k+3 (register staging)
    (64, 50, 1, 1.0, 0, "sym_rt"),
    (300, 41, 19, 9.0, 0, "sym_rt"),       # every read group in play, the outermost ones cut by the table
    (300, 41, 16, 8.0, 0, "sym_rt"),       # j_lo = 1: group 0 skipped
    (300, 41, 13, 6.0, 0, "sym_rt"),
    (132, 30, 7, 4.0, 0, "sym_rt"),
    (45, 33, 24, 12.0, 0, "generic"),      # radius beyond the LDS kernel's range
])
def test_filter_matches_oracle(gpu, oracle, W, H, radius, sd, force, variant):
    mc, disc, colour, gbs = stats_case(oracle, W, H, 8, seed=W + radius)
    ref = oracle.filter_image(mc, disc, colour, gbs, G_DR, -0.5 / sd ** 2, radius)
    out, v = run_filter(gpu, mc, disc, colour, gbs, G_DR, sd, radius, force=force)
    assert v == variant
    for c in range(3):
        assert rel_l2(out[..., c], ref[..., c]) <= TOL, c


@pytest.mark.parametrize("parts", [1, 2, 3, 7, 41, 64])
def test_filter_window_sweep_parts(gpu, oracle, parts):
    """The LDS kernel splits the window rows over `parts` workgroups per tile (load balance);
    partial sums are combined by a second kernel.  Any split must agree with the oracle."""
    mc, disc, colour, gbs = stats_case(oracle, 300, 41, 8, seed=320)
    ref = oracle.filter_image(mc, disc, colour, gbs, G_DR, -0.5 / FILTER_SD ** 2, RADIUS)
    gpu.force_filter_parts(parts)
    try:
        out, v = run_filter(gpu, mc, disc, colour, gbs, G_DR, FILTER_SD, RADIUS)
    finally:
        gpu.force_filter_parts(0)
    assert v == "sym_r20"
    assert max(rel_l2(out[..., c], ref[..., c]) for c in range(3)) <= TOL


def test_filter_low_spp_high_rejection(gpu, oracle):
    """4 spp: wide confidence intervals at some pixels, heavy rejection at edges."""
    mc, disc, colour, gbs = stats_case(oracle, 280, 30, 4, seed=77)
    ref = oracle.filter_image(mc, disc, colour, gbs, G_DR, -0.5 / FILTER_SD ** 2, RADIUS)
    out, v = run_filter(gpu, mc, disc, colour, gbs, G_DR, FILTER_SD, RADIUS)
    assert v == "sym_r20"
    assert max(rel_l2(out[..., c], ref[..., c]) for c in range(3)) <= TOL


def test_filter_special_pixels(gpu, oracle):
    """n < 2 (infinite discriminator), zero variance, NaN statistics (negative sample upstream),
    and a pixel with no member at all (falls back to its own colour)."""
    mc, disc, colour, gbs = stats_case(oracle, 270, 26, 8, seed=5)
    disc[3, 10] = np.inf
    disc[4, 100:140] = 0.0
    mc[7, 200] = np.nan
    mc[9, 50] = 1e6
    disc[9, 50] = 0.0                                                   # rejects everyone but itself
    mc[12, 60] = np.nan
    disc[12, 60] = np.nan                                               # not even a member of itself
    ref = oracle.filter_image(mc, disc, colour, gbs, G_DR, -0.5 / FILTER_SD ** 2, RADIUS)
    assert np.array_equal(ref[12, 60], colour[12, 60])
    for force in (0, 3, 2, 1):
        out, v = run_filter(gpu, mc, disc, colour, gbs, G_DR, FILTER_SD, RADIUS, force=force)
        assert np.isfinite(out).all(), v
        assert np.array_equal(out[12, 60], colour[12, 60]), v
        assert max(rel_l2(out[..., c], ref[..., c]) for c in range(3)) <= TOL, v


def test_filter_non_finite_corrected_mean(gpu, oracle):
    """A pixel whose corrected mean is +-inf (all channels or one) takes no part -- in particular it
    must not slip into the window of a neighbour with fewer than two samples (discriminator +inf),
    where the oracle's `inf <= inf` and the kernel's max3 form would disagree.  Found by
    tools/experiments/fuzz_gpu.py (tiny images, +inf mean next to +inf discriminator)."""
    for W, H, radius in ((2, 1, 1), (6, 2, 8), (270, 12, 20), (33, 9, 5)):
        rng = np.random.default_rng(W * 7 + H)
        mc = rng.standard_normal((H, W, 3)).astype(np.float32)
        disc = (rng.random((H, W, 3)) * 2).astype(np.float32)
        colour = rng.random((H, W, 3), dtype=np.float32) * 3
        gbs = [rng.random((H, W, 3), dtype=np.float32), rng.random((H, W, 3), dtype=np.float32)]
        mc[0, 0] = np.inf
        disc[0, W - 1] = np.inf                                        # a neighbour that accepts everything finite
        if W > 8:
            mc[H // 2, 5, 1] = -np.inf                                  # one channel only
            disc[H // 2, 7] = np.inf
            mc[H - 1, 3] = np.inf
            disc[H - 1, 3] = np.inf                                     # both at the same pixel
        g_dr = [-0.5 / 0.3 ** 2, -0.5 / 0.2 ** 2]
        ref = oracle.filter_image(mc, disc, colour, gbs, g_dr, -0.5 / FILTER_SD ** 2, radius)
        assert np.array_equal(ref[0, 0], colour[0, 0])                  # filters nothing: its own colour
        for force in (0, 3, 2, 1):
            out, v = run_filter(gpu, mc, disc, colour, gbs, g_dr, FILTER_SD, radius, force=force)
            assert np.isfinite(out).all(), v
            assert np.array_equal(out[0, 0], colour[0, 0]), v
            assert max(rel_l2(out[..., c], ref[..., c]) for c in range(3)) <= TOL, (v, W, H)
    # filter<float>: validity is per buffer
    W, H = 40, 6
    rng = np.random.default_rng(3)
    mcs = [rng.standard_normal((H, W)).astype(np.float32) for _ in range(3)]
    dcs = [(rng.random((H, W)) * 2).astype(np.float32) for _ in range(3)]
    cols = [rng.random((H, W), dtype=np.float32) for _ in range(3)]
    gbs = [rng.random((H, W, 3), dtype=np.float32), rng.random((H, W, 3), dtype=np.float32)]
    mcs[1][2, 10] = np.inf
    dcs[1][2, 12] = np.inf
    dcs[0][2, 12] = np.inf
    g_dr = [-0.5 / 0.3 ** 2, -0.5 / 0.2 ** 2]
    outs = [torch.zeros(H, W, device=DEV) for _ in range(3)]
    a, keep = gpu.make_filter_args([], [], [], [], [to_dev(c) for c in cols], [to_dev(m) for m in mcs], [to_dev(d) for d in dcs],
                                   outs, [to_dev(g) for g in gbs], g_dr=g_dr, filter_sd=FILTER_SD, radius=20)
    gpu.window_filter(a, 1)
    torch.cuda.synchronize()
    for b in range(3):
        ref = oracle.filter_image(mcs[b], dcs[b], cols[b], gbs, g_dr, -0.5 / FILTER_SD ** 2, 20)
        assert rel_l2(outs[b].cpu().numpy(), ref) <= TOL, b


def test_filters_on_two_streams(gpu, oracle):
    """Two window filters in flight on two streams (different images): the per-part partial sums
    live in a workspace per (device, stream), so neither call sees the other's."""
    cases = [stats_case(oracle, 600, 48, 8, seed=21), stats_case(oracle, 600, 48, 8, seed=22)]
    refs = [oracle.filter_image(mc, dc, col, gbs, G_DR, -0.5 / FILTER_SD ** 2, RADIUS) for mc, dc, col, gbs in cases]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs, keeps = [], []
    for rep in range(3):
        outs = []
        for (mc, dc, col, gbs), st in zip(cases, streams):
            with torch.cuda.stream(st):
                out = torch.zeros(48, 600, 3, device=DEV)
                a, keep = gpu.make_filter_args([], [], [], [], [to_dev(col)], [to_dev(mc)], [to_dev(dc)], [out],
                                               [to_dev(g) for g in gbs], g_dr=G_DR, filter_sd=FILTER_SD, radius=RADIUS)
                gpu.force_filter_parts(3)
                gpu.window_filter(a, 3)
                outs.append(out)
                keeps.append((a, keep))
        torch.cuda.synchronize()
        for out, ref in zip(outs, refs):
            assert max(rel_l2(out.cpu().numpy()[..., c], ref[..., c]) for c in range(3)) <= TOL
    gpu.force_filter_parts(0)


def test_filter_roi(gpu, oracle):
    """The multi-GPU block path: outputs only inside the ROI, window clipped to the local image."""
    mc, disc, colour, gbs = stats_case(oracle, 330, 60, 8, seed=6)
    roi = (20, 20, 310, 40)
    ref = oracle.filter_image(mc, disc, colour, gbs, G_DR, -0.5 / FILTER_SD ** 2, RADIUS, roi=roi)
    for force in (0, 1):
        out, v = run_filter(gpu, mc, disc, colour, gbs, G_DR, FILTER_SD, RADIUS, roi=roi, force=force)
        assert not out[:20].any() and not out[40:].any() and not out[:, :20].any() and not out[:, 310:].any()
        assert max(rel_l2(out[..., c], ref[..., c]) for c in range(3)) <= TOL, v


@pytest.mark.parametrize("channels", [1, 3])
def test_filter_generic_gbuffer_sets(gpu, oracle, channels):
    """G-buffer sets with float images (depth / material id): up to six channels in total are spread
    over the LDS kernel's six feature slots; more than that goes through the generic kernel."""
    rng = np.random.default_rng(21)
    H, W, r = 30, 300, 7
    mc = rng.random((H, W, channels), dtype=np.float32)
    disc = (0.1 * rng.random((H, W, channels))).astype(np.float32)
    colour = rng.random((H, W, channels), dtype=np.float32)
    gbs = [rng.random((H, W, 3), dtype=np.float32), rng.random((H, W, 1), dtype=np.float32),
           rng.integers(0, 3, (H, W, 1)).astype(np.float32)]
    g_dr = [-0.5 / 0.3 ** 2, -0.5 / 0.2 ** 2, -0.5 / 0.1 ** 2]
    ref = oracle.filter_image(mc, disc, colour, gbs, g_dr, -0.5 / 16.0, r)
    out, v = run_filter(gpu, mc, disc, colour, gbs, g_dr, 4.0, r, channels=channels)
    # 3 + 1 + 1 channels at r = 7: since round 4 the pair-symmetric kernel's eight-plane build has a runtime radius ...
    assert v == ("sym_rt_g8" if channels == 3 else "sym_rt_f_g8")
    assert rel_l2(out, ref) <= TOL
    out, v = run_filter(gpu, mc, disc, colour, gbs, g_dr, 4.0, r, channels=channels, force=2)
    assert v == ("lds_rt" if channels == 3 else "lds_rt_f")           # ... and the one-sided kernel still spreads the set over its six slots
    assert rel_l2(out, ref) <= TOL
    out, v = run_filter(gpu, mc, disc, colour, gbs, g_dr, 4.0, r, channels=channels, force=1)
    assert v == "generic" and rel_l2(out, ref) <= TOL
    # float image first, r = 20 instantiation
    gbs2, dr2 = [gbs[1], gbs[0]], [g_dr[1], g_dr[0]]
    ref2 = oracle.filter_image(mc, disc, colour, gbs2, dr2, -0.5 / 100.0, 20)
    out2, v2 = run_filter(gpu, mc, disc, colour, gbs2, dr2, 10.0, 20, channels=channels)
    # at r = 20 a set with 1-channel images runs the pair-symmetric kernel's eight-feature-plane build
    assert v2 == ("sym_r20_g8" if channels == 3 else "sym_r20_f_g8") and rel_l2(out2, ref2) <= TOL
    out2b, v2b = run_filter(gpu, mc, disc, colour, gbs2, dr2, 10.0, 20, channels=channels, force=3)
    assert v2b == ("lds_r20" if channels == 3 else "lds_r20_f") and rel_l2(out2b, ref2) <= TOL  # slot layout of the one-sided kernel
    # seven channels: two RGB + one 1-channel image is an eight-plane set of the pair-symmetric kernel (runtime radius since round 4)
    gbs7 = [gbs[0], rng.random((H, W, 3), dtype=np.float32), gbs[1]]
    ref7 = oracle.filter_image(mc, disc, colour, gbs7, g_dr, -0.5 / 16.0, r)
    out7, v7 = run_filter(gpu, mc, disc, colour, gbs7, g_dr, 4.0, r, channels=channels)
    assert v7 == ("sym_rt_g8" if channels == 3 else "sym_rt_f_g8") and rel_l2(out7, ref7) <= TOL
    out7g, v7g = run_filter(gpu, mc, disc, colour, gbs7, g_dr, 4.0, r, channels=channels, force=2)   # ... which the one-sided kernel has no slots for
    assert v7g == "generic" and rel_l2(out7g, ref7) <= TOL
    # three RGB images (nine channels) fit nowhere
    gbs9 = [gbs[0], gbs7[1], rng.random((H, W, 3), dtype=np.float32)]
    ref9 = oracle.filter_image(mc, disc, colour, gbs9, g_dr, -0.5 / 16.0, r)
    out9, v9 = run_filter(gpu, mc, disc, colour, gbs9, g_dr, 4.0, r, channels=channels)
    assert v9 == "generic" and rel_l2(out9, ref9) <= TOL
    # no G-buffers at all
    ref0 = oracle.filter_image(mc, disc, colour, [], [], -0.5 / 16.0, r)
    out0, _ = run_filter(gpu, mc, disc, colour, [], [], 4.0, r, channels=channels)
    assert rel_l2(out0, ref0) <= TOL


@pytest.mark.parametrize("channels,order,spec_kw,W,radius", [
    (3, ("normal", "albedo", "depth", "materialid"), dict(), 300, 20),
    (3, ("depth", "albedo", "materialid", "normal"), dict(), 300, 20),           # any order of the argument list
    (3, ("albedo", "normal", "depth"), dict(), 300, 20),                          # one 1-channel image
    (3, ("normal", "albedo", "depth", "materialid"), dict(), 301, 20),            # width not a multiple of 4: register staging
    (3, ("normal", "albedo", "depth", "materialid"), dict(border=1), 300, 20),
    (3, ("normal", "albedo", "depth", "materialid"), dict(gate=1, channel_rule=1), 300, 20),
    (1, ("normal", "albedo", "depth", "materialid"), dict(), 300, 20),            # filter<float>, two buffers per launch
    (3, ("materialid", "depth", "normal", "albedo"), dict(), 300, 6),             # the shipped small radius (glass-caustics), runtime-radius build
    (3, ("normal", "albedo", "depth", "materialid"), dict(), 301, 13),
    (3, ("normal", "albedo", "depth", "materialid"), dict(channel_rule=1, border=1), 300, 7),
    (1, ("normal", "albedo", "depth", "materialid"), dict(), 300, 3),
], ids=["nadm", "danm-order", "three", "unaligned", "clamp", "asym+joint", "float", "r6", "r13-unaligned", "r7-joint+clamp", "r3-float"])
def test_filter_eight_feature_channels(gpu, oracle, channels, order, spec_kw, W, radius):
    """normal + albedo + depth + material id as G-buffers (statpath.cpp:828-835, 1096-1130: `filterbuffers` may name all
    four): eight feature channels run the pair-symmetric kernel's eight-plane build -- compile-time radius 20 or, since
    round 4, the runtime-radius build (r = 6 is scenes/render-denoise-glass-caustics.pbrt:19-20) -- not the global-memory
    kernel; same result as the oracle and as the general kernel."""
    RADIUS = radius
    H = 44
    feats = ("radiance", "normal", "albedo", "depth", "materialid")
    _, smp, st = make_case(W, H, 6, seed=321, features=feats)
    rad = st["radiance"]
    pick = (lambda a: a) if channels == 3 else (lambda a: np.ascontiguousarray(a[..., :1]))
    ospec = oracle.FilterSpec(**spec_kw)
    mc, disc = oracle.prepass(rad["n"], pick(rad["mean"]), pick(rad["m2"]), pick(rad["m3"]), spec=ospec)
    colour = pick(rad["film_mean"])
    sds = dict(normal=SD_NORMAL, albedo=SD_ALBEDO, depth=2.0, materialid=0.5)
    gbs = [st[g]["mean"] for g in order]
    g_dr = [-0.5 / sds[g] ** 2 for g in order]
    ref = oracle.filter_image(mc, disc, colour, gbs, g_dr, -0.5 / FILTER_SD ** 2, RADIUS, spec=ospec)
    gpu.set_filter_spec(**spec_kw)
    try:
        out, v = run_filter(gpu, mc, disc, colour, gbs, g_dr, FILTER_SD, RADIUS, channels=channels)
        out_g, v_g = run_filter(gpu, mc, disc, colour, gbs, g_dr, FILTER_SD, RADIUS, channels=channels, force=1)
    finally:
        gpu.set_filter_spec()
    want = ("sym_r20" if radius == 20 else "sym_rt") + ("_f" if channels == 1 else "") + "_g8" + ("_asym" if spec_kw.get("gate") else "") + \
        ("_joint" if spec_kw.get("channel_rule") and channels == 3 else "") + ("_clamp" if spec_kw.get("border") else "")
    assert v == want, v
    assert v_g == "generic"
    for c in range(channels):
        assert rel_l2(out[..., c], ref[..., c]) <= TOL, c
        assert rel_l2(out_g[..., c], ref[..., c]) <= TOL, c


@pytest.mark.parametrize("channels,joint,radius,border,order,jump", [
    (3, 0, 20, 0, ("normal", "albedo", "depth"), False), (3, 1, 20, 0, ("depth", "normal", "materialid", "albedo"), True),
    (3, 0, 6, 1, ("materialid",), False), (1, 0, 20, 0, ("normal", "albedo", "depth", "materialid"), True), (1, 0, 7, 0, ("depth", "albedo"), False),
    (3, 0, 20, 0, ("albedo", "materialid", "depth"), "ones")],
    ids=["rgb", "rgb-joint-far-items", "rgb-r6-clamp-one-plane", "float-far-items", "float-r7", "rgb-counts-of-one-and-two"])
def test_filter_welch_with_one_channel_gbuffers(gpu, oracle, channels, joint, radius, border, order, jump):
    """Welch degrees of freedom x 1-channel G-buffers (depth / material id, statpath.cpp:828-835): the eight-plane Welch builds of
    the pair-symmetric kernel.  Beside eight feature planes the CU's LDS has room for one more plane per staged row, so the
    ring holds n - 1 and every tap forms E = v * v / (n - 1) itself, with the oracle's own rounding (an IEEE division) --
    against the oracle and the general kernel, with sample counts that jump inside a tile (work items that leave the quantile
    band and are computed again from the whole table) and with counts of 1 and 2 (E = x / 0: inf or NaN, as in the oracle).
    On block + halo images the product travels in the 18-channel layout (tests/test_peer_gpu.py, test_multirank_gpu.py)."""
    from statmc_amd import sharding
    feats = ("radiance", "normal", "albedo", "depth", "materialid")
    W, H = 280, 30
    _, smp, st = make_case(W, H, 5, seed=77 + radius + channels, features=feats)
    rad = st["radiance"]
    if jump == "ones":
        rad["n"][:, 100:180] = 1
        rad["n"][:, 180:230] = 2
    elif jump:
        rad["n"][:, 150:] = 2600
    pick = (lambda a: a) if channels == 3 else (lambda a: np.ascontiguousarray(a[..., :1]))
    spec_kw = dict(dof=1, channel_rule=joint, border=border)
    ospec = oracle.FilterSpec(**spec_kw)
    mc, disc = oracle.prepass(rad["n"], pick(rad["mean"]), pick(rad["m2"]), pick(rad["m3"]), spec=ospec)
    colour = pick(rad["film_mean"])
    sds = dict(normal=SD_NORMAL, albedo=SD_ALBEDO, depth=2.0, materialid=0.5)
    gbs = [st[g]["mean"] for g in order]
    g_dr = [-0.5 / sds[g] ** 2 for g in order]
    sd = radius / 2.0
    ref = oracle.filter_image(mc, disc, colour, gbs, g_dr, -0.5 / sd ** 2, radius, spec=ospec, n=rad["n"])
    lib = gpu.load()
    lib.statmc_debug_welch_far_items.restype = C.c_int
    gpu.set_filter_spec(**spec_kw)
    try:
        out, v = run_filter(gpu, mc, disc, colour, gbs, g_dr, sd, radius, channels=channels, n=rad["n"])
        far = lib.statmc_debug_welch_far_items()
        out_g, v_g = run_filter(gpu, mc, disc, colour, gbs, g_dr, sd, radius, channels=channels, n=rad["n"], force=1)
    finally:
        gpu.set_filter_spec()
    assert v == "sym_welch" + ("_f" if channels == 1 else "") + "_g8" + ("_joint" if joint and channels == 3 else "") + ("_clamp" if border else ""), v
    assert v_g == "generic"
    assert (far > 0) == (jump is True), far
    finite = np.isfinite(ref)
    assert np.array_equal(np.isfinite(out), finite)
    for c in range(channels):
        assert rel_l2(np.where(finite, out, 0)[..., c], np.where(finite, ref, 0)[..., c]) <= TOL, c
        assert rel_l2(np.where(finite, out_g, 0)[..., c], np.where(finite, ref, 0)[..., c]) <= TOL, c
    assert sharding.block_image_channels([3, 3, 1], welch=True) == 18


@pytest.mark.parametrize("channels", [1, 3])
@pytest.mark.parametrize("n_g", [0, 1])
def test_filter_fewer_gbuffers_on_the_lds_kernel(gpu, oracle, channels, n_g):
    """`filterbuffers ["albedo"]` or none at all: the LDS kernel is written for two RGB G-buffers and
    runs with an absent one as a slot of factor 0 that is never read; same result as the generic
    kernel and the oracle."""
    rng = np.random.default_rng(100 + 10 * channels + n_g)
    H, W, r = 40, 300, 20
    mc = rng.standard_normal((H, W, channels)).astype(np.float32)
    disc = ((rng.random((H, W, channels)) ** 2) * 2).astype(np.float32)
    colour = rng.random((H, W, channels), dtype=np.float32) * 4
    gbs = [rng.random((H, W, 3), dtype=np.float32)][:n_g]
    g_dr = [-0.5 / 0.2 ** 2][:n_g]
    ref = oracle.filter_image(mc, disc, colour, gbs, g_dr, -0.5 / FILTER_SD ** 2, r)
    out, v = run_filter(gpu, mc, disc, colour, gbs, g_dr, FILTER_SD, r, channels=channels)
    assert v == ("sym_r20" if channels == 3 else "sym_r20_f")
    for c in range(channels):
        assert rel_l2(out[..., c], ref[..., c]) <= TOL
    out_1, v_1 = run_filter(gpu, mc, disc, colour, gbs, g_dr, FILTER_SD, r, channels=channels, force=3)   # one-sided LDS kernel
    assert v_1 == ("lds_r20" if channels == 3 else "lds_r20_f")
    for c in range(channels):
        assert rel_l2(out_1[..., c], ref[..., c]) <= TOL
    out_g, v_g = run_filter(gpu, mc, disc, colour, gbs, g_dr, FILTER_SD, r, channels=channels, force=1)
    assert v_g == "generic"
    for c in range(channels):
        assert rel_l2(out_g[..., c], ref[..., c]) <= TOL


@pytest.mark.parametrize("n_buffers,radius,variant", [(5, 20, "sym_r20_f"), (12, 20, "sym_r20_f"), (4, 7, "sym_rt_f"), (1, 20, "sym_r20_f"),
                                                      (2, 20, "sym_r20_f")])
def test_filter_float_multibuffer_fast_path(gpu, oracle, n_buffers, radius, variant):
    """filter<float> as ACRR (nBuffers = trackedbounces = 5) and SMIS (2 x 6 = 12) call it
    (estimator.cpp:437-459): 1-channel buffers with their own statistics and colour, shared RGB
    G-buffers; the pair-symmetric kernel takes them two per launch (r = 20), the one-sided LDS kernel three."""
    W, H = 276, 27
    _, smp, st = make_case(W, H, 8, seed=50 + n_buffers)
    rng = np.random.default_rng(n_buffers)
    gbs = [st["normal"]["mean"], st["albedo"]["mean"]]
    lum = smp["radiance"].mean(axis=3, keepdims=True)            # luminance-like scalar samples
    refs, args = [], dict(n=[], mean=[], m2=[], m3=[], film=[], mean_corr=[], disc=[], film_filtered=[])
    for b in range(n_buffers):
        s = oracle.new_state(H, W, 1)
        scale = np.float32(1.0 / (1 + b))                           # deeper bounces carry less energy
        oracle.accumulate(s, np.ascontiguousarray(lum * scale + rng.random(lum.shape, dtype=np.float32) * 0.01), True, 3)
        mc, dc = oracle.prepass(s["n"], s["mean"], s["m2"], s["m3"])
        refs.append(oracle.filter_image(mc, dc, s["film_mean"], gbs, G_DR, -0.5 / FILTER_SD ** 2, radius))
        for k, v in (("n", s["n"]), ("mean", s["mean"]), ("m2", s["m2"]), ("m3", s["m3"]), ("film", s["film_mean"])):
            args[k].append(to_dev(v))
        for k in ("mean_corr", "disc", "film_filtered"):
            args[k].append(torch.zeros(H, W, 1, device=DEV))
    a, keep = gpu.make_filter_args(g_buffers=[to_dev(g) for g in gbs], g_sds=[SD_NORMAL, SD_ALBEDO],
                                   filter_sd=FILTER_SD, radius=radius, **args)
    gpu.filter_f32(a)
    torch.cuda.synchronize()
    assert gpu.last_filter_variant() == variant
    for b in range(n_buffers):
        assert rel_l2(args["film_filtered"][b].cpu().numpy(), refs[b]) <= TOL, b
    # same call through the generic kernel and through the one-sided LDS kernel
    for force, name in ((1, "generic"), (3 if radius == 20 else 2, "lds_r20_f" if radius == 20 else "lds_rt_f")):
        gpu.force_filter_variant(force)
        try:
            for t in args["film_filtered"]:
                t.zero_()
            gpu.filter_f32(a)
            torch.cuda.synchronize()
        finally:
            gpu.force_filter_variant(0)
        assert gpu.last_filter_variant() == name
        for b in range(n_buffers):
            assert rel_l2(args["film_filtered"][b].cpu().numpy(), refs[b]) <= TOL, (name, b)


@pytest.mark.parametrize("n_buffers,variant", [(5, "sym_r20_f+lds_r20_f"), (3, "lds_r20_f"), (7, "sym_r20_f+lds_r20_f"), (4, "sym_r20_f"), (1, "sym_r20_f")])
def test_filter_float_odd_buffer_counts_end_on_the_one_sided_kernel(gpu, oracle, n_buffers, variant):
    """ACRR's five float buffers (estimator.cpp:434-460): an odd count would end with a pair-symmetric launch that carries one
    buffer at the price of two; where the one-sided kernel sweeps the window in one part (films of 720p and up; here: the split
    pinned to 1) the last THREE buffers go to it -- it shares the range weight over three.  Every buffer within TOL of the oracle
    whichever kernel took it; an even count and a single buffer stay on the pair-symmetric kernel."""
    W, H = 300, 31
    _, smp, st = make_case(W, H, 8, seed=70 + n_buffers)
    rng = np.random.default_rng(700 + n_buffers)
    gbs = [st["normal"]["mean"], st["albedo"]["mean"]]
    lum = smp["radiance"].mean(axis=3, keepdims=True)
    refs, args = [], dict(n=[], mean=[], m2=[], m3=[], film=[], mean_corr=[], disc=[], film_filtered=[])
    for b in range(n_buffers):
        s = oracle.new_state(H, W, 1)
        oracle.accumulate(s, np.ascontiguousarray(lum * np.float32(1.0 / (1 + b)) + rng.random(lum.shape, dtype=np.float32) * 0.01), True, 3)
        mc, dc = oracle.prepass(s["n"], s["mean"], s["m2"], s["m3"])
        refs.append(oracle.filter_image(mc, dc, s["film_mean"], gbs, G_DR, -0.5 / FILTER_SD ** 2, 20))
        for k, v in (("n", s["n"]), ("mean", s["mean"]), ("m2", s["m2"]), ("m3", s["m3"]), ("film", s["film_mean"])):
            args[k].append(to_dev(v))
        for k in ("mean_corr", "disc", "film_filtered"):
            args[k].append(torch.zeros(H, W, 1, device=DEV))
    a, keep = gpu.make_filter_args(g_buffers=[to_dev(g) for g in gbs], g_sds=[SD_NORMAL, SD_ALBEDO], filter_sd=FILTER_SD, radius=20, **args)
    gpu.set_filter_split(1)
    try:
        gpu.filter_f32(a)
        torch.cuda.synchronize()
        v = gpu.last_filter_variant()
    finally:
        gpu.set_filter_split(0)
    assert v == variant
    for b in range(n_buffers):
        assert rel_l2(args["film_filtered"][b].cpu().numpy(), refs[b]) <= TOL, b


@pytest.mark.parametrize("radius,border,g8", [(20, 0, False), (6, 1, False), (20, 0, True), (9, 1, True)], ids=["r20", "r6-clamp", "r20-eight-planes", "r9-clamp-eight-planes"])
def test_filter_float_multibuffer_welch(gpu, oracle, radius, border, g8):
    """filter<float> under Welch degrees of freedom: three buffers with DIFFERENT sample counts (5, 9 and 2 400 samples; two
    buffers share a launch of the pair-symmetric kernel's Welch build, each with its own counts, its own test and its own
    band entries -- 5 next to 2 400 leaves the band: those work items take the far build), against the oracle per buffer
    and against the general kernel."""
    W, H = 276, 27
    _, smp, st = make_case(W, H, 9, seed=61, features=("radiance", "normal", "albedo", "depth", "materialid") if g8 else ("radiance", "normal", "albedo"))
    gbs = [st["normal"]["mean"], st["albedo"]["mean"]]
    g_drs, g_sds = list(G_DR), [SD_NORMAL, SD_ALBEDO]
    if g8:      # + depth and material id: the eight-plane Welch build, the second buffer's n - 1 in a plane of its own
        gbs += [st["depth"]["mean"], st["materialid"]["mean"]]
        g_sds += [2.0, 0.5]
        g_drs += [-0.5 / 2.0 ** 2, -0.5 / 0.5 ** 2]
    lum = smp["radiance"].mean(axis=3, keepdims=True)
    spec = oracle.FilterSpec(dof=1, border=border)
    refs, args = [], dict(n=[], mean=[], m2=[], m3=[], film=[], mean_corr=[], disc=[], film_filtered=[])
    for b, spp in enumerate((5, 9, 9)):
        s = oracle.new_state(H, W, 1)
        oracle.accumulate(s, np.ascontiguousarray(lum[:spp] * np.float32(1.0 / (1 + b))), True, 3)
        if b == 2:
            s["n"][...] = 2400          # (statistics of 9 samples under a count of 2 400: the filter only sees numbers)
        mc, dc = oracle.prepass(s["n"], s["mean"], s["m2"], s["m3"], spec=spec)
        refs.append(oracle.filter_image(mc, dc, s["film_mean"], gbs, g_drs, -0.5 / (radius / 2.0) ** 2, radius, spec=spec, n=s["n"]))
        for k, v in (("n", s["n"]), ("mean", s["mean"]), ("m2", s["m2"]), ("m3", s["m3"]), ("film", s["film_mean"])):
            args[k].append(to_dev(v))
        for k in ("mean_corr", "disc", "film_filtered"):
            args[k].append(torch.zeros(H, W, 1, device=DEV))
    # order: (5, 2 400) share the first launch, 9 the second
    for k in args:
        args[k] = [args[k][0], args[k][2], args[k][1]]
    refs = [refs[0], refs[2], refs[1]]
    a, keep = gpu.make_filter_args(g_buffers=[to_dev(g) for g in gbs], g_sds=g_sds,
                                   filter_sd=radius / 2.0, radius=radius, **args)
    lib = gpu.load()
    lib.statmc_debug_welch_far_items.restype = C.c_int
    gpu.set_filter_spec(dof=1, border=border)
    try:
        gpu.filter_f32(a)
        torch.cuda.synchronize()
        assert gpu.last_filter_variant() == "sym_welch_f" + ("_g8" if g8 else "") + ("_clamp" if border else "")
        assert lib.statmc_debug_welch_far_items() == 0          # (the last launch: the 9-sample buffer alone)
        got = [t.cpu().numpy() for t in args["film_filtered"]]
        gpu.force_filter_variant(1)
        for t in args["film_filtered"]:
            t.zero_()
        gpu.filter_f32(a)
        torch.cuda.synchronize()
        assert gpu.last_filter_variant() == "generic"
    finally:
        gpu.force_filter_variant(0)
        gpu.set_filter_spec()
    for b in range(3):
        assert rel_l2(got[b], refs[b]) <= TOL, b
        assert rel_l2(args["film_filtered"][b].cpu().numpy(), refs[b]) <= TOL, ("generic", b)


def test_filter_entry_point_reference_argument_order(gpu, oracle):
    """statmc_filter_f32x3 with the reference's argument block: nBuffers = 2, denoiseFilm set:
    buffer 0 filters `film` into `film-f`, buffer 1 filters its own film-mean (estimator.cpp:465-487)."""
    W, H = 264, 20
    _, smp, st = make_case(W, H, 8, seed=31)
    rad = st["radiance"]
    film_img = (rad["film_mean"] * 1.5).astype(np.float32)            # the "film" image differs from t0 film-mean
    d = {k: to_dev(v) for k, v in rad.items()}
    mc = [torch.zeros(H, W, 3, device=DEV) for _ in range(2)]
    dc = [torch.zeros(H, W, 3, device=DEV) for _ in range(2)]
    ff = [torch.zeros(H, W, 3, device=DEV) for _ in range(2)]
    film_f = torch.zeros(H, W, 3, device=DEV)
    gb = [to_dev(st["normal"]["mean"]), to_dev(st["albedo"]["mean"])]
    a, keep = gpu.make_filter_args([d["n"]] * 2, [d["mean"]] * 2, [d["m2"]] * 2, [d["m3"]] * 2,
                                   [d["film_mean"]] * 2, mc, dc, ff, gb, g_sds=[SD_NORMAL, SD_ALBEDO],
                                   filter_sd=FILTER_SD, radius=RADIUS, denoise_film=True,
                                   film_buffer=to_dev(film_img), film_filtered_buffer=film_f)
    gpu.filter_f32x3(a)
    torch.cuda.synchronize()
    mcr, dr = oracle.prepass(rad["n"], rad["mean"], rad["m2"], rad["m3"])
    gbn = [st["normal"]["mean"], st["albedo"]["mean"]]
    ref0 = oracle.filter_image(mcr, dr, film_img, gbn, G_DR, -0.5 / FILTER_SD ** 2, RADIUS)
    ref1 = oracle.filter_image(mcr, dr, rad["film_mean"], gbn, G_DR, -0.5 / FILTER_SD ** 2, RADIUS)
    assert np.array_equal(mc[1].cpu().numpy(), mcr) and np.array_equal(dc[0].cpu().numpy(), dr)
    assert rel_l2(film_f.cpu().numpy(), ref0) <= TOL
    assert rel_l2(ff[1].cpu().numpy(), ref1) <= TOL
    assert not ff[0].any()                                             # buffer 0 wrote film-f, not t0-b0-film-mean-f


def test_maximum_width_and_empty_calls(gpu, oracle):
    """The reference passes width/height as unsigned short (estimator.h:316-317): the widest film is
    65535 columns.  Also: zero buffers is a no-op, an empty image is an error."""
    W, H, S = 65535, 3, 4
    rng = np.random.default_rng(65535)
    smp = rng.lognormal(0, 1, size=(S, H, W, 3)).astype(np.float32)
    gb = [rng.random((H, W, 3), dtype=np.float32), rng.random((H, W, 3), dtype=np.float32)]
    ref = oracle.new_state(H, W, 3)
    oracle.accumulate(ref, smp, True, 3)
    st = dev_state(oracle.new_state(H, W, 3))
    gpu.accumulate(W, H, [gpu.make_stat_type(to_dev(smp), st, True, 3)])
    torch.cuda.synchronize()
    assert np.array_equal(st["n"].cpu().numpy(), ref["n"]) and np.array_equal(st["film_mean"].cpu().numpy(), ref["film_mean"])
    mc, dc = oracle.prepass(ref["n"], ref["mean"], ref["m2"], ref["m3"])
    g_dr = [-0.5 / 0.5 ** 2, -0.5 / 0.4 ** 2]
    want = oracle.filter_image(mc, dc, ref["film_mean"], gb, g_dr, -0.5 / 25.0, 8)
    for force, variant in ((0, "sym_rt"), (2, "lds_rt"), (1, "generic")):
        out, v = run_filter(gpu, mc, dc, ref["film_mean"], gb, g_dr, 5.0, 8, force=force)
        assert v == variant
        assert max(rel_l2(out[..., c], want[..., c]) for c in range(3)) <= TOL, v
    # zero buffers: nothing to do, no error; empty image: invalid
    z = torch.zeros(4, 4, 3, device=DEV)
    a, keep = gpu.make_filter_args([], [], [], [], [z], [z.clone()], [z.clone()], [z.clone()], [], g_sds=[], radius=2)
    a.n_buffers = 0
    gpu.filter_f32x3(a)
    a.n_buffers, a.width = 1, 0
    with pytest.raises(gpu.StatmcError) as e:
        gpu.window_filter(a, 3)
    assert e.value.code == gpu.ERR_INVALID


def test_packed_inputs_path(gpu, oracle):
    """statmc_pack_filter_inputs + packed_inputs (the multi-GPU block + halo image): pack a block
    into the middle of a larger 15-channel image whose margins hold the true neighbouring pixels,
    filter with the ROI = block; must equal the separate-image filter of the larger image."""
    W, H, m = 300, 56, 20                                             # larger image; block = interior minus a margin of 20
    mc, disc, colour, gbs = stats_case(oracle, W, H, 8, seed=404)
    ref = oracle.filter_image(mc, disc, colour, gbs, G_DR, -0.5 / FILTER_SD ** 2, RADIUS, roi=(m, m, W - m, H - m))
    full15 = np.concatenate([mc, disc, colour, gbs[0], gbs[1]], axis=2)
    packed = to_dev(full15).clone()
    packed[m:H - m, m:W - m] = float("nan")                           # the pack kernel must fill exactly this
    blk = lambda a: to_dev(np.ascontiguousarray(a[m:H - m, m:W - m]))
    dummy = torch.zeros(H - 2 * m, W - 2 * m, 3, device=DEV)
    a, keep = gpu.make_filter_args([], [], [], [], [blk(colour)], [blk(mc)], [blk(disc)], [dummy],
                                   [blk(gbs[0]), blk(gbs[1])], g_sds=[SD_NORMAL, SD_ALBEDO], radius=RADIUS)
    gpu.pack_filter_inputs(a, packed, m, m)
    torch.cuda.synchronize()
    assert np.array_equal(packed.cpu().numpy(), full15)
    out = torch.zeros(H, W, 3, device=DEV)
    a2, keep2 = gpu.make_filter_args([], [], [], [], [], [], [], [out], [], g_sds=[SD_NORMAL, SD_ALBEDO],
                                     filter_sd=FILTER_SD, radius=RADIUS, roi=(m, m, W - m, H - m), packed=packed)
    gpu.window_filter(a2, 3)
    torch.cuda.synchronize()
    assert gpu.last_filter_variant() == "sym_r20"
    got = out.cpu().numpy()
    assert max(rel_l2(got[..., c], ref[..., c]) for c in range(3)) <= TOL
    assert not got[:m].any() and not got[:, :m].any()
    # the packed path is T = float3 only, and the block must fit
    with pytest.raises(gpu.StatmcError) as e:
        gpu.window_filter(a2, 1)
    assert e.value.code == gpu.ERR_UNSUPPORTED
    with pytest.raises(gpu.StatmcError) as e:
        gpu.pack_filter_inputs(a, packed, W - 10, 0)
    assert e.value.code == gpu.ERR_INVALID


@pytest.mark.parametrize("spec_kw,variant", [(dict(border=1), "sym_r20_clamp"), (dict(border=1, gate=1), "sym_r20_asym_clamp"), (dict(gate=1), "sym_r20_asym")])
def test_packed_inputs_under_non_default_specs(gpu, oracle, spec_kw, variant):
    """Block + halo image at r = 20 under a clamped border (this combination once selected the pair-symmetric
    kernel while its border pass could only read the five separate images a packed call does not have; rounds 2 - 4 sent it to
    the one-sided LDS kernel; since round 5 the border pass reads the packed image, and a block is filtered by the kernel
    that filters the whole film): the result equals the oracle's on the same local image under the same spec."""
    W, H, m = 300, 56, 20
    mc, disc, colour, gbs = stats_case(oracle, W, H, 8, seed=405)
    ospec = oracle.FilterSpec(**spec_kw)
    roi = (m, 0, W - m, H)                                           # ROI touches the top and bottom image border
    ref = oracle.filter_image(mc, disc, colour, gbs, G_DR, -0.5 / FILTER_SD ** 2, RADIUS, roi=roi, spec=ospec)
    packed = to_dev(np.concatenate([mc, disc, colour, gbs[0], gbs[1]], axis=2))
    out = torch.zeros(H, W, 3, device=DEV)
    a, keep = gpu.make_filter_args([], [], [], [], [], [], [], [out], [], g_sds=[SD_NORMAL, SD_ALBEDO],
                                   filter_sd=FILTER_SD, radius=RADIUS, roi=roi, packed=packed)
    gpu.set_filter_spec(**spec_kw)
    try:
        gpu.window_filter(a, 3)
        torch.cuda.synchronize()
        assert gpu.last_filter_variant() == variant, gpu.last_filter_variant()
    finally:
        gpu.set_filter_spec()
    got = out.cpu().numpy()
    x0, y0, x1, y1 = roi
    for c in range(3):
        assert rel_l2(got[y0:y1, x0:x1, c], ref[y0:y1, x0:x1, c]) <= TOL, c


@pytest.mark.parametrize("radius,sd,variant,origin", [(6, 3.0, "sym_rt", (128, 40)), (6, 3.0, "sym_rt", (122, 34)), (13, 6.0, "sym_rt", (256, 8))])
def test_packed_inputs_at_small_radii(gpu, oracle, radius, sd, variant, origin):
    """The block + halo image of the multi-GPU path at the glass-caustics radius: the pair-symmetric kernel's
    runtime-radius build stages from the 15-channel image (LDS-DMA) like the r = 20 build."""
    W, H, m = 300, 56, radius
    mc, disc, colour, gbs = stats_case(oracle, W, H, 8, seed=406)
    roi = (m, m, W - m, H - m)
    ref = oracle.filter_image(mc, disc, colour, gbs, G_DR, -0.5 / sd ** 2, radius, roi=roi)
    packed = to_dev(np.concatenate([mc, disc, colour, gbs[0], gbs[1]], axis=2))
    out = torch.zeros(H, W, 3, device=DEV)
    a, keep = gpu.make_filter_args([], [], [], [], [], [], [], [out], [], g_sds=[SD_NORMAL, SD_ALBEDO],
                                   filter_sd=sd, radius=radius, roi=roi, packed=packed, film_origin=origin)
    gpu.window_filter(a, 3)
    torch.cuda.synchronize()
    assert gpu.last_filter_variant() == variant
    got = out.cpu().numpy()
    x0, y0, x1, y1 = roi
    for c in range(3):
        assert rel_l2(got[y0:y1, x0:x1, c], ref[y0:y1, x0:x1, c]) <= TOL, c
    assert not got[:m].any() and not got[:, :m].any()


def test_prepass_pack_equals_prepass_then_pack(gpu, oracle):
    """statmc_prepass_pack (the multi-GPU path's single pass) writes the same bits as statmc_prepass
    followed by statmc_pack_filter_inputs, with and without the mean_corr / discriminator by-products."""
    W, H, mx, my = 116, 34, 20, 7
    _, smp, st = make_case(W, H, 6, seed=77)
    rad = st["radiance"]
    rad["n"][3, 5] = 1                                                 # infinite discriminator
    rad["n"][4, 6] = 0
    rad["m2"][5, 7] = 0.0
    mc_ref, dc_ref = oracle.prepass(rad["n"], rad["mean"], rad["m2"], rad["m3"])
    dev = {k: to_dev(v) for k, v in rad.items()}
    g0, g1 = to_dev(st["normal"]["mean"]), to_dev(st["albedo"]["mean"])
    want = np.full((H + 2 * my, W + 2 * mx, 15), -7.0, np.float32)
    want[my:my + H, mx:mx + W] = np.concatenate([mc_ref, dc_ref, rad["film_mean"], st["normal"]["mean"], st["albedo"]["mean"]], axis=2)
    for with_outputs in (True, False):
        mc, dc = torch.zeros(H, W, 3, device=DEV), torch.zeros(H, W, 3, device=DEV)
        packed = torch.full((H + 2 * my, W + 2 * mx, 15), -7.0, device=DEV)
        a, keep = gpu.make_filter_args([dev["n"]], [dev["mean"]], [dev["m2"]], [dev["m3"]], [dev["film_mean"]],
                                       [mc] if with_outputs else [], [dc] if with_outputs else [], [torch.zeros(H, W, 3, device=DEV)],
                                       [g0, g1], g_sds=[SD_NORMAL, SD_ALBEDO], radius=RADIUS)
        gpu.prepass_pack(a, packed, mx, my)
        torch.cuda.synchronize()
        assert np.array_equal(packed.cpu().numpy(), want, equal_nan=True)
        if with_outputs:
            assert np.array_equal(mc.cpu().numpy(), mc_ref) and np.array_equal(dc.cpu().numpy(), dc_ref)
        else:
            assert not mc.any() and not dc.any()
    with pytest.raises(gpu.StatmcError) as e:
        gpu.prepass_pack(a, packed, W, 0)
    assert e.value.code == gpu.ERR_INVALID
    # statmc_prepass_pack_rows: the block in row ranges -- the two outer strips in one launch (what the multi-GPU step sends
    # first), then the middle -- leaves the bits of the one-launch call, and a range on its own touches nothing else
    mc, dc = torch.zeros(H, W, 3, device=DEV), torch.zeros(H, W, 3, device=DEV)
    packed = torch.full((H + 2 * my, W + 2 * mx, 15), -7.0, device=DEV)
    a, keep = gpu.make_filter_args([dev["n"]], [dev["mean"]], [dev["m2"]], [dev["m3"]], [dev["film_mean"]], [mc], [dc],
                                   [torch.zeros(H, W, 3, device=DEV)], [g0, g1], g_sds=[SD_NORMAL, SD_ALBEDO], radius=RADIUS)
    gpu.prepass_pack(a, packed, mx, my, rows=[(0, 5), (H - 9, H)])
    torch.cuda.synchronize()
    part = packed.cpu().numpy()
    assert np.array_equal(part[my:my + 5], want[my:my + 5], equal_nan=True) and np.array_equal(part[my + H - 9:], want[my + H - 9:], equal_nan=True)
    assert (part[my + 5:my + H - 9] == -7.0).all() and not mc[5:H - 9].any()
    gpu.prepass_pack(a, packed, mx, my, rows=[(5, H - 9)])
    gpu.prepass_pack(a, packed, mx, my, rows=[(3, 3)])                 # an empty range
    torch.cuda.synchronize()
    assert np.array_equal(packed.cpu().numpy(), want, equal_nan=True)
    assert np.array_equal(mc.cpu().numpy(), mc_ref) and np.array_equal(dc.cpu().numpy(), dc_ref)
    with pytest.raises(gpu.StatmcError):
        gpu.prepass_pack(a, packed, mx, my, rows=[(4, 9), (8, 12)])    # overlapping


def test_filter_argument_errors(gpu):
    z = lambda c=3: torch.zeros(8, 8, c, device=DEV)
    a, keep = gpu.make_filter_args([], [], [], [], [z()], [z()], [z()], [z()], [z()], g_sds=[0.1], radius=2)
    a.roi_x1, a.roi_y1 = 20, 4                                         # ROI outside the image
    with pytest.raises(gpu.StatmcError) as e:
        gpu.window_filter(a, 3)
    assert e.value.code == gpu.ERR_INVALID
    img = z()
    a, keep = gpu.make_filter_args([], [], [], [], [img], [z()], [z()], [img], [], g_sds=[], radius=2)
    with pytest.raises(gpu.StatmcError):                               # in-place filtering is refused
        gpu.window_filter(a, 3)
    a, keep = gpu.make_filter_args([], [], [], [], [z()], [z()], [z()], [z()], [], g_sds=[], radius=2)
    a.mean_corr[0].step = 8 * 3 * 4 - 16                               # a row pitch shorter than a row
    with pytest.raises(gpu.StatmcError) as e:
        gpu.window_filter(a, 3)
    assert e.value.code == gpu.ERR_INVALID


@pytest.mark.parametrize("radius,roi", [(20, None), (3, None), (20, (8, 4, 60, 21))])
def test_pitched_device_images(gpu, oracle, radius, roi):
    """Device images with a row pitch (what cv::cuda::GpuMat allocates; buffer.h:25): filter<float3> with the reference's
    two-buffer argument block, filter<float> and mean-vars give the bits of the same calls on packed images, and the
    bytes between the rows are neither read as pixels nor written."""
    W, H = 68, 25
    _, smp, st = make_case(W, H, 8, seed=77)
    rad = st["radiance"]
    rng = np.random.default_rng(5)

    def pitched(t, pad):
        wide = torch.full((t.shape[0], t.shape[1] + pad) + tuple(t.shape[2:]), float("nan"), device=DEV).to(t.dtype)
        if t.dtype == torch.int32:
            wide.fill_(-7)
        wide[:, :t.shape[1]] = t
        return wide[:, :t.shape[1]], wide

    def run(make):
        wides = []

        def m(t, pad):                                                  # the image the call sees: t itself, or a pitched twin of it
            if not make:
                return t
            v, wide = pitched(t, pad)
            wides.append((wide, t.shape[1]))
            return v
        z3 = lambda fill=0.0: torch.full((H, W, 3), fill, device=DEV)
        z1 = lambda fill=0.0: torch.full((H, W), fill, device=DEV)
        d = {k: to_dev(v) for k, v in rad.items()}
        film_img = to_dev((rad["film_mean"] * 1.5).astype(np.float32))
        gb = [to_dev(st["normal"]["mean"]), to_dev(st["albedo"]["mean"])]
        # filter<float3> with the reference's two-buffer block, denoiseFilm set
        mc, dc = [m(z3(), 2) for _ in range(2)], [m(z3(), 5) for _ in range(2)]
        ff, film_f = [m(z3(5.0), 1) for _ in range(2)], m(z3(5.0), 4)
        ins = {k: m(d[k], 3 + i) for i, k in enumerate(("n", "mean", "m2", "m3", "film_mean"))}
        a, keep = gpu.make_filter_args([ins["n"]] * 2, [ins["mean"]] * 2, [ins["m2"]] * 2, [ins["m3"]] * 2, [ins["film_mean"]] * 2,
                                       mc, dc, ff, [m(gb[0], 6), m(gb[1], 7)], g_sds=[SD_NORMAL, SD_ALBEDO], filter_sd=FILTER_SD,
                                       radius=radius, denoise_film=True, film_buffer=m(film_img, 2), film_filtered_buffer=film_f, roi=roi)
        gpu.filter_f32x3(a)
        # filter<float>: the channels of the radiance statistics as three float buffers
        chan = lambda t, c: t[..., c].contiguous()
        fmc, fdc, fff = [m(z1(), 3) for _ in range(3)], [m(z1(), 2) for _ in range(3)], [m(z1(5.0), 1) for _ in range(3)]
        a1, keep1 = gpu.make_filter_args([m(d["n"], 9)] * 3, [m(chan(d["mean"], c), 1 + c) for c in range(3)],
                                         [m(chan(d["m2"], c), 2 + c) for c in range(3)], [m(chan(d["m3"], c), 3 + c) for c in range(3)],
                                         [m(chan(d["film_mean"], c), 4 + c) for c in range(3)], fmc, fdc, fff, [m(gb[0], 6), m(gb[1], 7)],
                                         g_sds=[SD_NORMAL, SD_ALBEDO], filter_sd=FILTER_SD, radius=radius, roi=roi)
        gpu.filter_f32(a1)
        var = m(z3(), 4)
        gpu.calculate_mean_vars([m(d["n"], 2)], [m(d["film_m2"], 3)], [var], row_n_quirk=True)
        torch.cuda.synchronize()
        for wide, w in wides:                                           # the bytes between the rows are as they were
            pad = wide[:, w:]
            assert bool((pad == -7).all()) if wide.dtype == torch.int32 else bool(torch.isnan(pad).all())
        return [t.contiguous().clone() for t in mc + dc + ff + [film_f] + fmc + fdc + fff + [var]]

    packed_res, pitched_res = run(False), run(True)
    for i, (x, y) in enumerate(zip(packed_res, pitched_res)):
        assert torch.equal(x.view(torch.int32), y.view(torch.int32)), i
    film_f = packed_res[6]
    assert bool(film_f.isfinite().all()) and (roi is None) == bool((film_f != 5.0).all())
    mcr, dr = oracle.prepass(rad["n"], rad["mean"], rad["m2"], rad["m3"])
    assert np.array_equal(pitched_res[1].cpu().numpy(), mcr) and np.array_equal(pitched_res[2].cpu().numpy(), dr)


def test_film_update_matches_oracle(gpu, oracle):
    rng = np.random.default_rng(17)
    n = 1237
    px = np.zeros(n, dtype=oracle.FILM_PIXEL_DTYPE)
    px["xyz"] = rng.random((n, 3)) * 4
    px["filter_weight_sum"] = rng.integers(0, 5, n)            # includes zero weights
    px["splat_xyz"] = rng.random((n, 3)) * (rng.random((n, 1)) < 0.1)
    px["pad"] = np.nan                                          # the pad word must not leak
    ref = oracle.film_update(px, splat_scale=0.25, scale=1.5)
    out = torch.zeros(n, 3, device=DEV)
    gpu.film_update(torch.from_numpy(px.view(np.uint8)).to(DEV), n, out, splat_scale=0.25, scale=1.5)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref)


def test_device_memory_and_streams_roundtrip(gpu):
    """statmc_malloc / upload / download / memset / stream_create / synchronize: the GpuMat +
    cv::cuda::Stream roles of Buffer (buffer.h:25,57-63; estimator.h:326)."""
    lib = gpu.load()
    host = np.arange(1000, dtype=np.float32)
    back = np.zeros_like(host)
    dptr, stream = C.c_void_p(), C.c_void_p()
    gpu.check(lib.statmc_malloc(C.byref(dptr), host.nbytes))
    gpu.check(lib.statmc_stream_create(C.byref(stream)))
    try:
        gpu.check(lib.statmc_upload(dptr, host.ctypes.data, host.nbytes, stream))
        gpu.check(lib.statmc_download(back.ctypes.data, dptr, host.nbytes, stream))
        gpu.check(lib.statmc_synchronize(stream))
        assert np.array_equal(back, host)
        gpu.check(lib.statmc_memset(dptr, 0, host.nbytes, stream))
        gpu.check(lib.statmc_download(back.ctypes.data, dptr, host.nbytes, stream))
        gpu.check(lib.statmc_synchronize(stream))
        assert not back.any()
    finally:
        gpu.check(lib.statmc_stream_destroy(stream))
        gpu.check(lib.statmc_free(dptr))
    assert lib.statmc_malloc(None, 16) == gpu.ERR_INVALID
    assert lib.statmc_setup(99) == gpu.ERR_INVALID and b"out of range" in lib.statmc_last_error()


def test_filter_randomised_configurations(gpu, oracle):
    """40 seeded random configurations: image size, radius, filter sds, ROI, window-sweep parts,
    kernel variant -- every one against the oracle."""
    rng = np.random.default_rng(20240607)
    worst = 0.0
    for case in range(40):
        W = int(rng.integers(5, 420))
        H = int(rng.integers(3, 48))
        radius = int(rng.choice([1, 2, 3, 5, 6, 8, 11, 16, 20, 20, 20]))
        sd = float(rng.uniform(1.0, 12.0))
        g_sds = [float(rng.uniform(0.05, 0.5)), float(rng.uniform(0.01, 0.2))]
        g_dr = [-0.5 / s ** 2 for s in g_sds]
        mc = rng.standard_normal((H, W, 3)).astype(np.float32)
        disc = (rng.random((H, W, 3)) ** 3 * 2.0).astype(np.float32)
        if rng.random() < 0.3:
            disc[rng.integers(0, H), rng.integers(0, W)] = np.inf
        if rng.random() < 0.3:
            mc[rng.integers(0, H), rng.integers(0, W), rng.integers(0, 3)] = np.nan      # one NaN channel
        colour = rng.random((H, W, 3), dtype=np.float32) * 3
        gbs = [rng.random((H, W, 3), dtype=np.float32), rng.random((H, W, 3), dtype=np.float32)]
        roi = None
        if rng.random() < 0.5 and W > 8 and H > 4:
            x0, y0 = int(rng.integers(0, W // 2)), int(rng.integers(0, H // 2))
            roi = (x0, y0, int(rng.integers(x0 + 1, W + 1)), int(rng.integers(y0 + 1, H + 1)))
        ref = oracle.filter_image(mc, disc, colour, gbs, g_dr, -0.5 / sd ** 2, radius, roi=roi)
        force = int(rng.choice([0, 0, 2, 1]))
        gpu.force_filter_parts(int(rng.choice([0, 1, 2, 5])))
        try:
            out, v = run_filter(gpu, mc, disc, colour, gbs, g_dr, sd, radius, roi=roi, force=force)
        finally:
            gpu.force_filter_parts(0)
        assert np.isfinite(out).all(), (case, v)
        err = max(rel_l2(out[..., c], ref[..., c]) for c in range(3))
        worst = max(worst, err)
        assert err <= TOL, (case, v, W, H, radius, roi, err)
    assert worst > 0          # the GPU path really computed something different from a copy of the oracle


# ------------------------------------------------------------------ filter spec v2: every open choice, HIP == oracle
import itertools

# three gate forms (symmetric, one-sided, centre interval only = Moon et al. / -DMEMFNC=1) x 2^5
SPEC_VARIANTS = [dict(zip(("gate", "channel_rule", "sides", "dof", "border", "small_n"), v))
                 for v in itertools.product((0, 1, 2), (0, 1), (0, 1), (0, 1), (0, 1), (0, 1))]


def spec_id(v):
    return "".join(str(v[k]) for k in ("gate", "channel_rule", "sides", "dof", "border", "small_n"))


def run_spec(gpu, oracle, st, spec_kw, radius, sd, channels=3, alpha_index=0, force=0):
    """pre-pass + window filter through the C ABI under a spec; returns (mc, disc, out, variant, oracle triple)."""
    rad = st["radiance"]
    n = rad["n"]
    pick = (lambda a: a) if channels == 3 else (lambda a: np.ascontiguousarray(a[..., :1]))
    gbs = [st["normal"]["mean"], st["albedo"]["mean"]]
    ospec = oracle.FilterSpec(**spec_kw)
    omc, odc = oracle.prepass(n, pick(rad["mean"]), pick(rad["m2"]), pick(rad["m3"]), alpha_index=alpha_index, spec=ospec)
    oout = oracle.filter_image(omc, odc, pick(rad["film_mean"]), gbs, G_DR, -0.5 / sd ** 2, radius, spec=ospec, n=n,
                               alpha_index=alpha_index)
    h, w = n.shape
    mc, dc, out = (torch.zeros(h, w, channels, device=DEV) for _ in range(3))
    a, keep = gpu.make_filter_args([to_dev(n)], [to_dev(pick(rad["mean"]))], [to_dev(pick(rad["m2"]))], [to_dev(pick(rad["m3"]))],
                                   [to_dev(pick(rad["film_mean"]))], [mc], [dc], [out], [to_dev(g) for g in gbs],
                                   g_dr=G_DR, filter_sd=sd, radius=radius)
    gpu.set_filter_spec(**spec_kw)
    gpu.check(gpu.load().statmc_set_significance(alpha_index))
    gpu.force_filter_variant(force)
    try:
        (gpu.filter_f32x3 if channels == 3 else gpu.filter_f32)(a)
        torch.cuda.synchronize()
        variant = gpu.last_filter_variant()
    finally:
        gpu.force_filter_variant(0)
        gpu.set_filter_spec()
        gpu.load().statmc_set_significance(0)
    return mc.cpu().numpy(), dc.cpu().numpy(), out.cpu().numpy(), variant, (omc, odc, oout)


def expected_lds_variant(spec_kw, channels, radius):
    """Which kernel serves a spec: a per-pair Welch lookup runs the general kernel; at r = 20 the pair-symmetric kernel
    takes every other spec for an RGB buffer (the clamped border's taps beyond the image come from a second small
    kernel) and the symmetric gate for float buffers; the one-sided LDS kernel takes the rest (the non-default
    membership tests in its runtime-radius build)."""
    gate, joint, border = spec_kw.get("gate", 0), spec_kw.get("channel_rule", 0) and channels == 3, spec_kw.get("border", 0)
    if spec_kw.get("dof", 0):
        # Welch degrees of freedom: the pair-symmetric kernel's Welch builds (one per buffer type for every radius; the gate
        # field has no meaning under Welch) -- an RGB buffer, or two float buffers per launch
        return "sym_welch" + ("_f" if channels == 1 else "") + ("_joint" if joint else "") + ("_clamp" if border else "")
    f = "_f" if channels == 1 else ""
    g = ("", "_asym", "_centre")[gate]
    if channels == 3 or not gate:       # the pair-symmetric kernel: compile-time radius 20, runtime radius below
        return ("sym_r20" if radius == 20 else "sym_rt") + f + g + ("_joint" if joint else "") + ("_clamp" if border else "")
    return "lds_rt" + f + g + ("_joint" if joint else "")


@pytest.mark.parametrize("spec_kw", SPEC_VARIANTS, ids=[spec_id(v) for v in SPEC_VARIANTS])
def test_filter_spec_variants_match_oracle(gpu, oracle, spec_kw):
    """All 96 combinations of the six open choices (gate form, channel rule, quantile sides, dof, border,
    n < 2): pre-pass bit-exact, window filter <= 1e-5 per channel, on a 3-spp case (wide intervals, many
    decisions near the threshold) with a one-sample pixel and a non-default significance level."""
    _, smp, st = make_case(90, 30, 3, seed=17)
    st["radiance"]["n"][5, 7] = 1
    st["radiance"]["n"][20, 60:64] = 0
    mc, dc, out, variant, (omc, odc, oout) = run_spec(gpu, oracle, st, spec_kw, radius=7, sd=4.0, alpha_index=2)
    assert np.array_equal(mc, omc, equal_nan=True) and np.array_equal(dc, odc, equal_nan=True)
    assert variant == expected_lds_variant(spec_kw, 3, 7), variant
    for c in range(3):
        assert rel_l2(out[..., c], oout[..., c]) <= TOL, c
    if variant != "generic":       # the general kernel under the same spec: the two HIP paths agree as well
        _, _, out_g, variant_g, _ = run_spec(gpu, oracle, st, spec_kw, radius=7, sd=4.0, alpha_index=2, force=1)
        assert variant_g == "generic"
        for c in range(3):
            assert rel_l2(out[..., c], out_g[..., c]) <= TOL, c
        # ... and the one-sided LDS kernel's build for the spec (Welch: it has none -- the general kernel)
        _, _, out_l, variant_l, _ = run_spec(gpu, oracle, st, spec_kw, radius=7, sd=4.0, alpha_index=2, force=2)
        gate, joint = spec_kw.get("gate", 0), spec_kw.get("channel_rule", 0)
        assert variant_l == ("generic" if spec_kw.get("dof", 0) else "lds_rt" + ("", "_asym", "_centre")[gate] + ("_joint" if joint else "")), variant_l
        for c in range(3):
            assert rel_l2(out_l[..., c], oout[..., c]) <= TOL, c


@pytest.mark.parametrize("spec_kw", [dict(), dict(gate=1), dict(dof=1), dict(border=1, channel_rule=1), dict(sides=1, small_n=1),
                                     dict(border=1), dict(gate=1, channel_rule=1, border=1), dict(channel_rule=1)],
                         ids=["default", "asym", "welch", "clamp+joint", "one-sided+exclude", "clamp", "asym+joint+clamp", "joint"])
@pytest.mark.parametrize("channels", [1, 3])
def test_filter_spec_r20_and_float(gpu, oracle, spec_kw, channels):
    """The shipped radius / sd under a few specs, RGB and float buffers; the default spec must stay on the LDS kernel."""
    _, smp, st = make_case(280, 26, 4, seed=23)
    mc, dc, out, variant, (omc, odc, oout) = run_spec(gpu, oracle, st, spec_kw, radius=RADIUS, sd=FILTER_SD, channels=channels)
    assert np.array_equal(dc, odc, equal_nan=True)
    assert variant == expected_lds_variant(spec_kw, channels, RADIUS), variant
    for c in range(channels):
        assert rel_l2(out[..., c], oout[..., c]) <= TOL, c


@pytest.mark.parametrize("joint", [0, 1], ids=["per-channel", "joint"])
def test_filter_welch_quantile_band(gpu, oracle, joint):
    """Welch degrees of freedom on the pair-symmetric kernel read their quantiles from a band of the table in LDS, chosen per
    work item from the least sample count it touches; an item whose pairs ask for more than the band holds is flagged and
    computed again from the table in global memory.  Uniform counts (whatever their size) never leave the band; a film
    whose counts jump from 3 to thousands inside a tile does -- both against the oracle, with the number of items that
    took the second path read back (statmc_debug_welch_far_items)."""
    lib = gpu.load()
    lib.statmc_debug_welch_far_items.restype = C.c_int
    spec_kw = dict(dof=1, channel_rule=joint)
    for n_of, far_expected in ((lambda n: n, False),                                      # 4 everywhere
                               (lambda n: np.full_like(n, 900), False),                   # nu in 899 .. 1798: band from 898
                               (lambda n: np.full_like(n, 5000), False),                  # beyond the table: its last entry
                               (lambda n: np.where(np.arange(n.shape[1])[None, :] < 150, n, 2600).astype(n.dtype), True)):
        _, smp, st = make_case(280, 26, 4, seed=29)
        st["radiance"]["n"][...] = n_of(st["radiance"]["n"])
        mc, dc, out, variant, (omc, odc, oout) = run_spec(gpu, oracle, st, spec_kw, radius=RADIUS, sd=FILTER_SD)
        far = lib.statmc_debug_welch_far_items()
        assert variant == "sym_welch" + ("_joint" if joint else ""), variant
        assert (far > 0) == far_expected, far
        for c in range(3):
            assert rel_l2(out[..., c], oout[..., c]) <= TOL, (c, far)


def test_filter_spec_errors_and_per_device_state(gpu):
    lib = gpu.load()
    bad = gpu.FilterSpec(gate=3)
    assert lib.statmc_set_filter_spec(C.byref(bad)) == gpu.ERR_INVALID
    bad = gpu.FilterSpec(channel_rule=2)
    assert lib.statmc_set_filter_spec(C.byref(bad)) == gpu.ERR_INVALID
    assert lib.statmc_set_filter_spec(None) == gpu.ERR_INVALID
    gpu.set_filter_spec(border=1, sides=1)
    try:
        assert gpu.get_filter_spec().as_tuple() == (0, 0, 1, 0, 1, 0)
        gpu.check(lib.statmc_setup(0))                       # idempotent: the device keeps its settings
        assert gpu.get_filter_spec().as_tuple() == (0, 0, 1, 0, 1, 0)
        # Welch mode needs the sample counts at the window filter
        z = torch.zeros(8, 8, 3, device=DEV)
        a, keep = gpu.make_filter_args([], [], [], [], [z], [z.clone()], [z.clone()], [z.clone()], [], g_sds=[], radius=2)
        gpu.set_filter_spec(dof=1)
        assert lib.statmc_window_filter(C.byref(a), 3) == gpu.ERR_INVALID
    finally:
        gpu.set_filter_spec()
    assert gpu.get_filter_spec().as_tuple() == (0,) * 6
    # a device that was never set up is refused
    assert lib.statmc_set_device(torch.cuda.device_count() + 3) == gpu.ERR_NO_DEVICE


def test_filter_non_finite_feature(gpu, oracle):
    """ADVICE r3 / spec v2.1: a NaN or infinite G-buffer value used to make the range weight of every pair with that pixel NaN
    -- in the oracle too -- and, in the runtime-radius builds of the pair-symmetric kernel, the weight of taps just BEYOND a
    small radius (their exponent is -inf only while the feature term is a number).  Now such a pixel takes no part, like one
    with a non-finite colour: it keeps its own colour, every other pixel is what it would be without it; every kernel
    agrees with the oracle.  Small radii put the NaN just outside many windows; six and eight feature planes; a clamped border
    (the edge pixel that a clamped tap repeats is the bad one)."""
    for radius, sd, planes8, spec_kw in ((20, 10.0, False, {}), (3, 2.0, False, {}), (6, 3.0, True, {}), (7, 4.0, False, dict(border=1)),
                                         (20, 10.0, True, {}), (5, 3.0, False, dict(dof=1))):
        mc, disc, colour, gbs = stats_case(oracle, 300, 30, 6, seed=77 + radius)
        gbs = [g.copy() for g in gbs]
        g_dr = list(G_DR)
        if planes8:
            rng = np.random.default_rng(radius)
            gbs += [rng.random(mc.shape[:2] + (1,), dtype=np.float32) * 3, rng.integers(0, 4, mc.shape[:2] + (1,)).astype(np.float32)]
            g_dr += [-0.5 / 0.7 ** 2, -0.5 / 0.5 ** 2]
            gbs[2][11, 150, 0] = np.inf
            gbs[3][25, 7, 0] = np.nan
        gbs[0][4, 60, 1] = np.nan
        gbs[1][12, 200, 2] = -np.inf
        gbs[0][0, 0, 0] = np.nan                       # the corner pixel: what a clamped border repeats
        gbs[1][29, 140, 0] = np.nan                    # last row
        bad_px = [(4, 60), (12, 200), (0, 0), (29, 140)] + ([(11, 150), (25, 7)] if planes8 else [])
        n = np.full(mc.shape[:2], 6, np.int32)
        ospec = oracle.FilterSpec(**spec_kw)
        if spec_kw.get("dof"):                          # Welch: the discriminator image holds s^2 / n
            disc = (disc / np.float32(oracle.t_quantile(0, 5)) ** 2).astype(np.float32)
        ref = oracle.filter_image(mc, disc, colour, gbs, g_dr, -0.5 / sd ** 2, radius, spec=ospec, n=n)
        assert np.isfinite(ref).all()                  # nothing spreads
        for y, x in bad_px:
            assert np.array_equal(ref[y, x], colour[y, x])       # the pixel passes its own colour through
        clean = [np.nan_to_num(g, nan=0.5, posinf=0.5, neginf=0.5) for g in gbs]
        ref_clean = oracle.filter_image(mc, disc, colour, clean, g_dr, -0.5 / sd ** 2, radius, spec=ospec, n=n)
        far = np.ones(mc.shape[:2], bool)
        for y, x in bad_px:
            far[max(0, y - radius):y + radius + 1, max(0, x - radius):x + radius + 1] = False
        if not spec_kw.get("border"):
            assert np.array_equal(ref[far], ref_clean[far])      # pixels whose window does not hold a bad one are untouched
        gpu.set_filter_spec(**spec_kw)
        try:
            for force in (0, 2, 1):
                out, v = run_filter(gpu, mc, disc, colour, gbs, g_dr, sd, radius, force=force, n=n if spec_kw.get("dof") else None)
                assert np.isfinite(out).all(), (v, radius)
                for c in range(3):
                    assert rel_l2(out[..., c], ref[..., c]) <= TOL, (v, radius, c)
        finally:
            gpu.set_filter_spec()


def test_filter_non_finite_colour(gpu, oracle):
    """a NaN / inf colour at a pixel that is not a member (or not valid) used to reach the sums of
    the LDS kernel as 0 * NaN.  Spec v2: such a pixel takes no part; all three kernels agree with the oracle."""
    for radius, sd in ((20, 10.0), (7, 4.0)):
        mc, disc, colour, gbs = stats_case(oracle, 300, 30, 6, seed=31 + radius)
        colour = colour.copy()
        colour[3, 40, 1] = np.nan            # valid statistics, NaN colour
        colour[9, 250] = np.inf
        mc[15, 100] = np.nan                 # invalid statistics AND non-finite colour
        colour[15, 100] = np.nan
        mc[20, 20] = 1e9                     # valid, member of no other window, finite colour
        disc[20, 20] = 0.0
        colour[21, 21, 0] = -np.inf
        ref = oracle.filter_image(mc, disc, colour, gbs, G_DR, -0.5 / sd ** 2, radius)
        bad = ~np.isfinite(ref)
        assert bad.sum() == 1 + 3 + 3 + 1    # only the pixels themselves
        for force in (0, 3, 2, 1):
            out, v = run_filter(gpu, mc, disc, colour, gbs, G_DR, sd, radius, force=force)
            assert np.array_equal(~np.isfinite(out), bad), v
            ok = ~bad
            for c in range(3):
                assert rel_l2(out[..., c][ok[..., c]], ref[..., c][ok[..., c]]) <= TOL, (v, c)
    # float mode: per buffer
    W, H = 48, 6
    rng = np.random.default_rng(8)
    mcs = [rng.standard_normal((H, W)).astype(np.float32) for _ in range(3)]
    dcs = [(rng.random((H, W)) * 2).astype(np.float32) for _ in range(3)]
    cols = [rng.random((H, W), dtype=np.float32) for _ in range(3)]
    gbs = [rng.random((H, W, 3), dtype=np.float32), rng.random((H, W, 3), dtype=np.float32)]
    cols[1][2, 10] = np.nan
    cols[2][4, 30] = np.inf
    g_dr = [-0.5 / 0.3 ** 2, -0.5 / 0.2 ** 2]
    outs = [torch.zeros(H, W, device=DEV) for _ in range(3)]
    a, keep = gpu.make_filter_args([], [], [], [], [to_dev(c) for c in cols], [to_dev(m) for m in mcs], [to_dev(d) for d in dcs],
                                   outs, [to_dev(g) for g in gbs], g_dr=g_dr, filter_sd=FILTER_SD, radius=20)
    gpu.window_filter(a, 1)
    torch.cuda.synchronize()
    for b in range(3):
        ref = oracle.filter_image(mcs[b], dcs[b], cols[b], gbs, g_dr, -0.5 / FILTER_SD ** 2, 20)
        o = outs[b].cpu().numpy()
        assert np.array_equal(np.isfinite(o), np.isfinite(ref)), b
        ok = np.isfinite(ref)
        assert rel_l2(o[ok], ref[ok]) <= TOL, b


# ------------------------------------------------------------------ the reference's own build recipe fuses multiply-adds
def test_accumulate_against_both_contraction_modes(gpu, oracle):
    """scripts/_build.sh builds the reference with clang++ -O3 -march=native, whose default -ffp-contract=on fuses the
    m2 / m3 / filmM2 updates of StatTile<Float> (not of StatTile<Vec3>: the multiply and the add sit in different
    inlined operators).  The HIP kernel rounds every operation on its own (= a g++ build of the reference): it is
    bit-exact against the un-contracted oracle on the raw-sample chain and within 1e-5 of BOTH modes on a heavy-tailed
    1024-spp stream; the distance between the two modes themselves is reported."""
    rng = np.random.default_rng(42)
    S, H, W = 1024, 8, 64
    # log-normal radiance with 20 % zero paths and rare x1000 fireflies: the third moment is ill-conditioned
    smp = rng.lognormal(0.0, 1.0, size=(S, H, W, 1)).astype(np.float32)
    smp *= (rng.random((S, H, W, 1)) > 0.2)
    smp *= 1.0 + 999.0 * (rng.random((S, H, W, 1)) > 0.9995)
    smp = np.ascontiguousarray(smp, dtype=np.float32)
    st = {k: to_dev(v) for k, v in oracle.new_state(H, W, 1).items()}
    for a, b in ((0, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 128), (128, 256), (256, 512), (512, 1024)):   # the reference's schedule
        gpu.accumulate(W, H, [gpu.make_stat_type(to_dev(smp[a:b]), st, True, 3)])
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in st.items()}
    ref = {}
    try:
        for mode in (False, True):
            oracle.set_fp_contract(mode)
            s = oracle.new_state(H, W, 1)
            oracle.accumulate(s, smp, True, 3)
            ref[mode] = s
    finally:
        oracle.set_fp_contract(False)
    assert np.array_equal(got["n"], ref[False]["n"])
    assert np.array_equal(got["film_mean"], ref[False]["film_mean"]) and np.array_equal(got["film_m2"], ref[False]["film_m2"])
    inter = {}
    for k in ("mean", "m2", "m3", "film_mean", "film_m2"):
        for mode in (False, True):
            assert rel_l2(got[k], ref[mode][k]) <= TOL, (k, mode, rel_l2(got[k], ref[mode][k]))
        inter[k] = rel_l2(ref[True][k], ref[False][k])
    print("contracted vs un-contracted reference arithmetic, relative L2:", {k: "%.2e" % v for k, v in inter.items()})
    assert inter["mean"] == 0.0 and inter["film_mean"] == 0.0 and 0 < inter["m2"] < 1e-6 and 0 < inter["m3"] < TOL
    # the statistics the filter consumes: corrected mean and discriminator of both modes agree within the tolerance too
    pre = [oracle.prepass(ref[m]["n"], ref[m]["mean"], ref[m]["m2"], ref[m]["m3"]) for m in (False, True)]
    assert rel_l2(pre[1][0], pre[0][0]) <= TOL and rel_l2(pre[1][1], pre[0][1]) <= TOL
