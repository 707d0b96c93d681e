"""Committed golden vectors (tests/golden/*.npz, self-oracle: see make_golden.py) against the
CPU oracle (no GPU) and against the HIP path (gpu)."""
import glob
import os

import numpy as np
import pytest

from conftest import rel_l2

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "case_*.npz")))
G_DR = [-0.5 / 0.1 ** 2, -0.5 / 0.02 ** 2]


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path)
    h, w = g["n"].shape
    st = oracle.new_state(h, w, 3)
    oracle.accumulate(st, g["samples_radiance"], True, 3)
    for k in ("n", "mean", "m2", "m3", "film_mean", "film_m2"):
        assert np.array_equal(st[k], g[k]), k
    mc, disc = oracle.prepass(g["n"], g["mean"], g["m2"], g["m3"])
    assert np.array_equal(mc, g["mean_corr"]) and np.array_equal(disc, g["discriminator"])
    out = oracle.filter_image(mc, disc, g["film_mean"], [g["normal_mean"], g["albedo_mean"]], G_DR,
                              -0.5 / float(g["filter_sd"]) ** 2, int(g["radius"]))
    assert np.array_equal(out, g["film_f"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_hip_reproduces_golden(gpu, path):
    import torch
    from statmc_amd import film
    g = np.load(path)
    h, w = g["n"].shape
    dev = torch.device("cuda:0")
    fs = film.FilmStats(w, h, dev, filter_sd=float(g["filter_sd"]), radius=int(g["radius"]))
    fs.accumulate({t: torch.from_numpy(g["samples_" + t]).to(dev) for t in ("radiance", "normal", "albedo")})
    out = fs.denoise().cpu().numpy()
    rad = fs.state["radiance"]
    assert np.array_equal(rad["n"].cpu().numpy(), g["n"])
    assert np.array_equal(rad["film_mean"].cpu().numpy(), g["film_mean"])      # raw-sample Welford: exact
    assert np.array_equal(fs.g_buffer("normal").cpu().numpy(), g["normal_mean"])
    assert np.array_equal(fs.g_buffer("albedo").cpu().numpy(), g["albedo_mean"])
    for k in ("mean", "m2", "m3"):                                             # sqrt vs pow(x, .5): <= 1 ulp apart
        assert rel_l2(rad[k].cpu().numpy(), g[k]) <= 1e-5, k
    for c in range(3):
        assert rel_l2(out[..., c], g["film_f"][..., c]) <= 1e-5, c


# ---------------------------------------------------------------- SURVEY 8c golden sets (1) - (3), committed in round 4
GDIR = os.path.join(os.path.dirname(__file__), "golden")
VARIANTS = [(c, t, m) for c in (1, 3) for t in (0, 1) for m in (1, 2, 3)]
FIELDS = ("mean", "m2", "m3", "film_mean", "film_m2")


def test_oracle_reproduces_the_accumulate_edge_cases(oracle):
    """Zeros (Box-Cox -> -2), constants (m2 = m3 = 0), one firefly, n = 1, ragged counts through all twelve
    (T, transform, maxMoment) variants of StatTile<T>::Add[Transform]SampleM{1,2,3} (estimator.h:162-232), both
    contraction modes; plus what the fixture must say about those cases whatever produced it."""
    g = np.load(os.path.join(GDIR, "accumulate_edge_cases.npz"))
    count, smp = g["count"], g["samples"]
    H, W = count.shape
    try:
        for contract, prefix in ((False, ""), (True, "fma_")):
            oracle.set_fp_contract(contract)
            for c, t, m in VARIANTS:
                key = "%sc%d_t%d_m%d_" % (prefix, c, t, m)
                assert np.array_equal(g[key + "n"], count)
                for y in range(H):
                    for x in range(W):
                        px = oracle.add_samples_to_pixel(smp[:count[y, x], y, x, :c], c, t, m)
                        for f in FIELDS:
                            assert np.asarray(px[f], np.float32).tobytes() == g[key + f][y, x].tobytes(), (key, f, y, x)
    finally:
        oracle.set_fp_contract(False)
    full = g["c3_t1_m3_mean"]
    assert np.all(full[3] == -2.0) and np.all(g["c3_t1_m3_m2"][3] == 0) and np.all(g["c3_t1_m3_film_mean"][3] == 0)   # zero samples
    assert np.all(g["c3_t0_m3_m2"][4] == 0) and np.all(g["c3_t0_m3_m3"][4] == 0)                                    # constants
    assert np.all(g["c1_t0_m2_m2"][6] == 0) and np.array_equal(g["c1_t0_m1_mean"][6, :, 0], smp[0, 6, :, 0])       # n = 1
    assert np.all(g["c3_t0_m3_m3"][5] > 0)                                                                          # the firefly skews the pixel
    # non-transform types: film-mean == mean and film-m2 == m2 (estimator.h:209-210)
    assert np.array_equal(g["c3_t0_m2_film_mean"], g["c3_t0_m2_mean"]) and np.array_equal(g["c3_t0_m2_film_m2"], g["c3_t0_m2_m2"])
    # Vec3 tiles are not contracted in either build (the multiply and the add sit in different operator functions)
    for k in ("mean", "m2", "m3", "film_m2"):
        assert np.array_equal(g["c3_t1_m3_" + k], g["fma_c3_t1_m3_" + k]), k


def test_oracle_reproduces_the_mean_vars_quirk(oracle):
    g = np.load(os.path.join(GDIR, "mean_vars_row_quirk.npz"))
    q, p = oracle.mean_vars(g["n"], g["film_m2"], row_n_quirk=True), oracle.mean_vars(g["n"], g["film_m2"], row_n_quirk=False)
    assert np.array_equal(q, g["film_var_row_quirk"]) and np.array_equal(p, g["film_var_per_pixel"])
    assert np.array_equal(q[2], p[2]) and not np.array_equal(q, p)     # the uniform row agrees, the others do not
    nf = g["n"][:, :1].astype(np.float32)
    assert np.array_equal(q, (g["film_m2"] / ((nf - np.float32(1)) * nf)[..., None]).astype(np.float32))    # estimator.cpp:540,558: n of the row's first pixel


def test_host_side_catalogue_matches_the_fixture():
    """SURVEY 8c golden set (3) against the C++ host side's AllocateBuffers (product code, no GPU)."""
    import json
    import subprocess
    from statmc_amd import build
    exe = build.build_tools()
    cat = json.load(open(os.path.join(GDIR, "buffer_catalogue.json")))
    assert sorted(cat) == ["acrr", "denoise", "ours", "proden", "smis"]
    for cfg, lines in cat.items():
        got = subprocess.check_output([exe, "--catalogue", "--config", cfg, "--width", "32", "--height", "16"], text=True).splitlines()
        assert got == lines, cfg
    d = cat["denoise"]
    assert sum(l.startswith("buffer ") for l in d) == 32 and sum(l.startswith("upload ") for l in d) == 7 and sum(l.startswith("download ") for l in d) == 1


@pytest.mark.gpu
def test_hip_reproduces_the_accumulate_edge_cases(gpu):
    """The HIP accumulation over the same fixture: rows of full count through statmc_accumulate_rows, the n = 1 row as a
    batch of one, the ragged row through statmc_accumulate_tiles (one 1 x 1 tile per pixel, its own sample count).  Counts and
    every untransformed quantity bit for bit (against the un-contracted oracle), Box-Cox moments within 1e-5 (v_sqrt_f32
    against powf)."""
    import torch
    from statmc_amd import api, film
    g = np.load(os.path.join(GDIR, "accumulate_edge_cases.npz"))
    count, smp = g["count"], g["samples"]
    S, H, W, _ = smp.shape
    dev = torch.device("cuda:0")
    for c, t, m in VARIANTS:
        key = "c%d_t%d_m%d_" % (c, t, m)
        st = film.new_state(H, W, c, dev, transform=True)
        full = torch.from_numpy(np.ascontiguousarray(smp[..., :c])).to(dev)
        api.accumulate(W, H, [api.make_stat_type(full, st, t, m)], rows=(0, 6))
        api.accumulate(W, H, [api.make_stat_type(full[:1].contiguous(), st, t, m)], rows=(6, 7))
        # row 7: pixel x has x + 1 samples -- its own 1 x 1 tile block [count][1][1][c] in the arena
        blocks = [smp[:count[7, x], 7, x, :c].reshape(-1) for x in range(W)]
        arena = torch.from_numpy(np.concatenate(blocks)).to(dev)
        offs = torch.tensor(np.cumsum([0] + [int(count[7, x]) for x in range(W - 1)]), dtype=torch.int64, device=dev)
        bounds = torch.tensor([(x, 7, x + 1, 8) for x in range(W)], dtype=torch.int32, device=dev)
        cnt = torch.tensor([int(count[7, x]) for x in range(W)], dtype=torch.int32, device=dev)
        api.accumulate_tiles(W, H, [api.make_stat_type_arena(arena, c, st, t, m)], bounds, offs, cnt)
        torch.cuda.synchronize()
        assert np.array_equal(st["n"].cpu().numpy(), g[key + "n"]), key
        names = ["mean"] + (["m2"] if m >= 2 else []) + (["m3"] if m >= 3 else [])
        for f in names:
            a, b = st[f].cpu().numpy(), g[key + f]
            if t:
                assert rel_l2(a, b) <= 1e-5, (key, f)
            else:
                assert np.array_equal(a, b), (key, f)
        if t:      # the raw-sample Welford chain: no transform in it, exact
            assert np.array_equal(st["film_mean"].cpu().numpy(), g[key + "film_mean"]), key
            assert np.array_equal(st["film_m2"].cpu().numpy(), g[key + "film_m2"]), key


@pytest.mark.gpu
def test_hip_reproduces_the_mean_vars_quirk(gpu):
    import torch
    from statmc_amd import api
    g = np.load(os.path.join(GDIR, "mean_vars_row_quirk.npz"))
    dev = torch.device("cuda:0")
    n, m2 = torch.from_numpy(g["n"]).to(dev), torch.from_numpy(g["film_m2"]).to(dev)
    for quirk, key in ((True, "film_var_row_quirk"), (False, "film_var_per_pixel")):
        out = torch.zeros_like(m2)
        api.calculate_mean_vars([n], [m2], [out], row_n_quirk=quirk)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), g[key]), key
