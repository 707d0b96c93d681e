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
