import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) GPU")


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = np.sqrt((b ** 2).sum())
    return float(np.sqrt(((a - b) ** 2).sum()) / (den if den > 0 else 1.0))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def gpu():
    """The HIP library, set up on cuda:0.  GPU tests fail loudly if it cannot be loaded."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test started without a GPU"
    from statmc_amd import api
    api.setup(0)
    return api


# shipped filter parameters (scenes/render-denoise.pbrt:19-22)
FILTER_SD = 10.0
RADIUS = 20
SD_NORMAL, SD_ALBEDO = 0.1, 0.02


def make_case(width, height, spp, seed=1, features=("radiance", "normal", "albedo"), n_regions=6):
    """Small synthetic film: samples + oracle-accumulated statistics (numpy)."""
    from oracle import oracle as o
    from statmc_amd import synthetic
    from statmc_amd.film import STAT_TYPES
    scene = synthetic.Scene(width, height, n_regions=n_regions, seed=seed)
    smp = {k: v.numpy() for k, v in scene.samples(spp, seed=seed + 100, features=features).items()}
    st = {}
    for t in features:
        st[t] = o.new_state(height, width, STAT_TYPES[t]["channels"])
        o.accumulate(st[t], smp[t], STAT_TYPES[t]["transform"], STAT_TYPES[t]["max_moment"])
    return scene, smp, st


def edge_case_stream(W=8, H=8, S=24, seed=7):
    """SURVEY 8c's edge cases as one small sample file: rows 0-2 log-normal radiance with 20 % zero paths, row 3 all zeros
    (Box-Cox -> -2), row 4 constants (m2 = m3 = 0), row 5 one firefly among small values, row 6 a single sample (n = 1),
    row 7 ragged counts 1 .. 8.  Returns (count [H, W] int32, samples [S, H, W, 3] float32)."""
    rng = np.random.default_rng(seed)
    smp = np.exp(rng.normal(0.0, 1.0, (S, H, W, 3))).astype(np.float32)
    smp *= (rng.random((S, H, W, 1)) >= 0.2)
    smp[:, 3] = 0.0
    smp[:, 4] = (0.25 * (1 + np.arange(W, dtype=np.float32)))[None, :, None]
    smp[:, 5] = (0.01 + 0.001 * rng.random((S, W, 3))).astype(np.float32)
    smp[7, 5] *= 1000.0
    count = np.full((H, W), S, np.int32)
    count[6] = 1
    count[7] = 1 + np.arange(W)
    return count, np.ascontiguousarray(smp, np.float32)
