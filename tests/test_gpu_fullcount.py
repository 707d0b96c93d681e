"""BASELINE.json configs[2] (1920x1080, 256 spp) and configs[4] (3840x2160, 1024 spp) at their FULL sample counts,
HIP path against the oracle on the same samples.

The film is accumulated on the device the way the reference renders it (statpath.cpp:272-279: iterations of 4, 4, 8, 16, ...
samples on persistent statistics), from a seeded stream that is regenerated in 32-sample chunks (374 GB of samples at
4K / 1024 spp never exist at once).  The oracle (oracle/statmc_oracle.c, the restated StatTile<T>::Add*Sample*,
estimator.h:162-226) accumulates the SAME samples, in the same order, on eight full-width rows and the four 16 x 16
corner blocks of the film.  Then: sample counts and every moment of untransformed samples bit for bit, moments of
Box-Cox-transformed samples <= 1e-5 relative L2 (v_sqrt_f32 against powf, the one non-bit-exact step), the pre-pass
bit for bit on those regions (dof = 1023 of the t-table at 1024 spp), and strips of the window filter <= 1e-5 per channel."""
import numpy as np
import pytest
import torch

from conftest import FILTER_SD, RADIUS, SD_ALBEDO, SD_NORMAL, rel_l2

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
G_DR = [-0.5 / SD_NORMAL ** 2, -0.5 / SD_ALBEDO ** 2]
CHUNK = 32


def regions_of(W, H):
    """Eight full-width rows (top, bottom, and six in between) + the four 16 x 16 corner blocks: (y0, y1, x0, x1)."""
    rows = sorted({0, 1, H // 4, H // 2 - 1, H // 2, (3 * H) // 4, H - 2, H - 1})
    reg = [(y, y + 1, 0, W) for y in rows]
    reg += [(0, 16, 0, 16), (0, 16, W - 16, W), (H - 16, H, 0, 16), (H - 16, H, W - 16, W)]
    return reg


def run_full_count(gpu, oracle, W, H, spp, types, seed):
    from statmc_amd import film, synthetic
    from statmc_amd.film import STAT_TYPES
    scene = synthetic.Scene(W, H, seed=seed, device=DEV)
    fs = film.FilmStats(W, H, DEV, types=types)
    regs = regions_of(W, H)
    ost = [{t: oracle.new_state(y1 - y0, x1 - x0, STAT_TYPES[t]["channels"]) for t in types} for (y0, y1, x0, x1) in regs]
    batches = synthetic.sample_schedule(spp)
    assert sum(batches) == spp and batches[:3] == [4, 4, 8]
    # batch boundaries inside the stream of 32-sample chunks
    cuts = set(np.cumsum(batches).tolist())
    launches = 0
    for s0 in range(0, spp, CHUNK):
        n = min(CHUNK, spp - s0)
        part = scene.samples(n, seed=seed * 100000 + s0, features=types)
        # the device accumulates batch by batch (a batch never straddles two chunks unless it is a multiple of them)
        edges = sorted({0, n} | {c - s0 for c in cuts if s0 < c < s0 + n})
        for a, b in zip(edges[:-1], edges[1:]):
            fs.accumulate({t: v[a:b] for t, v in part.items()})
            launches += 1
        for (y0, y1, x0, x1), st in zip(regs, ost):
            for t in types:
                oracle.accumulate(st[t], np.ascontiguousarray(part[t][:, y0:y1, x0:x1].cpu().numpy()),
                                  STAT_TYPES[t]["transform"], STAT_TYPES[t]["max_moment"])
        del part
    torch.cuda.synchronize()
    assert launches >= len(batches)

    # ---- accumulated state, region by region
    for (y0, y1, x0, x1), st in zip(regs, ost):
        for t in types:
            got = {k: v[y0:y1, x0:x1].cpu().numpy() for k, v in fs.state[t].items() if v is not None}
            want = st[t]
            assert np.array_equal(got["n"], want["n"]) and int(got["n"].min()) == spp, (t, "n", y0, x0)
            if STAT_TYPES[t]["transform"]:
                # raw-sample Welford: bit for bit; moments of the Box-Cox'd sample: sqrt against powf
                for k in ("film_mean", "film_m2"):
                    assert np.array_equal(got[k], want[k]), (t, k, y0, x0)
                for k in ("mean", "m2", "m3"):
                    for c in range(got[k].shape[-1]):
                        assert rel_l2(got[k][..., c], want[k][..., c]) <= 1e-5, (t, k, c, y0, x0)
            else:
                keys = ["mean"] + (["m2"] if STAT_TYPES[t]["max_moment"] >= 2 else []) + (["m3"] if STAT_TYPES[t]["max_moment"] >= 3 else [])
                for k in keys:
                    assert np.array_equal(got[k], want[k]), (t, k, y0, x0)

    # ---- pre-pass on the regions: bit for bit on the device's own moments
    fs.prepass()
    torch.cuda.synchronize()
    rad = fs.state["radiance"]
    for (y0, y1, x0, x1) in regs:
        sl = (slice(y0, y1), slice(x0, x1))
        cp = lambda v: np.ascontiguousarray(v[sl].cpu().numpy())
        omc, odc = oracle.prepass(cp(rad["n"]), cp(rad["mean"]), cp(rad["m2"]), cp(rad["m3"]))
        assert np.array_equal(fs.mean_corr[sl].cpu().numpy(), omc, equal_nan=True), (y0, x0)
        assert np.array_equal(fs.disc[sl].cpu().numpy(), odc, equal_nan=True), (y0, x0)
    return fs


def check_filter_strips(gpu, oracle, fs, rois):
    colour = fs.state["radiance"]["film_mean"]
    a, keep = fs.filter_args()
    gpu.window_filter(a, 3)
    torch.cuda.synchronize()
    assert gpu.last_filter_variant() == "sym_r20"
    whole = fs.film_f
    assert torch.isfinite(whole).all()
    mc, dc, col = fs.mean_corr.cpu().numpy(), fs.disc.cpu().numpy(), colour.cpu().numpy()
    gb = [fs.g_buffer("normal").cpu().numpy(), fs.g_buffer("albedo").cpu().numpy()]
    for roi in rois:
        x0, y0, x1, y1 = roi
        ref = oracle.filter_image(mc, dc, col, gb, G_DR, -0.5 / FILTER_SD ** 2, RADIUS, roi=roi)[y0:y1, x0:x1]
        got = whole[y0:y1, x0:x1].cpu().numpy()
        for c in range(3):
            assert rel_l2(got[..., c], ref[..., c]) <= 1e-5, (roi, c)


@pytest.mark.parametrize("channels", [11, 9])
def test_config2_1080p_256spp_against_the_oracle(gpu, oracle, channels):
    from statmc_amd import synthetic
    types = list(synthetic.FEATURES) if channels == 11 else ["radiance", "normal", "albedo"]
    W, H = 1920, 1080
    fs = run_full_count(gpu, oracle, W, H, 256, types, seed=21)
    check_filter_strips(gpu, oracle, fs, [(0, 536, W, 542), (0, 0, 280, 6), (W - 280, H - 6, W, H)])


@pytest.mark.parametrize("channels", [11, 9])
def test_config4_4k_1024spp_against_the_oracle(gpu, oracle, channels):
    """9 chained iterations (4, 4, 8, ..., 512), n = 1024, dof 1023."""
    from statmc_amd import synthetic
    types = list(synthetic.FEATURES) if channels == 11 else ["radiance", "normal", "albedo"]
    W, H = 3840, 2160
    assert synthetic.sample_schedule(1024) == [4, 4, 8, 16, 32, 64, 128, 256, 512]
    fs = run_full_count(gpu, oracle, W, H, 1024, types, seed=22)
    check_filter_strips(gpu, oracle, fs, [(0, 1077, W, 1081), (0, 0, 280, 6), (W - 280, H - 6, W, H)])
