"""BASELINE.json full-size cases (1920x1080) through size-independent properties, since the CPU
oracle needs minutes for a whole 1080p window filter: linearity in the colour image, constant
fixed point, agreement of the two independent HIP kernels (LDS vs generic) and of both with the
oracle on a strip, block-decomposed filtering == whole-film filtering, batch chaining."""
import numpy as np
import pytest
import torch

from conftest import FILTER_SD, RADIUS, SD_ALBEDO, SD_NORMAL, rel_l2

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
W, H = 1920, 1080
G_DR = [-0.5 / SD_NORMAL ** 2, -0.5 / SD_ALBEDO ** 2]


@pytest.fixture(scope="module")
def film1080(gpu):
    from statmc_amd import film, synthetic
    scene = synthetic.Scene(W, H, seed=1, device=DEV)
    fs = film.FilmStats(W, H, DEV, types=synthetic.FEATURES)
    smp = scene.samples(16, seed=2)
    fs.accumulate(smp)
    fs.prepass()
    torch.cuda.synchronize()
    return fs, smp


def wf(gpu, fs, colour, out, roi=None, force=0):
    a, keep = fs.filter_args(roi=roi, colour=colour, out=out)
    gpu.force_filter_variant(force)
    try:
        gpu.window_filter(a, 3)
        torch.cuda.synchronize()
    finally:
        gpu.force_filter_variant(0)
    return out


def test_batches_chain_exactly(gpu, film1080):
    """16 spp in one launch == 4 + 4 + 8 (the reference's schedule): same per-pixel update order."""
    from statmc_amd import film, synthetic
    fs, smp = film1080
    fs2 = film.FilmStats(W, H, DEV, types=synthetic.FEATURES)
    for a, b in ((0, 4), (4, 8), (8, 16)):
        fs2.accumulate({k: v[a:b].contiguous() for k, v in smp.items()})
    torch.cuda.synchronize()
    for t in synthetic.FEATURES:
        for k, v in fs.state[t].items():
            if v is not None:
                assert torch.equal(v, fs2.state[t][k]), (t, k)
    assert int(fs.state["radiance"]["n"].min()) == 16 == int(fs.state["depth"]["n"].max())


def test_moments_match_torch_float64(gpu, film1080):
    fs, smp = film1080
    x = smp["radiance"][:, 500:520].double()
    assert rel_l2(fs.state["radiance"]["film_mean"][500:520].cpu().numpy(), x.mean(0).cpu().numpy()) < 1e-6
    tx = (x.sqrt() - 1) / 0.5
    m2 = ((tx - tx.mean(0)) ** 2).sum(0)
    assert rel_l2(fs.state["radiance"]["m2"][500:520].cpu().numpy(), m2.cpu().numpy()) < 1e-5
    assert rel_l2(fs.state["albedo"]["mean"][500:520].cpu().numpy(), smp["albedo"][:, 500:520].double().mean(0).cpu().numpy()) < 1e-6


def test_filter_fullsize_properties(gpu, film1080):
    fs, _ = film1080
    colour = fs.state["radiance"]["film_mean"]
    f1 = wf(gpu, fs, colour, torch.zeros_like(colour)).clone()
    assert gpu.last_filter_variant() == "lds_r20"
    assert torch.isfinite(f1).all()
    # constant image is a fixed point (weights are normalised)
    const = torch.full_like(colour, 0.375)
    fc = wf(gpu, fs, const, torch.zeros_like(colour))
    assert float((fc - 0.375).abs().max()) < 0.375 * 1e-5   # 1681 fp32 terms / their sum
    # linear in the colour image: weights depend on the statistics only
    c2 = torch.rand_like(colour)
    f2 = wf(gpu, fs, c2, torch.zeros_like(colour)).clone()
    f12 = wf(gpu, fs, colour + 2 * c2, torch.zeros_like(colour))
    assert rel_l2(f12.cpu().numpy(), (f1 + 2 * f2).cpu().numpy()) < 1e-6
    # denoising moves the image towards the noise-free mean of many more samples
    # (sanity of the whole path, not a parity claim)
    assert float((f1 - colour).abs().mean()) > 0


def test_lds_kernel_vs_generic_vs_oracle_on_strip(gpu, oracle, film1080):
    fs, _ = film1080
    colour = fs.state["radiance"]["film_mean"]
    roi = (0, 530, W, 546)                                   # 16 full-width rows in the middle of the film
    a = wf(gpu, fs, colour, torch.zeros_like(colour), roi=roi, force=0)[530:546].cpu().numpy()
    b = wf(gpu, fs, colour, torch.zeros_like(colour), roi=roi, force=1)[530:546].cpu().numpy()
    assert rel_l2(a, b) < 1e-6
    ref = oracle.filter_image(fs.mean_corr.cpu().numpy(), fs.disc.cpu().numpy(), colour.cpu().numpy(),
                              [fs.g_buffer("normal").cpu().numpy(), fs.g_buffer("albedo").cpu().numpy()], G_DR,
                              -0.5 / FILTER_SD ** 2, RADIUS, roi=roi)[530:546]
    for c in range(3):
        assert rel_l2(a[..., c], ref[..., c]) <= 1e-5
        assert rel_l2(b[..., c], ref[..., c]) <= 1e-5
    # top-left and bottom-right corners: clipped windows
    for roi in ((0, 0, 300, 8), (W - 300, H - 8, W, H)):
        x0, y0, x1, y1 = roi
        a = wf(gpu, fs, colour, torch.zeros_like(colour), roi=roi)[y0:y1, x0:x1].cpu().numpy()
        ref = oracle.filter_image(fs.mean_corr.cpu().numpy(), fs.disc.cpu().numpy(), colour.cpu().numpy(),
                                  [fs.g_buffer("normal").cpu().numpy(), fs.g_buffer("albedo").cpu().numpy()], G_DR,
                                  -0.5 / FILTER_SD ** 2, RADIUS, roi=roi)[y0:y1, x0:x1]
        assert rel_l2(a, ref) <= 1e-5


def test_block_decomposition_equals_whole_film(gpu, film1080):
    """What 4 GPUs compute (2x2 blocks, each with its r-pixel halo) is bit-identical to the
    single-GPU result: per-pixel tap order does not depend on the block origin."""
    from statmc_amd import sharding
    fs, _ = film1080
    colour = fs.state["radiance"]["film_mean"]
    gpu.force_filter_parts(2)      # same window-row split for the film and for the blocks
    whole = wf(gpu, fs, colour, torch.zeros_like(colour)).clone()
    imgs = dict(mean_corr=fs.mean_corr, disc=fs.disc, colour=colour, normal=fs.g_buffer("normal"), albedo=fs.g_buffer("albedo"))
    bw, bh = W // 2, H // 2
    for rank in range(4):
        L = sharding.BlockLayout(rank, 4, bw, bh, RADIUS)
        ox, oy = L.origin
        loc = {k: v[oy - L.pt:oy + bh + L.pb, ox - L.pl:ox + bw + L.pr].contiguous() for k, v in imgs.items()}
        out = torch.zeros_like(loc["colour"])
        a, keep = gpu.make_filter_args([], [], [], [], [loc["colour"]], [loc["mean_corr"]], [loc["disc"]], [out],
                                       [loc["normal"], loc["albedo"]], g_sds=[SD_NORMAL, SD_ALBEDO],
                                       filter_sd=FILTER_SD, radius=RADIUS, roi=L.roi)
        gpu.window_filter(a, 3)
        torch.cuda.synchronize()
        assert torch.equal(L.interior(out), whole[oy:oy + bh, ox:ox + bw]), rank
    gpu.force_filter_parts(0)
