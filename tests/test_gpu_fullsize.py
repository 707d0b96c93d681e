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
    assert gpu.last_filter_variant() == "sym_r20"
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


@pytest.mark.parametrize("spec_kw,variant", [(dict(), "sym_r20"), (dict(border=1), "sym_r20_clamp"),
                                             (dict(gate=1, channel_rule=1, border=1), "sym_r20_asym_joint_clamp"), (dict(channel_rule=1), "sym_r20_joint"),
                                             (dict(gate=1), "sym_r20_asym"), (dict(gate=1, channel_rule=1), "sym_r20_asym_joint")],
                         ids=["default", "clamp", "asym+joint+clamp", "joint", "asym", "asym+joint"])
def test_block_decomposition_equals_whole_film(gpu, film1080, spec_kw, variant):
    """What 4 GPUs compute (2x2 blocks, each with its r-pixel halo) is bit-identical to the
    single-GPU result: per-pixel tap order does not depend on the block origin -- under the default spec (pair-symmetric
    kernel, film-anchored tiles) and under specs the one-sided LDS kernel serves (a clamped border repeats the FILM's
    edge pixels: the local image ends where the film does on exactly those sides)."""
    from statmc_amd import sharding
    fs, _ = film1080
    colour = fs.state["radiance"]["film_mean"]
    gpu.set_filter_spec(**spec_kw)
    gpu.force_filter_parts(2)      # same window-row split for the film and for the blocks
    try:
        blocks_2x2_equal_whole(gpu, fs, colour, variant)
    finally:
        gpu.force_filter_parts(0)
        gpu.set_filter_spec()


def blocks_2x2_equal_whole(gpu, fs, colour, variant):
    from statmc_amd import sharding
    whole = wf(gpu, fs, colour, torch.zeros_like(colour)).clone()
    assert gpu.last_filter_variant() == variant
    imgs = dict(mean_corr=fs.mean_corr, disc=fs.disc, colour=colour, normal=fs.g_buffer("normal"), albedo=fs.g_buffer("albedo"))
    bw, bh = W // 2, H // 2
    for rank in range(4):
        L = sharding.BlockLayout(rank, 4, bw, bh, RADIUS)
        ox, oy = L.origin
        loc = {k: v[oy - L.pt:oy + bh + L.pb, ox - L.pl:ox + bw + L.pr].contiguous() for k, v in imgs.items()}
        out = torch.zeros_like(loc["colour"])
        a, keep = gpu.make_filter_args([], [], [], [], [loc["colour"]], [loc["mean_corr"]], [loc["disc"]], [out],
                                       [loc["normal"], loc["albedo"]], g_sds=[SD_NORMAL, SD_ALBEDO],
                                       filter_sd=FILTER_SD, radius=RADIUS, roi=L.roi, film_origin=(ox - L.pl, oy - L.pt))
        gpu.window_filter(a, 3)
        torch.cuda.synchronize()
        assert gpu.last_filter_variant() == variant
        assert torch.equal(L.interior(out), whole[oy:oy + bh, ox:ox + bw]), rank


def test_welch_fullsize_strips_band_and_blocks(gpu, oracle, film1080):
    """Welch degrees of freedom at the full 1080p size (the pin's other candidate for the dof): the pair-symmetric kernel's
    Welch build against oracle strips <= 1e-5 (middle rows, clipped corners); the film's uniform sample count keeps every
    work item inside its quantile band (no far items); and what 4 GPUs compute -- 2 x 2 blocks, the sample counts riding in
    the 16th channel of the block + halo image -- is the single-GPU result bit for bit under a pinned split."""
    import ctypes as C
    from statmc_amd import sharding
    fs, _ = film1080
    lib = gpu.load()
    lib.statmc_debug_welch_far_items.restype = C.c_int
    colour = fs.state["radiance"]["film_mean"]
    n = fs.state["radiance"]["n"]
    spec = oracle.FilterSpec(dof=1)
    gpu.set_filter_spec(dof=1)
    gpu.force_filter_parts(2)
    try:
        fs.prepass()                                  # (Welch: the discriminator image holds s^2 / n)
        whole = wf(gpu, fs, colour, torch.zeros_like(colour)).clone()
        assert gpu.last_filter_variant() == "sym_welch" and lib.statmc_debug_welch_far_items() == 0
        mc, dc, col, nn = fs.mean_corr.cpu().numpy(), fs.disc.cpu().numpy(), colour.cpu().numpy(), n.cpu().numpy()
        gbs = [fs.g_buffer("normal").cpu().numpy(), fs.g_buffer("albedo").cpu().numpy()]
        for roi in ((0, 534, W, 542), (0, 0, 300, 6), (W - 300, H - 6, W, H)):
            x0, y0, x1, y1 = roi
            ref = oracle.filter_image(mc, dc, col, gbs, G_DR, -0.5 / FILTER_SD ** 2, RADIUS, roi=roi, spec=spec, n=nn)[y0:y1, x0:x1]
            got = whole[y0:y1, x0:x1].cpu().numpy()
            for c in range(3):
                assert rel_l2(got[..., c], ref[..., c]) <= 1e-5, (roi, c)
        # 2 x 2 blocks through 16-channel block + halo images
        bw, bh = W // 2, H // 2
        for rank in range(4):
            L = sharding.BlockLayout(rank, 4, bw, bh, RADIUS)
            ox, oy = L.origin
            cut = lambda t: t[oy - L.pt:oy + bh + L.pb, ox - L.pl:ox + bw + L.pr].contiguous()
            packed = torch.zeros(L.ph, L.pw, 16, device=DEV)
            a, keep = gpu.make_filter_args([cut(n)], [], [], [], [cut(colour)], [cut(fs.mean_corr)], [cut(fs.disc)], [torch.zeros_like(cut(colour))],
                                           [cut(fs.g_buffer("normal")), cut(fs.g_buffer("albedo"))], g_sds=[SD_NORMAL, SD_ALBEDO],
                                           filter_sd=FILTER_SD, radius=RADIUS)
            gpu.pack_filter_inputs(a, packed, 0, 0)          # (block + halo cut out of the whole film: one pack, no exchange)
            out = torch.zeros(L.ph, L.pw, 3, device=DEV)
            a2, keep2 = gpu.make_filter_args(n=[], mean=[], m2=[], m3=[], film=[], mean_corr=[], disc=[], film_filtered=[out], g_buffers=[],
                                             g_sds=[SD_NORMAL, SD_ALBEDO], filter_sd=FILTER_SD, radius=RADIUS, roi=L.roi, packed=packed,
                                             film_origin=(ox - L.pl, oy - L.pt))
            gpu.window_filter(a2, 3)
            torch.cuda.synchronize()
            assert gpu.last_filter_variant() == "sym_welch"
            assert torch.equal(L.interior(out), whole[oy:oy + bh, ox:ox + bw]), rank
    finally:
        gpu.force_filter_parts(0)
        gpu.set_filter_spec()
        fs.prepass()                                  # the module's film goes back to the default spec's discriminator


def test_default_dispatch_blocks_vs_whole_film(gpu, oracle, film1080):
    """What N GPUs compute under the DEFAULT dispatch -- every block picks the window-sweep split that fits its own shape
    (1 part for the whole 1080p film, 3 for its 1920 x 135 / 270 / 540 strips on a 256-CU device) -- against the
    one-device result: the same film to <= 1e-6 relative L2 per channel (the split regroups a pixel's 1681 terms, nothing
    else), and the same strips of the oracle to <= 1e-5.  With the split PINNED to the whole film's
    (statmc_set_filter_split(statmc_filter_split_auto(W, H, r)): the declared, per-device setting) the blocks are the
    one-device result bit for bit.  Row strips for 2 / 4 / 8 devices, 2 x 2 and 4 x 2 blocks."""
    from statmc_amd import sharding
    fs, _ = film1080
    colour = fs.state["radiance"]["film_mean"]
    assert gpu.get_filter_split() == 0
    whole = wf(gpu, fs, colour, torch.zeros_like(colour)).clone()
    assert gpu.last_filter_variant() == "sym_r20"
    import ctypes as C

    def split_used():     # (parts, parts of the tail rows, tail rows): 0, 0 = every tile the same
        hi, rows = C.c_int(0), C.c_int(0)
        gpu.load().statmc_debug_last_filter_tail(C.byref(hi), C.byref(rows))
        return (gpu.load().statmc_debug_last_filter_parts(), hi.value, rows.value)
    parts_whole = gpu.load().statmc_debug_last_filter_parts()
    assert parts_whole == gpu.filter_split_auto(W, H, RADIUS) and split_used() == (parts_whole, 0, 0)
    imgs = dict(mean_corr=fs.mean_corr, disc=fs.disc, colour=colour, normal=fs.g_buffer("normal"), albedo=fs.g_buffer("albedo"))
    seam = (0, 536, W, 544)                         # 8 full-width rows across the y = 540 seam of every grid below
    ref = oracle_strip(oracle, fs, colour, seam)

    def assemble(gx, gy):
        out_film, used = torch.empty_like(whole), set()
        bw, bh = W // gx, H // gy
        for rank in range(gx * gy):
            L = sharding.BlockLayout(rank, gx * gy, bw, bh, RADIUS, grid=(gx, gy))
            ox, oy = L.origin
            loc = {k: v[oy - L.pt:oy + bh + L.pb, ox - L.pl:ox + bw + L.pr].contiguous() for k, v in imgs.items()}
            out = torch.zeros_like(loc["colour"])
            a, keep = gpu.make_filter_args([], [], [], [], [loc["colour"]], [loc["mean_corr"]], [loc["disc"]], [out],
                                           [loc["normal"], loc["albedo"]], g_sds=[SD_NORMAL, SD_ALBEDO],
                                           filter_sd=FILTER_SD, radius=RADIUS, roi=L.roi, film_origin=(ox - L.pl, oy - L.pt))
            gpu.window_filter(a, 3)
            torch.cuda.synchronize()
            assert gpu.last_filter_variant() == "sym_r20"
            used.add(split_used())
            out_film[oy:oy + bh, ox:ox + bw] = L.interior(out)
        return out_film, used

    differing = 0
    for gx, gy in ((1, 2), (1, 4), (1, 8), (2, 2), (4, 2)):
        got, used = assemble(gx, gy)
        differing += used != {(parts_whole, 0, 0)}
        g, w = got.cpu().numpy(), whole.cpu().numpy()
        for c in range(3):
            assert rel_l2(g[..., c], w[..., c]) <= 1e-6, (gx, gy, c, used)
            assert rel_l2(g[536:544, :, c], ref[..., c]) <= 1e-5, (gx, gy, c)
    assert differing > 0        # the case the bound is stated for: some grid did run under another split than the film
    gpu.set_filter_split(parts_whole)
    try:
        for gx, gy in ((1, 8), (4, 2)):
            got, used = assemble(gx, gy)
            assert used == {(parts_whole, 0, 0)} and torch.equal(got, whole), (gx, gy)
    finally:
        gpu.set_filter_split(0)


# ====================================================================== every BASELINE.json config
# configs[0] 256x256 / 16 spp, configs[1] 1280x720 / 64 spp, configs[2] 1920x1080 / 256 spp,
# configs[3] 1920x1080 / 64 spp cut 2x2, configs[4] 3840x2160 cut 4x2 (spp bounded here: the per-pixel update
# does not depend on how many samples came before, test_batches_chain_exactly / the 256-spp test below).
def oracle_strip(oracle, fs, colour, roi):
    x0, y0, x1, y1 = roi
    return oracle.filter_image(fs.mean_corr.cpu().numpy(), fs.disc.cpu().numpy(), colour.cpu().numpy(),
                               [fs.g_buffer("normal").cpu().numpy(), fs.g_buffer("albedo").cpu().numpy()], G_DR,
                               -0.5 / FILTER_SD ** 2, RADIUS, roi=roi)[y0:y1, x0:x1]


def check_strips(gpu, oracle, fs, rois, also_generic=True):
    colour = fs.state["radiance"]["film_mean"]
    whole = wf(gpu, fs, colour, torch.zeros_like(colour)).clone()
    variant = gpu.last_filter_variant()
    assert variant == "sym_r20", variant
    assert torch.isfinite(whole).all()
    for roi in rois:
        x0, y0, x1, y1 = roi
        ref = oracle_strip(oracle, fs, colour, roi)
        got = whole[y0:y1, x0:x1].cpu().numpy()
        for c in range(3):
            assert rel_l2(got[..., c], ref[..., c]) <= 1e-5, (roi, c)
        if also_generic:
            b = wf(gpu, fs, colour, torch.zeros_like(colour), roi=roi, force=1)[y0:y1, x0:x1].cpu().numpy()
            assert rel_l2(b, ref) <= 1e-5, roi
            # the ROI call of the LDS kernel gives what the whole-film call gave
            a = wf(gpu, fs, colour, torch.zeros_like(colour), roi=roi)[y0:y1, x0:x1].cpu().numpy()
            assert rel_l2(a, got) <= 1e-6, roi
    return whole


def blocks_equal_whole(gpu, fs, gx, gy, parts=2):
    """gx x gy block decomposition (each block + its r-pixel halo filtered on its own) == whole film, bit for bit."""
    from statmc_amd import sharding
    colour = fs.state["radiance"]["film_mean"]
    Wf, Hf = fs.width, fs.height
    gpu.force_filter_parts(parts)
    try:
        whole = wf(gpu, fs, colour, torch.zeros_like(colour)).clone()
        imgs = dict(mean_corr=fs.mean_corr, disc=fs.disc, colour=colour, normal=fs.g_buffer("normal"), albedo=fs.g_buffer("albedo"))
        bw, bh = Wf // gx, Hf // gy
        for rank in range(gx * gy):
            L = sharding.BlockLayout(rank, gx * gy, bw, bh, RADIUS, grid=(gx, gy))
            ox, oy = L.origin
            loc = {k: v[oy - L.pt:oy + bh + L.pb, ox - L.pl:ox + bw + L.pr].contiguous() for k, v in imgs.items()}
            out = torch.zeros_like(loc["colour"])
            a, keep = gpu.make_filter_args([], [], [], [], [loc["colour"]], [loc["mean_corr"]], [loc["disc"]], [out],
                                           [loc["normal"], loc["albedo"]], g_sds=[SD_NORMAL, SD_ALBEDO],
                                           filter_sd=FILTER_SD, radius=RADIUS, roi=L.roi, film_origin=(ox - L.pl, oy - L.pt))
            gpu.window_filter(a, 3)
            torch.cuda.synchronize()
            assert torch.equal(L.interior(out), whole[oy:oy + bh, ox:ox + bw]), (gx, gy, rank)
    finally:
        gpu.force_filter_parts(0)


def make_film(gpu, W, H, spp, seed=1, types=("radiance", "normal", "albedo"), chunk=32):
    from statmc_amd import film, synthetic
    scene = synthetic.Scene(W, H, seed=seed, device=DEV)
    fs = film.FilmStats(W, H, DEV, types=types)
    for s0 in range(0, spp, chunk):
        fs.accumulate(scene.samples(min(chunk, spp - s0), seed=seed * 1000 + s0, features=types))
    fs.prepass()
    torch.cuda.synchronize()
    return fs


def test_config0_256x256_16spp_end_to_end(gpu, oracle):
    """configs[0]: the CPU-plumbing shape, HIP against the oracle over the whole film, every stage."""
    from conftest import make_case
    from statmc_amd import film
    W0, H0, S = 256, 256, 16
    scene, smp, st = make_case(W0, H0, S, seed=4)
    fs = film.FilmStats(W0, H0, DEV)
    fs.accumulate({k: torch.from_numpy(v).to(DEV) for k, v in smp.items()})
    out = fs.denoise().cpu().numpy()
    torch.cuda.synchronize()
    rad = st["radiance"]
    assert np.array_equal(fs.state["radiance"]["n"].cpu().numpy(), rad["n"])
    assert np.array_equal(fs.state["radiance"]["film_mean"].cpu().numpy(), rad["film_mean"])
    mc, dc = oracle.prepass(rad["n"], rad["mean"], rad["m2"], rad["m3"])
    ref = oracle.filter_image(mc, dc, rad["film_mean"], [st["normal"]["mean"], st["albedo"]["mean"]], G_DR,
                              -0.5 / FILTER_SD ** 2, RADIUS)
    for c in range(3):
        assert rel_l2(out[..., c], ref[..., c]) <= 1e-5, c
    assert rel_l2(fs.mean_corr.cpu().numpy(), mc) <= 1e-5


def test_config1_1280x720_64spp(gpu, oracle):
    """configs[1]: 1280x720 / 64 spp.  5 tile columns: the shape whose default dispatch splits the window sweep
    over several workgroups per tile and sums the parts in a second kernel."""
    fs = make_film(gpu, 1280, 720, 64, seed=3)
    assert int(fs.state["radiance"]["n"].min()) == 64
    check_strips(gpu, oracle, fs, [(0, 352, 1280, 364), (0, 0, 300, 8), (1280 - 300, 720 - 8, 1280, 720), (1000, 0, 1280, 6)])
    parts = gpu.load().statmc_debug_last_filter_parts()
    assert parts >= 1
    # 10 x 90 = 900 tiles leave the fourth round of a 256-CU device half empty: the automatic split gives the last tile
    # rows more parts (tail split, round 4) -- and a call for a band of rows across that boundary is the whole-image
    # call's result bit for bit (the split belongs to the image, not to the region)
    import ctypes as C
    hi, rows = C.c_int(0), C.c_int(0)
    gpu.load().statmc_debug_last_filter_tail(C.byref(hi), C.byref(rows))
    if gpu.load().statmc_device_cus() == 256:
        assert parts == 1 and hi.value >= 2 and 0 < rows.value < 90, (parts, hi.value, rows.value)
    colour = fs.state["radiance"]["film_mean"]
    whole = wf(gpu, fs, colour, torch.zeros_like(colour)).clone()
    y_split = 720 - 8 * rows.value if rows.value else 360
    for roi in ((0, max(0, y_split - 40), 1280, min(720, y_split + 48)), (256, 0, 1024, 720), (0, 704, 1280, 720)):
        x0, y0, x1, y1 = roi
        part = wf(gpu, fs, colour, torch.zeros_like(colour), roi=roi)
        assert torch.equal(part[y0:y1, x0:x1], whole[y0:y1, x0:x1]), roi
    ref = oracle_strip(oracle, fs, colour, (0, 704, 1280, 720))          # the image's last rows: inside the tail
    for c in range(3):
        assert rel_l2(whole[704:720, :, c].cpu().numpy(), ref[..., c]) <= 1e-5, c
    # any other split of the sweep agrees with the default one
    colour = fs.state["radiance"]["film_mean"]
    base = wf(gpu, fs, colour, torch.zeros_like(colour)).clone()
    for forced in (1, 3):
        gpu.force_filter_parts(forced)
        try:
            other = wf(gpu, fs, colour, torch.zeros_like(colour))
        finally:
            gpu.force_filter_parts(0)
        assert rel_l2(other.cpu().numpy(), base.cpu().numpy()) <= 1e-6, forced


def test_config2_1080p_256spp(gpu, oracle):
    """configs[2] at its full sample count: one 256-sample launch == the reference's schedule 4+4+8+...+128
    chained (bit for bit), then strips of the filter against the oracle."""
    from statmc_amd import film, synthetic
    types = synthetic.FEATURES
    scene = synthetic.Scene(W, H, seed=1, device=DEV)
    smp = {t: [] for t in types}
    for s0 in range(0, 256, 32):
        part = scene.samples(32, seed=7000 + s0, features=types)
        for t in types:
            smp[t].append(part[t])
    smp = {t: torch.cat(v, dim=0) for t, v in smp.items()}
    one = film.FilmStats(W, H, DEV, types=types)
    one.accumulate(smp)
    chained = film.FilmStats(W, H, DEV, types=types)
    s0 = 0
    for b in synthetic.sample_schedule(256):
        chained.accumulate({t: v[s0:s0 + b] for t, v in smp.items()})
        s0 += b
    torch.cuda.synchronize()
    assert s0 == 256
    for t in types:
        for k, v in one.state[t].items():
            if v is not None:
                assert torch.equal(v, chained.state[t][k]), (t, k)
    assert int(one.state["radiance"]["n"].min()) == 256 == int(one.state["materialid"]["n"].max())
    x = smp["radiance"][:, 700:704].double()
    assert rel_l2(one.state["radiance"]["film_mean"][700:704].cpu().numpy(), x.mean(0).cpu().numpy()) < 1e-6
    del smp, chained
    one.prepass()
    check_strips(gpu, oracle, one, [(0, 536, W, 544), (W - 280, H - 6, W, H)], also_generic=False)


def test_config3_1080p_64spp_2x2_blocks(gpu):
    """configs[3]: one 1920x1080 film cut 2x2 (960x540 blocks + halo) == the whole film, bit for bit."""
    fs = make_film(gpu, W, H, 64, seed=5)
    blocks_equal_whole(gpu, fs, 2, 2)


def test_block_decomposition_at_the_shipped_small_radius(gpu, oracle):
    """scenes/render-denoise-glass-caustics.pbrt (filterradius 6, filtersd 3) on the pair-symmetric kernel's runtime-radius
    build: strips against the oracle, and 2 x 2 / 1 x 4 block decompositions (block + 6-pixel halo each) == the whole
    film bit for bit -- the film-anchored tile grid and the fixed gather order do not depend on the radius."""
    from statmc_amd import sharding
    r, sd = 6, 3.0
    fs = make_film(gpu, 1280, 720, 16, seed=11)
    colour = fs.state["radiance"]["film_mean"]

    def run(imgs, roi=None, origin=None, parts=0):
        out = torch.zeros_like(imgs["colour"])
        a, keep = gpu.make_filter_args([], [], [], [], [imgs["colour"]], [imgs["mean_corr"]], [imgs["disc"]], [out],
                                       [imgs["normal"], imgs["albedo"]], g_sds=[SD_NORMAL, SD_ALBEDO], filter_sd=sd, radius=r,
                                       roi=roi, film_origin=origin)
        gpu.force_filter_parts(parts)
        try:
            gpu.window_filter(a, 3)
            torch.cuda.synchronize()
        finally:
            gpu.force_filter_parts(0)
        assert gpu.last_filter_variant() == "sym_rt", gpu.last_filter_variant()
        return out

    imgs = dict(mean_corr=fs.mean_corr, disc=fs.disc, colour=colour, normal=fs.g_buffer("normal"), albedo=fs.g_buffer("albedo"))
    whole = run(imgs, parts=2)
    for roi in ((0, 352, 1280, 360), (0, 0, 300, 8), (1280 - 300, 720 - 8, 1280, 720)):
        x0, y0, x1, y1 = roi
        ref = oracle.filter_image(fs.mean_corr.cpu().numpy(), fs.disc.cpu().numpy(), colour.cpu().numpy(),
                                  [imgs["normal"].cpu().numpy(), imgs["albedo"].cpu().numpy()], G_DR, -0.5 / sd ** 2, r, roi=roi)[y0:y1, x0:x1]
        got = whole[y0:y1, x0:x1].cpu().numpy()
        for c in range(3):
            assert rel_l2(got[..., c], ref[..., c]) <= 1e-5, (roi, c)
    for gx, gy in ((2, 2), (1, 4)):
        bw, bh = 1280 // gx, 720 // gy
        for rank in range(gx * gy):
            L = sharding.BlockLayout(rank, gx * gy, bw, bh, r, grid=(gx, gy))
            ox, oy = L.origin
            loc = {k: v[oy - L.pt:oy + bh + L.pb, ox - L.pl:ox + bw + L.pr].contiguous() for k, v in imgs.items()}
            out = run(loc, roi=L.roi, origin=(ox - L.pl, oy - L.pt), parts=2)
            assert torch.equal(L.interior(out), whole[oy:oy + bh, ox:ox + bw]), (gx, gy, rank)


def test_config4_4k_strips_and_4x2_blocks(gpu, oracle):
    """configs[4]: 3840x2160.  Strip + corners against the oracle, and the 8-GPU partition (4x2 blocks of
    960x1080) against the whole film."""
    fs = make_film(gpu, 3840, 2160, 16, seed=6)
    check_strips(gpu, oracle, fs, [(0, 1076, 3840, 1082), (0, 0, 280, 6), (3840 - 280, 2160 - 6, 3840, 2160)], also_generic=False)
    blocks_equal_whole(gpu, fs, 4, 2)
