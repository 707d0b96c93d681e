"""Offline denoise of a statistics dump through the C++ host side (tools/statmc_denoise, the
counterpart of `pbrt --denoise`, statpath.cpp:456-550): PFM dumps in, film-f.pfm out."""
import glob
import os
import subprocess

import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "case_*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_dump_to_denoised_image(gpu, tmp_path, path):
    from statmc_amd import build, pfm
    exe = build.build_tools()
    g = np.load(path)
    spp = int(g["spp"])
    stem = str(tmp_path / "scene")
    dump = {"film": g["film_mean"], "t0-b0-n": g["n"], "t0-b0-mean": g["mean"], "t0-b0-m2": g["m2"],
            "t0-b0-m3": g["m3"], "t1-b0-film-mean": g["normal_mean"], "t2-b0-film-mean": g["albedo_mean"]}
    for name, img in dump.items():                    # the outputregex of scenes/render-for-ours.pbrt:24
        pfm.write_pfm("%s-%d-%s.pfm" % (stem, spp, name), img)
    out = subprocess.run([exe, "--stem", stem, "--spp", str(spp), "--filtersd", str(float(g["filter_sd"])),
                          "--filterradius", str(int(g["radius"])), "--warmup",
                          "--output", "film-f,t0-b0-mean-corr,t0-b0-discriminator"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "HIP time [ns]:" in out.stdout and "Warm-Up" in out.stdout
    film_f = pfm.read_pfm("%s-%d-film-f.pfm" % (stem, spp))
    assert np.array_equal(pfm.read_pfm("%s-%d-t0-b0-mean-corr.pfm" % (stem, spp)), g["mean_corr"])
    assert np.array_equal(pfm.read_pfm("%s-%d-t0-b0-discriminator.pfm" % (stem, spp)), g["discriminator"])
    for c in range(3):
        assert rel_l2(film_f[..., c], g["film_f"][..., c]) <= 1e-5


def test_compare_against_reference_dumps_and_custom_quantiles(gpu, tmp_path):
    """--compare: the route by which real StatMC output dumps (film-f written by the CUDA build) are
    checked against this build -- here the golden outputs stand in for them.  --tquantiles loads a
    caller's t table; with the built-in table written to a file the result must not change."""
    import re
    from statmc_amd import build, pfm
    from oracle import oracle
    exe = build.build_tools()
    g = np.load(GOLDEN[0])
    spp = int(g["spp"])
    stem, ref = str(tmp_path / "scene"), str(tmp_path / "ref")
    dump = {"film": g["film_mean"], "t0-b0-n": g["n"], "t0-b0-mean": g["mean"], "t0-b0-m2": g["m2"],
            "t0-b0-m3": g["m3"], "t1-b0-film-mean": g["normal_mean"], "t2-b0-film-mean": g["albedo_mean"]}
    for name, img in dump.items():
        pfm.write_pfm("%s-%d-%s.pfm" % (stem, spp, name), img)
    pfm.write_pfm("%s-%d-film-f.pfm" % (ref, spp), g["film_f"])
    pfm.write_pfm("%s-%d-t0-b0-discriminator.pfm" % (ref, spp), g["discriminator"])
    table = tmp_path / "tq.txt"
    table.write_text("\n".join("%.9g" % oracle.t_quantile(0, dof) for dof in range(1, 4097)))
    base = [exe, "--stem", stem, "--spp", str(spp), "--filtersd", str(float(g["filter_sd"])), "--filterradius",
            str(int(g["radius"])), "--output", "film-f,t0-b0-discriminator", "--compare", ref]
    for extra in ([], ["--tquantiles", str(table)]):
        out = subprocess.run(base + extra, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        errs = {(m.group(1), int(m.group(2))): float(m.group(3))
                for m in re.finditer(r"compare (\S+) ch(\d) rel_l2 (\S+)", out.stdout)}
        assert set(errs) == {("film-f", 0), ("film-f", 1), ("film-f", 2)} | {("t0-b0-discriminator", c) for c in range(3)}
        assert all(errs[("t0-b0-discriminator", c)] == 0.0 for c in range(3))
        assert all(errs[("film-f", c)] <= 1e-5 for c in range(3))
    # a different table must change the discriminator (the option is live)
    table.write_text("\n".join("%.9g" % (2.0 * oracle.t_quantile(0, dof)) for dof in range(1, 4097)))
    out = subprocess.run(base + ["--tquantiles", str(table)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert float(re.search(r"compare t0-b0-discriminator ch0 rel_l2 (\S+)", out.stdout).group(1)) > 0.5


def test_fit_spec_finds_the_spec_that_wrote_the_dumps(gpu, tmp_path):
    """tools/fit_spec.py: with output dumps of the CUDA denoiser in hand, pinning this build is one command.  Stand-in
    for those dumps here: the oracle under a non-default spec (joint channel rule, one-sided quantiles, alpha 0.05).
    The tool must rank exactly that spec first, within BASELINE's 1e-5."""
    import re
    import sys
    from statmc_amd import build, pfm
    from oracle import oracle
    build.build_tools()
    g = np.load(GOLDEN[1])                                    # caustics_r6: small window, quick to sweep
    spp = int(g["spp"])
    n = g["n"].copy()
    n[3, 5] = 1                                               # one pixel where the n < 2 rule matters
    spec = oracle.FilterSpec(channel_rule=oracle.CHANNELS_JOINT, sides=oracle.SIDES_ONE)
    mc, dc = oracle.prepass(n, g["mean"], g["m2"], g["m3"], alpha_index=2, spec=spec)
    film_f = oracle.filter_image(mc, dc, g["film_mean"], [g["normal_mean"], g["albedo_mean"]], [-50.0, -1250.0],
                                 -0.5 / float(g["filter_sd"]) ** 2, int(g["radius"]), spec=spec, n=n, alpha_index=2)
    stem, ref = str(tmp_path / "scene"), str(tmp_path / "cuda")
    for name, img in {"film": g["film_mean"], "t0-b0-n": n, "t0-b0-mean": g["mean"], "t0-b0-m2": g["m2"], "t0-b0-m3": g["m3"],
                      "t1-b0-film-mean": g["normal_mean"], "t2-b0-film-mean": g["albedo_mean"]}.items():
        pfm.write_pfm("%s-%d-%s.pfm" % (stem, spp, name), img)
    for name, img in {"film-f": film_f, "t0-b0-mean-corr": mc, "t0-b0-discriminator": dc}.items():
        pfm.write_pfm("%s-%d-%s.pfm" % (ref, spp, name), img)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fit_spec.py"), "--stem", stem, "--ref", ref, "--spp", str(spp),
                          "--filtersd", str(float(g["filter_sd"])), "--filterradius", str(int(g["radius"])), "--quick"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    best = re.search(r"best: significance (\d), --spec (\S+)\s+\(film-f worst channel (\S+);", out.stdout)
    assert best, out.stdout
    assert best.group(1) == "2" and best.group(2) == "gate=sym,channels=joint,sides=one", out.stdout
    assert float(best.group(3)) <= 1e-5
    rows = [l for l in out.stdout.splitlines() if re.match(r"^\d\.\d+e", l)]
    assert len(rows) == 3 * 12                                                   # 3 levels x (3 gate forms x 2^2) specs
    assert float(rows[1].split()[0]) > 1e-5 or "gate=asym,channels=joint,sides=one" in rows[1]   # only the rounding-twin gate form ties


def test_pin_from_dumps_end_to_end(gpu, tmp_path):
    """tools/pin_from_dumps.sh on a directory of dumps: the oracle under a non-default spec (pooled channels, clamped
    border, n < 2 excluded, alpha 0.002) stands in for the CUDA build.  Two stems; the tool must print the table, find
    exactly that spec over the full 96 x 3 grid and write it as the pinned default (to a scratch header: the tree's
    own default is not touched, nothing is rebuilt)."""
    import re
    import sys
    from statmc_amd import build, pfm
    from oracle import oracle
    build.build_tools()
    spec = oracle.FilterSpec(channel_rule=oracle.CHANNELS_JOINT, border=oracle.BORDER_CLAMP, small_n=oracle.SMALL_N_EXCLUDE)
    for stem, gi in (("sceneA", 1), ("sceneB", 1)):
        g = np.load(GOLDEN[gi])
        spp = int(g["spp"])
        n = g["n"].copy()
        n[3, 5] = 1
        if stem == "sceneB":
            n[10, 2:6] = 0
        mc, dc = oracle.prepass(n, g["mean"], g["m2"], g["m3"], alpha_index=1, spec=spec)
        film_f = oracle.filter_image(mc, dc, g["film_mean"], [g["normal_mean"], g["albedo_mean"]], [-50.0, -1250.0],
                                     -0.5 / float(g["filter_sd"]) ** 2, int(g["radius"]), spec=spec, n=n, alpha_index=1)
        for name, img in {"film": g["film_mean"], "t0-b0-n": n, "t0-b0-mean": g["mean"], "t0-b0-m2": g["m2"], "t0-b0-m3": g["m3"],
                          "t1-b0-film-mean": g["normal_mean"], "t2-b0-film-mean": g["albedo_mean"], "film-f": film_f,
                          "t0-b0-mean-corr": mc, "t0-b0-discriminator": dc}.items():
            pfm.write_pfm(str(tmp_path / ("%s-%d-%s.pfm" % (stem, spp, name))), img)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = tmp_path / "pinned.h"
    out = subprocess.run([os.path.join(root, "tools", "pin_from_dumps.sh"), str(tmp_path), "--filtersd", str(float(g["filter_sd"])),
                          "--filterradius", str(int(g["radius"])), "--header", str(header), "--no-rebuild"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    win = re.search(r"winner: significance (\d), spec (\S+): film-f (\S+) ", out.stdout)
    assert win and win.group(1) == "1" and float(win.group(3)) <= 1e-5, out.stdout
    # (the two forms of the gate differ by one rounding of a threshold: either may come out on top)
    assert win.group(2) in ("gate=sym,channels=joint,sides=two,dof=pixel,border=clamp,small_n=exclude",
                            "gate=asym,channels=joint,sides=two,dof=pixel,border=clamp,small_n=exclude"), out.stdout
    h = header.read_text()
    assert re.search(r"#define STATMC_PINNED_SPEC \{[01], 1, 0, 0, 1, 1\}", h) and "#define STATMC_PINNED_SIGNIFICANCE 1" in h
    table = (tmp_path / "pin_table.txt").read_text()
    assert table.count("==== scene") == 2 and len([l for l in table.splitlines() if re.match(r"^\d\.\d+e", l)]) == 2 * 3 * 96
    # a directory whose outputs no spec reproduces is reported, not pinned
    pfm.write_pfm(str(tmp_path / ("sceneA-%d-film-f.pfm" % spp)), film_f * 1.01)
    out = subprocess.run([os.path.join(root, "tools", "pin_from_dumps.sh"), str(tmp_path), "--filtersd", str(float(g["filter_sd"])),
                          "--filterradius", str(int(g["radius"])), "--header", str(tmp_path / "other.h"), "--no-rebuild", "--quick"],
                         capture_output=True, text=True)
    assert out.returncode == 3 and "NOT pinned" in out.stdout and not (tmp_path / "other.h").exists()


@pytest.mark.parametrize("grid", ["2x1", "2x2", "1x3"])
def test_sharded_denoise_through_the_cpp_host(gpu, tmp_path, grid):
    """The C++ host side without Python in the data path: statmc::FilmShards cuts the Estimator's images into film
    blocks, runs pre-pass + pack per block, statmc_halo_exchange (device-to-device copies ordered by events), the
    window filter per block, and pastes film-f together -- bit-identical to the unsharded Estimator::Denoise()."""
    from conftest import make_case
    from statmc_amd import build, pfm
    exe = build.build_tools()
    W, H, spp = 144, 96, 8
    _, smp, st = make_case(W, H, spp, seed=9)
    rad = st["radiance"]
    stem = str(tmp_path / "scene")
    for name, img in {"film": rad["film_mean"], "t0-b0-n": rad["n"], "t0-b0-mean": rad["mean"], "t0-b0-m2": rad["m2"],
                      "t0-b0-m3": rad["m3"], "t1-b0-film-mean": st["normal"]["mean"], "t2-b0-film-mean": st["albedo"]["mean"]}.items():
        pfm.write_pfm("%s-%d-%s.pfm" % (stem, spp, name), img)
    outs = {}
    for g in (None, grid):
        cmd = [exe, "--stem", stem, "--spp", str(spp), "--output", "film-f", "--parts", "2"] + (["--grid", g] if g else [])
        out = subprocess.run(cmd, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        outs[g] = pfm.read_pfm("%s-%d-film-f.pfm" % (stem, spp))
    assert np.isfinite(outs[None]).all() and np.abs(outs[None] - rad["film_mean"]).max() > 0
    assert np.array_equal(outs[grid], outs[None])
    # ... and with every device image a block of the placed allocator (statmc::usePlacedMemory(): the Estimator's tables and the
    # blocks' images; the cross-DEVICE case is tests/test_multidevice_gpu.py)
    out = subprocess.run([exe, "--stem", stem, "--spp", str(spp), "--output", "film-f", "--parts", "2", "--grid", grid, "--placed"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert np.array_equal(pfm.read_pfm("%s-%d-film-f.pfm" % (stem, spp)), outs[None])
    if grid == "2x2":     # Welch degrees of freedom: the sample count rides in the block + halo image's 16th channel
        for g in (None, grid):
            cmd = [exe, "--stem", stem, "--spp", str(spp), "--output", "film-f", "--parts", "2", "--spec", "dof=welch"] + (["--grid", g] if g else [])
            out = subprocess.run(cmd, capture_output=True, text=True)
            assert out.returncode == 0, out.stderr
            outs["welch", g] = pfm.read_pfm("%s-%d-film-f.pfm" % (stem, spp))
        assert np.array_equal(outs["welch", grid], outs["welch", None])
        assert not np.array_equal(outs["welch", None], outs[None])
        # ... and with depth among the G-buffers (`filterbuffers ["depth" "normal" "albedo"]`): the 18-channel image, the eight-plane
        # Welch builds; stat types in the Estimator's order: t1 depth, t2 normal, t3 albedo
        _, smp8, st8 = make_case(W, H, spp, seed=9, features=("radiance", "normal", "albedo", "depth"))
        stem8 = str(tmp_path / "scene8")
        rad8 = st8["radiance"]
        for name, img in {"film": rad8["film_mean"], "t0-b0-n": rad8["n"], "t0-b0-mean": rad8["mean"], "t0-b0-m2": rad8["m2"],
                          "t0-b0-m3": rad8["m3"], "t1-b0-film-mean": st8["depth"]["mean"], "t2-b0-film-mean": st8["normal"]["mean"],
                          "t3-b0-film-mean": st8["albedo"]["mean"]}.items():
            pfm.write_pfm("%s-%d-%s.pfm" % (stem8, spp, name), img)
        for g in (None, grid):
            cmd = [exe, "--stem", stem8, "--spp", str(spp), "--output", "film-f", "--parts", "2", "--spec", "dof=welch",
                   "--filterbuffers", "depth,normal,albedo", "--filterbuffersds", "2.0,0.1,0.02"] + (["--grid", g] if g else [])
            out = subprocess.run(cmd, capture_output=True, text=True)
            assert out.returncode == 0, out.stderr
            outs["welch8", g] = pfm.read_pfm("%s-%d-film-f.pfm" % (stem8, spp))
        assert np.isfinite(outs["welch8", None]).all() and np.abs(outs["welch8", None] - rad8["film_mean"]).max() > 0
        assert np.array_equal(outs["welch8", grid], outs["welch8", None])


def test_cv_adaptor_matches_the_library(gpu, tmp_path):
    """include/statmc_cv.hpp -- the cv:: names the reference's statistics path uses, on top of the C ABI -- driven the
    way the reference's Estimator drives OpenCV (Buffers, uploaded PtrStepSzb tables, filter<float3> with the argument
    list of estimator.cpp:465-487, PFM in BGR order): same film-f as the golden vector."""
    from statmc_amd import build, pfm
    build.build_tools()
    g = np.load([p for p in GOLDEN if "default_r20" in p][0])     # the binary runs the shipped sd 10 / radius 20
    spp = int(g["spp"])
    stem = str(tmp_path / "scene")
    for name, img in {"film": g["film_mean"], "t0-b0-n": g["n"], "t0-b0-mean": g["mean"], "t0-b0-m2": g["m2"], "t0-b0-m3": g["m3"],
                      "t1-b0-film-mean": g["normal_mean"], "t2-b0-film-mean": g["albedo_mean"]}.items():
        pfm.write_pfm("%s-%d-%s.pfm" % (stem, spp, name), img)
    out_path = str(tmp_path / "film-f.pfm")
    out = subprocess.run([build.CV_ADAPTOR_BIN, stem, str(spp), out_path], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stderr)
    assert "ok 40x28 dumps 7" in out.stdout and "bands 0" in out.stdout       # too small to cut into bands
    film_f = pfm.read_pfm(out_path)
    for c in range(3):
        assert rel_l2(film_f[..., c], g["film_f"][..., c]) <= 1e-5


@pytest.mark.parametrize("config,W,H", [("denoise", 640, 600), ("acrr", 328, 520), ("denoise", 200, 130), ("denoise", 1920, 1080)],
                         ids=["rgb-640x600", "acrr-float-328x520", "rgb-200x130", "rgb-1080p"])
def test_band_pipeline_gives_the_same_bits(gpu, tmp_path, config, W, H):
    """Upload / Denoise / Download as a pipeline of row bands on three streams (Estimator::SetPipelineBands): every
    output -- denoised images, corrected means, discriminators -- is bit-identical to the one-stream sequence, for RGB
    and for float buffers, for the automatic band count and for forced ones (uneven bands, bands at the minimum
    height)."""
    import re
    import torch
    from statmc_amd import build, film, pfm, synthetic
    exe = build.build_tools()
    spp = 8
    scene = synthetic.Scene(W, H, seed=3, device=torch.device("cuda:0"))
    fs = film.FilmStats(W, H, torch.device("cuda:0"))
    fs.accumulate(scene.samples(spp, seed=5, features=("radiance", "normal", "albedo")))
    torch.cuda.synchronize()
    rad = fs.state["radiance"]
    stem = str(tmp_path / "scene")
    cpu = lambda t: t.cpu().numpy()
    if config == "denoise":
        dump = {"film": cpu(rad["film_mean"]), "t0-b0-n": cpu(rad["n"]), "t0-b0-mean": cpu(rad["mean"]), "t0-b0-m2": cpu(rad["m2"]),
                "t0-b0-m3": cpu(rad["m3"]), "t1-b0-film-mean": cpu(fs.g_buffer("normal")), "t2-b0-film-mean": cpu(fs.g_buffer("albedo"))}
        outputs = "film-f,t0-b0-mean-corr,t0-b0-discriminator"
    else:   # ACRR: five luminance buffers (bounces), untransformed: mean == film-mean
        dump = {"film": cpu(rad["film_mean"]), "t1-b0-film-mean": cpu(fs.g_buffer("normal")), "t2-b0-film-mean": cpu(fs.g_buffer("albedo"))}
        for b in range(5):
            k = 1.0 / (1 + b)
            lum = cpu(rad["film_mean"]).mean(axis=2, keepdims=True).astype(np.float32) * np.float32(k)
            dump.update({"t0-b%d-n" % b: cpu(rad["n"]), "t0-b%d-film-mean" % b: lum, "t0-b%d-mean" % b: lum,
                         "t0-b%d-m2" % b: (cpu(rad["m2"]).mean(axis=2, keepdims=True) * k * k).astype(np.float32),
                         "t0-b%d-m3" % b: (cpu(rad["m3"]).mean(axis=2, keepdims=True) * k ** 3).astype(np.float32)})
        outputs = ",".join("t0-b%d-film-mean-f" % b for b in range(5)) + ",t0-b2-discriminator"
    for name, img in dump.items():
        pfm.write_pfm("%s-%d-%s.pfm" % (stem, spp, name), img)
    results = {}
    # (bands, copy-engine queues of the copies in: 1 or 2)
    variants = ((1, 1), (0, 1), (2, 1), (3, 1), (8, 1), ((0, 2), 2), ((3, 2), 2))
    if W * H > 1000000:   # the 1080p case is there for the automatic plan (five round-fitted bands): fewer variants, the dumps are large
        variants = ((1, 1), (0, 1), (8, 1), ((0, 2), 2))
    for bands, queues in variants:
        n_bands = bands[0] if isinstance(bands, tuple) else bands
        out = subprocess.run([exe, "--stem", stem, "--spp", str(spp), "--config", config, "--bands", str(n_bands), "--output", outputs],
                             capture_output=True, text=True, env=dict(os.environ, STATMC_UPLOAD_QUEUES=str(queues)))
        assert out.returncode == 0, out.stderr
        used = int(re.search(r"pipeline bands: (\d+)", out.stdout).group(1))
        results[bands] = (used, {n: pfm.read_pfm("%s-%d-%s.pfm" % (stem, spp, n)) for n in outputs.split(",")})
    assert results[1][0] == 1
    # automatic: bands fitted to the window filter's rounds of 256 workgroups (statmc_bands.hpp): 1080p is 4 x 248 rows + 88,
    # 640 x 600 two bands (384 + 216), the others fit one round and stay whole
    assert results[0][0] == {(640, 600): 2, (328, 520): 1, (200, 130): 1, (1920, 1080): 5}[(W, H)]
    assert results[8][0] == min(8, H // 64)                              # no band shorter than 64 rows
    base = results[1][1]
    assert all(np.isfinite(v).all() and float(np.abs(v).max()) > 0 for v in base.values())
    for bands, (used, imgs) in results.items():
        for n, v in imgs.items():
            assert np.array_equal(v, base[n]), (bands, used, n)
    if (W, H) in ((640, 600), (328, 520)):
        # Welch degrees of freedom (the pair-symmetric kernel's Welch builds: a quantile band per work item, flagged items
        # computed again): the bands of the pipeline give the bits of the one-stream sequence there too, RGB and float
        welch = {}
        for n_bands in (1, 3):
            out = subprocess.run([exe, "--stem", stem, "--spp", str(spp), "--config", config, "--bands", str(n_bands), "--output", outputs,
                                  "--spec", "dof=welch"], capture_output=True, text=True)
            assert out.returncode == 0, out.stderr
            welch[n_bands] = {n: pfm.read_pfm("%s-%d-%s.pfm" % (stem, spp, n)) for n in outputs.split(",")}
        first = outputs.split(",")[0]
        assert not np.array_equal(welch[1][first], base[first])
        for n, v in welch[3].items():
            assert np.array_equal(v, welch[1][n]), n


def test_cv_adaptor_band_pipeline_same_bits(gpu, tmp_path):
    """The reference's call order -- Buffer::upload x 7, filter<float3>, Buffer::download, synchronize -- through
    include/statmc_cv.hpp: the adaptor notes the uploads, filter<T> issues them band by band and filters each band as
    it lands, the downloads follow band by band.  Same bits as the one-stream order (STATMC_CV_BANDS=1), for the
    automatic band count and a forced one."""
    import re
    import torch
    from statmc_amd import build, film, pfm, synthetic
    build.build_tools()
    W, H, spp = 640, 600, 8
    scene = synthetic.Scene(W, H, seed=4, device=torch.device("cuda:0"))
    fs = film.FilmStats(W, H, torch.device("cuda:0"))
    fs.accumulate(scene.samples(spp, seed=6, features=("radiance", "normal", "albedo")))
    torch.cuda.synchronize()
    rad = fs.state["radiance"]
    stem = str(tmp_path / "scene")
    for name, img in {"film": rad["film_mean"], "t0-b0-n": rad["n"], "t0-b0-mean": rad["mean"], "t0-b0-m2": rad["m2"], "t0-b0-m3": rad["m3"],
                      "t1-b0-film-mean": fs.g_buffer("normal"), "t2-b0-film-mean": fs.g_buffer("albedo")}.items():
        pfm.write_pfm("%s-%d-%s.pfm" % (stem, spp, name), img.cpu().numpy())
    outs = {}
    for bands in ("1", "0", "3"):
        f, m = str(tmp_path / ("film-f-%s.pfm" % bands)), str(tmp_path / ("mean-corr-%s.pfm" % bands))
        out = subprocess.run([build.CV_ADAPTOR_BIN, stem, str(spp), f, m], capture_output=True, text=True,
                             env=dict(os.environ, STATMC_CV_BANDS=bands))
        assert out.returncode == 0, (out.returncode, out.stderr)
        used = int(re.search(r"bracket_ns \d+ bands (\d+)", out.stdout).group(1))
        assert used == {"1": 0, "0": 2, "3": 3}[bands], out.stdout
        outs[bands] = (pfm.read_pfm(f), pfm.read_pfm(m))
    assert np.isfinite(outs["1"][0]).all() and float(np.abs(outs["1"][0]).max()) > 0
    for bands in ("0", "3"):
        assert np.array_equal(outs[bands][0], outs["1"][0]) and np.array_equal(outs[bands][1], outs["1"][1]), bands
