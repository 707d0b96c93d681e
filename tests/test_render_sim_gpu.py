"""The accumulation side of the drop-in, driven the way StatPathIntegrator::Render drives it
(tools/statmc_render_sim.cpp: 16 x 16 StatTile sets from Estimator::GetTiles, Add*Sample* through
member-function pointers on worker threads, Merge[Transform]Tiles per tile, the exponential
iteration schedule, Upload / Denoise / Download / Synchronize) -- checked against the CPU oracle fed
with the same synthetic samples, which are restated here from the tool's generator."""
import os
import subprocess

import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu

K_ALBEDO = np.array([[0.80, 0.25, 0.20], [0.15, 0.55, 0.85], [0.60, 0.60, 0.10]], np.float32)
K_NORMAL = np.array([[0.0, 0.0, 1.0], [0.6, -0.8, 0.0], [-0.7071, 0.0, 0.7071]], np.float32)
M32 = np.uint64(0xFFFFFFFF)


def mix(a):
    a = a.astype(np.uint64)
    a ^= a >> np.uint64(16)
    a = (a * np.uint64(0x7feb352d)) & M32
    a ^= a >> np.uint64(15)
    a = (a * np.uint64(0x846ca68b)) & M32
    a ^= a >> np.uint64(16)
    return a


def draw(seed, x, y, sample, stream):
    base = (np.uint64(0x9e3779b9) * ((x + 65536 * y).astype(np.uint64) & M32)) & M32
    k = mix(np.uint64(seed) ^ base)
    k = mix((k + np.uint64(sample)) & M32)
    return mix((k + ((np.uint64(stream) * np.uint64(0x85ebca6b)) & M32)) & M32)


def unit(key):
    return (key >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def make_samples(seed, W, H, s0, S, x0=0, y0=0):
    """[S, H, W, 3] radiance, normal, albedo samples number s0 .. s0+S-1 of every pixel of the W x H window at (x0, y0)
    (fp32, same operation order as makeSample in tools/statmc_render_sim.cpp)."""
    y, x = np.mgrid[y0:y0 + H, x0:x0 + W]
    region = ((x // 24) + (y // 20)) % 3
    e = np.float32(0.5) + np.float32(0.5) * (((x * 7 + y * 3) % 32).astype(np.float32) / np.float32(32.0))
    rad = np.zeros((S, H, W, 3), np.float32)
    nrm = np.zeros_like(rad)
    alb = np.zeros_like(rad)
    for s in range(S):
        black = (draw(seed, x, y, s0 + s, 3) & np.uint64(7)) == 0
        for c in range(3):
            u = unit(draw(seed, x, y, s0 + s, c))
            t = np.float32(4.0) * (u * u)
            rad[s, ..., c] = np.where(black, np.float32(0), (K_ALBEDO[region, c] * e) * t)
            nrm[s, ..., c] = K_NORMAL[region, c] + (unit(draw(seed, x, y, s0 + s, 4 + c)) - np.float32(0.5)) * np.float32(0.02)
            alb[s, ..., c] = K_ALBEDO[region, c] + (unit(draw(seed, x, y, s0 + s, 7 + c)) - np.float32(0.5)) * np.float32(0.01)
    return rad, nrm, alb


@pytest.mark.parametrize("W,H,stage_mb,placed", [(88, 44, 1, False), (50, 37, 2048, False), (88, 44, 1, True)],
                         ids=["vector-tiles-small-staging", "scalar-tiles", "placed-memory"])
def test_render_loop_matches_oracle(gpu, oracle, tmp_path, W, H, stage_mb, placed):
    """placed: the C++ host side with statmc::usePlacedMemory() -- device images from statmc_malloc_placed(STATE), the sample
    arenas of the device-side accumulation from statmc_malloc_placed(STREAM): the same files."""
    from statmc_amd import build, pfm
    build.build_tools()
    spp, iterations, seed, radius, sd = 4, 3, 5, 20, 10.0
    stem = str(tmp_path / "sim")
    out = subprocess.run([build.RENDER_SIM_BIN, "--width", str(W), "--height", str(H), "--spp", str(spp),
                          "--iterations", str(iterations), "--threads", "4", "--seed", str(seed), "--stem", stem,
                          "--filtersd", str(sd), "--filterradius", str(radius), "--stage-mb", str(stage_mb), "--warmup",
                          "--tilestats"] + (["--placed"] if placed else []),
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.count("Noisiest tile:") == iterations
    # --warmup: one throw-away iteration whose statistics must not survive into the real run
    assert out.stdout.count("CUDA time [ns]:") == iterations + 1 and "Iteration: 3" in out.stdout and "Warm-Up End" in out.stdout

    st = {"rad": oracle.new_state(H, W, 3), "nrm": oracle.new_state(H, W, 3), "alb": oracle.new_state(H, W, 3)}
    done = 0
    for i in range(1, iterations + 1):
        target = spp if i == 1 else spp << (i - 2)                 # statpath.cpp:272-279
        rad, nrm, alb = make_samples(seed, W, H, done, target)
        oracle.accumulate(st["rad"], rad, True, 3)
        oracle.accumulate(st["nrm"], nrm, False, 1)
        oracle.accumulate(st["alb"], alb, False, 1)
        done += target
        rd = lambda name: pfm.read_pfm("%s-%d-%s.pfm" % (stem, done, name))
        assert np.array_equal(rd("t0-b0-n"), st["rad"]["n"].astype(np.float32))
        # raw-sample moments and the plain feature means carry no sqrt: bit for bit
        assert np.array_equal(rd("t0-b0-film-mean"), st["rad"]["film_mean"])
        assert np.array_equal(rd("t0-b0-film-m2"), st["rad"]["film_m2"])
        assert np.array_equal(rd("t1-b0-mean"), st["nrm"]["mean"])
        assert np.array_equal(rd("t1-b0-film-mean"), st["nrm"]["mean"])     # non-transform: film-mean is mean
        assert np.array_equal(rd("t2-b0-mean"), st["alb"]["mean"])
        assert np.array_equal(rd("t1-b0-n"), st["nrm"]["n"].astype(np.float32))
        for k in ("mean", "m2", "m3"):                                       # Box-Cox side: sqrt vs pow
            assert rel_l2(rd("t0-b0-" + k), st["rad"][k]) <= 1e-5, (i, k)
        # the denoised image from the device-side statistics vs the oracle's filter on its own
        r = st["rad"]
        mc, dc = oracle.prepass(r["n"], r["mean"], r["m2"], r["m3"])
        ref = oracle.filter_image(mc, dc, r["film_mean"], [st["nrm"]["mean"], st["alb"]["mean"]],
                                  [-0.5 / 0.1 ** 2, -0.5 / 0.02 ** 2], -0.5 / sd ** 2, radius)
        got = rd("t0-b0-film-mean-f")
        for c in range(3):
            assert rel_l2(got[..., c], ref[..., c]) <= 1e-5, (i, c)
        # --tilestats: tile-local pooled mean / variance of the radiance mean (wave-level Welford + Chan merges) against
        # the oracle's restatement of the same reduction tree on the same image: bit for bit
        tm = oracle.tile_moments(r["film_mean"], 16)
        assert np.array_equal(rd("t0-b0-tile-mean"), tm[..., 1])
        assert np.array_equal(rd("t0-b0-tile-var"), np.where(tm[..., 0] > 1, tm[..., 2] / np.maximum(tm[..., 0] - 1, 1), 0).astype(np.float32))


def test_render_loop_adaptive_budgets(gpu, oracle, tmp_path):
    """--adaptive: the tile-local moments steer the sampler.  After every iteration the tiles are ranked by
    Estimator::TileNoise (tile mean of film-m2 / ((n - 1) n), summed over the channels: statmc_calculate_mean_vars +
    statmc_tile_moments); the noisiest quarter gets twice the schedule's samples in the next iteration, the quietest quarter
    half.  The budgets are recomputed here from the oracle's statistics alone: the ranking is only the same if the device's
    per-tile noise equals the oracle's in order, and the ragged statistics that follow are checked per pixel."""
    from statmc_amd import build, pfm
    build.build_tools()
    W, H, spp, iterations, seed, T = 88, 44, 4, 3, 11, 16
    stem = str(tmp_path / "adaptive")
    out = subprocess.run([build.RENDER_SIM_BIN, "--width", str(W), "--height", str(H), "--spp", str(spp), "--iterations",
                          str(iterations), "--threads", "4", "--seed", str(seed), "--stem", stem, "--adaptive"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    tx, ty = (W + T - 1) // T, (H + T - 1) // T
    n_tiles = tx * ty
    keys = ("n", "mean", "m2", "m3", "film_mean", "film_m2")
    st = {k: oracle.new_state(H, W, 3) for k in ("rad", "nrm", "alb")}
    tile_done = np.zeros(n_tiles, np.int64)
    noise, done = None, 0
    for i in range(1, iterations + 1):
        target = spp if i == 1 else spp << (i - 2)
        budget = np.full(n_tiles, target, np.int64)
        if noise is not None:
            order = np.argsort(noise, kind="stable")
            q = n_tiles // 4
            budget[order[:q]] = max(1, target // 2)
            budget[order[n_tiles - q:]] = 2 * target
        for t in range(n_tiles):
            x0, y0 = (t % tx) * T, (t // tx) * T
            x1, y1 = min(x0 + T, W), min(y0 + T, H)
            smp = make_samples(seed, x1 - x0, y1 - y0, int(tile_done[t]), int(budget[t]), x0, y0)
            for name, s_t, (transform, mm) in zip(("rad", "nrm", "alb"), smp, ((True, 3), (False, 1), (False, 1))):
                sub = oracle.new_state(y1 - y0, x1 - x0, 3)
                for k in keys:
                    sub[k][...] = st[name][k][y0:y1, x0:x1]
                oracle.accumulate(sub, s_t, transform, mm)
                for k in keys:
                    st[name][k][y0:y1, x0:x1] = sub[k]
        tile_done += budget
        done += target
        r = st["rad"]
        tm = oracle.tile_moments(oracle.mean_vars(r["n"], r["film_m2"], row_n_quirk=False), T)     # [ty, tx, 3, {count, mean, M2}]
        noise = ((tm[..., 0, 1] + tm[..., 1, 1]) + tm[..., 2, 1]).reshape(-1)
        rd = lambda name: pfm.read_pfm("%s-%d-%s.pfm" % (stem, done, name))
        assert np.array_equal(rd("t0-b0-n"), r["n"].astype(np.float32)), i
        assert np.array_equal(rd("t1-b0-n"), st["nrm"]["n"].astype(np.float32)), i
        assert np.array_equal(rd("t0-b0-film-mean"), r["film_mean"]), i
        assert np.array_equal(rd("t0-b0-film-m2"), r["film_m2"]), i
        assert np.array_equal(rd("t1-b0-mean"), st["nrm"]["mean"]), i
        assert np.array_equal(rd("t2-b0-mean"), st["alb"]["mean"]), i
        for k in ("mean", "m2", "m3"):
            assert rel_l2(rd("t0-b0-" + k), r[k]) <= 1e-5, (i, k)
        assert "Adaptive: samples per pixel so far %d .. %d over %d tiles" % (tile_done.min(), tile_done.max(), n_tiles) in out.stdout
    assert tile_done.min() < done < tile_done.max()          # the budgets did move
    # the filter runs on ragged counts: the denoised image against the oracle's on its own statistics
    r = st["rad"]
    mc, dc = oracle.prepass(r["n"], r["mean"], r["m2"], r["m3"])
    ref = oracle.filter_image(mc, dc, r["film_mean"], [st["nrm"]["mean"], st["alb"]["mean"]],
                              [-0.5 / 0.1 ** 2, -0.5 / 0.02 ** 2], -0.5 / 10.0 ** 2, 20)
    got = pfm.read_pfm("%s-%d-t0-b0-film-mean-f.pfm" % (stem, done))
    for c in range(3):
        assert rel_l2(got[..., c], ref[..., c]) <= 1e-5, c


def test_render_loop_statistics_only(gpu, tmp_path):
    """`calcstats` configuration (scenes/render-for-ours.pbrt): statistics are accumulated and dumped,
    no filter call (runCUDA false, statpath.cpp:1050-1051)."""
    from statmc_amd import build, pfm
    build.build_tools()
    stem = str(tmp_path / "stats")
    out = subprocess.run([build.RENDER_SIM_BIN, "--width", "48", "--height", "32", "--spp", "2", "--iterations", "2",
                          "--stem", stem, "--no-denoise", "--outputregex", "film|t[0-9]+-b[0-9]+-(n|mean|m2|m3)"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "CUDA time [ns]: " in out.stdout
    n = pfm.read_pfm("%s-4-t0-b0-n.pfm" % stem)
    assert np.all(n == 4.0)
    # OutputBufferSelection: only the buffers the regular expression matches are written
    assert os.path.exists("%s-4-film.pfm" % stem) and os.path.exists("%s-4-t2-b0-mean.pfm" % stem)
    assert not os.path.exists("%s-4-t0-b0-film-mean-f.pfm" % stem)
    assert not os.path.exists("%s-4-t0-b0-film-m2.pfm" % stem)


def test_render_loop_acrr_float_buffers(gpu, oracle, tmp_path):
    """Render<Float> (`multichannelstats` false, `acrr`): the luminance up to each of 5 tracked
    bounces is one float stat buffer (StatTile<Float> per bounce, MergeTransformTiles over the
    bounces), all five filtered by one filter<float> call (estimator.cpp:434-460)."""
    from statmc_amd import build, pfm
    build.build_tools()
    W, H, spp, iterations, seed, radius, sd, nb = 64, 32, 4, 2, 9, 20, 10.0, 5
    stem = str(tmp_path / "acrr")
    out = subprocess.run([build.RENDER_SIM_BIN, "--config", "acrr", "--trackedbounces", str(nb), "--width", str(W),
                          "--height", str(H), "--spp", str(spp), "--iterations", str(iterations), "--seed", str(seed),
                          "--stem", stem], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    st = [oracle.new_state(H, W, 1) for _ in range(nb)]
    nrm_st, alb_st = oracle.new_state(H, W, 3), oracle.new_state(H, W, 3)
    done = 0
    for i in range(1, iterations + 1):
        target = spp if i == 1 else spp << (i - 2)
        rad, nrm, alb = make_samples(seed, W, H, done, target)
        oracle.accumulate(nrm_st, nrm, False, 1)
        oracle.accumulate(alb_st, alb, False, 1)
        for j in range(nb):
            share = np.float32(j + 1) / np.float32(nb)
            ls = rad * share
            lum = (np.float32(0.212671) * ls[..., 0] + np.float32(0.715160) * ls[..., 1]) + np.float32(0.072169) * ls[..., 2]
            oracle.accumulate(st[j], np.ascontiguousarray(lum[..., None]), True, 3)
        done += target
    rd = lambda name: pfm.read_pfm("%s-%d-%s.pfm" % (stem, done, name))
    for j in range(nb):
        pre = "t0-b%d-" % j
        assert np.array_equal(rd(pre + "n"), st[j]["n"].astype(np.float32))
        assert np.array_equal(rd(pre + "film-mean"), st[j]["film_mean"][..., 0]), j
        assert np.array_equal(rd(pre + "film-m2"), st[j]["film_m2"][..., 0]), j
        for k in ("mean", "m2", "m3"):
            assert rel_l2(rd(pre + k), st[j][k][..., 0]) <= 1e-5, (j, k)
        mc, dc = oracle.prepass(st[j]["n"], st[j]["mean"], st[j]["m2"], st[j]["m3"])
        ref = oracle.filter_image(mc, dc, st[j]["film_mean"], [nrm_st["mean"], alb_st["mean"]],
                                  [-0.5 / 0.1 ** 2, -0.5 / 0.02 ** 2], -0.5 / sd ** 2, radius)
        assert rel_l2(rd(pre + "film-mean-f"), ref[..., 0]) <= 1e-5, j


def test_render_loop_smis_tallies(gpu, oracle, tmp_path):
    """`smis`: per tracked bounce a BSDF and a light win-rate tally (float, plain M3, no radiance
    type), merged through MergeTiles(vector<vector<StatTile<Float>>>, {cfg, cfg}) (statpath.cpp:385-386)
    and filtered by one filter<float> call over 2 x bounces buffers."""
    from statmc_amd import build, pfm
    build.build_tools()
    W, H, spp, iterations, seed, radius, sd, nb = 48, 32, 4, 2, 3, 20, 10.0, 3
    stem = str(tmp_path / "smis")
    out = subprocess.run([build.RENDER_SIM_BIN, "--config", "smis", "--trackedbounces", str(nb), "--width", str(W),
                          "--height", str(H), "--spp", str(spp), "--iterations", str(iterations), "--seed", str(seed),
                          "--stem", stem], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    y, x = np.mgrid[0:H, 0:W]
    st = [[oracle.new_state(H, W, 1) for _ in range(nb)] for _ in range(2)]
    nrm_st, alb_st = oracle.new_state(H, W, 3), oracle.new_state(H, W, 3)
    done = 0
    for i in range(1, iterations + 1):
        target = spp if i == 1 else spp << (i - 2)
        _, nrm, alb = make_samples(seed, W, H, done, target)
        oracle.accumulate(nrm_st, nrm, False, 1)
        oracle.accumulate(alb_st, alb, False, 1)
        for ti, base in ((0, 20), (1, 40)):
            for j in range(nb):
                smp = np.stack([((draw(seed, x, y, done + s, base + j) >> np.uint64(3)) & np.uint64(1)).astype(np.float32)
                                for s in range(target)])[..., None]
                oracle.accumulate(st[ti][j], np.ascontiguousarray(smp), False, 3)
        done += target
    rd = lambda name: pfm.read_pfm("%s-%d-%s.pfm" % (stem, done, name))
    for ti in range(2):
        for j in range(nb):
            pre, s = "t%d-b%d-" % (ti, j), st[ti][j]
            assert np.array_equal(rd(pre + "n"), s["n"].astype(np.float32))
            for k in ("mean", "m2", "m3"):
                assert np.array_equal(rd(pre + k), s[k][..., 0]), (ti, j, k)
            mc, dc = oracle.prepass(s["n"], s["mean"], s["m2"], s["m3"])
            ref = oracle.filter_image(mc, dc, s["mean"], [nrm_st["mean"], alb_st["mean"]],
                                      [-0.5 / 0.1 ** 2, -0.5 / 0.02 ** 2], -0.5 / sd ** 2, radius)
            assert rel_l2(rd(pre + "film-mean-f"), ref[..., 0]) <= 1e-5, (ti, j)
    assert np.array_equal(rd("t2-b0-mean"), nrm_st["mean"]) and np.array_equal(rd("t3-b0-mean"), alb_st["mean"])
