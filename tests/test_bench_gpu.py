"""bench.py end to end on a small film: the one-GPU line with its secondary legs, and the N > 1 strong-scaling mode
(ONE film cut into blocks) with two ranks that share the test box's single GPU (gloo rendezvous, halos via the host)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--film", "512x256", "--spp", "8", "--steps", "3", "--warmup", "1"]


def _line(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout + out.stderr
    return json.loads(lines[0])


def test_single_gpu_line(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + COMMON + ["--cpu-acc-rows", "8"],
                         capture_output=True, text=True, timeout=360)
    assert out.returncode == 0, out.stderr[-2000:]
    r = _line(out)
    assert r["metric"] == "denoised_mpixels_per_s" and r["unit"] == "Mpixels/s" and r["n_gpus"] == 1 and r["steps"] == 3
    assert r["value"] > 0 and r["dtype"] == "f32" and r["data"] == "synthetic" and r["vs_baseline"] is None
    assert r["config"]["film"] == "512x256" and r["config"]["filter_variant"] == "sym_r20"
    assert abs(r["value"] - 512 * 256 / r["ms_per_step"] / 1e3) < 1e-2 * r["value"]
    rl, rf = r["roofline"], r["roofline_filter"]
    assert rl["bound"] == "hbm" and rl["unit"] == "GB/s" and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3
    # the window filter is a VALU-bound stencil: its roofline is the fp32 issue rate, the HBM figures ride along
    assert rf["bound"] == "valu" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["hbm"]["unit"] == "GB/s" and abs(rf["hbm"]["frac"] - rf["hbm"]["achieved"] / rf["hbm"]["peak"]) < 1e-3
    assert r["n_ranks_seen"] == 1 and r["config"]["schedule"] == "single" and r["config"]["resident_pool_spp"] == 8
    assert 0.5 < r["shader_clock"]["during_filter_GHz"] < 3.0 and 0.5 < r["shader_clock"]["during_accumulate_GHz"] < 3.0
    assert r["cpu_baseline"]["kind"] == "port" and r["cpu_baseline"]["cores"] >= 1 and r["cpu_baseline"]["value"] > 0
    par = r["cpu_baseline"]["parity_of_the_same_run"]
    assert par["prepass_bit_exact"] and max(par["rel_l2_per_channel"]) <= 1e-5
    # secondary legs: the reference's own bracket through the C++ host side, tile-fed accumulation, copy rates
    assert r["cuda_time_bracket"]["cuda_time_bracket_ms"] > 0 and r["cuda_time_bracket"]["iterations"] == 12
    assert r["cuda_time_bracket"]["two_upload_queues"]["best_ms"] > 0
    assert r["tile_fed_accumulate"]["achieved_GBs"] > 0
    assert r["host_copies"]["upload_bytes_per_px"] == 76 and r["host_copies"]["download_bytes_per_px"] == 12
    assert r["filter_8_feature_channels"]["filter_variant"] == "sym_r20_g8" and r["filter_8_feature_channels"]["avg_ms"] > 0
    assert r["pcie_inclusive"]["GBs"] > 0 and r["pcie_inclusive"]["mpixels_per_s_if_samples_cross_pcie"] < r["value"]
    ff = r["filter_float_buffers"]
    assert ff["filter_variant"] == "sym_r20_f" and 0 < ff["1_buffers_ms"] <= ff["2_buffers_ms"] * 1.3 and ff["smis_12_buffers_ms"] > ff["acrr_5_buffers_ms"] > ff["2_buffers_ms"]
    # round 6: the filter parameters as keys of their own (the driver keeps 120 characters of `workload`), the placed allocator trimmed
    # before the timed region and A/B'd against torch's allocator in the same process, the reference's progressive schedule as a leg
    cfg = r["config"]
    assert len(cfg["workload"]) <= 120 and cfg["radius"] == 20 and cfg["filter_sd"] == 10.0 and cfg["g_sds"] == [0.1, 0.02]
    pl = r["placement"]
    if pl.get("active"):
        assert pl["slots_idle"] == 0 and pl["trimmed_before_timing"] is not None and pl["slots"] <= 12, pl
        assert "skipped" in pl["check"]       # (a test-sized pool: too short to tell two sets of buffers apart)
    ab = r["accumulate_placement_ab"]
    assert ab["placed_ms"] > 0 and ab["unplaced_ms"] > 0 and r["kernels"]["accumulate"]["unplaced_frac_hbm"] == ab["unplaced_frac_hbm"]
    # ... and the pre-pass in the accumulation's epilogue: no launch of its own in the step, the same bits (the CPU leg's
    # `prepass_bit_exact` above compared mean-corr / discriminator of the timed step with the oracle's)
    assert r["kernels"]["prepass"]["launches_per_step"] == 0 and r["kernels"]["prepass"]["ms_per_step"] == 0.0
    assert "epilogue" in r["config"]["step_order"]
    rs = r["reference_schedule"]
    assert rs["batches"] == [4, 4] and rs["iterations"] == 2 and rs["ms_per_step"] > 0 and 0 < rs["filter_share"] < 1
    assert abs(rs["accumulate_ms"] + rs["prepass_ms"] + rs["filter_ms"] - rs["ms_per_step"]) < 0.5 * rs["ms_per_step"]


def test_placement_check_can_hand_the_step_to_the_allocators_memory(gpu):
    """bench.py runs the step on the placed buffers and on copies from torch's allocator before it times, and the faster set goes on
    (one card in six gives the placed allocator little to choose from).  The other branch, forced: the line says
    which set ran, the placed blocks are gone, every later leg runs on plain memory."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + COMMON + ["--no-cpu-baseline"], capture_output=True, text=True,
                         timeout=360, env=dict(os.environ, STATMC_BENCH_CHECK_PICKS="allocator"))
    assert out.returncode == 0, out.stderr[-2000:]
    r = _line(out)
    pl = r["placement"]
    if "check" in pl and "skipped" not in pl["check"]:      # (a device without virtual-memory management has nothing to check)
        assert pl["check"]["chosen"] == "torch's allocator" and pl["check"]["forced"] and pl["timed_region_ran_on"] == "torch's allocator"
        assert pl["check"]["placed_ms"] > 0 and pl["check"]["allocator_ms"] > 0
        assert set(pl["map_after_release"]) <= set("#_")          # only the allocator's own slots are still backed
        assert "skipped" in r["accumulate_placement_ab"]
    assert r["value"] > 0 and r["reference_schedule"]["ms_per_step"] > 0


def test_step_fed_by_tile_blocks(gpu, tmp_path):
    """`--feed tiles`: the timed step's samples reach the accumulation as 16 x 16 tile blocks (statmc_accumulate_tiles), the way
    Render<T> hands them over; the line says so and the step leaves the same film-f as the film-major feed, bit for bit."""
    films = {}
    for feed in ("film", "tiles"):
        dump = str(tmp_path / (feed + ".npy"))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--feed", feed, "--no-host-legs", "--no-cpu-baseline",
                              "--dump-film-f", dump] + COMMON, capture_output=True, text=True, timeout=360)
        assert out.returncode == 0, out.stderr[-2000:]
        r = _line(out)
        assert r["config"]["feed"].startswith("16 x 16 tile blocks" if feed == "tiles" else "film-major")
        assert r["roofline"]["kernel"] == ("accumulate_tiles_kernel" if feed == "tiles" else "accumulate_kernel") and r["value"] > 0
        import numpy as np
        films[feed] = np.load(dump)
    assert films["film"].shape == (256, 512, 3) and (films["film"] == films["tiles"]).all()


@pytest.mark.parametrize("grid,blocks", [("rows", "1x2"), ("blocks", "2x1")])
def test_strong_scaling_two_ranks_share_the_device(gpu, grid, blocks):
    """`python bench.py --gpus 2` with NO launcher around it: bench.py starts its own ranks (fresh processes, before
    any GPU call in the parent) and relays rank 0's line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device",
           "--grid", grid] + COMMON
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=360, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    r = _line(out)
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["n_ranks_seen"] == 2
    assert r["config"]["film"] == "512x256" and r["config"]["block_grid"] == blocks      # one film, two blocks
    assert abs(r["value"] - 512 * 256 / r["ms_per_step"] / 1e3) < 1e-2 * r["value"]      # film pixels per step time
    k = r["kernels"]
    if grid == "rows":      # 512 x 128 strips: the overlapped order, with keys of its own (nothing shares a name with the plain order)
        assert r["config"]["step_order"].startswith("border rows first")
        assert k["border_chain"]["ms_per_step"] > 0 and k["border_chain"]["rows"] == 20
        assert k["interior"]["accumulate_ms_per_step"] > 0 and k["interior"]["rows"] == 108
        assert k["interior_exposed"]["ms_per_step"] >= 0 and k["exchange_exposed"]["ms_per_step"] >= 0
        assert "accumulate" not in k and "halo_exchange" not in k
    else:
        assert k["halo_exchange"]["ms_per_step"] > 0 and k["accumulate"]["ms_per_step"] > 0 and "border_chain" not in k
    assert r["backend"] == "gloo"
    assert r["overlap_self_check"]["overlapped_vs_plain_order"] == "bit-identical" and r["overlap_self_check"]["ranks"] == 2
    assert r["gather_ms"] > 0 and r["gather"]["bytes"] == 12 * 512 * 256 and not r["gather"]["in_step"]
    assert "cpu_baseline" not in r


def test_two_ranks_with_placed_buffers(gpu):
    """The N > 1 step with every rank's moments and sample arenas from statmc_malloc_placed (what the driver's multi-GPU runs use;
    here two ranks share the box's GPU, each with its own slots): same self-checks as the plain two-rank run."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["STATMC_BENCH_PLACED_ON_SHARED_DEVICE"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device"] + COMMON
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=360, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    r = _line(out)
    assert r["n_gpus"] == 2 and r["n_ranks_seen"] == 2 and r["placement"]["requested"] and r["placement"]["error"] is None
    assert r["placement"]["virtual_memory"] if "virtual_memory" in r["placement"] else True
    assert r["overlap_self_check"]["overlapped_vs_plain_order"] == "bit-identical"
    assert r["config"]["step_order"].startswith("border rows first")


def test_launched_under_torchrun_and_gather_in_step(gpu):
    """The driver's N > 1 launch form (python -m torch.distributed.run ... bench.py --gpus N) still works, and --gather
    puts the assembly of film-f on rank 0 inside the step."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device",
           "--gather"] + COMMON
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=360)
    assert out.returncode == 0, out.stderr[-3000:]
    r = _line(out)
    assert r["n_gpus"] == 2 and r["n_ranks_seen"] == 2 and r["gather"]["in_step"] and r["kernels"]["gather"]["ms_per_step"] > 0


def test_reference_schedule_with_a_sample_pool(gpu):
    """configs[4]'s shape in small: the reference's 4, 4, 8, 16 schedule (statpath.cpp:272-279) with the denoiser after every
    iteration, samples drawn from a resident pool smaller than the sample count."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--film", "512x256", "--spp", "32", "--steps", "2",
                          "--warmup", "1", "--schedule", "reference", "--pool-spp", "12", "--no-host-legs", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=360)
    assert out.returncode == 0, out.stderr[-2000:]
    r = _line(out)
    c = r["config"]
    assert c["schedule"] == "reference" and c["iterations_per_step"] == 4 and c["resident_pool_spp"] == 12
    # 4 | 4 | 8 (wraps the pool: 8..12, 0..4) | 16 (4..12, 0..8): 1 + 1 + 2 + 2 launches
    assert c["accumulate_launches_per_step"] == 6 and r["roofline_filter"]["launches_per_step"] == 4
    assert abs(r["value"] - 512 * 256 / r["ms_per_step"] / 1e3) < 1e-2 * r["value"]


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra)
    return env


def test_peer_leg_one_process_all_blocks_and_same_bits_as_the_rank_leg(gpu, tmp_path):
    """`--backend peer`: ONE process drives both blocks through the C ABI (statmc_halo_exchange's device-to-device copies;
    here both blocks share the box's one GPU).  Its film-f equals the one-process-per-block leg's (gloo, halos via the
    host) bit for bit, and -- the blocks being filtered under the default dispatch -- the whole film's within 1e-6."""
    import numpy as np
    dumps = {}
    for backend in ("peer", "gloo"):
        dumps[backend] = str(tmp_path / ("film_f_%s.npy" % backend))
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", backend, "--share-device",
               "--dump-film-f", dumps[backend]] + COMMON
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=360, env=_clean_env())
        assert out.returncode == 0, out.stderr[-3000:]
        r = _line(out)
        assert r["backend"] == backend and r["n_gpus"] == 2 and r["n_ranks_seen"] == 2
        assert r["overlap_self_check"]["overlapped_vs_plain_order"] == "bit-identical"
        # all four feature types as G-buffers travel in a 17-channel block + halo image and stay on the pair-symmetric kernel
        e8 = r["filter_8_feature_channels"]
        assert e8["filter_variant"] == "sym_r20_g8" and e8["packed_channels"] == 17 and e8["avg_ms"] > 0
        if backend == "peer":
            assert "fallback_from" not in r and r["devices"] == [0, 0] and r["host_enqueue_ms_per_step"] > 0
            assert r["config"]["parallelism"].startswith("film blocks x2, ONE process")
            k = r["kernels"]
            assert k["border_chain"]["ms_per_step"] > 0 and k["interior"]["accumulate_ms_per_step"] > 0 and k["filter"]["ms_per_step"] > 0
            assert abs(r["value"] - 512 * 256 / r["ms_per_step"] / 1e3) < 1e-2 * r["value"]
            assert r["gather"]["bytes"] == 12 * 512 * 256
    a, b = np.load(dumps["peer"]), np.load(dumps["gloo"])
    assert a.shape == (256, 512, 3) and np.array_equal(a, b)


@pytest.mark.parametrize("launcher,mode", [("self", "1"), ("torchrun", "1"), ("self", "hang"), ("torchrun", "hang")],
                         ids=["self", "torchrun", "self-hang", "torchrun-hang"])
def test_nccl_failure_falls_back_to_the_peer_leg(gpu, launcher, mode):
    """The nccl leg cannot come up (forced: STATMC_BENCH_FAIL_NCCL=1 raises in the bring-up, =hang never returns from it and
    leaves it to the watchdog; on this one-GPU box RCCL would refuse two ranks on a device anyway) -> a FRESH process runs
    the peer leg and the line records where it came from.  Both launch forms: the self-launcher's parent owns the fallback,
    under a foreign launcher (the driver's form) rank 0 does."""
    if launcher == "self":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")]
    else:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py")]
    cmd += ["--gpus", "2", "--backend", "nccl", "--share-device"] + COMMON + (["--bringup-timeout", "6"] if mode == "hang" else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=_clean_env(STATMC_BENCH_FAIL_NCCL=mode))
    assert out.returncode == 0, out.stderr[-3000:]
    r = _line(out)
    assert r["backend"] == "peer" and r["fallback_from"] == "nccl" and r["n_gpus"] == 2 and r["n_ranks_seen"] == 2
    assert "why" in r["nccl_error"] and r["value"] > 0
    if mode == "hang" and launcher == "torchrun":
        assert "stuck" in r["nccl_error"]["why"]


def test_stalled_run_under_a_foreign_launcher_falls_back(gpu):
    """A run that stalls AFTER a good bring-up (forced: STATMC_BENCH_FAIL_NCCL=stall; gloo ranks here, the box has one GPU)
    under the driver's launch form, where no parent owns a time limit: past --run-timeout the ranks leave and rank 0 runs
    the peer leg in a fresh process."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device",
           "--run-timeout", "5"] + COMMON
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=_clean_env(STATMC_BENCH_FAIL_NCCL="stall"))
    assert out.returncode == 0, out.stderr[-3000:]
    r = _line(out)
    assert r["backend"] == "peer" and r["fallback_from"] == "gloo" and r["n_ranks_seen"] == 2 and r["value"] > 0
    assert "did not finish" in r["nccl_error"]["why"]

