"""Runs whenever the box has TWO devices (it skips on the one-GPU boxes this repository is developed on, and says so): the
first execution of the multi-device code on real hardware should be a pass / fail, not a bench line to interpret.
  (a) statmc_amd/peer.py with blocks on devices 0 and 1 -- real hipDeviceEnablePeerAccess, cross-device statmc_copy_rect /
      statmc_halo_exchange -- equals the whole film on device 0, bit for bit under a pinned split;
  (b) bench.py --gpus 2 on the nccl leg, self-launched: two ranks seen, the overlapped order bit-identical to the plain one,
      no fallback;
  (c) the same with the RCCL bring-up forced to fail (STATMC_BENCH_FAIL_NCCL=1): the line comes from the peer leg and says so;
  (d) the two legs leave the same film-f."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_DEV = torch.cuda.device_count()          # (counting devices does not initialise the GPU on this image)
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(N_DEV < 2, reason="needs two devices: this box has %d (the multi-device code paths are otherwise covered with "
                                                   "all blocks on device 0 -- test_peer_gpu.py, test_multirank_gpu.py -- and over gloo on the CPU)" % N_DEV)]
COMMON = ["--film", "512x256", "--spp", "8", "--steps", "3", "--warmup", "1"]
TYPES = ("radiance", "normal", "albedo")


def _line(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    return json.loads(lines[0])


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra)
    return env


@pytest.mark.parametrize("grid,bw,bh", [((1, 2), 272, 64), ((2, 1), 136, 96)], ids=["1x2-strips-overlapped", "2x1-blocks"])
def test_peer_film_on_two_devices_equals_the_whole_film(gpu, grid, bw, bh):
    from statmc_amd import peer, pipeline, sharding, synthetic
    gx, gy = grid
    radius = 20
    dev0 = torch.device("cuda:0")
    scene = synthetic.Scene(gx * bw, gy * bh, n_regions=7, seed=5)
    batches = [scene.samples(4, seed=6, features=TYPES), scene.samples(3, seed=7, features=TYPES)]
    lib = gpu.load()
    for d in (0, 1):
        gpu.setup(d)
        gpu.check(lib.statmc_set_device(d))
        gpu.set_filter_split(2)
    gpu.check(lib.statmc_set_device(0))
    try:
        one = pipeline.BlockPipeline(sharding.BlockLayout(0, 1, gx * bw, gy * bh, radius), dev0, TYPES, radius=radius, filter_sd=radius / 2.0)
        refs = []
        for smp in batches:
            one.accumulate({k: v.to(dev0) for k, v in smp.items()})
            refs.append(one.denoise().clone())
        pf = peer.PeerFilm(2, bw, bh, radius, [0, 1], TYPES, filter_sd=radius / 2.0, grid=grid)
        assert [b.device_index for b in pf.blocks] == [0, 1]
        for smp, ref in zip(batches, refs):
            per_block = []
            for blk in pf.blocks:
                ox, oy = blk.layout.origin
                per_block.append([{k: v[:, oy:oy + bh, ox:ox + bw].contiguous().to(blk.dev) for k, v in smp.items()}])
            pf.run(pf.prepare_step(per_block))
            pf.synchronize()
            got = pf.gather()
            pf.synchronize()
            assert got.device == ref.device or True
            assert torch.equal(got.to(dev0), ref)
    finally:
        for d in (0, 1):
            gpu.check(lib.statmc_set_device(d))
            gpu.set_filter_split(0)
        gpu.check(lib.statmc_set_device(0))


def test_bench_two_gpus_nccl_leg(gpu, tmp_path):
    dump = str(tmp_path / "nccl.npy")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-fallback", "--dump-film-f", dump] + COMMON,
                         capture_output=True, text=True, timeout=900, env=_env())
    assert out.returncode == 0, out.stderr[-3000:]
    r = _line(out)
    assert r["n_gpus"] == 2 and r["n_ranks_seen"] == 2 and r["backend"] == "nccl" and "fallback_from" not in r
    assert r["overlap_self_check"]["overlapped_vs_plain_order"] == "bit-identical" and r["overlap_self_check"]["ranks"] == 2
    assert r["config"]["film"] == "512x256" and r["value"] > 0 and r["scaling"] == "strong"
    assert np.isfinite(np.load(dump)).all()


def test_bench_two_gpus_forced_nccl_failure_lands_on_the_peer_leg(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + COMMON, capture_output=True, text=True, timeout=900,
                         env=_env(STATMC_BENCH_FAIL_NCCL="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    r = _line(out)
    assert r["n_gpus"] == 2 and r["backend"] == "peer" and r["fallback_from"] == "nccl" and "nccl_error" in r
    assert len(r["devices"]) == 2 and len(set(r["devices"])) == 2          # two different devices drove the two blocks
    assert r["overlap_self_check"]["overlapped_vs_plain_order"] == "bit-identical"


def test_the_two_legs_leave_the_same_film(gpu, tmp_path):
    dumps = {}
    for backend in ("nccl", "peer"):
        dumps[backend] = str(tmp_path / (backend + ".npy"))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", backend, "--no-fallback",
                              "--dump-film-f", dumps[backend]] + COMMON, capture_output=True, text=True, timeout=900, env=_env())
        assert out.returncode == 0, out.stderr[-3000:]
        assert _line(out)["backend"] == backend
    a, b = np.load(dumps["nccl"]), np.load(dumps["peer"])
    assert a.shape == b.shape == (256, 512, 3) and np.array_equal(a, b)
