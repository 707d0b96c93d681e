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


_RCCL_RANK = r"""
import os, sys, time
sys.path.insert(0, {root!r})
import numpy as np, torch
from statmc_amd import api, pipeline, sharding, synthetic
rank, gx, gy, bw, bh, radius, work = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7])
n = gx * gy
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
api.setup(rank)
api.set_filter_split(2)
idf = os.path.join(work, "id.bin")
if rank == 0:
    with open(idf + ".tmp", "wb") as f:
        f.write(api.RcclComm.unique_id())
    os.replace(idf + ".tmp", idf)
t0 = time.time()
while not os.path.exists(idf):
    assert time.time() - t0 < 120, "no id"
    time.sleep(0.05)
comm = api.RcclComm(n, rank, open(idf, "rb").read())
types = ("radiance", "normal", "albedo")
layout = sharding.BlockLayout(rank, n, bw, bh, radius, grid=(gx, gy))
pipe = pipeline.BlockPipeline(layout, dev, types, radius=radius, filter_sd=radius / 2.0)
scene = synthetic.Scene(gx * bw, gy * bh, n_regions=7, seed=5)
ox, oy = layout.origin
for seed, s in ((6, 4), (7, 3)):
    smp = scene.samples(s, seed=seed, features=types)
    pipe.accumulate({{k: v[:, oy:oy + bh, ox:ox + bw].contiguous().to(dev) for k, v in smp.items()}})
    pipe.prepass()
    pipe.exchange_rccl(comm)
    out = pipe.window_filter()
    torch.cuda.synchronize()
np.save(os.path.join(work, "block%d.npy" % rank), out.cpu().numpy())
comm.destroy()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("grid,bw,bh", [((1, 2), 272, 64), ((2, 1), 136, 96)], ids=["1x2-strips", "2x1-blocks"])
def test_rccl_halo_exchange_through_the_c_abi(gpu, tmp_path, grid, bw, bh):
    """statmc_halo_exchange_rccl: two PROCESSES, one per device, a communicator made through the C ABI
    (statmc_rccl_unique_id / statmc_rccl_comm_create), the blocks' halos exchanged by RCCL send / recv -- the assembled film equals
    the whole film filtered on one device, bit for bit under a pinned split."""
    from statmc_amd import pipeline, sharding, synthetic
    gx, gy = grid
    radius = 20
    script = tmp_path / "rank.py"
    script.write_text(_RCCL_RANK.format(root=ROOT))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(gx), str(gy), str(bw), str(bh), str(radius), str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=_env()) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-1000:] + se[-3000:]
    dev0 = torch.device("cuda:0")
    gpu.set_filter_split(2)
    try:
        one = pipeline.BlockPipeline(sharding.BlockLayout(0, 1, gx * bw, gy * bh, radius), dev0, TYPES, radius=radius, filter_sd=radius / 2.0)
        scene = synthetic.Scene(gx * bw, gy * bh, n_regions=7, seed=5)
        for seed, s in ((6, 4), (7, 3)):
            one.accumulate({k: v.to(dev0) for k, v in scene.samples(s, seed=seed, features=TYPES).items()})
            ref = one.denoise().clone()
    finally:
        gpu.set_filter_split(0)
    ref = ref.cpu().numpy()
    for r in range(2):
        blk = np.load(str(tmp_path / ("block%d.npy" % r)))
        ox, oy = (r % gx) * bw, (r // gx) * bh
        assert np.array_equal(blk, ref[oy:oy + bh, ox:ox + bw]), "block %d differs from the whole film" % r


def test_sharded_denoise_on_two_devices_with_placed_memory(gpu, tmp_path):
    """with statmc::usePlacedMemory() every DeviceImage -- the Estimator's tables AND the block images of statmc::FilmShards
    -- is a block of the placed allocator (hipMemCreate / hipMemMap), which hipDeviceEnablePeerAccess does not cover: the allocator
    grants the mapping to the peers itself (hipMemSetAccess).  Blocks on devices 0 and 1, cut / halo exchange / paste across devices:
    the same film-f as the unsharded run, with and without placed memory."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
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
    for key, extra in (("one", []), ("two", ["--grid", "2x1", "--devices", "0,1"]), ("two-placed", ["--grid", "2x1", "--devices", "0,1", "--placed"]),
                       ("strips-placed", ["--grid", "1x2", "--devices", "0,1", "--placed"])):
        out = subprocess.run([exe, "--stem", stem, "--spp", str(spp), "--output", "film-f", "--parts", "2"] + extra, capture_output=True, text=True, env=_env())
        assert out.returncode == 0, key + ": " + out.stderr[-2000:]
        outs[key] = pfm.read_pfm("%s-%d-film-f.pfm" % (stem, spp))
    for key in ("two", "two-placed", "strips-placed"):
        assert np.array_equal(outs[key], outs["one"]), key
