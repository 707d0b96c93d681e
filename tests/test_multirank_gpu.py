"""The N > 1 path end to end on the GPU: several ranks (gloo rendezvous, all on cuda:0 because
the test box has one GPU; halos staged through the host) each run BlockPipeline on their block of
a film; the assembled blocks must equal what one process computes on the whole film.  RCCL itself
is exercised by the driver's multi-GPU bench; this pins everything around it."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


BW, BH, SPP, RADIUS = 272, 40, 6, 20
TYPES = ("radiance", "normal", "albedo")
# all four feature types as G-buffers (statpath.cpp:828-835, 1096-1130): eight feature planes, a 17-channel block + halo image
TYPES8 = ("radiance", "normal", "albedo", "depth", "materialid")
G8 = ("materialid", "depth", "normal", "albedo")     # the reference's stat-type order (statpath.cpp:1096-1160)


def _grid(world, rows):
    from statmc_amd import sharding
    return sharding.row_strips(world) if rows else sharding.grid_for(world)


def _film_samples(world, rows, BH=BH, g8=False):
    """Whole-film sample stream, identical in every process (CPU generator, fixed seed)."""
    from statmc_amd import synthetic
    g8 = g8 is True
    gx, gy = _grid(world, rows)
    scene = synthetic.Scene(gx * BW, gy * BH, n_regions=9, seed=21)
    return scene.samples(SPP, seed=22, features=TYPES8 if g8 else TYPES)


def _worker(rank, world, rows, port, q, BH=BH, g8=False):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from statmc_amd import api, pipeline, sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _run(rank, world, rows, q, dist, api, pipeline, sharding, BH, g8)
    except Exception as e:                      # report instead of leaving the parent waiting
        q.put((rank, "error", repr(e), None))
        raise
    finally:
        dist.destroy_process_group()


def _run(rank, world, rows, q, dist, api, pipeline, sharding, BH=BH, g8=False):
    if True:
        dev = torch.device("cuda:0")
        api.setup(0)
        api.force_filter_parts(2)            # same window-row split as the single-process run
        welch = g8 in ("welch", "welch8")    # Welch degrees of freedom: the sample count travels in a 16-channel image (18 with depth / material id)
        g8 = g8 is True or g8 == "welch8"
        if welch:
            api.set_filter_spec(dof=1)
        L = sharding.BlockLayout(rank, world, BW, BH, RADIUS, grid=_grid(world, rows))
        ox, oy = L.origin
        smp = {k: v[:, oy:oy + BH, ox:ox + BW].contiguous().to(dev) for k, v in _film_samples(world, rows, BH, g8).items()}
        pipe = pipeline.BlockPipeline(L, dev, TYPES8 if g8 else TYPES, radius=RADIUS, via_host=True,
                                      **(dict(g_buffers=G8) if g8 else {}))
        assert pipe.packed.shape[2] == ((18 if welch else 17) if g8 else 16 if welch else 15)
        # row strips tall enough for it take the overlapped order: the rows a neighbour needs first, the exchange started,
        # the rest accumulated behind it (BlockPipeline.accumulate_and_denoise); everything else the plain order
        overlapped = bool(pipe.border_rows())
        out = pipe.accumulate_and_denoise(smp).clone()
        torch.cuda.synchronize()
        assert api.last_filter_variant() == (("sym_welch_g8" if welch else "sym_r20_g8") if g8 else "sym_welch" if welch else "sym_r20")
        assert overlapped == (rows and BH >= 2 * RADIUS + 8)
        q.put((rank, ox, oy, out.cpu().numpy()))
        dist.barrier()


@pytest.mark.parametrize("world,rows,BH,g8", [(2, False, 40, False), (4, False, 40, False), (3, True, 40, False), (3, True, 56, False),
                                              (2, True, 64, False), (2, False, 40, True), (3, True, 56, True), (2, True, 64, "welch"), (2, True, 64, "welch8")],
                         ids=["2x1", "2x2", "1x3-rows", "1x3-rows-overlapped", "1x2-rows-overlapped", "2x1-eight-planes",
                              "1x3-rows-overlapped-eight-planes", "1x2-rows-overlapped-welch", "1x2-rows-overlapped-welch-eight-planes"])
def test_blocks_equal_whole_film(gpu, world, rows, BH, g8):
    """(The window-sweep split is pinned to the same value on both sides -- statmc_set_filter_split, the declared
    per-device setting: that is what makes the comparison bit for bit; tests/test_gpu_fullsize.py has the default dispatch.)"""
    from statmc_amd import pipeline, sharding
    dev = torch.device("cuda:0")
    gx, gy = _grid(world, rows)
    mode = g8
    welch = g8 in ("welch", "welch8")
    g8 = g8 is True or g8 == "welch8"
    whole = _film_samples(world, rows, BH, g8)
    gpu.force_filter_parts(2)
    if welch:
        gpu.set_filter_spec(dof=1)
    try:
        one = pipeline.BlockPipeline(sharding.BlockLayout(0, 1, gx * BW, gy * BH, RADIUS), dev, TYPES8 if g8 else TYPES, radius=RADIUS,
                                     **(dict(g_buffers=G8) if g8 else {}))
        one.accumulate({k: v.to(dev) for k, v in whole.items()})
        ref = one.denoise().cpu().numpy()
    finally:
        gpu.force_filter_parts(0)
        gpu.set_filter_spec()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(rk, world, rows, port, q, BH, mode)) for rk in range(world)]
    for p in procs:
        p.start()
    got = []
    try:
        for _ in range(world):
            item = q.get(timeout=120)
            assert item[1] != "error", item
            got.append(item)
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    assert sorted(g[0] for g in got) == list(range(world))
    for rank, ox, oy, blk in got:
        assert np.array_equal(blk, ref[oy:oy + BH, ox:ox + BW]), rank     # bit-identical
