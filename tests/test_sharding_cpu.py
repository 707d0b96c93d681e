"""Multi-GPU path on CPU: block layout + halo exchange over gloo (world_size 2 and 4), and the
filter applied per block on the exchanged halos equals the filter of the whole film."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bw, bh, r, rows, q, welch=False):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle
    from statmc_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L = sharding.BlockLayout(rank, world, bw, bh, r, grid=sharding.row_strips(world) if rows else None)
        fw, fh = L.film_size
        rng = np.random.default_rng(123)                      # every rank builds the same film
        film = {
            "mc": rng.random((fh, fw, 3), dtype=np.float32),
            "disc": (0.2 * rng.random((fh, fw, 3))).astype(np.float32),
            "colour": rng.random((fh, fw, 3), dtype=np.float32),
            "g0": rng.random((fh, fw, 3), dtype=np.float32),
            "g1": rng.random((fh, fw, 3), dtype=np.float32),
        }
        ox, oy = L.origin
        # Welch degrees of freedom: the block + halo image has a 16th channel, the bits of the pixel's int32 sample count;
        # welch == "g8": + a depth G-buffer -- 18 channels: 15, 16 the 1-channel features (the second slot empty), 17 the count
        n_film = rng.integers(2, 40, size=(fh, fw)).astype(np.int32)
        depth = (3 * rng.random((fh, fw, 1))).astype(np.float32)
        g8 = welch == "g8"
        welch = bool(welch)
        ch = sharding.block_image_channels([3, 3, 1] if g8 else [3, 3], welch)
        assert ch == (18 if g8 else 16 if welch else 15)
        n_at = 17 if g8 else 15
        packed = L.new_padded(ch, "cpu")
        packed.fill_(float("nan"))
        inner = L.interior(packed)
        for i, k in enumerate(("mc", "disc", "colour", "g0", "g1")):
            inner[..., 3 * i:3 * i + 3] = torch.from_numpy(film[k][oy:oy + bh, ox:ox + bw])
        if g8:
            inner[..., 15] = torch.from_numpy(depth[oy:oy + bh, ox:ox + bw, 0].copy())
            inner[..., 16] = 0.0
        if welch:
            inner[..., n_at] = torch.from_numpy(n_film[oy:oy + bh, ox:ox + bw].copy()).view(torch.float32)
        if rows:   # row strips: the exchange in two halves (started, something else done, waited for) as the multi-GPU step runs it
            in_flight = sharding.exchange_halo_start(L, packed)
            busy = float(torch.ones(1000).sum())          # (what the rank does meanwhile: the rest of its accumulation)
            in_flight.wait()
            assert busy == 1000.0 and not in_flight.reqs
        else:
            sharding.exchange_halo(L, packed)
        # the padded block must now equal the film window around the block
        want = np.concatenate([film[k] for k in ("mc", "disc", "colour", "g0", "g1")], axis=2)[
            oy - L.pt:oy + bh + L.pb, ox - L.pl:ox + bw + L.pr]
        ok_halo = np.array_equal(packed.numpy()[..., :15], want)
        n_loc = None
        if welch:
            n_loc = np.ascontiguousarray(packed.numpy()[..., n_at]).view(np.int32)
            ok_halo = ok_halo and np.array_equal(n_loc, n_film[oy - L.pt:oy + bh + L.pb, ox - L.pl:ox + bw + L.pr])
        if g8:
            ok_halo = ok_halo and np.array_equal(packed.numpy()[..., 15], depth[oy - L.pt:oy + bh + L.pb, ox - L.pl:ox + bw + L.pr, 0])
        # filter the block with ROI = owned pixels, compare with the whole-film filter
        p = packed.numpy()
        loc = [np.ascontiguousarray(p[..., 3 * i:3 * i + 3]) for i in range(5)]
        drs, ds = [-0.5 / 0.3 ** 2, -0.5 / 0.5 ** 2], -0.5 / 4.0 ** 2
        g_loc, g_film = [loc[3], loc[4]], [film["g0"], film["g1"]]
        if g8:
            drs = drs + [-0.5 / 1.5 ** 2]
            g_loc.append(np.ascontiguousarray(p[..., 15:16]))
            g_film.append(depth)
        spec = oracle.FilterSpec(dof=oracle.DOF_WELCH) if welch else None
        out = oracle.filter_image(loc[0], loc[1], loc[2], g_loc, drs, ds, r, roi=L.roi, threads=1, spec=spec, n=n_loc)
        ref = oracle.filter_image(film["mc"], film["disc"], film["colour"], g_film, drs, ds, r,
                                  roi=(ox, oy, ox + bw, oy + bh), threads=1, spec=spec, n=n_film if welch else None)
        x0, y0, x1, y1 = L.roi
        ok_filter = np.array_equal(out[y0:y1, x0:x1], ref[oy:oy + bh, ox:ox + bw])
        # final gather: the blocks assembled on rank 0 are the whole-film filter output
        whole = sharding.gather_blocks(L, torch.from_numpy(np.ascontiguousarray(out[y0:y1, x0:x1])))
        if rank == 0:
            full = oracle.filter_image(film["mc"], film["disc"], film["colour"], g_film, drs, ds, r, threads=1,
                                       spec=spec, n=n_film if welch else None)
            ok_filter = ok_filter and whole.shape == full.shape and np.array_equal(whole.numpy(), full)
        else:
            ok_filter = ok_filter and whole is None
        q.put((rank, ok_halo, ok_filter))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,bw,bh,r,rows,welch", [(2, 24, 18, 5, False, False), (4, 16, 14, 6, False, False), (3, 20, 9, 4, True, False),
                                                      (4, 12, 10, 5, True, False), (2, 20, 12, 5, True, True), (4, 14, 12, 4, False, True),
                                                      (2, 20, 12, 5, True, "g8")])
def test_halo_exchange_and_block_filter(world, bw, bh, r, rows, welch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(rk, world, port, bw, bh, r, rows, q, welch)) for rk in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert res == [(rk, True, True) for rk in range(world)]


def test_layouts():
    from statmc_amd import sharding
    assert sharding.grid_for(1) == (1, 1) and sharding.grid_for(2) == (2, 1)
    assert sharding.grid_for(4) == (2, 2) and sharding.grid_for(8) == (4, 2)
    L = sharding.BlockLayout(5, 8, 1920, 1080, 20)          # block (1, 1) of a 4 x 2 grid
    assert (L.bx, L.by) == (1, 1) and (L.left, L.right, L.up, L.down) == (4, 6, 1, None)
    assert (L.pw, L.ph) == (1960, 1100) and L.roi == (20, 20, 1940, 1100)
    assert L.film_size == (7680, 2160) and L.origin == (1920, 1080)
    S = sharding.BlockLayout(3, 8, 1920, 1080, 20, grid=sharding.row_strips(8))   # bench.py's default grid
    assert (S.left, S.right, S.up, S.down) == (None, None, 2, 4) and (S.pw, S.ph) == (1920, 1120)
    assert S.film_size == (1920, 8640) and S.origin == (0, 3240) and S.roi == (0, 20, 1920, 1100)
    one = sharding.BlockLayout(0, 1, 1920, 1080, 20)
    assert (one.pw, one.ph) == (1920, 1080) and one.roi == (0, 0, 1920, 1080)


def test_blocks_smaller_than_the_radius_are_refused():
    """The exchange fetches one ring of neighbours: a block narrower than the radius would need the next ring."""
    from statmc_amd import sharding
    with pytest.raises(ValueError, match="smaller than the filter radius"):
        sharding.BlockLayout(0, 4, 64, 12, 20, grid=(1, 4))
    with pytest.raises(ValueError, match="smaller than the filter radius"):
        sharding.BlockLayout(1, 2, 16, 64, 20, grid=(2, 1))
    sharding.BlockLayout(0, 1, 8, 8, 20)                     # a single block has no neighbours: any size
    sharding.BlockLayout(0, 2, 20, 8, 20, grid=(2, 1))       # only the split direction counts


def test_block_image_channel_rules():
    """15 channels: exactly two RGB G-buffers (the shipped configurations); 17: every other set of up to two RGB and two
    1-channel images, absent slots zero (what statmc::FilmShards packs); Welch degrees of freedom: + the sample count, 16 and 18."""
    from statmc_amd import sharding
    assert sharding.block_image_channels([3, 3]) == 15
    for g in ([3], [], [3, 1], [3, 3, 1, 1], [1, 3, 1], [1]):
        assert sharding.block_image_channels(g) == 17, g
    assert sharding.block_image_channels([3, 3], welch=True) == 16
    for g in ([3], [3, 3, 1], [1, 1], [3, 3, 1, 1]):
        assert sharding.block_image_channels(g, welch=True) == 18, g
