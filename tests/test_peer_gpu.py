"""ONE process, every block (statmc_amd/peer.py): the whole step through the C ABI -- statmc_accumulate_row_ranges ->
statmc_prepass_pack_rows -> statmc_halo_exchange (device-to-device copies) -> statmc_window_filter -- for row strips in the
overlapped order and for 2-D grids (two exchange phases), six and eight feature planes, several steps in a row (the packed
images are rewritten every step).  All blocks live on cuda:0 here (the test box has one GPU): the same code drives one
block per device on a multi-GPU node.  The assembled film equals the whole film, bit for bit under a pinned split."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TYPES = ("radiance", "normal", "albedo")
TYPES8 = ("radiance", "normal", "albedo", "depth", "materialid")
G8 = ("materialid", "depth", "normal", "albedo")


@pytest.mark.parametrize("grid,bw,bh,g8,radius", [((1, 3), 272, 56, False, 20), ((2, 2), 144, 40, False, 20), ((2, 1), 136, 48, True, 20),
                                                  ((1, 2), 260, 64, True, 6), ((1, 4), 128, 24, False, 20), ((1, 2), 272, 64, "welch", 20),
                                                  ((2, 2), 144, 40, "welch", 7), ((1, 2), 272, 64, "one-rgb", 20), ((1, 2), 272, 64, "welch8", 20),
                                                  ((2, 2), 144, 40, "welch8", 9), ((2, 2), 144, 40, "welch-clamp", 9), ((1, 2), 272, 64, "welch8-clamp", 20),
                                                  ((2, 1), 136, 48, "g8-clamp", 20)],
                         ids=["1x3-overlapped", "2x2", "2x1-eight-planes", "1x2-overlapped-eight-planes-r6", "1x4-short-strips",
                              "1x2-overlapped-welch", "2x2-welch-r7", "1x2-one-rgb-gbuffer-in-the-17-channel-image",
                              "1x2-overlapped-welch-eight-planes", "2x2-welch-eight-planes-r9", "2x2-welch-clamped-border-r9",
                              "1x2-overlapped-welch-eight-planes-clamped-border", "2x1-eight-planes-clamped-border"])
def test_peer_film_equals_whole_film(gpu, grid, bw, bh, g8, radius):
    from statmc_amd import peer, pipeline, sharding, synthetic
    gx, gy = grid
    world = gx * gy
    # "welch": Welch degrees of freedom (the device's filter spec): the block + halo image has a 16th channel, the sample count
    mode = g8 if isinstance(g8, str) else ""
    welch = "welch" in mode              # "welch8": + depth and material id among the G-buffers -- the 18-channel image, eight-plane Welch builds
    clamp = "clamp" in mode              # a clamped border on film blocks under Welch: the border kernel reads the block + halo image
    one_rgb = g8 == "one-rgb"      # any set other than exactly two RGB G-buffers travels in the 17-channel image, 1-channel slots empty
    g8 = g8 is True or "welch8" in mode or "g8" in mode
    types = TYPES8 if g8 else TYPES
    kw = dict(g_buffers=G8) if g8 else dict(g_buffers=("albedo",)) if one_rgb else {}
    scene = synthetic.Scene(gx * bw, gy * bh, n_regions=7, seed=5)
    batches = [scene.samples(4, seed=6, features=types), scene.samples(3, seed=7, features=types)]     # two steps: 4 + 3 samples
    gpu.set_filter_split(2)
    if welch or clamp:
        gpu.set_filter_spec(dof=1 if welch else 0, border=1 if clamp else 0)
    try:
        one = pipeline.BlockPipeline(sharding.BlockLayout(0, 1, gx * bw, gy * bh, radius), DEV, types, radius=radius, filter_sd=radius / 2.0, **kw)
        refs = []
        for smp in batches:
            one.accumulate({k: v.to(DEV) for k, v in smp.items()})
            refs.append(one.denoise().clone())
        pf = peer.PeerFilm(world, bw, bh, radius, [0] * world, types, filter_sd=radius / 2.0, grid=grid, **kw)
        assert pf.overlap == (gx == 1 and bh >= 2 * radius + 8)
        assert pf.blocks[0].packed.shape[2] == ((18 if welch else 17) if (g8 or one_rgb) else 16 if welch else 15)
        for smp, ref in zip(batches, refs):
            per_block = []
            for blk in pf.blocks:
                ox, oy = blk.layout.origin
                per_block.append([{k: v[:, oy:oy + bh, ox:ox + bw].contiguous().to(DEV) for k, v in smp.items()}])
            pf.run(pf.prepare_step(per_block))
            pf.synchronize()
            got = pf.gather()
            pf.synchronize()
            assert torch.equal(got, ref)
            if welch:
                assert gpu.last_filter_variant() == ("sym_welch_g8" if g8 else "sym_welch") + ("_clamp" if clamp else "")
            elif clamp:
                assert gpu.last_filter_variant() == "sym_r20_g8_clamp"
        # the plain order on a grid that overlaps by default: same bits
        if pf.overlap:
            pf.reset()
            pf.synchronize()
            for smp in batches:
                per_block = []
                for blk in pf.blocks:
                    ox, oy = blk.layout.origin
                    per_block.append([{k: v[:, oy:oy + bh, ox:ox + bw].contiguous().to(DEV) for k, v in smp.items()}])
                pf.run(pf.prepare_step(per_block, overlap=False))
            pf.synchronize()
            assert torch.equal(pf.gather(), refs[-1])
            pf.synchronize()
    finally:
        gpu.set_filter_split(0)
        gpu.set_filter_spec()


def test_rccl_binds_at_run_time_and_makes_a_communicator(gpu):
    """The C ABI's RCCL entry points on ONE device: the library has no link-time dependency on RCCL, binds the copy the process
    already has (torch's) or librccl.so.1 on first use, and a one-rank communicator made through statmc_rccl_unique_id /
    statmc_rccl_comm_create exists and goes away again; with one block there is nothing to exchange.  (Two ranks need two
    devices: tests/test_multidevice_gpu.py::test_rccl_halo_exchange_through_the_c_abi.)"""
    import ctypes as C
    from statmc_amd.peer import Block
    lib = gpu.load()
    assert lib.statmc_rccl_available() == 1, gpu.load().statmc_last_error()
    comm = gpu.RcclComm(1, 0, gpu.RcclComm.unique_id())
    assert comm.handle.value
    packed = torch.zeros(32, 48, 15, device=DEV)
    gpu.halo_exchange_rccl(packed, 0, 1, 1, 48, 32, 20, comm)                 # 1 x 1 grid: returns at once
    blk = Block()
    blk.device, blk.packed, blk.stream = 0, gpu.image_of(packed), gpu.current_stream_handle()
    # a grid the communicator does not match is refused before anything is sent
    assert lib.statmc_halo_exchange_rccl(C.byref(blk), 1, 2, 48, 32, 20, comm.handle, 0) == gpu.ERR_INVALID
    assert b"communicator" in lib.statmc_last_error()
    torch.cuda.synchronize()
    comm.destroy()
