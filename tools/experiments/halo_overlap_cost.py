"""What overlapping the halo exchange with the filter of the block interior would cost on the compute side: the window
filter of one rank's block + halo image (N = 2, 4, 8 row strips of a 1080p film, a middle rank) as ONE launch over the
owned rows against THREE (the rows that need no halo first, then the two r-row strips that do) -- same bits, since the
tile grid is anchored in film coordinates.  The exchange such a split could hide is 2.3 MB per neighbour.
python tools/experiments/halo_overlap_cost.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, pipeline, sharding, synthetic

FW, FH, spp, R = 1920, 1080, 16, 20
SPP_ACC = 256
dev = torch.device("cuda:0")
api.setup(0)
types = ["radiance", "normal", "albedo"]


def timed(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for world in (2, 4, 8):
    grid = sharding.row_strips(world)
    W, H = FW // grid[0], FH // grid[1]
    L = sharding.BlockLayout(world // 2, world, W, H, R, grid=grid)
    ox, oy = L.origin
    scene = synthetic.Scene(W, H, n_regions=12, seed=1, device=dev, x_offset=ox, y_offset=oy, full_width=FW, full_height=FH)
    pipe = pipeline.BlockPipeline(L, dev, types)
    pipe.accumulate(scene.samples(spp, seed=3, features=types))
    pipe.prepass()
    pipe.packed[:L.pt].copy_(pipe.packed[L.pt:2 * L.pt])          # stand-ins for the neighbours' rows
    pipe.packed[L.pt + H:].copy_(pipe.packed[H:H + L.pb])
    torch.cuda.synchronize()
    x0, y0, x1, y1 = L.roi

    def launch(roi, out):
        a, keep = api.make_filter_args(n=[], mean=[], m2=[], m3=[], film=[], mean_corr=[], disc=[], film_filtered=[out],
                                       g_buffers=[], g_sds=pipe.fs.g_sds, filter_sd=pipe.filter_sd, radius=R, roi=roi,
                                       packed=pipe.packed, film_origin=(ox - L.pl, oy - L.pt))
        api.window_filter(a, 3)

    one, three = L.new_padded(3, dev), L.new_padded(3, dev)
    t_one = timed(lambda: launch((x0, y0, x1, y1), one))

    def split():
        launch((x0, y0 + R, x1, y1 - R), three)                    # needs no halo row
        launch((x0, y0, x1, y0 + R), three)
        launch((x0, y1 - R, x1, y1), three)
    t_three = timed(split)
    t_interior = timed(lambda: launch((x0, y0 + R, x1, y1 - R), three))
    split()
    torch.cuda.synchronize()
    same = torch.equal(one.view(torch.int32), three.view(torch.int32))
    print("N=%d block %dx%d (+%d halo rows): one launch %.3f ms | interior %.3f + two strips = %.3f ms (+%.3f) | same bits: %s"
          % (world, W, H, L.pt + L.pb, t_one, t_interior, t_three, t_three - t_one, same), flush=True)


# ---- the other way of hiding the exchange: the rows a neighbour needs accumulated, pre-passed and sent first, the rest of
# the block accumulated while they travel (BlockPipeline.accumulate_and_denoise).  Compute-side cost: the accumulation +
# pre-pass of a middle rank's block in one piece against border rows + interior (three + three launches).
for world in (2, 4, 8):
    grid = sharding.row_strips(world)
    W, H = FW // grid[0], FH // grid[1]
    L = sharding.BlockLayout(world // 2, world, W, H, R, grid=grid)
    ox, oy = L.origin
    scene = synthetic.Scene(W, H, n_regions=12, seed=1, device=dev, x_offset=ox, y_offset=oy, full_width=FW, full_height=FH)
    pipe = pipeline.BlockPipeline(L, dev, list(synthetic.FEATURES))
    smp = {t: torch.cat([scene.samples(32, seed=5 + i, features=synthetic.FEATURES)[t] for i in range(SPP_ACC // 32)]) for t in synthetic.FEATURES}

    def one_piece():
        pipe.accumulate(smp)
        pipe.prepass()

    def split():
        pipe.accumulate(smp, rows=pipe.border_rows())
        pipe.prepass(rows=pipe.border_rows())
        rows = pipe.interior_rows()
        pipe.accumulate(smp, rows=rows)
        pipe.prepass(rows=rows)

    def borders_only():
        pipe.accumulate(smp, rows=pipe.border_rows())
        pipe.prepass(rows=pipe.border_rows())

    t_one, t_split, t_b = timed(one_piece, 10), timed(split, 10), timed(borders_only, 10)
    print("N=%d block %dx%d, %d spp, 11 channels: accumulate + pre-pass in one piece %.3f ms | border rows %s first %.3f ms, then the rest: %.3f ms (+%.3f); the exchange has %.3f ms to hide in"
          % (world, W, H, SPP_ACC, t_one, pipe.border_rows(), t_b, t_split, t_split - t_one, t_split - t_b), flush=True)
