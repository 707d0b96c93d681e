"""Window-filter time of one rank's block (N = 2, 4, 8 row strips of a 1080p film) against the forced number of
window-sweep parts; the automatic choice is marked.  python tools/experiments/block_parts.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, pipeline, sharding, synthetic

FW, FH, spp = 1920, 1080, 16
dev = torch.device("cuda:0")
api.setup(0)
types = ["radiance", "normal", "albedo"]
for world in (2, 4, 8):
    grid = sharding.row_strips(world)
    W, H = FW // grid[0], FH // grid[1]
    L = sharding.BlockLayout(world // 2, world, W, H, 20, grid=grid)
    ox, oy = L.origin
    scene = synthetic.Scene(W, H, n_regions=12, seed=1, device=dev, x_offset=ox, y_offset=oy, full_width=FW, full_height=FH)
    pipe = pipeline.BlockPipeline(L, dev, types)
    pipe.accumulate(scene.samples(spp, seed=3, features=types))
    pipe.prepass()
    torch.cuda.synchronize()
    line = []
    for parts in (0, 1, 2, 3, 4, 5, 6, 7):
        api.force_filter_parts(parts)
        for _ in range(3):
            pipe.window_filter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            pipe.window_filter()
        e1.record()
        torch.cuda.synchronize()
        import ctypes
        used = api.load().statmc_debug_last_filter_parts()
        hi, rows = ctypes.c_int(0), ctypes.c_int(0)
        api.load().statmc_debug_last_filter_tail(ctypes.byref(hi), ctypes.byref(rows))
        tail = " (+%d parts on the last %d tile rows)" % (hi.value, rows.value) if hi.value else ""
        line.append("%s%d%s: %.3f" % ("auto=" if parts == 0 else "", used, tail, e0.elapsed_time(e1) / 30))
    api.force_filter_parts(0)
    print("N=%d block %dx%d: %s ms" % (world, W, H, "  ".join(line)), flush=True)
