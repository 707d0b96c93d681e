"""Tile-fed accumulation at the reference's batch lengths (4 .. 64 samples per 16 x 16 tile), 1080p and 4K, placed buffers: which
(tile, type) item a wave takes (order 0: the waves of a workgroup = the types of one tile; 1: tiles innermost; 2: a workgroup = four
consecutive tiles of ONE type, so a state row of the workgroup is 768 contiguous bytes instead of 192) and how many workgroups
the grid holds per CU (0 = the default: a cap of 8 up to ~ 4 Mpixels, one item per wave beyond; 4096 = one item per wave, no grid-stride walk),
and whether the first rows of the LDS-DMA ring are requested before the state loads (f1: no gain).
python tools/experiments/time_tiles_order.py [4k]"""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from statmc_amd import api, synthetic

dev = torch.device("cuda:0")
api.setup(0)
bench.torch = torch
bench.PLACED["on"] = os.environ.get("PLACED", "1") == "1"
lib = api.load()
lib.statmc_debug_accumulate_tiles_variant.argtypes = [ctypes.c_int] * 3
types = list(synthetic.FEATURES)
for W, H in ((3840, 2160),) if "4k" in sys.argv else ((1920, 1080), (3840, 2160)):
    S_all = 64
    scene = synthetic.Scene(W, H, n_regions=12, seed=1, device=dev)
    smp = {t: bench.new_arena((S_all, H, W, synthetic.CHANNELS[t]), dev) for t in types}
    for s0 in range(0, S_all, 16):
        part = scene.samples(16, seed=77 + s0, features=types)
        for t in types:
            smp[t][s0:s0 + 16] = part[t]
        del part
    for S in (4, 8, 16, 64):
        line = "%dx%d %2d spp:" % (W, H, S)
        for order, wg, first in ((0, 0, 0), (2, 0, 0), (2, 0, 1), (2, 4096, 0), (2, 4096, 1), (2, 0, 0), (2, 0, 1)):
            lib.statmc_debug_accumulate_tiles_variant(2, order, wg)
            lib.statmc_debug_accumulate_launch(-1, first)      # (first: the ring's first rows requested before the state loads)
            r = bench._tile_fed_measure(W, H, dev, smp, types, S, reps=6)
            line += "  o%d/wg%-4d/f%d %.4f ms %.3f" % (order, wg, first, r["avg_ms"], r["frac_hbm"])
        lib.statmc_debug_accumulate_launch(-1, 0)
        print(line, flush=True)
    lib.statmc_debug_accumulate_tiles_variant(2, 2, 0)
    del smp
    torch.cuda.empty_cache()
