"""A/B of the film-major accumulation's launch shape (round 4): capped grid with slots per type and a
grid-stride walk (0) against one pass per workgroup with the types round-robin (1); the LDS-DMA ring's first rows requested
behind (0) or before (1) the state loads.  Films 720p / 1080p / 4K, launches of 4 .. 256 samples per pixel; the four
shapes must leave the same bits.
python tools/experiments/time_accumulate_launch.py        (PLACED=1, round 5: moments and arenas from statmc_malloc_placed, one state per
                                                          shape so that every launch shape runs on the same buffers)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

dev = torch.device("cuda:0")
api.setup(0)
lib = api.load()
types = list(synthetic.FEATURES)
PLACED = os.environ.get("PLACED") == "1"


def bpp(S):
    t = 0
    for x in types:
        c = film.STAT_TYPES[x]
        planes = c["max_moment"] + (2 if c["transform"] else 0)
        t += 4 * c["channels"] * S + 2 * (4 + 4 * c["channels"] * planes)
    return t


SHAPES = ((1280, 720, (64,)), (1920, 1080, (4, 8, 16, 32, 64, 256)), (3840, 2160, (4, 8, 16, 32, 64)))
if os.environ.get("LONG"):      # long batches only (round 5: where does the resident grid win?)
    SHAPES = ((1280, 720, (128, 256, 512)), (1920, 1080, (96, 128, 256, 512)), (3840, 2160, (96, 128)))
for W, H, spps in SHAPES:
    scene = synthetic.Scene(W, H, seed=1, device=dev)
    Smax = max(spps)
    smp = {t: (api.empty_placed((Smax, H, W, synthetic.CHANNELS[t]), torch.float32, dev, api.MEM_STREAM) if PLACED else
               torch.empty((Smax, H, W, synthetic.CHANNELS[t]), device=dev)) for t in types}
    fs_shared = film.FilmStats(W, H, dev, types=types, placed=True) if PLACED else None
    for s0 in range(0, Smax, 32):
        part = scene.samples(min(32, Smax - s0), seed=7 + s0, features=types)
        for t in types:
            smp[t][s0:s0 + part[t].shape[0]] = part[t]
        del part
    for S in spps:
        part = {t: v[:S] for t, v in smp.items()}
        ref = None
        line = "%dx%d %3d spp (%5d B/px):" % (W, H, S, bpp(S))
        # (grid 2, round 5: the resident grid -- one workgroup per CU, each walking every stat type -- through statmc_debug_accumulate_resident_blocks)
        for grid_mode, dma_first in ((0, 0), (1, 0), (2, 0), (0, 1), (1, 1)) if os.environ.get("RESIDENT") else ((0, 0), (1, 0), (0, 1), (1, 1)):
            api.check(lib.statmc_debug_accumulate_resident_blocks(int(os.environ.get("RESIDENT", 256)) if grid_mode == 2 else 0))
            api.check(lib.statmc_debug_accumulate_launch(0 if grid_mode == 2 else grid_mode, dma_first))
            fs = fs_shared if PLACED else film.FilmStats(W, H, dev, types=types)
            fs.reset()
            fs.accumulate(part)
            torch.cuda.synchronize()
            state = torch.cat([v.reshape(-1).view(torch.int32) for st in fs.state.values() for v in st.values() if v is not None])
            if ref is None:
                ref = state
            assert torch.equal(ref, state), (W, H, S, grid_mode, dma_first)
            reps = 20 if S <= 64 else 6
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fs.accumulate(part)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / reps)
            line += "  grid %d dma_first %d: %.4f ms %.2f TB/s" % (grid_mode, dma_first, best, bpp(S) * W * H / best / 1e9)
            if not PLACED:
                del fs
        print(line, flush=True)
    del smp
api.check(lib.statmc_debug_accumulate_launch(-1, 0))
api.check(lib.statmc_debug_accumulate_resident_blocks(0))
