"""Round 6: the tile-fed accumulation at the batch lengths the reference's schedule opens with, with and without the
prefetch of the next item's tile record (statmc_debug_accumulate_tiles_variant order 2 | 18), interleaved in one process; the film-major
launch on the same samples beside it.  python tools/experiments/time_tiles_prefetch.py [W H]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
dev = torch.device("cuda:0")
api.setup(0)
lib = api.load()
lib.statmc_debug_accumulate_tiles_variant.argtypes = [ctypes.c_int] * 3
types = list(synthetic.FEATURES)
tiles = [(x, y, min(x + 16, W), min(y + 16, H)) for y in range(0, H, 16) for x in range(0, W, 16)]
bounds = torch.tensor(tiles, dtype=torch.int32, device=dev)
npx = torch.tensor([(x1 - x0) * (y1 - y0) for x0, y0, x1, y1 in tiles], dtype=torch.int64)


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    runs = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        runs.append(e0.elapsed_time(e1) / n)
    return sorted(runs)[1]


for S in (4, 8, 16, 32, 64):
    smp = {t: torch.empty((S, H, W, synthetic.CHANNELS[t]), device=dev).uniform_() for t in types}
    bpp = sum(4 * film.STAT_TYPES[t]["channels"] * S + 2 * (4 + 4 * film.STAT_TYPES[t]["channels"] * (film.STAT_TYPES[t]["max_moment"] + (2 if film.STAT_TYPES[t]["transform"] else 0))) for t in types)
    frac = lambda ms: bpp * W * H / ms / 8e9
    fs = film.FilmStats(W, H, dev, types=types)
    t_film = timeit(lambda: fs.accumulate(smp))
    offs = (torch.cumsum(npx * S, 0) - npx * S).to(dev)
    cnt = torch.full((len(tiles),), S, dtype=torch.int32, device=dev)
    st2 = film.FilmStats(W, H, dev, types=types)
    sts, keep = [], []
    for t in types:
        c = film.STAT_TYPES[t]["channels"]
        arena = torch.empty((int((npx * S).sum()) * c,), device=dev)
        pos = 0
        for y in range(0, H, 16):
            th = min(16, H - y)
            band = smp[t][:, y:y + th].reshape(S, th, W // 16, 16, c).permute(2, 0, 1, 3, 4).contiguous().reshape(-1)
            arena[pos:pos + band.numel()] = band
            pos += band.numel()
        keep.append(arena)
        sts.append(api.make_stat_type_arena(arena, c, st2.state[t], film.STAT_TYPES[t]["transform"], film.STAT_TYPES[t]["max_moment"]))
    run = lambda: api.accumulate_tiles(W, H, sts, bounds, offs, cnt)
    res = {}
    for rep in range(2):
        for order in (18, 2):
            lib.statmc_debug_accumulate_tiles_variant(2, order, 0)
            res.setdefault(order, []).append(timeit(run))
    lib.statmc_debug_accumulate_tiles_variant(2, 2, 0)
    # same bits either way
    a_, b_ = film.FilmStats(W, H, dev, types=types), film.FilmStats(W, H, dev, types=types)
    outs = []
    for order, f in ((18, a_), (2, b_)):
        lib.statmc_debug_accumulate_tiles_variant(2, order, 0)
        s2 = [api.make_stat_type_arena(keep[i], film.STAT_TYPES[t]["channels"], f.state[t], film.STAT_TYPES[t]["transform"], film.STAT_TYPES[t]["max_moment"]) for i, t in enumerate(types)]
        api.accumulate_tiles(W, H, s2, bounds, offs, cnt)
    torch.cuda.synchronize()
    lib.statmc_debug_accumulate_tiles_variant(2, 2, 0)
    same = all(torch.equal(a_.state[t][k], b_.state[t][k]) for t in types for k in a_.state[t] if a_.state[t][k] is not None)
    print("%dx%d S=%2d  film-major %.4f ms (%.3f)  tile-fed without prefetch %.4f ms (%.3f)  with %.4f ms (%.3f)  same bits %s"
          % (W, H, S, t_film, frac(t_film), min(res[18]), frac(min(res[18])), min(res[2]), frac(min(res[2])), same), flush=True)
    del smp, keep, sts
    torch.cuda.empty_cache()
