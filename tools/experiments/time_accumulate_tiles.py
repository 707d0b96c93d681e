"""Scratch: statmc_accumulate_tiles (16x16 tile blocks) against the film-major statmc_accumulate at 1080p."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
W, H = 1920, 1080
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
types = list(synthetic.FEATURES)
sc = synthetic.Scene(W, H, seed=1, device=dev)
chunks = [sc.samples(32, seed=10 + i, features=types) for i in range(S // 32)]
smp = {t: torch.cat([c[t] for c in chunks]) for t in types}
del chunks
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
bpp = sum(4 * film.STAT_TYPES[t]["channels"] * S + 2 * (4 + 4 * film.STAT_TYPES[t]["channels"] * (film.STAT_TYPES[t]["max_moment"] + (2 if film.STAT_TYPES[t]["transform"] else 0))) for t in types)
fs = film.FilmStats(W, H, dev, types=types)
t_film = min(timeit(lambda: fs.accumulate(smp)) for _ in range(2))
api.load().statmc_debug_accumulate_dma(0)
t_film_reg = min(timeit(lambda: fs.accumulate(smp)) for _ in range(2))
api.load().statmc_debug_accumulate_dma(1)
print("S=%d film-major: RGB planes by LDS-DMA %.3f ms (%.0f GB/s) | loads into registers %.3f ms (%.0f GB/s)"
      % (S, t_film, bpp * W * H / t_film / 1e6, t_film_reg, bpp * W * H / t_film_reg / 1e6), flush=True)
# the same samples as tile blocks [tile][S][16][16][C] (1080 = 67.5 tiles: the last tile row is 8 high)
tiles = [(x, y, min(x + 16, W), min(y + 16, H)) for y in range(0, H, 16) for x in range(0, W, 16)]
bounds = torch.tensor(tiles, dtype=torch.int32, device=dev)
npx = torch.tensor([(x1 - x0) * (y1 - y0) for x0, y0, x1, y1 in tiles], dtype=torch.int64)
offs = torch.cumsum(npx * S, 0) - npx * S
arenas = {}
for t in types:
    c = film.STAT_TYPES[t]["channels"]
    a = torch.empty(int((npx * S).sum()) * c, device=dev)
    for k, (x0, y0, x1, y1) in enumerate(tiles):
        blk = smp[t][:, y0:y1, x0:x1].reshape(-1) if smp[t].dim() == 4 else smp[t][:, y0:y1, x0:x1].reshape(-1)
        a[int(offs[k]) * c:int(offs[k]) * c + blk.numel()] = blk
    arenas[t] = a
fs2 = film.FilmStats(W, H, dev, types=types)
sts = [api.make_stat_type_arena(arenas[t], film.STAT_TYPES[t]["channels"], fs2.state[t], film.STAT_TYPES[t]["transform"], film.STAT_TYPES[t]["max_moment"]) for t in types]
offs_d = offs.to(dev); cnt = torch.full((len(tiles),), S, dtype=torch.int32, device=dev)
import ctypes
lib = api.load()
lib.statmc_debug_accumulate_tiles_variant.argtypes = [ctypes.c_int] * 3
for umul, order, wg, dma in ((2, 2, 0, 1), (2, 0, 0, 1), (2, 2, 0, 0), (1, 2, 0, 1), (2, 2, 4, 1)):
    lib.statmc_debug_accumulate_dma(dma)
    lib.statmc_debug_accumulate_tiles_variant(umul, order, wg)
    tt = min(timeit(lambda: api.accumulate_tiles(W, H, sts, bounds, offs_d, cnt)) for _ in range(2))
    print("S=%d tiles dma %d umul %d order %d wg/cu %2d: %.3f ms (%.0f GB/s, %.3f of 8 TB/s)" % (S, dma, umul, order, wg, tt, bpp * W * H / tt / 1e6, bpp * W * H / tt / 8e9), flush=True)
lib.statmc_debug_accumulate_dma(1)
lib.statmc_debug_accumulate_tiles_variant(2, 2, 0)
t_tiles = min(timeit(lambda: api.accumulate_tiles(W, H, sts, bounds, offs_d, cnt)) for _ in range(2))
fs3 = film.FilmStats(W, H, dev, types=types); fs3.accumulate(smp)
fs4 = film.FilmStats(W, H, dev, types=types)
sts4 = [api.make_stat_type_arena(arenas[t], film.STAT_TYPES[t]["channels"], fs4.state[t], film.STAT_TYPES[t]["transform"], film.STAT_TYPES[t]["max_moment"]) for t in types]
api.accumulate_tiles(W, H, sts4, bounds, offs_d, cnt); torch.cuda.synchronize()
same = all(torch.equal(fs3.state[t][k], fs4.state[t][k]) for t in types for k in fs3.state[t] if fs3.state[t][k] is not None)
print("S=%d: film-major %.3f ms (%.0f GB/s) | tile blocks %.3f ms (%.0f GB/s) | results identical: %s"
      % (S, t_film, bpp * W * H / t_film / 1e6, t_tiles, bpp * W * H / t_tiles / 1e6, same))
