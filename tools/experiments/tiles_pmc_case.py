"""Round 6: the workload of tiles_pmc.sh -- at 1080p, all stat types, S = $S samples per pixel (default 4): three
launches of the film-major accumulation (accumulate_kernel), then three of the tile-fed one (accumulate_tiles_kernel) on the same samples
in 16 x 16 tile blocks.  The two kernels have different names, so the counter rows tell them apart."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

dev = torch.device("cuda:0")
api.setup(0)
types = list(synthetic.FEATURES)
W, H, S = 1920, 1080, int(os.environ.get("S", "4"))
smp = {t: torch.empty((S, H, W, synthetic.CHANNELS[t]), device=dev).uniform_() for t in types}
fs = film.FilmStats(W, H, dev, types=types)
for _ in range(3):
    fs.accumulate(smp)
torch.cuda.synchronize()
tiles = [(x, y, min(x + 16, W), min(y + 16, H)) for y in range(0, H, 16) for x in range(0, W, 16)]
bounds = torch.tensor(tiles, dtype=torch.int32, device=dev)
npx = torch.tensor([(x1 - x0) * (y1 - y0) for x0, y0, x1, y1 in tiles], dtype=torch.int64)
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
for _ in range(3):
    api.accumulate_tiles(W, H, sts, bounds, offs, cnt)
torch.cuda.synchronize()
bpp = sum(4 * film.STAT_TYPES[t]["channels"] * S + 2 * (4 + 4 * film.STAT_TYPES[t]["channels"] * ({1: 1, 2: 2, 3: 3}[film.STAT_TYPES[t]["max_moment"]] + (2 if film.STAT_TYPES[t]["transform"] else 0))) for t in types)
print("S=%d algorithmic bytes per launch %d (reads %d, writes %d)" % (S, bpp * W * H, (bpp - 112) * W * H, 112 * W * H), flush=True)
