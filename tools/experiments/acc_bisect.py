"""Round 5: what separates acc_placed.py (capped-grid launch at 0.842 of the peak) from every other harness (0.81)?  V = letters:
  a  nothing but the placed buffers;  p  five written torch arenas of the same sizes allocated first, the placed arenas filled by copy_;
  t  + a torch FilmStats alive;  x  + the torch launch timed first.   Result: p alone does it (0.809 -> 0.842); cause unknown.
V=p python tools/experiments/acc_bisect.py"""
import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
types = list(synthetic.FEATURES)
W, H, S = 1920, 1080, 256
V = os.environ.get("V", "a")
def bpp(S):
    t = 0
    for x in types:
        c = film.STAT_TYPES[x]; planes = c["max_moment"] + (2 if c["transform"] else 0)
        t += 4 * c["channels"] * S + 2 * (4 + 4 * c["channels"] * planes)
    return t
def timed(fs, a, reps):
    fs.accumulate(a); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fs.accumulate(a)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best
lib = api.load()
plain = None
if "p" in V:      # torch arenas first (written)
    plain = {t: torch.empty((S, H, W, synthetic.CHANNELS[t]), device=dev) for t in types}
    for t in types:
        for s0 in range(0, S, 16): plain[t][s0:s0 + 16].uniform_()
placed = {t: api.empty_placed((S, H, W, synthetic.CHANNELS[t]), torch.float32, dev, api.MEM_STREAM) for t in types}
fs_p = film.FilmStats(W, H, dev, types=types, placed=True)
for t in types:
    if plain is not None: placed[t].copy_(plain[t])
    else:
        for s0 in range(0, S, 16): placed[t][s0:s0 + 16].uniform_()
fs_t = film.FilmStats(W, H, dev, types=types) if "t" in V else None
if "x" in V and plain is not None:      # time the torch launch first
    print("torch", timed(fs_t, plain, 7))
row = []
for g in (0, 1, 0, 1):
    api.check(lib.statmc_debug_accumulate_launch(g, 0))
    ms = timed(fs_p, placed, 7)
    row.append("grid %d %.3f ms %.3f" % (g, ms, bpp(S) * W * H / ms / 8e9))
print(V, "  ".join(row), api.placement_info()["map"])
