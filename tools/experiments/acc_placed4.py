"""Round 5: bench.py (moments allocated before the arenas) measures the placed accumulation at 0.80 - 0.81 of the peak,
acc_placed.py (arenas first) mostly at 0.84.  Is it where in the state role's memory the moments sit?  One process, one set
of arenas, all stat types, 1080p / 256 spp: a state allocated before the arenas, one after, one after a spacer (another
slot), and one in torch's memory.  python tools/experiments/acc_placed4.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

dev = torch.device("cuda:0")
api.setup(0)
types = list(synthetic.FEATURES)
W, H, S = 1920, 1080, 256


def bpp(S):
    t = 0
    for x in types:
        c = film.STAT_TYPES[x]
        planes = c["max_moment"] + (2 if c["transform"] else 0)
        t += 4 * c["channels"] * S + 2 * (4 + 4 * c["channels"] * planes)
    return t


def timed(fs, a, reps=8):
    fs.accumulate(a)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fs.accumulate(a)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


order = os.environ.get("ORDER", "state-first")
states = {}
if order == "state-first":
    states["state before the arenas"] = film.FilmStats(W, H, dev, types=types, placed=True)
arenas = {t: api.empty_placed((S, H, W, synthetic.CHANNELS[t]), torch.float32, dev, api.MEM_STREAM) for t in types}
for t in types:
    for s0 in range(0, S, 16):
        arenas[t][s0:s0 + 16].uniform_()
states["state after the arenas"] = film.FilmStats(W, H, dev, types=types, placed=True)
spacer = api.empty_placed((1 << 28,), torch.float32, dev, api.MEM_STATE)          # 1 GiB: the next state lies in another slot
states["state in another slot"] = film.FilmStats(W, H, dev, types=types, placed=True)
states["state in torch's memory"] = film.FilmStats(W, H, dev, types=types)
info = api.placement_info()
print("order:", order, " map:", info["map"], flush=True)
base = min(a.data_ptr() for a in arenas.values())
p0 = None
for name, fs in states.items():
    ptr = fs.state["radiance"]["n"].data_ptr()
    print("%-28s radiance n at %+8.3f GiB from the first arena" % (name, (ptr - base) / 2 ** 30), flush=True)
for rnd in range(2):
    for name, fs in states.items():
        ms = timed(fs, arenas)
        print("%-28s %.3f ms  %.3f of 8 TB/s" % (name, ms, bpp(S) * W * H / ms / 8e9), flush=True)
