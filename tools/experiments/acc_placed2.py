"""Round 5: does it matter whether the five sample arenas of a launch sit in ONE run of class-B slots or in several?
One process, one state (class A): first one placed block of the arenas' total size, the arenas carved out of it; then five
placed allocations (what is left of the B runs); then torch's allocator.  1080p / 256 spp, all stat types.
python tools/experiments/acc_placed2.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

dev = torch.device("cuda:0")
api.setup(0)
types = list(synthetic.FEATURES)
W, H, S = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 256


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


fs = film.FilmStats(W, H, dev, types=types, placed=True)
n_el = {t: S * H * W * synthetic.CHANNELS[t] for t in types}
big = api.empty_placed((sum(n_el.values()),), torch.float32, dev, api.MEM_STREAM)
one, pos = {}, 0
for t in types:
    one[t] = big[pos:pos + n_el[t]].view(S, H, W, synthetic.CHANNELS[t])
    pos += n_el[t]
    for s0 in range(0, S, 16):
        one[t][s0:s0 + 16].uniform_()
print("map after the one block:   ", api.placement_info()["map"], flush=True)
five = {t: api.empty_placed((S, H, W, synthetic.CHANNELS[t]), torch.float32, dev, api.MEM_STREAM) for t in types}
for t in types:
    five[t].copy_(one[t])
print("map after five more blocks:", api.placement_info()["map"], flush=True)
plain = {t: one[t].clone() for t in types}
fs_t = film.FilmStats(W, H, dev, types=types)
for rnd in range(2):
    for name, f, a in (("arenas in one placed block", fs, one), ("five placed blocks", fs, five), ("torch arenas, placed state", fs, plain), ("torch arenas, torch state", fs_t, plain),
                       ("placed block, torch state", fs_t, one)):
        ms = timed(f, a)
        print("%-28s %.3f ms  %.3f of 8 TB/s" % (name, ms, bpp(S) * W * H / ms / 8e9), flush=True)
