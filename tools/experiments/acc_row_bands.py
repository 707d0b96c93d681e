"""Does the film-major accumulation of a 4K film lose to the 1080p one because of what a launch spans?  One launch over the
whole film against the same film in 2 / 4 / 8 / 16 bands of rows (statmc_accumulate_row_ranges), each band its own launch,
back to back on one stream; and a 1080p film for scale.  64 and 16 samples per pixel, all stat types.
python tools/experiments/acc_row_bands.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

dev = torch.device("cuda:0")
api.setup(0)
types = list(synthetic.FEATURES)


def bpp(S):
    t = 0
    for x in types:
        c = film.STAT_TYPES[x]
        planes = c["max_moment"] + (2 if c["transform"] else 0)
        t += 4 * c["channels"] * S + 2 * (4 + 4 * c["channels"] * planes)
    return t


for W, H in ((1920, 1080), (3840, 2160)):
    scene = synthetic.Scene(W, H, seed=1, device=dev)
    Smax = 64
    smp = {t: torch.empty((Smax, H, W, synthetic.CHANNELS[t]), device=dev) for t in types}
    for s0 in range(0, Smax, 16):
        part = scene.samples(16, seed=7 + s0, features=types)
        for t in types:
            smp[t][s0:s0 + 16] = part[t]
        del part
    for S in (64, 16):
        part = {t: v[:S] for t, v in smp.items()}
        line = "%dx%d %2d spp:" % (W, H, S)
        for bands in (1, 2, 4, 8, 16):
            fs = film.FilmStats(W, H, dev, types=types)
            edges = [H * b // bands for b in range(bands + 1)]
            def run():
                if bands == 1:
                    fs.accumulate(part)
                else:
                    for b in range(bands):
                        fs.accumulate(part, rows=(edges[b], edges[b + 1]))
            run()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8):
                    run()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 8)
            line += "  %2d band%s %.3f ms %.2f TB/s" % (bands, " " if bands == 1 else "s", best, bpp(S) * W * H / best / 1e9)
            del fs
        print(line, flush=True)
    del smp
