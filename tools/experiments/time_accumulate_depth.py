"""Round 5: ring depth of the film-major accumulation's LDS-DMA walk (rows of the RGB sample planes in flight per wave), A/B in
ONE process on the same buffers (placement moves the same launch by up to 12 %: acc_gap2.py) -- needs a library built with
-DSTATMC_ACC_DMA_DEPTHS=1 (STATMC_VARIANT=tools/experiments/variants/acc_depths.so).  Depths are walked round-robin,
best of the rounds; every depth must leave the bits of depth 3.
python tools/experiments/time_accumulate_depth.py [depths]      e.g. 3,4,5,6,0"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import build
if os.environ.get("STATMC_VARIANT"):
    os.environ.setdefault("STATMC_ALLOW_DIAGNOSTIC_BUILD", "1")
    build.SO = os.path.abspath(os.environ["STATMC_VARIANT"])
from statmc_amd import api, film, synthetic

dev = torch.device("cuda:0")
api.setup(0)
lib = api.load()
types = list(synthetic.FEATURES)
# a depth >= 100 means: depth - 100 (the default ring) on the build for three waves per SIMD (-DSTATMC_ACC_OCC_AB=1)
depths = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3,4,5,6,0").split(",")]
shapes = [tuple(int(x) for x in sh.split("x")) for sh in (sys.argv[2] if len(sys.argv) > 2 else "1920x1080x256,1920x1080x64,1920x1080x16,3840x2160x64,3840x2160x16,1280x720x64").split(",")]


def bpp(S):
    t = 0
    for x in types:
        c = film.STAT_TYPES[x]
        planes = c["max_moment"] + (2 if c["transform"] else 0)
        t += 4 * c["channels"] * S + 2 * (4 + 4 * c["channels"] * planes)
    return t


for W, H, S in shapes:
    a = {}
    for t in types:
        x = api.empty_placed((S, H, W, synthetic.CHANNELS[t]), torch.float32, dev, api.MEM_STREAM)
        for s0 in range(0, S, 16):
            x[s0:s0 + 16].uniform_()
        a[t] = x
    ref = None
    same = {}
    def select(d):
        api.check(lib.statmc_debug_accumulate_dma(1 if d >= 100 else d))
        api.check(lib.statmc_debug_accumulate_occupancy(3 if d >= 100 else 2))

    for d in depths:
        select(d)
        fs = film.FilmStats(W, H, dev, types=types, placed=True)
        fs.accumulate(a)
        fs.accumulate(a)
        torch.cuda.synchronize()
        bits = [v.view(torch.int32).clone() for t in types for v in fs.state[t].values() if v is not None]
        if ref is None:
            ref = bits
        same[d] = all(torch.equal(x, y) for x, y in zip(ref, bits))
        del fs
    fs = film.FilmStats(W, H, dev, types=types, placed=True)
    best = {d: 1e9 for d in depths}
    reps = max(3, min(20, int(30 / (bpp(S) * W * H / 6e9))))
    for rnd in range(4):
        for d in depths:
            select(d)
            fs.accumulate(a)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fs.accumulate(a)
            e1.record()
            torch.cuda.synchronize()
            best[d] = min(best[d], e0.elapsed_time(e1) / reps)
    select(1)
    print("%dx%d %3d spp: " % (W, H, S) + "  ".join("D=%d %.3f ms %.2f TB/s%s" % (d, best[d], bpp(S) * W * H / best[d] / 1e9, "" if same[d] else " BITS DIFFER")
                                                      for d in depths), flush=True)
    del fs, a, x
    torch.cuda.empty_cache()
