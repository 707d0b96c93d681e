"""Round 5: why does the film-major accumulation stream the same bytes slower at 4K / 64 spp than at 1080p / 256 spp?
Three candidate causes, one case each (all stat types, back-to-back launches, HIP events):
  * what the caches keep between back-to-back launches: ONE 1080p film + arena repeated against K films + arenas in rotation
    (K x 464 MB of state, K x 5.8 GB of samples: nothing comes back within the Infinity Cache's 256 MB)
  * the stride between a pixel's consecutive samples (a whole film plane: 25 MB at 1080p, 100 MB at 4K): the 4K film as
    B bands of rows, each band with its own arena [S][rows][W][C] -- the row-band layout -- B launches back to back
  * the state's share of the bytes: the same film at 16 / 64 / 256 spp
STATMC_VARIANT=path.so times a variant library (e.g. -DSTATMC_ACC_SKIP_STORES=1: what the state stores cost per shape).
python tools/experiments/acc_gap.py [cases]   cases: comma list out of rot,band,spp (default: all)"""
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
types = list(synthetic.FEATURES)
cases = (sys.argv[1] if len(sys.argv) > 1 else "rot,band,spp").split(",")


def bpp(S):
    t = 0
    for x in types:
        c = film.STAT_TYPES[x]
        planes = c["max_moment"] + (2 if c["transform"] else 0)
        t += 4 * c["channels"] * S + 2 * (4 + 4 * c["channels"] * planes)
    return t


def arena(S, H, W):
    """uniform (0, 1) samples: the kernel's time does not depend on the values (tools/microbench/acc_model.hip)"""
    out = {}
    for t in types:
        a = torch.empty((S, H, W, synthetic.CHANNELS[t]), device=dev)
        for s0 in range(0, S, 16):
            a[s0:s0 + 16].uniform_()
        out[t] = a
    return out


def timed(run, reps=6, rounds=3):
    run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def report(tag, W, H, S, ms):
    print("%-46s %8.3f ms  %5.2f TB/s  %.3f of 8 TB/s" % (tag, ms, bpp(S) * W * H / ms / 1e9, bpp(S) * W * H / ms / 1e9 / 8), flush=True)


if "rot" in cases:
    W, H = 1920, 1080
    for S in (16, 64):
        K = 4
        sets = [(film.FilmStats(W, H, dev, types=types), arena(S, H, W)) for _ in range(K)]
        fs0, a0 = sets[0]
        report("1080p %3d spp, one film + arena repeated" % S, W, H, S, timed(lambda: fs0.accumulate(a0)))
        def rot():
            for fs, a in sets:
                fs.accumulate(a)
        report("1080p %3d spp, %d films + arenas in rotation" % (S, K), W, H, S, timed(rot, reps=3) / K)
        del sets, fs0, a0
        torch.cuda.empty_cache()

if "band" in cases:
    W, H = 3840, 2160
    for S in (16, 64):
        fs = film.FilmStats(W, H, dev, types=types)
        a = arena(S, H, W)
        report("4K %3d spp, film-major planes, one launch" % S, W, H, S, timed(lambda: fs.accumulate(a)))
        del fs, a
        torch.cuda.empty_cache()
        for B in (4, 16, 60):
            rows = H // B
            sets = [(film.FilmStats(W, rows, dev, types=types), arena(S, rows, W)) for _ in range(B)]
            def bands():
                for f, x in sets:
                    f.accumulate(x)
            report("4K %3d spp, %2d row bands of %d rows, own arenas" % (S, B, rows), W, H, S, timed(bands, reps=3))
            del sets
            torch.cuda.empty_cache()

if "spp" in cases:
    for W, H, name in ((1920, 1080, "1080p"), (3840, 2160, "4K")):
        fs = film.FilmStats(W, H, dev, types=types)
        a = arena(256, H, W)
        for S in (16, 64, 128, 256):
            part = {t: v[:S] for t, v in a.items()}
            report("%s %3d spp, one film + arena" % (name, S), W, H, S, timed(lambda: fs.accumulate(part), reps=4))
        del fs, a, part
        torch.cuda.empty_cache()
