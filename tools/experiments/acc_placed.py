"""Round 5: the film-major accumulation with its buffers from statmc_malloc_placed (moments in one HBM rank, sample arenas in
the other two) against the same launch on torch's allocations; same samples, same bits.
python tools/experiments/acc_placed.py [shapes]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

dev = torch.device("cuda:0")
api.setup(0)
types = list(synthetic.FEATURES)
shapes = [tuple(int(x) for x in sh.split("x")) for sh in (sys.argv[1] if len(sys.argv) > 1 else "1920x1080x256,1920x1080x64,1920x1080x16,3840x2160x64,3840x2160x16,1280x720x64").split(",")]


def bpp(S):
    t = 0
    for x in types:
        c = film.STAT_TYPES[x]
        planes = c["max_moment"] + (2 if c["transform"] else 0)
        t += 4 * c["channels"] * S + 2 * (4 + 4 * c["channels"] * planes)
    return t


def timed(fs, a, reps):
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


ONLY = os.environ.get("ONLY")      # "torch" | "placed": one allocator only (shapes whose arenas do not fit twice: 4K / 256 spp)
for W, H, S in shapes:
    if ONLY:
        if ONLY == "placed":
            a = {t: api.empty_placed((S, H, W, synthetic.CHANNELS[t]), torch.float32, dev, api.MEM_STREAM) for t in types}
        else:
            a = {t: torch.empty((S, H, W, synthetic.CHANNELS[t]), device=dev) for t in types}
        for t in types:
            for s0 in range(0, S, 16):
                a[t][s0:s0 + 16].uniform_()
        fs = film.FilmStats(W, H, dev, types=types, placed=ONLY == "placed")
        ms = timed(fs, a, 3)
        print("%dx%d %3d spp: %s %.3f ms %.3f" % (W, H, S, ONLY, ms, bpp(S) * W * H / ms / 8e9), flush=True)
        del a, fs
        torch.cuda.empty_cache()
        continue
    plain = {t: torch.empty((S, H, W, synthetic.CHANNELS[t]), device=dev) for t in types}
    if os.environ.get("SCENE"):       # bench.py's stream (log-normal radiance, 20 % zero paths, fireflies) instead of uniform numbers
        scene = synthetic.Scene(W, H, n_regions=12, seed=1, device=dev)
        for s0 in range(0, S, 32):
            part = scene.samples(min(32, S - s0), seed=1000 + s0, features=types)
            for t in types:
                plain[t][s0:s0 + part[t].shape[0]] = part[t]
            del part
    else:
        for t in types:
            for s0 in range(0, S, 16):
                plain[t][s0:s0 + 16].uniform_()
    t0 = time.perf_counter()
    if os.environ.get("STATE_FIRST"):
        fs_p = film.FilmStats(W, H, dev, types=types, placed=True)
    placed = {t: api.empty_placed((S, H, W, synthetic.CHANNELS[t]), torch.float32, dev, api.MEM_STREAM) for t in types}
    if not os.environ.get("STATE_FIRST"):
        fs_p = film.FilmStats(W, H, dev, types=types, placed=True)
    torch.cuda.synchronize()
    t_alloc = time.perf_counter() - t0
    for t in types:
        placed[t].copy_(plain[t])
    if os.environ.get("NO_PLAIN"):      # the torch arenas go away before anything is timed (is it their presence?)
        plain = {t: placed[t] for t in types}
        torch.cuda.empty_cache()
    fs_t = film.FilmStats(W, H, dev, types=types)
    reps = max(3, min(20, int(30 / (bpp(S) * W * H / 6e9))))
    row = []
    for rnd in range(2):
        row.append("torch %.3f ms %.3f" % ((lambda ms: (ms, bpp(S) * W * H / ms / 8e9))(timed(fs_t, plain, reps))))
        row.append("placed %.3f ms %.3f" % ((lambda ms: (ms, bpp(S) * W * H / ms / 8e9))(timed(fs_p, placed, reps))))
    if os.environ.get("GRIDS"):      # the placed launch under both launch shapes (statmc_debug_accumulate_launch), same buffers
        lib = api.load()
        for g in (0, 1, 0, 1):
            api.check(lib.statmc_debug_accumulate_launch(g, 0))
            ms = timed(fs_p, placed, reps)
            row.append("placed grid %d %.3f ms %.3f" % (g, ms, bpp(S) * W * H / ms / 8e9))
        api.check(lib.statmc_debug_accumulate_launch(-1, 0))
    # the same launch inside the step of bench.py: accumulate, pre-pass + window filter (VALU-bound, 1.7 ms at 1080p), repeat;
    # only the accumulations are timed
    if W * H <= 1920 * 1080:
        for name, fs, a in (("torch", fs_t, plain), ("placed", fs_p, placed)):
            fs.accumulate(a); fs.denoise(); torch.cuda.synchronize()
            evs = []
            for _ in range(8):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fs.accumulate(a); e1.record()
                fs.denoise()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ms = sorted(x.elapsed_time(y) for x, y in evs)[len(evs) // 2]
            row.append("%s between filters %.3f ms %.3f" % (name, ms, bpp(S) * W * H / ms / 8e9))
    # same bits: fresh state, two launches each
    fs_t.reset(); fs_p.reset()
    for _ in range(2):
        fs_t.accumulate(plain); fs_p.accumulate(placed)
    torch.cuda.synchronize()
    same = all(torch.equal(fs_t.state[t][k].view(torch.int32), fs_p.state[t][k].view(torch.int32)) for t in types for k in fs_t.state[t] if fs_t.state[t][k] is not None)
    print("%dx%d %3d spp: %s | same bits %s | placed allocation %.2f s" % (W, H, S, "  ".join(row), same, t_alloc), flush=True)
    del plain, placed, fs_p, fs_t
    torch.cuda.empty_cache()
print(api.placement_info(), flush=True)
