"""Round 5: what separates acc_placed.py (capped-grid launch at 0.842 of the peak) from every other harness (0.81)?  V = letters:
  a  nothing but the placed buffers;  p  five written torch arenas of the same sizes allocated first, the placed arenas filled by copy_;
  t  + a torch FilmStats alive;  x  + the torch launch timed first.   Result: p alone does it (0.809 -> 0.842); cause unknown.
  second pass:  f  (with p) the torch arenas freed before the timing;  b  an UNWRITTEN torch ballast of the arenas' total size first, placed
  arenas filled in place;  w  the same ballast written;  c  placed arenas filled by copy_ from a 16-sample torch buffer (no big torch arenas).
V=p python tools/experiments/acc_bisect.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
ballast = None
if "b" in V or "w" in V:
    ballast = torch.empty(sum(S * H * W * synthetic.CHANNELS[t] for t in types), device=dev)
    if "w" in V:
        for i in range(0, ballast.numel(), 1 << 28): ballast[i:i + (1 << 28)].uniform_()
if "p" in V:      # torch arenas first (written)
    plain = {t: torch.empty((S, H, W, synthetic.CHANNELS[t]), device=dev) for t in types}
    for t in types:
        for s0 in range(0, S, 16): plain[t][s0:s0 + 16].uniform_()
placed = {t: api.empty_placed((S, H, W, synthetic.CHANNELS[t]), torch.float32, dev, api.MEM_STREAM) for t in types}
fs_p = film.FilmStats(W, H, dev, types=types, placed=True)
for t in types:
    if plain is not None: placed[t].copy_(plain[t])
    elif "c" in V:
        small = torch.empty((16, H, W, synthetic.CHANNELS[t]), device=dev)
        for s0 in range(0, S, 16):
            small.uniform_(); placed[t][s0:s0 + 16].copy_(small)
        del small
    else:
        for s0 in range(0, S, 16): placed[t][s0:s0 + 16].uniform_()
if "f" in V and plain is not None:
    plain = None; torch.cuda.empty_cache(); torch.cuda.synchronize()
fs_t = film.FilmStats(W, H, dev, types=types) if "t" in V else None
if "x" in V and plain is not None:      # time the torch launch first
    print("torch", timed(fs_t, plain, 7))
row = []
for g in (0, 1, 0, 1):
    api.check(lib.statmc_debug_accumulate_launch(g, 0))
    ms = timed(fs_p, placed, 7)
    row.append("grid %d %.3f ms %.3f" % (g, ms, bpp(S) * W * H / ms / 8e9))
# the allocator's probe on the buffers themselves: the first GiB of every arena streamed beside read-modify-writes of a state plane
import ctypes as C
lib.statmc_debug_interference_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_float)]
tgt = fs_p.state["radiance"]["m2"]
pr = []
for t in types:
    ms = C.c_float()
    api.check(lib.statmc_debug_interference_probe(C.c_void_p(placed[t].data_ptr()), 1 << 30, C.c_void_p(tgt.data_ptr()), (tgt.numel() * 4) & ~15, C.byref(ms)))
    pr.append("%.4f" % ms.value)
ms = C.c_float()
other = fs_p.state["normal"]["mean"]
api.check(lib.statmc_debug_interference_probe(C.c_void_p(other.data_ptr()), (other.numel() * 4) & ~15, C.c_void_p(tgt.data_ptr()), (tgt.numel() * 4) & ~15, C.byref(ms)))
print(V, "  ".join(row), "probe arena GiB vs state plane:", " ".join(pr), "(state vs state %.4f)" % ms.value, api.placement_info()["map"])
