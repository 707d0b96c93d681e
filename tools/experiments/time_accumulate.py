"""Scratch: per-type accumulate bandwidth at 1080p."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
W, H = 1920, 1080
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sc = synthetic.Scene(W, H, seed=1, device=dev)
smp = sc.samples(S, seed=2)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def bpp(t):
    cfg = film.STAT_TYPES[t]; c = cfg["channels"]
    planes = cfg["max_moment"] + (2 if cfg["transform"] else 0)
    return 4 * c * S + 2 * (4 + 4 * c * planes)
for types in (["radiance"], ["normal"], ["depth"], ["normal", "albedo"], ["depth", "materialid"],
              ["normal", "albedo", "depth", "materialid"], list(synthetic.FEATURES)):
    fs = film.FilmStats(W, H, dev, types=types)
    sub = {t: smp[t] for t in types}
    t = timeit(lambda: fs.accumulate(sub))
    b = sum(bpp(x) for x in types) * W * H
    print("%-50s %.3f ms  %.0f GB/s" % ("+".join(types), t, b / t / 1e6))
# plain torch copy for reference
x = smp["radiance"]; y = torch.empty_like(x)
t = timeit(lambda: y.copy_(x))
print("torch copy (read+write) %.0f GB/s" % (2 * x.numel() * 4 / t / 1e6))
t = timeit(lambda: x.sum())
print("torch sum (read) %.0f GB/s" % (x.numel() * 4 / t / 1e6))
