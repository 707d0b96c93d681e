"""Scratch: time the all-types accumulate (1080p, S spp) and the r=20 filter with a variant library.
usage: time_variant.py path/to/variant.so [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from statmc_amd import build
os.environ.setdefault("STATMC_ALLOW_DIAGNOSTIC_BUILD", "1"); build.SO = os.path.abspath(sys.argv[1])
import torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
W, H = 1920, 1080
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
sc = synthetic.Scene(W, H, seed=1, device=dev)
smp = sc.samples(S, seed=2, features=synthetic.FEATURES)
def timeit(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def bpp(t):
    cfg = film.STAT_TYPES[t]; c = cfg["channels"]
    planes = cfg["max_moment"] + (2 if cfg["transform"] else 0)
    return 4 * c * S + 2 * (4 + 4 * c * planes)
res = []
for types in (["radiance"], list(synthetic.FEATURES)):
    fs = film.FilmStats(W, H, dev, types=types)
    sub = {t: smp[t] for t in types}
    t = min(timeit(lambda: fs.accumulate(sub)) for _ in range(3))
    res.append("%s %.3f ms %.0f GB/s" % ("rad" if len(types) == 1 else "all", t, sum(bpp(x) for x in types) * W * H / t / 1e6))
fs.prepass()
t = min(timeit(lambda: fs.window_filter(), 5) for _ in range(2))
res.append("filter %.3f ms" % t)
print(os.path.basename(sys.argv[1]), " | ".join(res))
