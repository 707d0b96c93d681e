"""Scratch: accumulate (HBM-bound) beside the window filter (VALU-bound) on two streams, with a
variant library.  usage: overlap_variant.py path/to/variant.so [spp] [grid_override]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from statmc_amd import build
os.environ.setdefault("STATMC_ALLOW_DIAGNOSTIC_BUILD", "1"); build.SO = os.path.abspath(sys.argv[1])
import torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
W, H = 1920, 1080
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
grid = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sc = synthetic.Scene(W, H, seed=1, device=dev)
chunks = [sc.samples(32, seed=10 + i, features=synthetic.FEATURES) for i in range(S // 32)]
smp = {t: torch.cat([c[t] for c in chunks]) for t in synthetic.FEATURES}
del chunks
fs = film.FilmStats(W, H, dev, types=synthetic.FEATURES)
fs.accumulate(smp); fs.prepass()
snap = dict(colour=fs.state["radiance"]["film_mean"].clone(), normal=fs.g_buffer("normal").clone(),
            albedo=fs.g_buffer("albedo").clone(), mc=fs.mean_corr.clone(), dc=fs.disc.clone())
out = torch.zeros_like(snap["colour"])
def run_filter():
    a, keep = api.make_filter_args([], [], [], [], [snap["colour"]], [snap["mc"]], [snap["dc"]], [out],
                                   [snap["normal"], snap["albedo"]], g_sds=[0.1, 0.02])
    api.window_filter(a, 3)
def wall(fn, n=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
api.accumulate_resident_blocks(grid)
acc = lambda: fs.accumulate(smp)
t_a = min(wall(acc) for _ in range(2)); t_f = min(wall(run_filter) for _ in range(2))
bpp = 44 * S + 224
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both(first):
    def f():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        order = [(s2, run_filter), (s1, acc)] if first == "filter" else [(s1, acc), (s2, run_filter)]
        for st, fn in order:
            with torch.cuda.stream(st):
                fn()
        cur.wait_stream(s1); cur.wait_stream(s2)
    return f
t_bf = min(wall(both("filter")) for _ in range(2))
t_ba = min(wall(both("acc")) for _ in range(2))
print("%s grid=%d: acc %.3f ms (%.0f GB/s) | filter %.3f | sum %.3f | concurrent filter-first %.3f, acc-first %.3f"
      % (os.path.basename(sys.argv[1]), grid, t_a, bpp * W * H / t_a / 1e6, t_f, t_a + t_f, t_bf, t_ba))
