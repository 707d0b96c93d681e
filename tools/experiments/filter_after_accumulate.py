"""Scratch: the window filter right after the 256-spp accumulation is ~15 % slower than back to back.
Cold inputs or clocks?  Insert (a) nothing, (b) a pass that re-reads the five filter inputs (warms
L2 / Infinity Cache, ~0.1 ms), (c) ~1 ms of VALU-heavy work on a small tensor (no cache effect),
between the accumulation and the filter, and time the filter alone (events around it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
W, H, S = 1920, 1080, 256
sc = synthetic.Scene(W, H, seed=1, device=dev)
chunks = [sc.samples(32, seed=10 + i, features=synthetic.FEATURES) for i in range(S // 32)]
smp = {t: torch.cat([c[t] for c in chunks]) for t in synthetic.FEATURES}
del chunks
fs = film.FilmStats(W, H, dev, types=synthetic.FEATURES)
fs.accumulate(smp); fs.prepass()
inputs = [fs.mean_corr, fs.disc, fs.state["radiance"]["film_mean"], fs.g_buffer("normal"), fs.g_buffer("albedo")]
small = torch.rand(1 << 22, device=dev) + 1.0
def touch():
    for t in inputs: t.sum()
def valu(n=12):
    x = small
    for _ in range(n): x = torch.lgamma(x) + 2.0
def timed(pre, n=10):
    ts = []
    for _ in range(n):
        fs.accumulate(smp); fs.prepass(); pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fs.window_filter(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = ts[2:]
    return sum(ts) / len(ts), min(ts)
def back_to_back(n=10):
    fs.window_filter(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fs.window_filter()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("filter back to back            %.3f ms" % back_to_back())
for name, pre in (("after accumulate", lambda: None), ("after accumulate + input touch", touch),
                  ("after accumulate + VALU work", valu), ("after accumulate + both", lambda: (valu(), touch())),
                  ("after accumulate + 1 ms VALU", lambda: valu(70)), ("after accumulate + 3 ms VALU", lambda: valu(210)),
                  ("after accumulate + 3 ms filter x1", lambda: fs.window_filter())):
    print("%-32s avg %.3f  min %.3f ms" % ((name,) + timed(pre)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); valu(); e1.record(); torch.cuda.synchronize(); print("VALU filler takes %.3f ms" % e0.elapsed_time(e1))
e0.record(); touch(); e1.record(); torch.cuda.synchronize(); print("input touch takes %.3f ms" % e0.elapsed_time(e1))
