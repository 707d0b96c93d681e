"""Scratch: why is the window filter slower inside bench.py (after the accumulation) than back to back?
Times the filter (events around the filter only) after: nothing, a 4 GiB streaming torch kernel
(flushes L2 / Infinity Cache, HBM-heavy), a short cache flush only (1 GiB), and the real accumulate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
W, H, S = 1920, 1080, 128
sc = synthetic.Scene(W, H, seed=1, device=dev)
chunks = [sc.samples(32, seed=10 + i, features=synthetic.FEATURES) for i in range(S // 32)]
smp = {t: torch.cat([c[t] for c in chunks]) for t in synthetic.FEATURES}
del chunks
fs = film.FilmStats(W, H, dev, types=synthetic.FEATURES)
fs.accumulate(smp); fs.prepass()
big = torch.ones(1 << 30, dtype=torch.float32, device=dev)
small = torch.ones(1 << 28, dtype=torch.float32, device=dev)
def timed_filter(pre, n=8):
    ts = []
    for _ in range(n):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fs.window_filter(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sum(ts[2:]) / len(ts[2:]), min(ts)
for name, pre in (("back to back", lambda: None),
                  ("after 8 GiB stream (torch mul_)", lambda: big.mul_(1.0001)),
                  ("after 2 GiB stream", lambda: small.mul_(1.0001)),
                  ("after accumulate 128 spp", lambda: fs.accumulate(smp)),
                  ("after accumulate + 200 us idle", lambda: (fs.accumulate(smp), torch.cuda.synchronize(), torch.cuda._sleep(400000))),
                  ("after sync + 2 ms host sleep", lambda: (fs.accumulate(smp), torch.cuda.synchronize(), __import__("time").sleep(0.002)))):
    avg, mn = timed_filter(pre)
    print("%-36s filter avg %.3f ms  min %.3f ms" % (name, avg, mn))
