"""Window-filter time against what ran before it and what the film holds: sample count, number of stat types,
repetitions, a preceding accumulate.  python tools/experiments/filter_ctx.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

W, H = 1920, 1080
dev = torch.device("cuda:0")
api.setup(0)


def timed(fn, reps, warm=1):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for types, spp in ((("radiance", "normal", "albedo"), 32), (("radiance", "normal", "albedo"), 256), (tuple(synthetic.FEATURES), 32),
                   (tuple(synthetic.FEATURES), 256)):
    scene = synthetic.Scene(W, H, n_regions=12, seed=1, device=dev)
    parts = [scene.samples(32, seed=1000 + s0, features=types) for s0 in range(0, spp, 32)]
    samples = {t: torch.cat([p[t] for p in parts], dim=0) for t in types}
    del parts
    fs = film.FilmStats(W, H, dev, types=types)
    fs.accumulate(samples)
    fs.prepass()
    torch.cuda.synchronize()
    line = "%2d types %3d spp: filter x5 %.3f  x20 %.3f  x100 %.3f ms" % (
        len(types), spp, timed(fs.window_filter, 5), timed(fs.window_filter, 20), timed(fs.window_filter, 100))
    # a filter timed on its own right after an accumulate
    ts = []
    for _ in range(5):
        fs.accumulate(samples)
        fs.prepass()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fs.window_filter()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    # ... and after an idle gap
    torch.cuda.synchronize()
    import time
    time.sleep(0.2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fs.window_filter(); e1.record(); torch.cuda.synchronize()
    print(line, "| after accumulate+prepass:", " ".join("%.3f" % t for t in ts), "| after 0.2 s idle: %.3f" % e0.elapsed_time(e1), flush=True)
    del fs, samples
    torch.cuda.empty_cache()
