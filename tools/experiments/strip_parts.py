import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
W = 1920
def t(fs, roi, n=20):
    fs.window_filter(roi=roi); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fs.window_filter(roi=roi)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
for rows in (135, 270):
    H = rows + 40
    sc = synthetic.Scene(W, H, seed=1, device=dev)
    fs = film.FilmStats(W, H, dev)
    fs.accumulate(sc.samples(8, seed=2)); fs.prepass()
    roi = (0, 20, W, 20 + rows)
    line = "%d rows:" % rows
    for k in (1, 2, 3, 4, 5, 6, 7, 8, 0):
        api.force_filter_parts(k)
        line += "  p%s %.3f" % (k or "auto", t(fs, roi))
    api.force_filter_parts(0)
    import ctypes as C
    ph, tr = C.c_int(), C.c_int()
    api.load().statmc_debug_last_filter_tail(C.byref(ph), C.byref(tr))
    print(line, " auto = %d parts, tail %d x %d rows" % (api.load().statmc_debug_last_filter_parts(), ph.value, tr.value), flush=True)
