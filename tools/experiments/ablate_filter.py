"""Where does the window filter's time go?  Timing-only ablation builds of lds_r20 (outputs wrong):
no LDS reads in the sweep (pure VALU), LDS reads with token VALU work (operand feed)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
W, H = 1920, 1080
sc = synthetic.Scene(W, H, seed=1, device=dev)
fs = film.FilmStats(W, H, dev)
fs.accumulate(sc.samples(16, seed=2, features=("radiance", "normal", "albedo"))); fs.prepass()
def wall(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
lib = api.load()
for parts in (1, 3):
    api.force_filter_parts(parts)
    for abl, name in ((0, "real"), (0, "real")):   # the ablation builds belonged to the previous kernel generation (git history)
        lib.statmc_debug_filter_ablation(abl)
        t = wall(fs.window_filter)
        print("parts=%d %-28s %.3f ms  (%s)" % (parts, name, t, api.last_filter_variant()))
lib.statmc_debug_filter_ablation(0); api.force_filter_parts(0)

