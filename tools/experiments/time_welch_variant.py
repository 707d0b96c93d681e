"""Welch build of the pair-symmetric kernel, timed with a variant library (timing-only ablations: STATMC_SYM_WELCH_ABLATE).
usage: time_welch_variant.py path/to/variant.so"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from statmc_amd import build
if len(sys.argv) > 1:
    os.environ.setdefault("STATMC_ALLOW_DIAGNOSTIC_BUILD", "1"); build.SO = os.path.abspath(sys.argv[1])
import torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
W, H = 1920, 1080
scene = synthetic.Scene(W, H, seed=1, device=dev)
fs = film.FilmStats(W, H, dev)
fs.accumulate(scene.samples(32, seed=2, features=("radiance", "normal", "albedo")))
out = []
for kw in (dict(), dict(dof=1)):
    api.set_filter_spec(**kw)
    fs.prepass()
    a, keep = fs.filter_args()
    for _ in range(2):
        api.window_filter(a, 3)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            api.window_filter(a, 3)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 8)
    out.append("%s %.3f ms" % (api.last_filter_variant(), best))
api.set_filter_spec()
print(os.path.basename(build.SO), " | ".join(out), flush=True)
