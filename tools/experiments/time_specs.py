"""Window-filter time at 1080p, r = 20 under the filter-spec options: which kernel serves each and what it costs.
python tools/experiments/time_specs.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

W, H = 1920, 1080
dev = torch.device("cuda:0")
api.setup(0)
scene = synthetic.Scene(W, H, seed=1, device=dev)
fs = film.FilmStats(W, H, dev)
fs.accumulate(scene.samples(32, seed=2, features=("radiance", "normal", "albedo")))
torch.cuda.synchronize()
for kw in (dict(), dict(border=1), dict(gate=1), dict(channel_rule=1), dict(gate=1, channel_rule=1, border=1), dict(sides=1, small_n=1), dict(gate=2), dict(gate=2, channel_rule=1), dict(dof=1)):
    api.set_filter_spec(**kw)
    fs.prepass()
    a, keep = fs.filter_args()
    reps = 3 if kw.get("dof") else 20
    for _ in range(2):
        api.window_filter(a, 3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        api.window_filter(a, 3)
    e1.record()
    torch.cuda.synchronize()
    print("%-45s %-18s %.3f ms" % (kw or "default", api.last_filter_variant(), e0.elapsed_time(e1) / reps), flush=True)
api.set_filter_spec()
