"""Scratch: A/B the 1080p r=20 window filter of several variant libraries on one box (each in its
own process, interleaved, best of N).  usage: ab_filter.py a.so b.so [...]"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(sys.argv[2])))
from statmc_amd import build
os.environ.setdefault("STATMC_ALLOW_DIAGNOSTIC_BUILD", "1"); build.SO = os.path.abspath(sys.argv[1])
import torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
W, H = 1920, 1080
sc = synthetic.Scene(W, H, seed=1, device=dev)
fs = film.FilmStats(W, H, dev)
fs.accumulate(sc.samples(32, seed=2)); fs.prepass()
def t(n=10):
    fs.window_filter(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fs.window_filter()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("%.4f" % min(t() for _ in range(4)))
'''
best = {}
for rep in range(3):
    for so in sys.argv[1:]:
        out = subprocess.run([sys.executable, "-c", CHILD, so, HERE], capture_output=True, text=True)
        try:
            v = float(out.stdout.strip().splitlines()[-1])
        except Exception:
            print(so, "failed:", out.stderr[-400:]); continue
        best.setdefault(so, []).append(v)
for so, v in best.items():
    print("%-40s filter ms: %s  (min %.4f)" % (os.path.basename(so), " ".join("%.4f" % x for x in v), min(v)))
