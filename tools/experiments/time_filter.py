"""Back-to-back timing of the window-filter kernels on one film: pair-symmetric (auto / forced parts) against the
one-sided r = 20 kernel.  python tools/experiments/time_filter.py [W H [spp]]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from statmc_amd import build
CASES = ((3, 0), (0, 0), (0, 1), (0, 2), (0, 3), (0, 4), (3, 0), (0, 0))
if "--lib" in sys.argv:      # a variant library (tools/experiments/build_variant.sh): sym kernel only, two part counts
    k = sys.argv.index("--lib")
    os.environ.setdefault("STATMC_ALLOW_DIAGNOSTIC_BUILD", "1"); build.SO = os.path.abspath(sys.argv[k + 1])
    del sys.argv[k:k + 2]
    CASES = ((0, 1), (0, 1), (0, 1), (0, 2))
import torch
from statmc_amd import api, film, synthetic

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda:0")
api.setup(0)
scene = synthetic.Scene(W, H, seed=1, device=dev)
fs = film.FilmStats(W, H, dev)
fs.accumulate(scene.samples(spp, seed=2, features=("radiance", "normal", "albedo")))
fs.prepass()
torch.cuda.synchronize()


def run(force, parts, reps=20):
    api.force_filter_variant(force)
    api.force_filter_parts(parts)
    a, keep = fs.filter_args()
    for _ in range(3):
        api.window_filter(a, 3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        api.window_filter(a, 3)
    e1.record()
    torch.cuda.synchronize()
    v, p = api.last_filter_variant(), api.load().statmc_debug_last_filter_parts()
    api.force_filter_variant(0)
    api.force_filter_parts(0)
    return e0.elapsed_time(e1) / reps, v, p, fs.film_f.clone()


base = None
for force, parts in CASES:
    ms, v, p, out = run(force, parts)
    if base is None:
        base = out
    err = float(((out - base).double().pow(2).sum() / base.double().pow(2).sum()).sqrt())
    print(os.path.basename(build.SO), "%dx%d  %-8s parts %d : %.3f ms  (%.0f Mpx/s)  rel L2 vs first %.2e" % (W, H, v, p, ms, W * H / ms / 1e3, err), flush=True)
