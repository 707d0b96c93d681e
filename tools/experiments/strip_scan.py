"""Window-filter time of 1920-wide strips (what one of N GPUs filters under strong scaling: 1080 / N rows + the halo) at parts = 1, 2, 3
and the host's choice, to split the strip's time into a fixed part and a part per round of 256 workgroups.
python tools/experiments/strip_scan.py      (STATMC_VARIANT=notrim: the build that sweeps every window row of the halo tiles)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get("STATMC_VARIANT"):          # tools/experiments/build_variant.sh NAME ... -> variants/NAME.so
    from statmc_amd import build
    build.SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", os.environ["STATMC_VARIANT"] + ".so")
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


for rows in (135, 270, 540, 1080):
    halo = 0 if rows == 1080 else 20
    H = rows + 2 * halo
    sc = synthetic.Scene(W, H, seed=1, device=dev)
    fs = film.FilmStats(W, H, dev)
    fs.accumulate(sc.samples(8, seed=2)); fs.prepass()
    roi = (0, halo, W, halo + rows)
    tiles = 15 * ((rows + 7) // 8)
    line = "%4d rows (+ halo): %4d tiles = %.2f rounds:" % (rows, tiles, tiles / 256.0)
    for k in (1, 2, 3, 0):
        api.force_filter_parts(k)
        ms = t(fs, roi)
        line += "  parts %s %.3f ms (%s)" % (k or "auto", ms, api.last_filter_variant())
    api.force_filter_parts(0)
    print(line + "   per 135 rows: %.3f ms" % (ms * 135 / rows), flush=True)
    del fs, sc
