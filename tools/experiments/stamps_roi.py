"""When do the workgroups of ONE pair-symmetric launch start, and how long does each take?  (diagnostic build with
-DSTATMC_SYM_STAMPS=1: tools/experiments/build_variant.sh stamps -DSTATMC_SYM_STAMPS=1)  An ROI of c x k tiles of a 1920 x 360
film at parts = 1: tools/experiments/strip_scan3.py measured one tile's time up to "200 tiles" and two from "208" -- the launch holds
c x (k + 3) items (the tile rows above the ROI whose windows reach into it), so that is 230 against 247 .. 285 items on 256 CUs.
python tools/experiments/stamps_roi.py C K"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from statmc_amd import build
os.environ.setdefault("STATMC_ALLOW_DIAGNOSTIC_BUILD", "1"); build.SO = os.path.join(ROOT, "tools", "experiments", "variants", "stamps.so")
import ctypes as C
import numpy as np
import torch
from statmc_amd import api, film, synthetic
c, k = int(sys.argv[1]), int(sys.argv[2])
W, H = 1920, 360
dev = torch.device("cuda:0")
api.setup(0)
scene = synthetic.Scene(W, H, seed=1, device=dev)
fs = film.FilmStats(W, H, dev)
fs.accumulate(scene.samples(8, seed=2)); fs.prepass()
api.force_filter_parts(1)
roi = (0, 24, 128 * c, 24 + 8 * k)
for _ in range(3):
    fs.window_filter(roi=roi)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fs.window_filter(roi=roi); e1.record(); torch.cuda.synchronize()
ptr, nbytes = C.c_void_p(), C.c_size_t()
api.load().statmc_debug_last_workspace(C.byref(ptr), C.byref(nbytes))
# items of the launch: the ROI's tiles AND the tiles above it whose window rows reach into it (sym_geometry: rows from ry0 - r)
tiles, stride = min(15, c + 1) * (k + 3), 8 * 128 + (21 + 8 - 1) * 168   # ry0 = 24, r = 20: tile rows 0 .. 2 lie above the ROI's first; + the tile column right of it
ws = torch.empty(tiles * stride, 4, device=dev)
api.check(api.load().statmc_download(C.c_void_p(ws.data_ptr()), ptr, tiles * stride * 16, api.current_stream_handle()))
torch.cuda.synchronize()
edge = ws.view(tiles, stride, 4)[:, stride - 16:stride - 8, :].cpu().numpy()      # [item][wave][prologue, after, whole (shader clocks), start (10 ns)]
start = edge[:, 0, 3]
start = (start - start.min()) % (1 << 24)
whole = edge[:, 0, 2]
print("ROI of %d x %d tiles = %d items: launch %.3f ms; item start (us after the first): median %.1f, 90 %% %.1f, max %.1f; items starting later than 50 us: %d"
      % (c, k, tiles, e0.elapsed_time(e1), np.median(start) / 100, np.percentile(start, 90) / 100, start.max() / 100, int((start > 5000).sum())))
print("   whole item (shader clocks, wave 0): min %.0f median %.0f max %.0f" % (whole.min(), np.median(whole), whole.max()))
late = np.nonzero(start > 5000)[0]
print("   late items:", late[:40].tolist())
loc = ws.view(tiles, stride, 4)[:, stride - 24:stride - 16, :].cpu().numpy().view(np.uint32)      # [item][wave][xcc, hw_id, block, -]
# HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (se 3 bits on gfx94x/95x)
cu_of = lambda i: (int(loc[i, 0, 0] & 15), int((loc[i, 0, 1] >> 13) & 7), int((loc[i, 0, 1] >> 12) & 1), int((loc[i, 0, 1] >> 8) & 15))
where = {}
for i in range(tiles):
    where.setdefault(cu_of(i), []).append(i)
print("   distinct (xcc, se, sh, cu): %d for %d items; CUs that ran two items: %d" % (len(where), tiles, sum(1 for v in where.values() if len(v) > 1)))
for i in late[:12]:
    print("   late item %d (block %d) ran on %s together with items %s" % (i, loc[i, 0, 2], cu_of(i), [j for j in where[cu_of(i)] if j != i]))
per_xcc = {}
for key in where:
    per_xcc.setdefault(key[0], set()).add(key[1:])
print("   CUs used per XCC:", {x: len(v) for x, v in sorted(per_xcc.items())})
if "--map" in sys.argv:
    for x in sorted(per_xcc):
        print("   xcc %d:" % x, sorted(per_xcc[x]))
