"""Per-wave cycle budget of the pair-symmetric filter (diagnostic build with -DSTATMC_SYM_STAMPS=1): shader clocks per
step spent in housekeeping (flush + staging), in the sweep, and waiting at the barrier, averaged over all workgroups.
  tools/experiments/build_variant.sh stamps -DSTATMC_SYM_STAMPS=1   (container)
  python tools/experiments/stamps_sym.py                             (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from statmc_amd import build
os.environ.setdefault("STATMC_ALLOW_DIAGNOSTIC_BUILD", "1"); build.SO = os.path.join(ROOT, "tools", "experiments", "variants", (sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "stamps") + ".so")
import ctypes as C
import torch
from statmc_amd import api, film, synthetic
W, H = 1920, 1080
dev = torch.device("cuda:0")
api.setup(0)
scene = synthetic.Scene(W, H, seed=1, device=dev)
fs = film.FilmStats(W, H, dev)
fs.accumulate(scene.samples(32, seed=2, features=("radiance", "normal", "albedo")))
fs.prepass()
api.force_filter_parts(1)
a, keep = fs.filter_args()
for _ in range(3):
    api.window_filter(a, 3)
torch.cuda.synchronize()
# the patch workspace of the launch: find it through a second, identical launch on a known-size buffer is not possible
# from here, so the library exposes nothing: read it back through the debug hook instead
ptr, nbytes = C.c_void_p(), C.c_size_t()
api.load().statmc_debug_last_workspace(C.byref(ptr), C.byref(nbytes))
n4 = nbytes.value // 16
ws = torch.empty(n4, 4, device=dev)
api.check(api.load().statmc_download(C.c_void_p(ws.data_ptr()), ptr, n4 * 16, api.current_stream_handle()))
torch.cuda.synchronize()
tiles = (W // 128) * (H // 8)
stride = n4 // tiles
full = ws[:tiles * stride].view(tiles, stride, 4)
edge = full[:, stride - 16:stride - 8, :].cpu()                                # [tile][wave][before first step, after last, whole item]
ws = full[:, stride - 8:, :].cpu()                                            # [tile][wave][hk, sweep, barrier, steps]
e = edge.mean((0, 1))
print("per item (clocks, mean over waves and workgroups): before the first step %.0f, after the last step %.0f, whole item %.0f"
      % (e[0], e[1], e[2]))
steps = ws[..., 3].clamp(min=1)
per = ws[..., :3] / steps[..., None]
print("tiles %d, stride %d float4" % (tiles, stride))
print("wave   housekeeping   sweep   barrier   (shader clocks per step, mean over workgroups)")
for w in range(8):
    m = per[:, w].mean(0)
    print("  %d   %10.0f %10.0f %9.0f" % (w, m[0], m[1], m[2]))
m = per.mean((0, 1))
print("all   %10.0f %10.0f %9.0f   total %.0f clocks/step" % (m[0], m[1], m[2], m.sum()))
if "--raw" in sys.argv:
    for t in (0, 700, 2024):
        print("tile", t, "step sums [hk, sweep, barrier, steps] wave 0:", ws[t, 0].tolist(), "wave 4:", ws[t, 4].tolist())
        print("        edge [before, after, whole] wave 0:", edge[t, 0].tolist(), "wave 4:", edge[t, 4].tolist())
