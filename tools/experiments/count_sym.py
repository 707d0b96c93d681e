"""How often would a wave-level "all rejected" skip fire?  Diagnostic build -DSTATMC_SYM_COUNT=1: every wave counts its
full read groups (4 taps x 4 pixels x 64 lanes) and those in which no lane has a member pair -- the groups whose
v_exp_f32 and accumulation a ballot in front of them would save -- on the bench's film (1080p, 256 spp, 12 regions).
  tools/experiments/build_variant.sh count -DSTATMC_SYM_COUNT=1      (container)
  python tools/experiments/count_sym.py [spp]                          (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from statmc_amd import build
os.environ.setdefault("STATMC_ALLOW_DIAGNOSTIC_BUILD", "1"); build.SO = os.path.join(ROOT, "tools", "experiments", "variants", "count.so")
import ctypes as C
import torch
from statmc_amd import api, film, synthetic
W, H = 1920, 1080
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
api.setup(0)
types = ("radiance", "normal", "albedo")
scene = synthetic.Scene(W, H, n_regions=12, seed=1, device=dev)
fs = film.FilmStats(W, H, dev, types=types)
for s0 in range(0, spp, 32):
    fs.accumulate(scene.samples(min(32, spp - s0), seed=1000 + s0, features=types))
fs.prepass()
api.force_filter_parts(1)
a, keep = fs.filter_args()
api.window_filter(a, 3)
torch.cuda.synchronize()
ptr, nbytes = C.c_void_p(), C.c_size_t()
api.load().statmc_debug_last_workspace(C.byref(ptr), C.byref(nbytes))
n4 = nbytes.value // 16
ws = torch.empty(n4, 4, device=dev)
api.check(api.load().statmc_download(C.c_void_p(ws.data_ptr()), ptr, n4 * 16, api.current_stream_handle()))
torch.cuda.synchronize()
tiles = (W // 128) * (H // 8)
stride = n4 // tiles
c = ws[:tiles * stride].view(tiles, stride, 4)[:, stride - 8:, :3].double().sum((0, 1)).cpu()
print("%d spp: %.0f full read groups per launch; no member pair in any lane: %.3f %% of the groups, %.3f %% of their halves"
      % (spp, c[0], 100 * c[1] / c[0], 100 * c[2] / (2 * c[0])))
