"""A/B of the film-major accumulation: RGB sample planes by LDS-DMA (default) against loads into registers
(statmc_debug_accumulate_dma), per type set at 1080p, interleaved repeats; the two walks must leave the same bits."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import build
if os.environ.get("STATMC_VARIANT"):
    os.environ.setdefault("STATMC_ALLOW_DIAGNOSTIC_BUILD", "1"); build.SO = os.path.abspath(os.environ["STATMC_VARIANT"])
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
lib = api.load()
W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sc = synthetic.Scene(W, H, seed=1, device=dev)
smp = sc.samples(S, seed=2)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def bpp(t):
    cfg = film.STAT_TYPES[t]; c = cfg["channels"]
    planes = cfg["max_moment"] + (2 if cfg["transform"] else 0)
    return 4 * c * S + 2 * (4 + 4 * c * planes)
RESIDENT = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
MODES = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 0]
for types in (["radiance"], ["normal"], ["depth", "materialid"], ["normal", "albedo"], ["radiance", "normal", "albedo"], ["normal", "albedo", "depth", "materialid"], list(synthetic.FEATURES)):
    sub = {t: smp[t] for t in types}
    b = sum(bpp(x) for x in types) * W * H
    states = {}
    for dma in (1, 0):
        lib.statmc_debug_accumulate_dma(dma)
        fs = film.FilmStats(W, H, dev, types=types)
        fs.accumulate(sub); torch.cuda.synchronize()
        states[dma] = {(t, k): v for t in types for k, v in fs.state[t].items() if v is not None}
    same = all(torch.equal(states[1][k].view(torch.int32), states[0][k].view(torch.int32)) for k in states[1])
    fs = film.FilmStats(W, H, dev, types=types)
    for resident in RESIDENT:
        lib.statmc_debug_accumulate_resident_blocks(resident)
        best = {1: 1e9, 0: 1e9}
        for rep in range(3):
            for dma in (1, 0):
                lib.statmc_debug_accumulate_dma(dma)
                best[dma] = min(best[dma], timeit(lambda: fs.accumulate(sub)))
        lib.statmc_debug_accumulate_dma(1)
        print("%-42s resident %4d LDS-DMA %.3f ms %5.0f GB/s | registers %.3f ms %5.0f GB/s | same bits: %s"
              % ("+".join(types), resident, best[1], b / best[1] / 1e6, best[0], b / best[0] / 1e6, same), flush=True)
    lib.statmc_debug_accumulate_resident_blocks(0)
