"""Does the step's accumulate (HBM-bound) keep its rate on a subset of the CUs, and does the window filter (VALU-bound)
scale on the complement?  Streams with CU masks (hipExtStreamCreateWithCUMask), each kernel alone and both together.
python tools/experiments/cu_mask.py [spp]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

W, H = 1920, 1080
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
api.setup(0)
hip = C.CDLL("libamdhip64.so")
N_CU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(lo, hi, period=None):
    """Stream whose kernels run on the CU-mask bits b with lo <= b % period < hi (period None: lo <= b < hi)."""
    words = (N_CU + 31) // 32
    mask = (C.c_uint32 * words)()
    for b in range(N_CU):
        if lo <= (b % period if period else b) < hi:
            mask[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), words, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


types = list(synthetic.FEATURES)
scene = synthetic.Scene(W, H, n_regions=12, seed=1, device=dev)
samples = {t: [] for t in types}
for s0 in range(0, spp, 32):
    part = scene.samples(min(32, spp - s0), seed=1000 + s0, features=types)
    for t in types:
        samples[t].append(part[t])
samples = {t: torch.cat(v, dim=0) for t, v in samples.items()}
fs = film.FilmStats(W, H, dev, types=types)
fs.accumulate(samples)
fs.prepass()
fs.window_filter()
torch.cuda.synchronize()
print("CUs", N_CU, "spp", spp, flush=True)


def timed(stream, fn, reps):
    with torch.cuda.stream(stream):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


acc = lambda: fs.accumulate(samples)
flt = lambda: fs.window_filter()


def together(sa, sf, label, reps=10):
    torch.cuda.synchronize()
    ea = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ef = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t0 = torch.cuda.Event(enable_timing=True)
    t0.record()
    sa.wait_event(t0)
    sf.wait_event(t0)
    with torch.cuda.stream(sa):
        ea[0].record()
        for _ in range(reps):
            acc()
        ea[1].record()
    with torch.cuda.stream(sf):
        ef[0].record()
        for _ in range(reps):
            flt()
        ef[1].record()
    torch.cuda.synchronize()
    print("%-28s accumulate %.3f ms, filter %.3f ms; pair %.3f ms" % (
        label, ea[0].elapsed_time(ea[1]) / reps, ef[0].elapsed_time(ef[1]) / reps,
        max(t0.elapsed_time(ea[1]), t0.elapsed_time(ef[1])) / reps), flush=True)


full = masked_stream(0, N_CU)
print("alone, all CUs: accumulate %.3f ms  filter %.3f ms" % (timed(full, acc, 5), timed(full, flt, 5)), flush=True)
for period in (32, 16, 8, 4):
    for k in (5, 6):        # accumulate on k/8 of every group of `period` mask bits
        a = period * k // 8
        sa, sf = masked_stream(0, a, period), masked_stream(a, period, period)
        print("period %2d  %d/%d : alone accumulate %.3f ms  filter %.3f ms" % (period, a, period - a, timed(sa, acc, 5), timed(sf, flt, 5)), flush=True)
        together(sa, sf, "   together")
# no masks at all: two plain streams, the dispatcher decides
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
together(s1, s2, "unmasked streams")
together(s2, s1, "unmasked streams (swapped)")
