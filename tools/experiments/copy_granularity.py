"""Host -> device rate of the bracket's seven uploads (157.6 MB at 1080p) against the number of pieces they are cut
into, and as 2-D copies of a band of six equally sized images.  python tools/experiments/copy_granularity.py"""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api

api.setup(0)
lib = api.load()
hip = C.CDLL("libamdhip64.so")
W, H = 1920, 1080
sizes = [W * H * 4] + [W * H * 12] * 6          # n, then six RGB images
total = sum(sizes)
host = C.c_void_p()
api.check(lib.statmc_malloc_host(C.byref(host), total))
C.memset(host, 1, total)
dev = torch.empty(total, dtype=torch.uint8, device="cuda:0")
stream = api.current_stream_handle()
offs = [sum(sizes[:i]) for i in range(len(sizes))]


def timed(fn, reps=7):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


for nb in (1, 2, 4, 6, 8, 16):
    def run():
        for k in range(nb):
            for o, s in zip(offs, sizes):
                rows = H // nb
                row = s // H
                a, b = k * rows * row, ((k + 1) * rows if k < nb - 1 else H) * row
                api.check(lib.statmc_upload(C.c_void_p(dev.data_ptr() + o + a), C.c_void_p(host.value + o + a), b - a, stream))
    ms = timed(run)
    print("%2d bands = %3d copies: %.3f ms  (%.1f GB/s)" % (nb, nb * 7, ms, total / ms / 1e6), flush=True)

# 2-D copy: one band of the six RGB images per call (pitch = image size)
hip.hipMemcpy2DAsync.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
for nb in (4, 6, 8):
    def run2d():
        for k in range(nb):
            rows = H // nb
            y0, y1 = k * rows, ((k + 1) * rows if k < nb - 1 else H)
            a, b = y0 * W * 4, y1 * W * 4
            api.check(lib.statmc_upload(C.c_void_p(dev.data_ptr() + a), C.c_void_p(host.value + a), b - a, stream))
            a, b = y0 * W * 12, y1 * W * 12
            rc = hip.hipMemcpy2DAsync(C.c_void_p(dev.data_ptr() + offs[1] + a), sizes[1], C.c_void_p(host.value + offs[1] + a), sizes[1],
                                      b - a, 6, 1, stream)   # hipMemcpyHostToDevice = 1
            assert rc == 0, rc
    ms = timed(run2d)
    print("%2d bands, 1 + one 2-D copy each: %.3f ms  (%.1f GB/s)" % (nb, ms, total / ms / 1e6), flush=True)

# (hipMemcpyBatchAsync would take the seven pieces of a band in one call, but the HIP runtime PyTorch 2.10 ships --
# the one a process that imports torch has loaded -- does not export it: not an option for the library.)

# the same pieces dealt over two or three copy streams (does the per-copy gap overlap?)
for ns in (2, 3):
    extra = [torch.cuda.Stream() for _ in range(ns)]
    for nb in (6, 8):
        def run_multi():
            for k in range(nb):
                for i, (o, s) in enumerate(zip(offs, sizes)):
                    rows = H // nb
                    row = s // H
                    a, b = k * rows * row, ((k + 1) * rows if k < nb - 1 else H) * row
                    st = C.c_void_p(extra[i % ns].cuda_stream)
                    api.check(lib.statmc_upload(C.c_void_p(dev.data_ptr() + o + a), C.c_void_p(host.value + o + a), b - a, st))
        ms = timed(run_multi)
        print("%2d bands = %3d copies over %d streams: %.3f ms  (%.1f GB/s)" % (nb, nb * 7, ns, ms, total / ms / 1e6), flush=True)
