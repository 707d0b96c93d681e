"""Is a GiB slot of the placed allocator of ONE interference class throughout?  (The classes come in runs of 4 .. 64 GiB of physical
memory; nothing aligns their boundaries to the allocator's GiB slots.)  A state block that fills its slot, a 6-GiB arena; every
GiB of the arena streamed beside read-modify-writes of each QUARTER of the state slot (statmc_debug_interference_probe), and
each quarter of every arena GiB beside the state's first quarter.  python tools/experiments/slot_quarters.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api

dev = torch.device("cuda:0")
api.setup(0)
lib = api.load()
lib.statmc_debug_interference_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_float)]
MiB = 1 << 20
state = api.empty_placed((1000 * MiB // 4,), torch.float32, dev, api.MEM_STATE)      # (a slot holds 1024 MiB; the block leaves its tail free)
state.zero_()
arena = api.empty_placed((6 << 28,), torch.float32, dev, api.MEM_STREAM)
arena.zero_()
torch.cuda.synchronize()


def probe(stream_ptr, stream_bytes, rmw_ptr, rmw_bytes):
    ms = C.c_float()
    api.check(lib.statmc_debug_interference_probe(C.c_void_p(stream_ptr), stream_bytes, C.c_void_p(rmw_ptr), rmw_bytes, C.byref(ms)))
    return ms.value


print(api.placement_info()["map"])
q = 250 * MiB
print("arena GiB (rows) streamed beside read-modify-writes of the state slot's quarters (columns), ms:")
for g in range(6):
    print("  GiB %d: " % g + "  ".join("%.4f" % probe(arena.data_ptr() + (g << 30), 1 << 30, state.data_ptr() + k * q, q) for k in range(4)))
print("quarters of every arena GiB (256 MiB streams) beside the state's first quarter, ms:")
for g in range(6):
    print("  GiB %d: " % g + "  ".join("%.4f" % probe(arena.data_ptr() + (g << 30) + (k << 28), 1 << 28, state.data_ptr(), q) for k in range(4)))
