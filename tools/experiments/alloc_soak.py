"""A soak of the placed allocator's address handling (windows used again, blocks carved from slots, trims in between): every live block is
filled with its own number when it is made; after every operation every live block must still hold only that number.
python tools/experiments/alloc_soak.py [steps] [library.so]   (tests/test_placement_gpu.py runs it; what it guards against: HISTORY.md 4.1d)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from statmc_amd import build
if len(sys.argv) > 2:
    build.SO = os.path.abspath(sys.argv[2])
from statmc_amd import api
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 48
api.setup(0)
dev = torch.device("cuda:0")
rng = np.random.default_rng(11)
MiB = 1 << 18
sizes = [64 * MiB, 700 * MiB, 1536 * MiB, 3072 * MiB, 5000 * MiB]
live, next_id, windows = {}, 1, 0


def check(step):
    for k, t in live.items():
        v = t[::65521]
        lo, hi = float(v.min().item()), float(v.max().item())
        if lo != float(k) or hi != float(k):
            print("step %d: block %d (%d MiB) holds values %.1f .. %.1f" % (step, k, t.numel() // MiB, lo, hi))
            print(api.placement_info()["map"])
            sys.exit(1)


for step in range(steps):
    total = sum(t.numel() for t in live.values()) * 4
    if live and (rng.random() < 0.45 or total > (26 << 30)):
        k = list(live)[int(rng.integers(0, len(live)))]
        del live[k]
    else:
        n = int(sizes[int(rng.integers(0, len(sizes)))])
        role = api.MEM_STREAM if rng.random() < 0.8 else api.MEM_STATE
        t = api.empty_placed((n,), torch.float32, dev, role)
        t.fill_(float(next_id))
        live[next_id] = t
        windows += n >= 3072 * MiB
        next_id += 1
    if step % 13 == 12:
        api.load().statmc_placement_trim()
    torch.cuda.synchronize()
    check(step)
print("ok", steps, "steps,", next_id - 1, "blocks,", windows, "of them windows;", api.placement_info()["map"])
