"""Round 5, fourth look: is the placement effect (the same launch 5.35 .. 6.15 TB/s by allocation) physical fragmentation?
Pool A is allocated first, on whatever the box's VRAM looks like; then the rest of the card is filled with 16 MiB
allocations, every other one freed, and pool B allocated out of the holes; A, B, A again.  4K, 64 spp, all stat types.
python tools/experiments/acc_frag.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

dev = torch.device("cuda:0")
api.setup(0)
types = list(synthetic.FEATURES)
W, H, S = 3840, 2160, 64
MB2 = 2 << 20


def bpp(S):
    t = 0
    for x in types:
        c = film.STAT_TYPES[x]
        planes = c["max_moment"] + (2 if c["transform"] else 0)
        t += 4 * c["channels"] * S + 2 * (4 + 4 * c["channels"] * planes)
    return t


need = sum((S * H * W * synthetic.CHANNELS[t] * 4 // MB2 + 3) * MB2 + 7 * ((H * W * synthetic.CHANNELS[t] * 4) // MB2 + 3) * MB2 for t in types)


def make_pool():
    pool = torch.empty(need // 4 + MB2, dtype=torch.float32, device=dev)
    for i in range(0, pool.numel(), 1 << 28):
        pool[i:i + (1 << 28)].uniform_()
    return pool


def carve(pool):
    pos = (-pool.data_ptr()) % MB2 // 4
    sts, keep = [], []

    def take(nfloats, dtype=torch.float32):
        nonlocal pos
        v = pool[pos:pos + nfloats]
        pos += (nfloats * 4 + MB2 - 1) // MB2 * (MB2 // 4)
        return v if dtype == torch.float32 else v.view(dtype)

    for t in types:
        cfgt = film.STAT_TYPES[t]
        c = cfgt["channels"]
        smp = take(S * H * W * c).view(S, H, W, c)
        st = {"n": take(H * W, torch.int32).view(H, W)}
        st["n"].zero_()
        for name, on in (("mean", True), ("m2", cfgt["max_moment"] >= 2), ("m3", cfgt["max_moment"] >= 3),
                         ("film_mean", cfgt["transform"]), ("film_m2", cfgt["transform"])):
            st[name] = take(H * W * c).view(H, W, c).zero_() if on else None
        sts.append(api.make_stat_type(smp, st, cfgt["transform"], cfgt["max_moment"]))
        keep.append((smp, st))
    return sts, keep


def timed(sts):
    if os.environ.get("SUBSETS"):
        for sub, nm in (([1], "normal only"), ([3], "depth only"), ([0], "radiance only"), ([1, 2], "normal + albedo"), ([3, 4], "depth + materialid")):
            ms = timed_one([sts[i] for i in sub])
            c = sum(film.STAT_TYPES[types[i]]["channels"] for i in sub)
            print("      %-20s %.3f ms  %.2f TB/s (samples only)" % (nm, ms, 4 * c * S * W * H / ms / 1e9), flush=True)
    return timed_one(sts)


def timed_one(sts):
    api.accumulate(W, H, sts)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            api.accumulate(W, H, sts)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 8)
    return best


def report(tag, pool):
    sts, keep = carve(pool)
    if os.environ.get("PMC"):            # counter passes: exactly three launches per report, no timing
        for _ in range(3):
            api.accumulate(W, H, sts)
        torch.cuda.synchronize()
        print(tag, "3 launches", flush=True)
        return
    ms = timed(sts)
    print("%-44s %.3f ms  %.2f TB/s   (pool at %x)" % (tag, ms, bpp(S) * W * H / ms / 1e9, pool.data_ptr()), flush=True)


free0, total = torch.cuda.mem_get_info(dev)
print("free %.1f GB of %.1f GB" % (free0 / 1e9, total / 1e9), flush=True)
A = make_pool()
report("pool A (allocated first)", A)
chunks = []
chunk = 16 << 20
while torch.cuda.mem_get_info(dev)[0] > (6 << 30):
    chunks.append(torch.empty(chunk // 4, dtype=torch.float32, device=dev))
print("filled the card with %d x 16 MiB; free %.1f GB" % (len(chunks), torch.cuda.mem_get_info(dev)[0] / 1e9), flush=True)
chunks = [c for i, c in enumerate(chunks) if i % 2]
torch.cuda.empty_cache()
print("every other one freed; free %.1f GB" % (torch.cuda.mem_get_info(dev)[0] / 1e9), flush=True)
B = make_pool()
report("pool B (out of 16 MiB holes)", B)
report("pool A again", A)
del chunks
torch.cuda.empty_cache()
C = make_pool()
report("pool C (after freeing everything else)", C)
report("pool B again", B)
