"""Round 5, ninth look (class matrix of the real launch): the same launch lands on one of several time levels (4.29 .. 4.71 ms at 4K / 64 spp) by allocation, and
offsets up to 266 MB inside one pool change nothing (acc_place.py).  Do GB-scale offsets?  One fresh pool of POOL_GB; every
trial places the five arenas and the five state blocks at random 2 MiB-aligned offsets, non-overlapping; prints the
time and the offsets in GiB.  python tools/experiments/acc_place3.py [trials]"""
import os
import random
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

dev = torch.device("cuda:0")
api.setup(0)
types = list(synthetic.FEATURES)
W, H, S = 1920, 1080, 40
MB2 = 2 << 20
POOL = int(os.environ.get("POOL_GB", 280)) << 30
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 24
pool = torch.empty(POOL // 4, dtype=torch.float32, device=dev)
for i in range(0, pool.numel(), 1 << 28):
    pool[i:i + (1 << 28)].uniform_()
base_off = (-pool.data_ptr()) % MB2
rng = random.Random(5)


def sizes(t):
    c = synthetic.CHANNELS[t]
    cfgt = film.STAT_TYPES[t]
    planes = 1 + cfgt["max_moment"] + (2 if cfgt["transform"] else 0)      # n + moments
    smp = -(-S * H * W * c * 4 // MB2) * MB2
    st = -(-H * W * c * 4 // MB2) * MB2
    return smp, st, planes


def bpp():
    t = 0
    for x in types:
        c = film.STAT_TYPES[x]
        planes = c["max_moment"] + (2 if c["transform"] else 0)
        t += 4 * c["channels"] * S + 2 * (4 + 4 * c["channels"] * planes)
    return t


def view(off_bytes, shape, dtype=torch.float32):
    n = 1
    for d in shape:
        n *= d
    v = pool[(base_off + off_bytes) // 4:(base_off + off_bytes) // 4 + n]
    return (v if dtype == torch.float32 else v.view(dtype)).view(*shape)


def place(mode):
    """-> {(type, 'a' | 's'): byte offset}; the state planes of a type sit back to back at its state offset"""
    blocks = []
    for t in types:
        smp, st, planes = sizes(t)
        blocks.append((t, "a", smp))
        blocks.append((t, "s", st * planes))
    if mode == "packed":
        order = blocks
        gaps = [0] * len(blocks)
    else:
        order = blocks[:]
        rng.shuffle(order)
        slack = POOL - MB2 - sum(b[2] for b in blocks) - (64 << 20)
        cuts = sorted(rng.randrange(0, slack // MB2) for _ in blocks)
        gaps = [(cuts[0]) * MB2] + [(cuts[i] - cuts[i - 1]) * MB2 for i in range(1, len(cuts))]
    pos, out = 0, {}
    for (t, kind, size), gap in zip(order, gaps):
        pos += gap
        out[(t, kind)] = pos
        pos += size
    return out


def build(pl):
    sts = []
    for t in types:
        cfgt = film.STAT_TYPES[t]
        c = cfgt["channels"]
        smp, stsz, planes = sizes(t)
        a = view(pl[(t, "a")], (S, H, W, c))
        so = pl[(t, "s")]
        st = {"n": view(so, (H, W), torch.int32)}
        k = 1
        for name, on in (("mean", True), ("m2", cfgt["max_moment"] >= 2), ("m3", cfgt["max_moment"] >= 3),
                         ("film_mean", cfgt["transform"]), ("film_m2", cfgt["transform"])):
            if on:
                st[name] = view(so + k * stsz, (H, W, c))
                k += 1
            else:
                st[name] = None
        for v in st.values():
            if v is not None:
                v.zero_()
        sts.append(api.make_stat_type(a, st, cfgt["transform"], cfgt["max_moment"]))
    return sts


def timed(sts):
    api.accumulate(W, H, sts)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            api.accumulate(W, H, sts)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 6)
    return best



print("pool %d GiB at %x" % (POOL >> 30, pool.data_ptr()), flush=True)
GiB = 1 << 30
# 1. map the pool at 1 GiB resolution with a 1 GB single-type stream (acc_map.py)
types[:] = ["normal"]
smp = sizes("normal")[0]
speed = []
off = 0
while off + GiB <= POOL - MB2:
    ms = timed(build({("normal", "a"): off + (32 << 20) if off == 0 else off, ("normal", "s"): 0}))
    speed.append(12 * S * W * H / ms / 1e9)
    off += GiB
lo, hi = min(speed), max(speed)
cut = (lo + hi) / 2
print("map: min %.2f max %.2f TB/s, cut %.2f; fast GiB %d, slow GiB %d" % (lo, hi, cut, sum(v > cut for v in speed), sum(v <= cut for v in speed)), flush=True)
print("".join("F" if v > cut else "s" for v in speed), flush=True)


def longest(pred):
    best, cur, start = (0, 0), 0, 0
    for i, v in enumerate(speed + [None]):
        if v is not None and pred(v):
            if cur == 0:
                start = i
            cur += 1
        else:
            if cur > best[1]:
                best = (start, cur)
            cur = 0
    return best


fast = longest(lambda v: v > cut + 0.25 * (hi - cut))
slow = longest(lambda v: v < cut - 0.25 * (cut - lo))
print("longest fast run: GiB %d .. %d; longest slow run: GiB %d .. %d" % (fast[0], fast[0] + fast[1], slow[0], slow[0] + slow[1]), flush=True)
# 2. class matrix at the level of the real launch: 1080p / 64 spp, all stat types (6.1 GiB of arenas, 0.5 GiB of state);
# arenas packed at the start of every run of >= 8 equal letters, state in one slot of every run
runs = []
start = 0
letters = ["F" if v > cut else "s" for v in speed]
for i in range(1, len(letters) + 1):
    if i == len(letters) or letters[i] != letters[start]:
        runs.append((start, i - start, letters[start]))
        start = i
print("runs:", " ".join("%s%d@%d" % (l, n, a) for a, n, l in runs), flush=True)
globals().update(W=1920, H=1080, S=64)
types[:] = list(synthetic.FEATURES)
need_a = sum(sizes(t)[0] for t in types)
need_s = sum(sizes(t)[1] * sizes(t)[2] for t in types)


def packed(a0, s0):
    pl, pa, ps = {}, a0, s0
    for t in types:
        smp_, st_, planes_ = sizes(t)
        pl[(t, "a")] = pa
        pa += smp_
        pl[(t, "s")] = ps
        ps += st_ * planes_
    return pl


arena_runs = [r for r in runs if r[1] >= 9][:10]
state_runs = [r for r in runs if r[1] >= 3][:14]
print("state at the LAST slot of run ->   " + " ".join("%6s" % ("%s@%d" % (l, a)) for a, n, l in state_runs), flush=True)
for a, n, l in arena_runs:
    row = []
    for sa, sn, sl in state_runs:
        a0 = (a + 1) * GiB if a == 0 else a * GiB
        s0 = (sa + sn - 1) * GiB
        if s0 < a0 + need_a and s0 + need_s > a0:
            row.append("   -  ")
            continue
        ms = timed(build(packed(a0, s0)))
        row.append("%6.3f" % (bpp() * W * H / ms / 8e9))
    print("arenas at the start of run %s%d@%-4d " % (l, n, a) + " ".join(row), flush=True)
