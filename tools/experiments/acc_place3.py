"""Round 5, fifth look: the same launch lands on one of several time levels (4.29 .. 4.71 ms at 4K / 64 spp) by allocation, and
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
W, H, S = 3840, 2160, 64
MB2 = 2 << 20
POOL = int(os.environ.get("POOL_GB", 96)) << 30
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
for trial in range(trials):
    mode = "packed" if trial in (0, trials - 1) else "random"
    pl = place(mode)
    ms = timed(build(pl))
    print("trial %2d %-6s %.3f ms %.2f TB/s | arenas GiB: %s | state GiB: %s" % (
        trial, mode, ms, bpp() * W * H / ms / 1e9,
        " ".join("%s %.3f" % (t[:3], pl[(t, "a")] / 2 ** 30) for t in types),
        " ".join("%s %.3f" % (t[:3], pl[(t, "s")] / 2 ** 30) for t in types)), flush=True)
