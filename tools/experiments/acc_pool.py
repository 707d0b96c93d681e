"""Round 5: the film-major accumulation with every buffer carved out of ONE pool at chosen offsets (the experiments that found the
interference classes).  python tools/experiments/acc_pool.py MODE
  random    the five arenas and the five state blocks at random 2 MiB-aligned offsets of a 96 GiB pool, 30 trials (4K / 64 spp): the
            same launch lands anywhere between 4.14 and 4.48 ms                                  (profiles/r05_acc_place3.log)
  sweep     ONE stat type, its arena sliding through the pool (state fixed); then a second arena sliding against a fixed first
            one: two levels with sharp edges                                                     (profiles/r05_acc_place4.log)
  map       a 1 GB arena (1080p / 40 spp, one type) sliding through a 280 GiB pool at 0.5 GiB steps   (profiles/r05_acc_map.log)
  fastslow  the 1-GiB map, then the whole 11-channel launch with arenas / state packed into the longest run of each letter:
            same class 0.76, different classes 0.85 at 1080p / 256 spp                            (profiles/r05_acc_fastslow.log)
  classes   the 1-GiB map, then arenas at the start of every long run x state in every run (1080p / 64 spp): three classes, the
            relation symmetric                                                                    (profiles/r05_acc_classes.log)
POOL_GB sizes the pool (default by mode)."""
import os
import random
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

dev = torch.device("cuda:0")
api.setup(0)
types = list(synthetic.FEATURES)
MODE = sys.argv[1] if len(sys.argv) > 1 else "fastslow"
W, H, S = (3840, 2160, 64) if MODE in ("random", "sweep") else (1920, 1080, 40)
MB2 = 2 << 20
POOL = int(os.environ.get("POOL_GB", 96 if MODE in ("random", "sweep") else 280)) << 30
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 30
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


if MODE == "random":
    print("pool %d GiB at %x" % (POOL >> 30, pool.data_ptr()), flush=True)
    for trial in range(trials):
        mode = "packed" if trial in (0, trials - 1) else "random"
        pl = place(mode)
        ms = timed(build(pl))
        print("trial %2d %-6s %.3f ms %.2f TB/s | arenas GiB: %s | state GiB: %s" % (
            trial, mode, ms, bpp() * W * H / ms / 1e9,
            " ".join("%s %.3f" % (t[:3], pl[(t, "a")] / 2 ** 30) for t in types),
            " ".join("%s %.3f" % (t[:3], pl[(t, "s")] / 2 ** 30) for t in types)), flush=True)

if MODE == "sweep":
    print("pool %d GiB at %x" % (POOL >> 30, pool.data_ptr()), flush=True)
    # Sweep: ONE stat type (mean-only RGB: a pure stream + 16 B/px of state); the arena slides through the pool in steps,
    # the state stays at the pool's end.  Then two types, the second arena sliding against a fixed first one.
    types[:] = ["normal"]
    smp, stsz, planes = sizes("normal")
    state_at = POOL - (256 << 20)
    step = int(float(os.environ.get("STEP_GB", 0.75)) * 2 ** 30) // MB2 * MB2
    off = 0
    while off + smp < state_at - (64 << 20):
        pl = {("normal", "a"): off, ("normal", "s"): state_at}
        ms = timed(build(pl))
        print("one type, arena at %7.3f GiB: %.3f ms %.2f TB/s" % (off / 2 ** 30, ms, 12 * S * W * H / ms / 1e9), flush=True)
        off += step
    types[:] = ["normal", "albedo"]
    state_at2 = state_at - (256 << 20)
    off = smp
    while off + smp < state_at2 - (64 << 20):
        pl = {("normal", "a"): 0, ("normal", "s"): state_at, ("albedo", "a"): off, ("albedo", "s"): state_at2}
        ms = timed(build(pl))
        print("two types, second arena at %7.3f GiB (first at 0): %.3f ms %.2f TB/s" % (off / 2 ** 30, ms, 24 * S * W * H / ms / 1e9), flush=True)
        off += step

if MODE == "map":
    print("pool %d GiB at %x" % (POOL >> 30, pool.data_ptr()), flush=True)
    # Map: ONE stat type (mean-only RGB: a pure stream), a 1 GB arena (1080p, 40 spp) sliding through the whole pool; the 33 MB
    # of state stay at the pool's start.
    types[:] = ["normal"]
    smp, stsz, planes = sizes("normal")
    step = int(float(os.environ.get("STEP_GB", 0.5)) * 2 ** 30) // MB2 * MB2
    off = 256 << 20
    line = []
    while off + smp < POOL - (64 << 20):
        pl = {("normal", "a"): off, ("normal", "s"): 0}
        ms = timed(build(pl))
        line.append("%.1f:%.2f" % (off / 2 ** 30, 12 * S * W * H / ms / 1e9))
        if len(line) == 16:
            print(" ".join(line), flush=True)
            line = []
        off += step
    print(" ".join(line), flush=True)

if MODE in ("fastslow", "classes"):
    print("pool %d GiB at %x" % (POOL >> 30, pool.data_ptr()), flush=True)
    GiB = 1 << 30
    # 1. map the pool at 1 GiB resolution with a 1 GB single-type stream (mode map)
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

if MODE == "fastslow":
    # 2. the whole 11-channel launch with everything packed inside the fast run / the slow run / arenas fast + state slow / ...
    for (W_, H_, S_) in ((3840, 2160, 64), (1920, 1080, 256), (1920, 1080, 64), (3840, 2160, 16)):
        globals().update(W=W_, H=H_, S=S_)
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

        f0, s0_ = (fast[0] + 1) * GiB, (slow[0] + 1) * GiB
        if (fast[1] - 2) * GiB < need_a + need_s or (slow[1] - 2) * GiB < need_a + need_s:
            print("%dx%d %d spp: a run is too short (%d / %d GiB for %.1f GiB)" % (W, H, S, fast[1], slow[1], (need_a + need_s) / GiB), flush=True)
            continue
        rows = []
        for rep in range(2):
            for tag, pl in (("arenas + state in FAST memory", packed(f0, f0 + need_a)), ("arenas + state in SLOW memory", packed(s0_, s0_ + need_a)),
                            ("arenas FAST, state SLOW", packed(f0, s0_)), ("arenas SLOW, state FAST", packed(s0_, f0))):
                ms = timed(build(pl))
                rows.append("%dx%d %3d spp  %-32s %.3f ms  %.2f TB/s  %.3f of 8 TB/s" % (W, H, S, tag, ms, bpp() * W * H / ms / 1e9, bpp() * W * H / ms / 8e9))
        print("\n".join(rows), flush=True)

if MODE == "classes":
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
