"""Round 5: three follow-ups on statmc_malloc_placed (all stat types unless said, 1080p / 256 spp).
python tools/experiments/acc_placed_more.py MODE
  block   the five arenas carved out of ONE placed block against five placed blocks against torch's allocator: contiguity of
          the arenas is not a factor                                                    (profiles/r05_acc_placed2.log)
  pairs   one stat type, eight placed arenas x three placed states, every pair: 6.75 .. 6.79 TB/s -- "apart" is uniform
                                                                                        (profiles/r05_acc_placed3.log)
  where   a state allocated before the arenas, one after, one in another slot, one in torch's memory (ORDER=arenas-first for
          the other order): where in the state role the moments sit does not matter     (profiles/r05_acc_placed4.log)
  ranks   the allocator's probe (statmc_debug_interference_probe) on every GiB of every arena against the state and against every
          arena: all apart from the state, all in one class among themselves            (profiles/r05_acc_ranks.log)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

dev = torch.device("cuda:0")
api.setup(0)
types = list(synthetic.FEATURES)
MODE = sys.argv[1] if len(sys.argv) > 1 else "block"
W, H, S = 1920, 1080, 256


def bpp(S):
    t = 0
    for x in types:
        c = film.STAT_TYPES[x]
        planes = c["max_moment"] + (2 if c["transform"] else 0)
        t += 4 * c["channels"] * S + 2 * (4 + 4 * c["channels"] * planes)
    return t


def timed(fs, a, reps=8):
    fs.accumulate(a)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fs.accumulate(a)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


if MODE == "block":
    fs = film.FilmStats(W, H, dev, types=types, placed=True)
    n_el = {t: S * H * W * synthetic.CHANNELS[t] for t in types}
    big = api.empty_placed((sum(n_el.values()),), torch.float32, dev, api.MEM_STREAM)
    one, pos = {}, 0
    for t in types:
        one[t] = big[pos:pos + n_el[t]].view(S, H, W, synthetic.CHANNELS[t])
        pos += n_el[t]
        for s0 in range(0, S, 16):
            one[t][s0:s0 + 16].uniform_()
    print("map after the one block:   ", api.placement_info()["map"], flush=True)
    five = {t: api.empty_placed((S, H, W, synthetic.CHANNELS[t]), torch.float32, dev, api.MEM_STREAM) for t in types}
    for t in types:
        five[t].copy_(one[t])
    print("map after five more blocks:", api.placement_info()["map"], flush=True)
    plain = {t: one[t].clone() for t in types}
    fs_t = film.FilmStats(W, H, dev, types=types)
    for rnd in range(2):
        for name, f, a in (("arenas in one placed block", fs, one), ("five placed blocks", fs, five), ("torch arenas, placed state", fs, plain), ("torch arenas, torch state", fs_t, plain),
                           ("placed block, torch state", fs_t, one)):
            ms = timed(f, a)
            print("%-28s %.3f ms  %.3f of 8 TB/s" % (name, ms, bpp(S) * W * H / ms / 8e9), flush=True)

if MODE == "pairs":
    W, H, S = 1920, 1080, 256
    states, spacers = [], []
    for k in range(3):
        states.append(film.FilmStats(W, H, dev, types=["normal"], placed=True))
        spacers.append(api.empty_placed((900 << 18,), torch.float32, dev, api.MEM_STATE))      # 900 MiB: the next state starts in another slot
    arenas = []
    for k in range(8):
        a = api.empty_placed((S, H, W, 3), torch.float32, dev, api.MEM_STREAM)
        for s0 in range(0, S, 32):
            a[s0:s0 + 32].uniform_()
        arenas.append(a)
    info = api.placement_info()
    base = None
    print("map:", info["map"], flush=True)
    print("state blocks at (GiB from the first):", [round((st.state["normal"]["mean"].data_ptr() - states[0].state["normal"]["mean"].data_ptr()) / 2 ** 30, 2) for st in states])
    print("arena blocks at (GiB from the first state):", [round((a.data_ptr() - states[0].state["normal"]["mean"].data_ptr()) / 2 ** 30, 2) for a in arenas], flush=True)
    for rnd in range(2):
        for si, st in enumerate(states):
            row = []
            for a in arenas:
                smp = {"normal": a}
                st.accumulate(smp)
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(6):
                        st.accumulate(smp)
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / 6)
                row.append("%.2f" % (12 * S * W * H / best / 1e9))
            print("state %d: TB/s per arena: %s" % (si, " ".join(row)), flush=True)

if MODE == "where":
    order = os.environ.get("ORDER", "state-first")
    states = {}
    if order == "state-first":
        states["state before the arenas"] = film.FilmStats(W, H, dev, types=types, placed=True)
    arenas = {t: api.empty_placed((S, H, W, synthetic.CHANNELS[t]), torch.float32, dev, api.MEM_STREAM) for t in types}
    for t in types:
        for s0 in range(0, S, 16):
            arenas[t][s0:s0 + 16].uniform_()
    states["state after the arenas"] = film.FilmStats(W, H, dev, types=types, placed=True)
    spacer = api.empty_placed((1 << 28,), torch.float32, dev, api.MEM_STATE)          # 1 GiB: the next state lies in another slot
    states["state in another slot"] = film.FilmStats(W, H, dev, types=types, placed=True)
    states["state in torch's memory"] = film.FilmStats(W, H, dev, types=types)
    info = api.placement_info()
    print("order:", order, " map:", info["map"], flush=True)
    base = min(a.data_ptr() for a in arenas.values())
    p0 = None
    for name, fs in states.items():
        ptr = fs.state["radiance"]["n"].data_ptr()
        print("%-28s radiance n at %+8.3f GiB from the first arena" % (name, (ptr - base) / 2 ** 30), flush=True)
    for rnd in range(2):
        for name, fs in states.items():
            ms = timed(fs, arenas)
            print("%-28s %.3f ms  %.3f of 8 TB/s" % (name, ms, bpp(S) * W * H / ms / 8e9), flush=True)

if MODE == "ranks":
    # Are the arenas of a placed launch in ONE class among themselves?  The allocator's probe on every GiB of every arena against the
    # first 64 MiB of every arena (and of the state): '=' same class, '.' apart.  (The probed words change: run this last.)
    import ctypes as C
    lib = api.load()
    lib.statmc_debug_interference_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_float)]
    fs = film.FilmStats(W, H, dev, types=types, placed=os.environ.get("ORDER") != "arenas-first")
    arenas = {t: api.empty_placed((S, H, W, synthetic.CHANNELS[t]), torch.float32, dev, api.MEM_STREAM) for t in types}
    for t in types:
        for s0 in range(0, S, 16):
            arenas[t][s0:s0 + 16].uniform_()
    if os.environ.get("ORDER") == "arenas-first":
        fs = film.FilmStats(W, H, dev, types=types, placed=True)
    ms = timed(fs, arenas)
    print("map:", api.placement_info()["map"])
    print("the launch: %.3f ms  %.3f of 8 TB/s" % (ms, bpp(S) * W * H / ms / 8e9), flush=True)
    GiB = 1 << 30
    targets = [("state", fs.state["radiance"]["mean"].data_ptr())] + [(t, arenas[t].data_ptr()) for t in types]
    for t in types:
        n_gib = arenas[t].numel() * 4 // GiB
        for g in range(max(n_gib, 1)):
            row = []
            for name, ptr in targets:
                out = C.c_float()
                nbytes = min(GiB, arenas[t].numel() * 4 - g * GiB) // 16 * 16
                if name == t and g == 0:
                    row.append("  #  ")
                    continue
                api.check(lib.statmc_debug_interference_probe(C.c_void_p(arenas[t].data_ptr() + g * GiB), nbytes, C.c_void_p(ptr), 64 << 20, C.byref(out)))
                row.append("%.3f" % (out.value * GiB / nbytes))
            print("%-10s GiB %d against the first 64 MiB of [%s]: %s" % (t, g, " ".join(n for n, _ in targets), " ".join(row)), flush=True)

if MODE == "depth":
    # Does it matter WHERE on the card a placed launch's buffers lie?  Set after set of (state, five arenas) is allocated and kept,
    # each one further into the card's memory than the one before; every set is timed (both launch shapes) when it is made and all
    # of them again at the end.  BALLAST_GB: torch memory allocated (and written) before the first set.
    lib = api.load()
    ballast = None
    if os.environ.get("BALLAST_GB"):
        ballast = torch.empty(int(float(os.environ["BALLAST_GB"]) * 2 ** 28), dtype=torch.float32, device=dev)
        ballast.zero_()
    sets = []

    def measure(k):
        fs, a = sets[k]
        out = []
        for g in (0, 1):
            api.check(lib.statmc_debug_accumulate_launch(g, 0))
            ms = timed(fs, a, 7)
            out.append("grid %d %.3f ms %.3f" % (g, ms, bpp(S) * W * H / ms / 8e9))
        api.check(lib.statmc_debug_accumulate_launch(-1, 0))
        return "  ".join(out)

    for k in range(int(os.environ.get("SETS", 5))):
        a = {t: api.empty_placed((S, H, W, synthetic.CHANNELS[t]), torch.float32, dev, api.MEM_STREAM) for t in types}
        for t in types:
            if os.environ.get("FILL") == "copy":      # written by one device-to-device copy of a torch tensor (what acc_placed.py does)
                src = torch.empty((S, H, W, synthetic.CHANNELS[t]), device=dev)
                for s0 in range(0, S, 16):
                    src[s0:s0 + 16].uniform_()
                a[t].copy_(src)
                del src
            else:
                for s0 in range(0, S, 16):
                    a[t][s0:s0 + 16].uniform_()
        sets.append((film.FilmStats(W, H, dev, types=types, placed=True), a))
        print("set %d (slots so far %d): %s" % (k, api.placement_info()["slots"], measure(k)), flush=True)
    print("map:", api.placement_info()["map"])
    for k in range(len(sets)):
        print("set %d again: %s" % (k, measure(k)), flush=True)
