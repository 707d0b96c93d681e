"""Round 5, third look: the same film-major launch runs 2 - 9 % apart depending on where its buffers were allocated
(acc_gap.py, acc_gap2.py).  Which address bits matter?  Everything is carved out of ONE device buffer at chosen offsets:
  pad   rows appended to every sample plane and state image (the launch accumulates rows [0, H) of a film H + pad rows
        tall: statmc_accumulate_rows), i.e. the stride between a pixel's consecutive samples modulo the powers of two
  skew  byte offset added to the i-th stat type's arena (and, `sskew`, to every state plane), on top of a 2 MiB-aligned base
All stat types, S samples per pixel, best of 3 x 8 launches per configuration, the list walked twice.
python tools/experiments/acc_place.py [W H S]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

dev = torch.device("cuda:0")
api.setup(0)
types = list(synthetic.FEATURES)
W, H, S = (int(v) for v in (sys.argv[1:4] + ["3840", "2160", "64"][len(sys.argv) - 1:]))
MB2 = 2 << 20


def bpp(S):
    t = 0
    for x in types:
        c = film.STAT_TYPES[x]
        planes = c["max_moment"] + (2 if c["transform"] else 0)
        t += 4 * c["channels"] * S + 2 * (4 + 4 * c["channels"] * planes)
    return t


MAXPAD = int(os.environ.get("MAXPAD", 16))
MAXSKEW = int(os.environ.get("MAXSKEW", 0))
need = 0
for t in types:
    c = synthetic.CHANNELS[t]
    need += MAXSKEW * 5 + (S * (H + MAXPAD) * W * c * 4 // MB2 + 3) * MB2 + 7 * (((H + MAXPAD) * W * c * 4) // MB2 + 3) * MB2
pool = torch.empty(need // 4 + MB2, dtype=torch.float32, device=dev)
for i in range(0, pool.numel(), 1 << 28):
    pool[i:i + (1 << 28)].uniform_()
base = pool.data_ptr()
base_off = (-base) % MB2 // 4           # first 2 MiB-aligned float of the pool


def carve(cfg):
    """-> list of api.StatType for this configuration (pad rows, arena skew bytes, state skew bytes)"""
    pad, skew, sskew = cfg
    Hp = H + pad
    pos = base_off
    sts, keep = [], []
    k = 0

    def take(nfloats, off_bytes, dtype=torch.float32):
        nonlocal pos
        start = pos + off_bytes // 4
        v = pool[start:start + nfloats]
        pos += (nfloats * 4 + off_bytes + MB2 - 1) // MB2 * (MB2 // 4)
        return v if dtype == torch.float32 else v.view(dtype)

    for i, t in enumerate(types):
        cfgt = film.STAT_TYPES[t]
        c = cfgt["channels"]
        smp = take(S * Hp * W * c, skew * i).view(S, Hp, W, c)
        st = {}
        for name in ("n", "mean", "m2", "m3", "film_mean", "film_m2"):
            k += 1
            if name == "n":
                st[name] = take(Hp * W, sskew * k, torch.int32).view(Hp, W)
                st[name].zero_()
            elif name in ("mean",) or (name == "m2" and cfgt["max_moment"] >= 2) or (name == "m3" and cfgt["max_moment"] >= 3) \
                    or (name in ("film_mean", "film_m2") and cfgt["transform"]):
                st[name] = take(Hp * W * c, sskew * k).view(Hp, W, c)
                st[name].zero_()
            else:
                st[name] = None
        sts.append(api.make_stat_type(smp, st, cfgt["transform"], cfgt["max_moment"]))
        keep.append((smp, st))
    assert pos * 4 <= pool.numel() * 4, "pool too small"
    return sts, keep, Hp


def timed(sts, Hp):
    def run():
        api.accumulate(W, Hp, sts, rows=(0, H))
    run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 8)
    return best


configs = [(0, 0, 0)]
if os.environ.get("BIG"):
    configs = [(0, 0, 0)] + [(p, 0, 0) for p in (45, 91, 182, 364, 728)] + [(0, k * MB2, 0) for k in (1, 3, 7, 31, 127)] + [(91, 7 * MB2, 0), (0, 0, 3 * MB2)]
elif os.environ.get("QUICK"):
    configs = [(0, 0, 0), (1, 4096, 4096)]
else:
  configs += [(p, 0, 0) for p in (1, 2, 3, 4, 5, 8, 16)]
  configs += [(0, k, 0) for k in (256, 1024, 4096, 16384, 65536, 262144, 1 << 20)]
  configs += [(0, 0, k) for k in (256, 4096, 65536, 1 << 20)]
  configs += [(1, 4096, 4096), (3, 65536 + 4096, 4096 + 256)]
print("film %dx%d, %d spp, pool %.1f GB at %x" % (W, H, S, pool.numel() * 4 / 1e9, base), flush=True)
for walk in range(2):
    for cfg in configs:
        sts, keep, Hp = carve(cfg)
        ms = timed(sts, Hp)
        stride = Hp * W * 12
        print("walk %d  pad %2d rows (RGB plane stride = 2^%d x %d)  arena skew %7d B  state skew %7d B : %.3f ms  %.2f TB/s"
              % (walk, cfg[0], (stride & -stride).bit_length() - 1, stride // (stride & -stride), cfg[1], cfg[2], ms,
                 bpp(S) * W * H / ms / 1e9), flush=True)
        del sts, keep
