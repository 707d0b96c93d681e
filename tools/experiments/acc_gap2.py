"""Round 5, second look: acc_gap.py measured the SAME launch (4K / 64 spp, one film + arena) at 5.63 and 6.15 TB/s in two sections
of one process.  Time since start, or where the allocations landed?  A time series of one case (a line per 8 launches, for
`secs` seconds), then the same film over a second and third arena allocated later, then the first arena again; the memory /
fabric clocks and the power from sysfs next to every line when the box lets an ordinary user read them.
python tools/experiments/acc_gap2.py [W H S secs]"""
import glob
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

dev = torch.device("cuda:0")
api.setup(0)
types = list(synthetic.FEATURES)
W, H, S, secs = (int(v) for v in (sys.argv[1:5] + ["3840", "2160", "64", "4"][len(sys.argv) - 1:]))


def bpp(S):
    t = 0
    for x in types:
        c = film.STAT_TYPES[x]
        planes = c["max_moment"] + (2 if c["transform"] else 0)
        t += 4 * c["channels"] * S + 2 * (4 + 4 * c["channels"] * planes)
    return t


def sysfs():
    out = []
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for f in ("pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_sclk", "pp_dpm_socclk"):
            try:
                cur = [l for l in open(os.path.join(card, f)).read().splitlines() if l.endswith("*")]
                out.append("%s %s" % (f[7:], cur[0].split(":")[1].strip(" *") if cur else "?"))
            except OSError:
                pass
        for p in glob.glob(os.path.join(card, "hwmon/hwmon*/power1_average")) + glob.glob(os.path.join(card, "hwmon/hwmon*/power1_input")):
            try:
                out.append("power %.0f W" % (int(open(p).read()) / 1e6))
            except OSError:
                pass
        break
    return ", ".join(out) if out else "sysfs clocks not readable"


def arena(S):
    out = {}
    for t in types:
        a = torch.empty((S, H, W, synthetic.CHANNELS[t]), device=dev)
        for s0 in range(0, S, 16):
            a[s0:s0 + 16].uniform_()
        out[t] = a
    return out


def series(tag, fs, a, secs):
    part = {t: v[:S] for t, v in a.items()}
    t_end = time.time() + secs
    k = 0
    while time.time() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            fs.accumulate(part)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 8
        if k < 6 or k % 16 == 0:
            print("%-38s #%4d  %.3f ms  %.2f TB/s   [%s]  ptr %x" % (tag, k, ms, bpp(S) * W * H / ms / 1e9, sysfs(), a["radiance"].data_ptr()), flush=True)
        k += 1


print("film %dx%d, %d spp; process start" % (W, H, S), flush=True)
fs = film.FilmStats(W, H, dev, types=types)
a1 = arena(S)
series("arena 1 (S planes), right after start", fs, a1, secs)
a2 = arena(4 * S)
series("arena 2 (4 S planes, first S used)", fs, a2, secs / 2)
series("arena 1 again", fs, a1, secs / 2)
del a2
torch.cuda.empty_cache()
a3 = arena(S)
series("arena 3 (S planes, allocated last)", fs, a3, secs / 2)
fs2 = film.FilmStats(W, H, dev, types=types)
series("arena 1, second film state", fs2, a1, secs / 2)
time.sleep(3)
series("arena 1, first film, after 3 s idle", fs, a1, secs / 2)
