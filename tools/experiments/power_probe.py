"""Scratch: socket power and shader clock while (a) the filter, (b) the accumulation, (c) the bench
step loop over and over (sysfs hwmon, sampled every ~20 ms from a second thread)."""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
def find(name):
    c = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/" + name))
    return c[0] if c else None
P = find("power1_average") or find("power1_input"); F = find("freq1_input")
cap = find("power1_cap")
print("power node", P, "freq node", F, "cap", open(cap).read().strip() if cap else None)
def rd(p):
    try: return float(open(p).read())
    except Exception: return float("nan")
W, H, S = 1920, 1080, 256
sc = synthetic.Scene(W, H, seed=1, device=dev)
chunks = [sc.samples(32, seed=10 + i, features=synthetic.FEATURES) for i in range(S // 32)]
smp = {t: torch.cat([c[t] for c in chunks]) for t in synthetic.FEATURES}
del chunks
fs = film.FilmStats(W, H, dev, types=synthetic.FEATURES)
fs.accumulate(smp); fs.prepass()
def loop(fn, seconds):
    samples, stop = [], [False]
    def poll():
        while not stop[0]:
            samples.append((rd(P) / 1e6 if P else float("nan"), rd(F) / 1e6 if F else float("nan")))
            time.sleep(0.02)
    th = threading.Thread(target=poll); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n += 20
    dt = time.time() - t0
    stop[0] = True; th.join()
    s = samples[len(samples) // 3:]
    pw = sum(x[0] for x in s) / len(s); fq = sum(x[1] for x in s) / len(s)
    return dt / n * 1e3, pw, fq, max(x[0] for x in s)
def step():
    fs.accumulate(smp); fs.prepass(); fs.window_filter()
for name, fn in (("filter only", fs.window_filter), ("accumulate only", lambda: fs.accumulate(smp)), ("bench step", step)):
    ms, pw, fq, pmax = loop(fn, 4.0)
    print("%-16s %.3f ms per call | socket power avg %.0f W (max %.0f) | sclk avg %.0f MHz" % (name, ms, pw, pmax, fq))
    fs.reset()
    fs.accumulate(smp); fs.prepass()
