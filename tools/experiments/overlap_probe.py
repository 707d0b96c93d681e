"""Scratch: do accumulate (HBM-bound) and window filter (VALU-bound) overlap on two streams?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
W, H, S = 1920, 1080, int(os.environ.get('S', '256'))
sc = synthetic.Scene(W, H, seed=1, device=dev)
chunks = [sc.samples(32, seed=10 + i) for i in range(S // 32)]
smp = {t: torch.cat([c[t] for c in chunks]) for t in synthetic.FEATURES}
fs = film.FilmStats(W, H, dev, types=synthetic.FEATURES)
fs.accumulate(smp); fs.prepass()
# snapshot of the filter inputs (what the pipelined bench would filter while the next batch accumulates)
snap = dict(colour=fs.state["radiance"]["film_mean"].clone(), normal=fs.g_buffer("normal").clone(),
            albedo=fs.g_buffer("albedo").clone(), mc=fs.mean_corr.clone(), dc=fs.disc.clone())
out = torch.zeros_like(snap["colour"])
def run_filter():
    a, keep = api.make_filter_args([], [], [], [], [snap["colour"]], [snap["mc"]], [snap["dc"]], [out],
                                   [snap["normal"], snap["albedo"]], g_sds=[0.1, 0.02])
    api.window_filter(a, 3)
def wall(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
import sys as _s
for rb in (0, 256, 512, 1024):
    api.accumulate_resident_blocks(rb)
    print("resident_blocks=%d: accumulate alone %.3f ms" % (rb, wall(lambda: fs.accumulate(smp))))
api.accumulate_resident_blocks(int(_s.argv[1]) if len(_s.argv) > 1 else 256)
t_a = wall(lambda: fs.accumulate(smp)); t_f = wall(run_filter)
print("accumulate %.3f ms, filter %.3f ms, sequential sum %.3f" % (t_a, t_f, t_a + t_f))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        fs.accumulate(smp)
    with torch.cuda.stream(s2):
        run_filter()
    cur.wait_stream(s1); cur.wait_stream(s2)
print("concurrent (2 streams): %.3f ms" % wall(both))
def both_rev():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        run_filter()
    with torch.cuda.stream(s1):
        fs.accumulate(smp)
    cur.wait_stream(s1); cur.wait_stream(s2)
print("concurrent, filter launched first: %.3f ms" % wall(both_rev))
