import sys, torch
sys.path.insert(0, '/root/repo')
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0"); api.setup(0)
free, total = torch.cuda.mem_get_info()
hog = torch.empty(int((free - (40 << 30)) // 4), dtype=torch.float32, device=dev)     # leave 40 GiB
print("free after the hog: %.1f GiB" % (torch.cuda.mem_get_info()[0] / 2**30), flush=True)
types = list(synthetic.FEATURES)
fs = film.FilmStats(1920, 1080, dev, types=types, placed=True)
S = 256
a = {}
try:
    for t in types:
        a[t] = api.empty_placed((S, 1080, 1920, synthetic.CHANNELS[t]), torch.float32, dev, api.MEM_STREAM)
        a[t].zero_()
    fs.accumulate(a); torch.cuda.synchronize()
    print("ok under pressure", flush=True)
except Exception as e:
    print("error:", type(e).__name__, str(e)[:300], flush=True)
i = api.placement_info(); print({k: i[k] for k in i if k != "map"}); print(i["map"])
print("free at the end: %.1f GiB" % (torch.cuda.mem_get_info()[0] / 2**30))
