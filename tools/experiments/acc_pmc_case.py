"""Round 5: the workload of acc_pmc.sh -- three launches of the film-major accumulation per shape, in this order:
1080p / 256 spp, 1080p / 64 spp, 4K / 64 spp, 4K / 16 spp (all stat types; uniform samples).  Prints the grid of every
shape so that the counter rows (same kernel name) can be told apart."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic

dev = torch.device("cuda:0")
api.setup(0)
types = list(synthetic.FEATURES)
for W, H, S in ((1920, 1080, 256), (1920, 1080, 64), (3840, 2160, 64), (3840, 2160, 16)):
    fs = film.FilmStats(W, H, dev, types=types)
    a = {}
    for t in types:
        x = torch.empty((S, H, W, synthetic.CHANNELS[t]), device=dev)
        for s0 in range(0, S, 16):
            x[s0:s0 + 16].uniform_()
        a[t] = x
    for _ in range(3):
        fs.accumulate(a)
    torch.cuda.synchronize()
    print("shape %dx%d S=%d done" % (W, H, S), flush=True)
    del fs, a, x
    torch.cuda.empty_cache()
