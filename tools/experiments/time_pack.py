"""Scratch: cost of the pack / unpack copies around the halo exchange (no communication)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, pipeline, sharding
dev = torch.device("cuda:0"); api.setup(0)
L = sharding.BlockLayout(5, 8, 1920, 1080, 20)   # interior block of a 4x2 grid: halos on 3 sides
sharding.exchange_halo = lambda *a, **k: None
pipe = pipeline.BlockPipeline(L, dev, ("radiance", "normal", "albedo"))
def wall(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("pack (HIP kernel, no exchange): %.3f ms" % wall(pipe.exchange))
print("filter on padded block (ROI): %.3f ms" % wall(pipe.window_filter, 5))
one = pipeline.BlockPipeline(sharding.BlockLayout(0, 1, 1920, 1080, 20), dev, ("radiance", "normal", "albedo"))
print("filter on plain block: %.3f ms" % wall(one.window_filter, 5))
