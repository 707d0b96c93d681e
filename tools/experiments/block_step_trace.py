"""One rank's overlapped step at N = 8 (middle rank of a row-strip grid) for a rocprofv3 --kernel-trace timeline.
rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/experiments/block_step_trace.py [side]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, pipeline, sharding, synthetic
side = len(sys.argv) > 1 and sys.argv[1] == "side"
dev = torch.device("cuda:0"); api.setup(0)
types = list(synthetic.FEATURES)
world = 8
L = sharding.BlockLayout(world // 2, world, 1920, 135, 20, grid=sharding.row_strips(world))
ox, oy = L.origin
scene = synthetic.Scene(1920, 135, n_regions=12, seed=1, device=dev, x_offset=ox, y_offset=oy, full_width=1920, full_height=1080)
samples = {t: torch.cat([scene.samples(32, seed=1000 + s0, features=types)[t] for s0 in range(0, 256, 32)]) for t in types}
pipe = pipeline.BlockPipeline(L, dev, types)
border, interior = pipe.border_rows(), pipe.interior_rows()
for _ in range(12):
    if side:
        in_flight = pipe.border_first(samples, exchange=False)
        pipe.interior_beside(samples)
        pipe.join_interior(in_flight)
    else:
        pipe.accumulate(samples, rows=border)
        pipe.accumulate(samples, rows=interior)
        pipe.prepass(rows=border)
        pipe.prepass(rows=interior)
    pipe.window_filter()
torch.cuda.synchronize()
