"""One rank's share of a strong-scaling step, measured on one GPU: the block a middle rank of an N-rank row-strip
grid owns (accumulate -> pre-pass + pack -> window filter; the halo exchange itself is left out, the halo holds
zeros).  Shows what the per-rank step costs next to 1/N of the whole-film step.
python tools/experiments/block_step.py [spp]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, pipeline, sharding, synthetic

FW, FH = int(os.environ.get("FW", 1920)), int(os.environ.get("FH", 1080))
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
api.setup(0)
types = list(synthetic.FEATURES)


def ev():
    return torch.cuda.Event(enable_timing=True)


base = None
res, detail = {}, {}
for world in (1, 2, 4, 8):
    grid = sharding.row_strips(world)
    W, H = FW // grid[0], FH // grid[1]
    rank = world // 2
    L = sharding.BlockLayout(rank, world, W, H, 20, grid=grid)
    ox, oy = L.origin
    scene = synthetic.Scene(W, H, n_regions=12, seed=1, device=dev, x_offset=ox, y_offset=oy, full_width=FW, full_height=FH)
    parts = [scene.samples(32, seed=1000 + s0, features=types) for s0 in range(0, spp, 32)]
    samples = {t: torch.cat([p[t] for p in parts], dim=0) for t in types}
    del parts
    pipe = pipeline.BlockPipeline(L, dev, types)

    # the order bench.py --gpus N runs on row strips: border rows first, (the exchange started), the rest, the filter
    border, interior = pipe.border_rows(), pipe.interior_rows()

    def step():
        if border:
            in_flight = pipe.border_first(samples, exchange=False)       # (world of its own here: nothing is sent)
            pipe.interior_beside(samples)
            pipe.join_interior(in_flight)
        else:
            pipe.accumulate(samples)
            pipe.prepass()
        pipe.window_filter()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    host = (time.perf_counter() - t0) / reps * 1e3       # time the host needs to issue a step
    torch.cuda.synchronize()
    e = [ev() for _ in range(4)]
    acc = pre = flt = 0.0
    for _ in range(10):
        e[0].record(); pipe.accumulate(samples); e[1].record(); pipe.prepass(); e[2].record(); pipe.window_filter(); e[3].record()
        torch.cuda.synchronize()
        acc += e[0].elapsed_time(e[1]) / 10; pre += e[1].elapsed_time(e[2]) / 10; flt += e[2].elapsed_time(e[3]) / 10
    if base is None:
        base = wall
    print("N=%d block %dx%d (+halo %dx%d): step %.3f ms (1/N of N=1: %.3f, efficiency %.2f)  host issue %.3f ms | accumulate %.3f  prepass %.3f  filter %.3f  parts %d"
          % (world, W, H, L.pw, L.ph, wall, base / world, base / world / wall, host, acc, pre, flt,
             api.load().statmc_debug_last_filter_parts()), flush=True)
    res[str(world)] = round(wall, 4)
    detail[str(world)] = {"accumulate_ms": round(acc, 4), "prepass_ms": round(pre, 4), "filter_ms": round(flt, 4), "host_issue_ms": round(host, 4),
                          "block": "%dx%d" % (W, H), "filter_parts": api.load().statmc_debug_last_filter_parts()}
    del pipe, samples
    torch.cuda.empty_cache()

import json
out = {"per_rank_step_ms": res, "detail": detail, "spp": spp, "film": "%dx%d" % (FW, FH),
       "source": "tools/experiments/block_step.py on one MI355X: the middle rank's block of an N-strip grid in the order bench.py runs (border rows first, then the rest, then the filter; detail: the one-piece kernels), halo exchange left out"}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/block_step.json", "w"), indent=1)
