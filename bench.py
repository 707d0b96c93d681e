#!/usr/bin/env python3
"""bench.py -- denoised Mpixels/s of the StatMC statistics hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input, per GPU:
    accumulate   S samples/pixel x 11 channels (radiance RGB, normal, albedo, depth, material id)
                 into the running moments          (StatTile::Add*Sample* + Merge*Tile)
    pre-pass     (n, mean, m2, m3) -> Johnson-corrected mean + discriminator
    [halo]       N > 1 only: r-pixel border of the 5 filter inputs from the neighbour blocks (RCCL)
    filter       (2r+1)^2 statistics-gated cross-bilateral window over the colour image
with every input already resident in HBM when the timed region starts.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 1920x1080 film,
256 spp, 11-channel sample stream, shipped filter parameters (filtersd 10, filterradius 20,
normal sd 0.1, albedo sd 0.02).  For N > 1 every rank owns one 1920x1080 block of an
N-block film (1xN row strips by default, --grid blocks for 2x1 / 2x2 / 4x2) -- weak scaling, value = all blocks' pixels / step time.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TOPS = 78.6          # 10^12 fp32 lane-ops/s: 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz
FILTER_BYTES_PER_PX = 72       # 60 B in (5 float3 images) + 12 B out
PREPASS_BYTES_PER_PX = 64      # 40 B in + 24 B out


def accumulate_bytes_per_px(spp, types):
    from statmc_amd.film import STAT_TYPES
    total = 0
    for t in types:
        cfg = STAT_TYPES[t]
        c = cfg["channels"]
        planes = {1: 1, 2: 2, 3: 3}[cfg["max_moment"]] + (2 if cfg["transform"] else 0)
        total += 4 * c * spp + 2 * (4 + 4 * c * planes)   # samples + RMW of n and the moment planes
    return total


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--radius", type=int, default=20)
    ap.add_argument("--filtersd", type=float, default=10.0)
    ap.add_argument("--channels", type=int, default=11, choices=(9, 11))
    ap.add_argument("--grid", default="rows", choices=("rows", "blocks"),
                    help="N > 1: rows = 1xN strips of full-width blocks (contiguous halo rows, default); "
                         "blocks = 2x1 / 2x2 / 4x2 grid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="gloo (+ --share-device) exercises the N > 1 code path on a 1-GPU box; halos go via the host")
    ap.add_argument("--share-device", action="store_true", help="debug: every rank uses cuda:0")
    ap.add_argument("--cpu-acc-rows", type=int, default=32, help="rows of the film the CPU baseline accumulates")
    return ap.parse_args()


def cpu_baseline(args, fs, samples, types, budget_s=7.0):
    """Times the CPU oracle (the restated reference algorithm, OpenMP over tiles / rows, all host
    cores) on a bounded sample of the same workload: each leg is sized from a short probe so that
    it does about `budget_s` seconds of work.  Reported, never used by the GPU path."""
    import numpy as np
    from oracle import oracle
    from statmc_amd.film import STAT_TYPES
    W, H, S = args.width, args.height, args.spp
    cores = oracle.num_threads()

    # ---- accumulate: a strip of rows, all spp, all channels, repeated on fresh state
    ar = min(args.cpu_acc_rows, H)
    y0 = (H - ar) // 2
    host = {t: samples[t][:, y0:y0 + ar].contiguous().cpu().numpy() for t in types}

    def acc_once():
        t0 = time.perf_counter()
        for t in types:
            st = oracle.new_state(ar, W, STAT_TYPES[t]["channels"])
            oracle.accumulate(st, host[t], STAT_TYPES[t]["transform"], STAT_TYPES[t]["max_moment"])
        return time.perf_counter() - t0

    acc_once()  # page in / warm the thread pool
    t_acc, acc_reps = 0.0, 0
    while t_acc < budget_s and acc_reps < 200:
        t_acc += acc_once()
        acc_reps += 1
    acc_s_per_px = t_acc / (acc_reps * ar * W)

    # ---- pre-pass (full frame) + filter (rows sized from a 8-row probe, full window)
    rad = {k: v.cpu().numpy() for k, v in fs.state["radiance"].items() if v is not None}
    gb = [fs.g_buffer(g).cpu().numpy() for g in fs.g_names]
    g_dr = [-0.5 / (sd * sd) for sd in fs.g_sds]
    ds = -0.5 / (args.filtersd ** 2)
    t0 = time.perf_counter()
    mc, dc = oracle.prepass(rad["n"], rad["mean"], rad["m2"], rad["m3"])
    t_pre = time.perf_counter() - t0

    def flt(rows):
        fy0 = (H - rows) // 2
        t0 = time.perf_counter()
        oracle.filter_image(mc, dc, rad["film_mean"], gb, g_dr, ds, args.radius, roi=(0, fy0, W, fy0 + rows))
        return time.perf_counter() - t0

    probe_rows = min(max(cores // 4, 8), H)
    t_probe = flt(probe_rows)
    fr = int(min(H, max(probe_rows, budget_s / max(t_probe / probe_rows, 1e-9))))
    t_flt, flt_reps = 0.0, 0
    while t_flt < budget_s and flt_reps < 50:
        t_flt += flt(fr)
        flt_reps += 1
    flt_s_per_px = t_flt / (flt_reps * fr * W)
    s_per_px = acc_s_per_px + t_pre / (W * H) + flt_s_per_px

    # ---- the same two legs on ONE thread (small strips), for the single-core figure
    ar1 = min(2, ar)
    t0 = time.perf_counter()
    for t in types:
        st = oracle.new_state(ar1, W, STAT_TYPES[t]["channels"])
        oracle.accumulate(st, np.ascontiguousarray(host[t][:, :ar1]), STAT_TYPES[t]["transform"], STAT_TYPES[t]["max_moment"],
                          threads=1)
    acc1 = (time.perf_counter() - t0) / (ar1 * W)
    fy1 = H // 2
    t0 = time.perf_counter()
    oracle.filter_image(mc, dc, rad["film_mean"], gb, g_dr, ds, args.radius, roi=(0, fy1, W, fy1 + 2), threads=1)
    flt1 = (time.perf_counter() - t0) / (2 * W)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    return {
        "value": round(1e-6 / s_per_px, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
        "cpu_model": cpu_model, "host_logical_cpus": os.cpu_count(),
        "sample": "oracle (C restatement of the reference algorithm, OpenMP, %d threads): accumulate %d rows x %d px "
                  "x %d spp x %d ch, %d repetitions (%.1f s); pre-pass full frame (%.2f s); filter %d rows x %d px, "
                  "full %dx%d window, %d repetitions (%.1f s); per-pixel times summed and inverted"
                  % (cores, ar, W, S, args.channels, acc_reps, t_acc, t_pre, fr, W, 2 * args.radius + 1,
                     2 * args.radius + 1, flt_reps, t_flt),
        "accumulate_s_per_mpx": round(acc_s_per_px * 1e6, 4),
        "filter_s_per_mpx": round(flt_s_per_px * 1e6, 4),
        "single_thread": {"value": round(1e-6 / (acc1 + t_pre / (W * H) + flt1), 5), "unit": "Mpixels/s", "cores": 1,
                          "sample": "accumulate %d rows, filter 2 rows, one thread" % ar1},
    }


def host_copy_times(fs, dev):
    """What the reference's `CUDA time` bracket adds when statistics are produced on the host
    (statpath.cpp:409-417): Upload() of the 7 filter inputs (76 B/px) and Download() of film-f
    (12 B/px), pinned host memory, through statmc_upload / statmc_download.  Never part of `value`."""
    import ctypes as C
    from statmc_amd import api
    lib = api.load()
    rad = fs.state["radiance"]
    ups = [rad["film_mean"], rad["n"], rad["mean"], rad["m2"], rad["m3"], fs.g_buffer("normal"), fs.g_buffer("albedo")]
    host_up = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in ups]
    host_dn = torch.empty(fs.film_f.shape, dtype=torch.float32).pin_memory()
    stream = api.current_stream_handle()

    def upload():
        for h, d in zip(host_up, ups):
            api.check(lib.statmc_upload(C.c_void_p(d.data_ptr()), C.c_void_p(h.data_ptr()), h.numel() * 4, stream))

    def download():
        api.check(lib.statmc_download(C.c_void_p(host_dn.data_ptr()), C.c_void_p(fs.film_f.data_ptr()), host_dn.numel() * 4, stream))

    out = {}
    for name, fn, nbytes in (("upload", upload, sum(h.numel() * 4 for h in host_up)), ("download", download, host_dn.numel() * 4)):
        for h in host_up:
            h.zero_()
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        out[name + "_ms"] = round(ms, 3)
        out[name + "_GBs"] = round(nbytes / ms / 1e6, 1)
        out[name + "_bytes_per_px"] = nbytes // (fs.width * fs.height)
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run "
                             "--nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU (no CPU fallback exists for the product path)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from statmc_amd import api, film, pipeline, sharding, synthetic
    api.setup(local_rank)

    W, H, S, r = args.width, args.height, args.spp, args.radius
    types = list(synthetic.FEATURES) if args.channels == 11 else ["radiance", "normal", "albedo"]
    grid = sharding.row_strips(world) if args.grid == "rows" else sharding.grid_for(world)
    layout = sharding.BlockLayout(rank, world, W, H, r, grid=grid)
    fw, fh = layout.film_size
    ox, oy = layout.origin

    # ---- synthetic inputs, generated in place in HBM (seeded; same generator as the tests)
    scene = synthetic.Scene(W, H, n_regions=min(12 * world, 32), seed=1, device=dev, x_offset=ox, y_offset=oy,
                            full_width=fw, full_height=fh)
    samples = {t: [] for t in types}
    chunk = 32
    for s0 in range(0, S, chunk):
        part = scene.samples(min(chunk, S - s0), seed=1000 * (rank + 1) + s0, features=types)
        for t in types:
            samples[t].append(part[t])
    samples = {t: torch.cat(v, dim=0) for t, v in samples.items()}
    pipe = pipeline.BlockPipeline(layout, dev, types, filter_sd=args.filtersd, radius=r,
                                  via_host=args.backend == "gloo")
    fs = pipe.fs

    ev = lambda: torch.cuda.Event(enable_timing=True)
    k_events = {"accumulate": [], "prepass": [], "halo": [], "filter": []}

    def step(record):
        e = [ev() for _ in range(5)] if record else None
        if record: e[0].record()
        pipe.accumulate(samples)
        if record: e[1].record()
        pipe.prepass()
        if record: e[2].record()
        if world > 1:
            pipe.exchange()
        if record: e[3].record()
        pipe.window_filter()
        if record:
            e[4].record()
            for name, i in (("accumulate", 0), ("prepass", 1), ("halo", 2), ("filter", 3)):
                k_events[name].append((e[i], e[i + 1]))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    variant = api.last_filter_variant()
    ms = {k: (sum(a.elapsed_time(b) for a, b in v) / max(len(v), 1)) for k, v in k_events.items()}
    px_block = W * H
    ms_per_step = elapsed * 1e3 / args.steps
    value = world * px_block * args.steps / elapsed / 1e6

    result = None
    if rank == 0:
        flt_gbs = FILTER_BYTES_PER_PX * px_block / (ms["filter"] * 1e-3) / 1e9
        acc_bpp = accumulate_bytes_per_px(S, types)
        acc_gbs = acc_bpp * px_block / (ms["accumulate"] * 1e-3) / 1e9
        pre_gbs = PREPASS_BYTES_PER_PX * px_block / (ms["prepass"] * 1e-3) / 1e9
        taps = (2 * r + 1) ** 2
        # fp32 VALU work of the window filter: 20 instructions per (tap, pixel) pair, 9 of them
        # packed (v_pk_*_f32 = 2 issue slots) and one v_exp_f32 (quarter rate): 33 slots
        valu_slots = 33.0 * taps * px_block / 64.0
        valu_rate = valu_slots / (ms["filter"] * 1e-3) / 1e12
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        # the committed PMC traffic figures were collected on the default workload only
        default_cfg = (W, H, S, args.channels, r) == (1920, 1080, 256, 11, 20)
        if default_cfg and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath))
            except Exception:
                traffic = {}
        result = {
            "metric": "denoised_mpixels_per_s", "value": round(value, 3), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "StatMC accumulate+prepass+filter, %dx%d block/GPU, %d spp, %d-channel samples, "
                            "filterradius %d, filtersd %g, G-buffers normal(sd 0.1)+albedo(sd 0.02) "
                            "[BASELINE.json configs[2] shape, synthetic stream]" % (W, H, S, args.channels, r, args.filtersd),
                "film": "%dx%d" % (fw, fh), "block_grid": "%dx%d" % (layout.gx, layout.gy),
                "spp": S, "sample_channels": args.channels, "filter_variant": variant,
                "parallelism": "film blocks x%d, RCCL halo exchange" % world if world > 1 else "single GPU",
            },
            # the kernel that dominates the step: the sample-stream accumulation (HBM-bound)
            "roofline": {
                "kernel": "accumulate_kernel", "bound": "hbm",
                "achieved": round(acc_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(acc_gbs / HBM_PEAK_GBS, 4),
                "traffic": traffic.get("accumulate_kernel"),
                "algorithmic_bytes_per_launch": acc_bpp * px_block, "bytes_per_px": acc_bpp,
                "avg_launch_ms": round(ms["accumulate"], 4),
            },
            # the kernel BASELINE.json's metric names: HBM GB/s of the window filter.  It is a
            # (2r+1)^2-tap fp32 stencil -- bound by VALU issue, not by HBM -- so its fraction of the fp32
            # VALU peak is given next to the (necessarily small) HBM fraction.
            "roofline_filter": {
                "kernel": "window_filter_lds<%d> + combine_parts_kernel (%s)" % (r, variant), "bound": "hbm",
                "achieved": round(flt_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(flt_gbs / HBM_PEAK_GBS, 5),
                "traffic": traffic.get("window_filter_lds"),
                "algorithmic_bytes_per_launch": FILTER_BYTES_PER_PX * px_block, "bytes_per_px": FILTER_BYTES_PER_PX,
                "avg_launch_ms": round(ms["filter"], 4),
                "valu": {"achieved": round(valu_rate, 2), "peak": VALU_PEAK_TOPS, "unit": "T wave-slots x64 lanes/s",
                         "frac": round(valu_rate / VALU_PEAK_TOPS * 64.0, 4),
                         "note": "33 issue slots per (tap, pixel) pair; peak = 256 CU x 4 SIMD x 1 slot / 2 cycles x 2.4 GHz x 64 lanes"},
            },
            "kernels": {
                "accumulate": {"avg_ms": round(ms["accumulate"], 4), "bytes_per_px": acc_bpp,
                               "achieved_GBs": round(acc_gbs, 1), "frac_hbm": round(acc_gbs / HBM_PEAK_GBS, 4)},
                "prepass": {"avg_ms": round(ms["prepass"], 4), "bytes_per_px": PREPASS_BYTES_PER_PX,
                            "achieved_GBs": round(pre_gbs, 1), "frac_hbm": round(pre_gbs / HBM_PEAK_GBS, 4)},
                "halo_exchange": {"avg_ms": round(ms["halo"], 4)},
                "filter": {"avg_ms": round(ms["filter"], 4), "mpixels_per_s": round(px_block / ms["filter"] / 1e3, 2)},
            },
        }
        if world == 1:
            result["host_copies"] = host_copy_times(fs, dev)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args, fs, samples, types)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
