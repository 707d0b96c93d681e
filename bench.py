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
normal sd 0.1, albedo sd 0.02).  For N > 1 the SAME film is cut into N blocks (1xN row strips by
default -- 1920x540, 1920x270, 1920x135 -- or --grid blocks for 2x1 / 2x2 / 4x2: 960x1080, 960x540,
480x540), one per rank: strong scaling, value = film pixels / step time ("1920x1080 @1/2/4/8 GPU").
--film 3840x2160 is configs[4]'s film; --scaling weak gives every rank its own --film-sized block instead.

Two ways to run N > 1, same step order, same JSON line (`backend` says which):
    --backend nccl   one process per GPU (torch.distributed; the driver's launch form, or self-launched), halo rows by
                     RCCL send/recv.  The bring-up is checked under a watchdog; if the leg gives no result, a FRESH
                     process runs the peer leg and the line carries `fallback_from` / `nccl_error`.
    --backend peer   ONE process drives all N devices through the C ABI: statmc_accumulate_row_ranges ->
                     statmc_prepass_pack_rows -> statmc_halo_exchange (device-to-device copies, peer access over
                     xGMI) -> statmc_window_filter.  No torch.distributed, no RCCL.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

torch = None   # imported by main() AFTER the launch decision: the parent of a self-launched N > 1 run never touches the GPU
dist = None

PLACED = {"on": False, "error": None}   # set by main(): buffers of the timed step come from statmc_malloc_placed
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


def new_film_stats(W, H, dev, types, **kw):
    """FilmStats for a secondary leg, its moments placed like the timed step's."""
    from statmc_amd import film
    return film.FilmStats(W, H, dev, types=types, placed=PLACED["on"], **kw)


def new_arenas(S, H, W, dev, types):
    """One sample arena per stat type, their total announced first (statmc_placement_expect: ONE class is searched for all of them)."""
    from statmc_amd import api, synthetic
    if PLACED["on"]:
        try:
            api.placement_expect(api.MEM_STREAM, sum(4 * S * H * W * synthetic.CHANNELS[t] for t in types), dev)
        except Exception:      # noqa: BLE001
            pass
    try:
        return {t: new_arena((S, H, W, synthetic.CHANNELS[t]), dev) for t in types}
    finally:
        if PLACED["on"]:
            try:
                api.placement_expect(api.MEM_STREAM, 0, dev)
            except Exception:      # noqa: BLE001
                pass


def new_arena(shape, dev):
    """A sample arena (float32), placed like the timed step's."""
    from statmc_amd import api
    if PLACED["on"]:
        return api.empty_placed(shape, torch.float32, dev, api.MEM_STREAM)
    return torch.empty(shape, dtype=torch.float32, device=dev)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="default: about 2 s of timed work on one GPU")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--film", default=None, help="WxH of the film (default 1920x1080; 3840x2160 = BASELINE configs[4])")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-overlap-halo", dest="overlap_halo", action="store_false",
                    help="N > 1: accumulate the whole block, then exchange the halo (default: border rows first, the exchange behind the rest of the accumulation)")
    ap.add_argument("--scaling", default="strong", choices=("strong", "weak"),
                    help="N > 1: strong = one film cut into N blocks (default); weak = one film-sized block per rank")
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--radius", type=int, default=20)
    ap.add_argument("--filtersd", type=float, default=10.0)
    ap.add_argument("--channels", type=int, default=11, choices=(9, 11))
    ap.add_argument("--grid", default="rows", choices=("rows", "blocks"),
                    help="N > 1: rows = 1xN strips of full-width blocks (contiguous halo rows, default); "
                         "blocks = 2x1 / 2x2 / 4x2 grid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo", "peer"),
                    help="N > 1: nccl = one process per GPU, halo exchange over torch.distributed / RCCL (default; if it gives no "
                         "result, the peer leg runs in a fresh process and the line says so); peer = ONE process drives all N devices "
                         "through the C ABI, statmc_halo_exchange's device-to-device copies -- no torch.distributed, no RCCL; "
                         "gloo (+ --share-device) exercises the per-rank code path on a 1-GPU box, halos via the host")
    ap.add_argument("--no-fallback", action="store_true", help="nccl: do not fall back to the peer leg")
    ap.add_argument("--run-timeout", type=int, default=600, help="nccl ranks under a foreign launcher: seconds after the bring-up before a "
                    "run that has not printed its line is given up (the ranks leave, rank 0 runs the peer leg in a fresh process)")
    ap.add_argument("--rank-timeout", type=int, default=900, help="self-launched ranks: seconds before the leg is ended (and the peer leg tried)")
    ap.add_argument("--bringup-timeout", type=int, default=150, help="seconds the process-group bring-up (init, first all-reduce, first "
                                                                      "neighbour exchange) may take before it counts as hung")
    ap.add_argument("--dump-film-f", default=None, help="write film-f of a fresh run (empty statistics, three steps) to this .npy (tests: legs against each other)")
    ap.add_argument("--share-device", action="store_true", help="debug: every rank uses cuda:0")
    ap.add_argument("--cpu-acc-rows", type=int, default=32, help="rows of the film the CPU baseline accumulates")
    ap.add_argument("--no-host-legs", action="store_true", help="skip the secondary host-side measurements (N = 1)")
    ap.add_argument("--feed", choices=("film", "tiles"), default="film",
                    help="how the timed step's samples reach the accumulation: film-major planes through statmc_accumulate (default), or "
                         "16 x 16 tile blocks through statmc_accumulate_tiles, the way Render<T> hands them over (N = 1, --schedule single)")
    ap.add_argument("--schedule", default="single", choices=("single", "reference"),
                    help="single = one accumulate of --spp samples + one pre-pass + filter per step (default); reference = the "
                         "reference's progressive schedule (statpath.cpp:272-279): iterations of 4, 4, 8, 16, ... samples up to "
                         "--spp, the denoiser after every iteration, statistics reset at the start of a step")
    ap.add_argument("--pool-spp", type=int, default=0,
                    help="samples per pixel kept resident in HBM (0 = all of --spp if they fit in 60 %% of the free memory, else "
                         "what fits); sample s of a step is pool sample s mod pool: every sample is still read from HBM")
    ap.add_argument("--gather", action="store_true",
                    help="N > 1: assemble film-f on rank 0 inside every step (default: measured after the timed region as gather_ms)")
    ap.add_argument("--no-bind", action="store_true", help="do not bind the rank to the CPUs of its GPU's NUMA node")
    ap.add_argument("--separate-prepass", dest="fused_prepass", action="store_false",
                    help="N = 1: run the pre-pass as a launch of its own (rounds 1 - 5; default: the accumulation's epilogue writes mean-corr and "
                         "discriminator from the registers that hold the new moments -- statmc_stat_type::mean_corr, the same bits)")
    ap.add_argument("--no-placement", dest="placement", action="store_false",
                    help="running moments and sample arenas from torch's allocator (default: statmc_malloc_placed -- the moments in one "
                         "interference class of the card's memory, the arenas in another: include/statmc.h)")
    ap.add_argument("--no-placement-check", dest="placement_check", action="store_false",
                    help="N = 1: keep the placed buffers whatever they measure (default: before the warm-up the step is run a few times on the "
                         "placed buffers and on copies from torch's allocator, and the timed region uses the faster set -- a card whose GiB slots "
                         "mostly straddle interference classes gives the allocator nothing to choose from; the line says which set ran and both times)")
    args = ap.parse_args()
    if args.film:
        args.width, args.height = (int(v) for v in args.film.lower().split("x"))
    return args


def cpu_share():
    """The CPUs this process may use: the affinity mask, and the cgroup's CPU quota where one is set (a GPU box of the pool
    shows 256 logical CPUs in the mask and a quota of 16: threads beyond the quota are throttled, which is what made 128
    OpenMP threads deliver 3.3 x one thread in rounds 3 and 4)."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]           # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except Exception:      # noqa: BLE001
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())       # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:  # noqa: BLE001
            quota = None
    usable = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return {"affinity_cpus": aff, "cgroup_quota_cpus": quota, "usable_cpus": usable}


def cpu_baseline(args, fs, samples, types, budget_s=6.0):
    """Times the CPU oracle (the restated reference algorithm, OpenMP over tiles / rows) on a bounded sample of the same
    workload, on as many threads as the box gives this process (cpu_share): each leg is sized from a short probe so that it
    does about `budget_s` seconds of work.  The accumulation reads its samples the way the reference's render loop produces
    them -- tile after tile, pixel after pixel, a pixel's samples one after the other (oracle_accumulate_tile_stream) -- and
    not as film-major planes, which cost a CPU a cache miss per sample.  Reported, never used by the GPU path."""
    import numpy as np
    from oracle import oracle
    from statmc_amd.film import STAT_TYPES
    W, H = args.width, args.height
    S = next(iter(samples.values())).shape[0]          # the resident samples (all of --spp unless a pool was needed)
    # the binding to the GPU's NUMA node (bind_to_gpu_numa) is for the GPU path: the CPU legs get every CPU the box allows
    mask_before = os.sched_getaffinity(0)
    affinity_before = len(mask_before)
    try:
        os.sched_setaffinity(0, range(os.cpu_count() or 1))
    except Exception:      # noqa: BLE001
        pass
    try:
        return _cpu_baseline_legs(args, fs, samples, types, budget_s, affinity_before)
    finally:
        try:                # the host-side legs that follow run under the GPU's NUMA binding again
            os.sched_setaffinity(0, mask_before)
        except Exception:      # noqa: BLE001
            pass


def _cpu_baseline_legs(args, fs, samples, types, budget_s, affinity_before):
    import numpy as np
    from oracle import oracle
    from statmc_amd.film import STAT_TYPES
    W, H = args.width, args.height
    S = next(iter(samples.values())).shape[0]
    share = cpu_share()
    cores = min(share["usable_cpus"], oracle.num_threads())

    # ---- accumulate: a strip of >= 128 rows (>= 960 tiles at 1080p), all channels, a bounded number of samples per pixel
    # (the per-sample cost does not depend on the count), repeated on fresh state
    ar = min(max(args.cpu_acc_rows, 128), H)
    ar -= ar % 16 if ar >= 16 else 0
    y0 = (H - ar) // 2
    S_cpu = min(S, 64)
    host = {t: samples[t][:S_cpu, y0:y0 + ar].contiguous().cpu().numpy() for t in types}
    stream = {t: oracle.to_tile_major(host[t]) for t in types}

    def acc_once(threads):
        t0 = time.perf_counter()
        for t in types:
            st = oracle.new_state(ar, W, STAT_TYPES[t]["channels"])
            oracle.accumulate_tile_stream(st, stream[t], S_cpu, STAT_TYPES[t]["transform"], STAT_TYPES[t]["max_moment"], threads=threads)
        return time.perf_counter() - t0

    def acc_leg(threads, budget):
        acc_once(threads)  # page in / warm the thread pool
        t, reps = 0.0, 0
        while t < budget and reps < 200:
            t += acc_once(threads)
            reps += 1
        return t / (reps * ar * W) * (S / S_cpu), reps, t       # seconds per pixel at the step's sample count

    acc_s_per_px, acc_reps, t_acc = acc_leg(cores, budget_s)

    # ---- pre-pass (full frame) + filter (rows sized from a probe, full window)
    rad = {k: v.cpu().numpy() for k, v in fs.state["radiance"].items() if v is not None}
    gb = [fs.g_buffer(g).cpu().numpy() for g in fs.g_names]
    g_dr = [-0.5 / (sd * sd) for sd in fs.g_sds]
    ds = -0.5 / (args.filtersd ** 2)
    t0 = time.perf_counter()
    mc, dc = oracle.prepass(rad["n"], rad["mean"], rad["m2"], rad["m3"])
    t_pre = time.perf_counter() - t0

    last = {}

    def flt(rows, threads, keep=False):
        fy0 = (H - rows) // 2
        t0 = time.perf_counter()
        out = oracle.filter_image(mc, dc, rad["film_mean"], gb, g_dr, ds, args.radius, roi=(0, fy0, W, fy0 + rows), threads=threads)
        dt = time.perf_counter() - t0
        if keep:
            last["out"], last["rows"] = out, (fy0, fy0 + rows)
        return dt

    def flt_leg(threads, budget, keep=False):
        probe_rows = min(max(2 * threads, 8), H)          # every thread has rows to work on (one row = one work item)
        t_probe = flt(probe_rows, threads)
        fr = int(min(H, max(4 * threads, probe_rows, budget / max(t_probe / probe_rows, 1e-9))))
        t, reps = 0.0, 0
        while t < budget and reps < 50:
            t += flt(fr, threads, keep=keep)
            reps += 1
        return t / (reps * fr * W), reps, t, fr

    flt_s_per_px, flt_reps, t_flt, fr = flt_leg(cores, budget_s, keep=True)
    s_per_px = acc_s_per_px + t_pre / (W * H) + flt_s_per_px
    # The rows the oracle has just filtered, against what the HIP path left in film-f for the same statistics (the
    # checker at work, not the product: BASELINE's bound is 1e-5 relative L2 per channel; "oracle" = this build's own
    # restatement, whose filter half is parity-unpinned against the CUDA denoiser -- DESIGN.md section 2)
    parity = None
    try:
        y0, y1 = last["rows"]
        got = fs.film_f[y0:y1].cpu().numpy().astype(np.float64)
        ref = last["out"][y0:y1].astype(np.float64)
        errs = [float(np.sqrt(((got[..., c] - ref[..., c]) ** 2).sum() / max((ref[..., c] ** 2).sum(), 1e-300))) for c in range(3)]
        parity = {"rows": [y0, y1], "rel_l2_per_channel": [float("%.3e" % e) for e in errs], "bound": 1e-5,
                  "prepass_bit_exact": bool(np.array_equal(fs.mean_corr.cpu().numpy(), mc, equal_nan=True)
                                            and np.array_equal(fs.disc.cpu().numpy(), dc, equal_nan=True)),
                  "against": "oracle/ (CPU restatement; its filter half is this build's spec: parity with the CUDA denoiser unpinned)"}
    except Exception as e:      # noqa: BLE001
        parity = {"error": repr(e)[:200]}

    # ---- how the two legs scale with threads on this box (about 1.2 s per point and leg): the evidence behind `cores`
    scaling = []
    for th in [t for t in (1, 4, 16, 32, 64, 128) if t <= share["affinity_cpus"]]:
        if th == cores:
            a_s, f_s = acc_s_per_px, flt_s_per_px
        else:
            a_s = acc_leg(th, 1.2)[0]
            f_s = flt_leg(th, 1.2)[0]
        scaling.append({"threads": th, "accumulate_s_per_mpx": round(a_s * 1e6, 4), "filter_s_per_mpx": round(f_s * 1e6, 4),
                        "mpixels_per_s": round(1e-6 / (a_s + t_pre / (W * H) + f_s), 5)})
    if not any(r["threads"] == cores for r in scaling):
        scaling.append({"threads": cores, "accumulate_s_per_mpx": round(acc_s_per_px * 1e6, 4), "filter_s_per_mpx": round(flt_s_per_px * 1e6, 4),
                        "mpixels_per_s": round(1e-6 / s_per_px, 5)})
        scaling.sort(key=lambda r: r["threads"])
    one = next(r for r in scaling if r["threads"] == 1)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    throttled = None
    try:
        throttled = int(open("/sys/fs/cgroup/cpu.stat").read().split("nr_throttled")[1].split()[0])
    except Exception:      # noqa: BLE001
        pass
    return {
        "value": round(1e-6 / s_per_px, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
        "cpu_model": cpu_model, "host_logical_cpus": os.cpu_count(), "cpu_share": share,
        "affinity": {"cpus_during_gpu_legs": affinity_before, "cpus_during_cpu_legs": len(os.sched_getaffinity(0))},
        "cgroup_throttled_periods_so_far": throttled,
        "sample": "oracle (C restatement of the reference algorithm, OpenMP, %d threads = the CPUs this box gives the process: %d in the "
                  "affinity mask, cgroup quota %s): accumulate %d rows x %d px (%d tiles) x %d of the %d spp x %d ch, samples in tile-major "
                  "pixel-major order, %d repetitions (%.1f s), scaled to %d spp; pre-pass full frame (%.2f s); filter %d rows x %d px, full "
                  "%dx%d window, %d repetitions (%.1f s); per-pixel times summed and inverted"
                  % (cores, share["affinity_cpus"], ("%.1f CPUs" % share["cgroup_quota_cpus"]) if share["cgroup_quota_cpus"] else "none",
                     ar, W, (ar // 16) * (W // 16), S_cpu, S, args.channels, acc_reps, t_acc, S, t_pre, fr, W, 2 * args.radius + 1,
                     2 * args.radius + 1, flt_reps, t_flt),
        "parity_of_the_same_run": parity,
        "accumulate_s_per_mpx": round(acc_s_per_px * 1e6, 4),
        "filter_s_per_mpx": round(flt_s_per_px * 1e6, 4),
        "threads_scaling": scaling,
        "speedup_over_one_thread": round(one["mpixels_per_s"] and (1e-6 / s_per_px) / one["mpixels_per_s"], 2),
        "single_thread": {"value": one["mpixels_per_s"], "unit": "Mpixels/s", "cores": 1},
    }


def numa_facts(dev_index):
    """Where the GPU and this process sit: a pinned buffer on the far socket is what a 10x slower Upload looks like."""
    out = {}
    try:
        import glob
        bus = torch.cuda.get_device_properties(dev_index).pci_bus_id if hasattr(torch.cuda.get_device_properties(dev_index), "pci_bus_id") else None
        nodes = {}
        for pth in glob.glob("/sys/bus/pci/devices/*/numa_node"):
            d = os.path.basename(os.path.dirname(pth))
            try:
                cls = open(os.path.join(os.path.dirname(pth), "class")).read().strip()
            except OSError:
                continue
            if cls.startswith("0x0302") or cls.startswith("0x0380") or cls.startswith("0x1200"):   # display / accelerators
                nodes[d] = int(open(pth).read().strip())
        out["gpu_numa_nodes"] = nodes
        out["pci_bus_id"] = bus
        out["process_cpus"] = len(os.sched_getaffinity(0))
        cur = open("/proc/self/stat").read().split()[38]
        out["cpu_now"] = int(cur)
        for n in glob.glob("/sys/devices/system/node/node*/cpulist"):
            out.setdefault("node_cpulists", {})[os.path.basename(os.path.dirname(n))] = open(n).read().strip()
    except Exception as e:                                      # diagnostics only
        out["error"] = repr(e)
    return out


def host_copy_times(fs, dev):
    """What the reference's `CUDA time` bracket adds when statistics are produced on the host
    (statpath.cpp:409-417): Upload() of the 7 filter inputs (76 B/px) and Download() of film-f
    (12 B/px) through statmc_upload / statmc_download from buffers of statmc_malloc_host (page-locked by the
    library, written once before timing so that every page exists).  Best and median of 7.  Never part of `value`."""
    import ctypes as C
    from statmc_amd import api
    lib = api.load()
    rad = fs.state["radiance"]
    ups = [rad["film_mean"], rad["n"], rad["mean"], rad["m2"], rad["m3"], fs.g_buffer("normal"), fs.g_buffer("albedo")]
    stream = api.current_stream_handle()
    host = []
    for t in ups + [fs.film_f]:
        p = C.c_void_p()
        api.check(lib.statmc_malloc_host(C.byref(p), t.numel() * 4))
        C.memset(p, 1, t.numel() * 4)
        host.append(p)

    def upload():
        for h, d in zip(host[:-1], ups):
            api.check(lib.statmc_upload(C.c_void_p(d.data_ptr()), h, d.numel() * 4, stream))

    def download():
        api.check(lib.statmc_download(host[-1], C.c_void_p(fs.film_f.data_ptr()), fs.film_f.numel() * 4, stream))

    keep = [t.clone() for t in ups]                      # the uploads overwrite the statistics: put them back after
    out = {}
    for name, fn, nbytes in (("upload", upload, sum(t.numel() * 4 for t in ups)), ("download", download, fs.film_f.numel() * 4)):
        fn()
        torch.cuda.synchronize()
        times = []
        for _ in range(7):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        times.sort()
        out[name + "_ms"] = round(times[0], 3)
        out[name + "_ms_median"] = round(times[len(times) // 2], 3)
        out[name + "_GBs"] = round(nbytes / times[0] / 1e6, 1)
        out[name + "_bytes_per_px"] = nbytes // (fs.width * fs.height)
    for t, k in zip(ups, keep):
        t.copy_(k)
    torch.cuda.synchronize()
    for h in host:
        lib.statmc_free_host(h)
    out["numa"] = numa_facts(dev.index or 0)
    return out


def host_bracket(fs, args):
    """The reference's own metric for this path: the `CUDA time [ns]` bracket = Upload + Denoise + Download +
    Synchronize (statpath.cpp:409-417, 520-527), through the C++ host side (statmc::Estimator, the thing a pbrt build
    links): the statistics are written as a dump and tools/bin/statmc_denoise runs the bracket on it (one warm-up,
    four timed iterations).  Secondary: never part of `value`."""
    import re
    import shutil
    import subprocess
    import tempfile
    from statmc_amd import build, pfm
    exe = build.build_tools()
    rad = fs.state["radiance"]
    d = tempfile.mkdtemp(prefix="statmc_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        stem = os.path.join(d, "scene")
        dump = {"film": rad["film_mean"], "t0-b0-n": rad["n"], "t0-b0-mean": rad["mean"], "t0-b0-m2": rad["m2"],
                "t0-b0-m3": rad["m3"], "t1-b0-film-mean": fs.g_buffer("normal"), "t2-b0-film-mean": fs.g_buffer("albedo")}
        for name, img in dump.items():
            pfm.write_pfm("%s-%d-%s.pfm" % (stem, args.spp, name), img.cpu().numpy())
        res = {}
        for key, bands, queues in (("pipelined", "0", "1"), ("two_queues", "0", "2"), ("one_stream", "1", "1")):
            out = subprocess.run([exe, "--stem", stem, "--spp", ",".join([str(args.spp)] * 12), "--filtersd", str(args.filtersd),
                                  "--filterradius", str(args.radius), "--warmup", "--bands", bands, "--output", "film-f"],
                                 capture_output=True, text=True, timeout=300, env=dict(os.environ, STATMC_UPLOAD_QUEUES=queues))
            if out.returncode != 0:
                return {"error": out.stderr.strip()[-300:]}
            ns = [int(v) for v in re.findall(r"HIP time \[ns\]: (\d+)", out.stdout)][1:]     # drop the warm-up
            ns.sort()
            res[key] = {"best_ms": round(ns[0] / 1e6, 3), "median_ms": round(ns[len(ns) // 2] / 1e6, 3), "worst_ms": round(ns[-1] / 1e6, 3),
                        "mean_ms": round(sum(ns) / len(ns) / 1e6, 3), "iterations": len(ns),
                        "bands": int(re.search(r"pipeline bands: (\d+)", out.stdout).group(1))}
        return {"cuda_time_bracket_ms": res["pipelined"]["best_ms"], "median_ms": res["pipelined"]["median_ms"],
                "worst_ms": res["pipelined"]["worst_ms"], "mean_ms": res["pipelined"]["mean_ms"], "iterations": res["pipelined"]["iterations"],
                "pipeline_bands": res["pipelined"]["bands"], "one_stream_ms": res["one_stream"]["best_ms"],
                "two_upload_queues": res["two_queues"],
                "what": "Estimator::Upload (76 B/px) + Denoise + Download (12 B/px) + Synchronize, C++ host side "
                        "(tools/bin/statmc_denoise), page-locked host images; the three phases run as a pipeline of row "
                        "bands on three streams (same bits; bands fitted to the window filter's rounds of 256 workgroups), one copy queue; "
                        "one_stream_ms: the same calls one after the other; "
                        "two_upload_queues: the copies dealt over two queues (faster when nothing stalls, with a tail: DESIGN.md 4.5)"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _tile_fed_measure(W, H, dev, samples, types, S, reps=5):
    """statmc_accumulate_tiles over the whole film with S samples per 16 x 16 tile (median of three `reps`-launch averages)."""
    from statmc_amd import api, film
    tiles = [(x, y, min(x + 16, W), min(y + 16, H)) for y in range(0, H, 16) for x in range(0, W, 16)]
    bounds = torch.tensor(tiles, dtype=torch.int32, device=dev)
    npx = torch.tensor([(x1 - x0) * (y1 - y0) for x0, y0, x1, y1 in tiles], dtype=torch.int64)
    offs = torch.cumsum(npx * S, 0) - npx * S
    st2 = new_film_stats(W, H, dev, types)
    sts, keep = [], []
    for t in types:
        c = film.STAT_TYPES[t]["channels"]
        src = samples[t][:S]
        # [S, H, W, C] -> per tile [S, th, tw, C]: full 16-row bands by reshape, the ragged last band by hand
        arena = new_arena((int((npx * S).sum()) * c,), dev)
        pos = 0
        for y in range(0, H, 16):
            th = min(16, H - y)
            band = src[:, y:y + th].reshape(S, th, W // 16, 16, c).permute(2, 0, 1, 3, 4).contiguous().reshape(-1)
            arena[pos:pos + band.numel()] = band
            pos += band.numel()
        keep.append(arena)
        sts.append(api.make_stat_type_arena(arena, c, st2.state[t], film.STAT_TYPES[t]["transform"], film.STAT_TYPES[t]["max_moment"]))
    offs_d = offs.to(dev)
    cnt = torch.full((len(tiles),), S, dtype=torch.int32, device=dev)
    api.accumulate_tiles(W, H, sts, bounds, offs_d, cnt)
    torch.cuda.synchronize()
    runs = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            api.accumulate_tiles(W, H, sts, bounds, offs_d, cnt)
        e1.record()
        torch.cuda.synchronize()
        runs.append(e0.elapsed_time(e1) / reps)
    ms = sorted(runs)[1]              # median of three averages
    bpp = accumulate_bytes_per_px(S, types)
    return {"spp": S, "avg_ms": round(ms, 4), "best_ms": round(min(runs), 4), "bytes_per_px": bpp,
            "achieved_GBs": round(bpp * W * H / ms / 1e6, 1), "frac_hbm": round(bpp * W * H / ms / 1e6 / HBM_PEAK_GBS, 4)}


def tile_fed_accumulate(fs, samples, types):
    """The accumulation fed the way Render<T> produces samples: 16 x 16 tile blocks through statmc_accumulate_tiles
    (Estimator::Merge[Transform]Tiles).  GB/s on the same byte count as the film-major kernel, at the step's sample count
    (the figure the film-major roofline line is compared with) and at 64 samples per tile.  Secondary."""
    W, H, dev = fs.width, fs.height, fs.device
    if W % 16:
        return {"skipped": "film width is not a multiple of the 16-pixel tile"}
    S_all = next(iter(samples.values())).shape[0]
    out = _tile_fed_measure(W, H, dev, samples, types, S_all)
    if S_all > 64:
        out["at_64_spp"] = _tile_fed_measure(W, H, dev, samples, types, 64)
    return out


def accumulate_by_batch(fs, samples, types, batch_sizes=(4, 8, 16, 32, 64), reps=10):
    """The accumulation at the batch sizes the reference's progressive schedule launches (statpath.cpp:272-279: 4, 4, 8, 16,
    ... samples per iteration): one launch of S samples per pixel moves 44 S + 224 B/px (11 channels), so at small S the
    224 B/px read-modify-write of the state dominates.  Film-major (statmc_accumulate) and tile-fed
    (statmc_accumulate_tiles) launches, `reps` back to back, median of three averages.  Secondary."""
    from statmc_amd import film
    W, H, dev = fs.width, fs.height, fs.device
    S_all = next(iter(samples.values())).shape[0]
    rows = []
    for S in batch_sizes:
        if S > S_all:
            continue
        st2 = new_film_stats(W, H, dev, types)
        part = {t: v[:S] for t, v in samples.items()}
        st2.accumulate(part)
        torch.cuda.synchronize()
        runs = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                st2.accumulate(part)
            e1.record()
            torch.cuda.synchronize()
            runs.append(e0.elapsed_time(e1) / reps)
        ms = sorted(runs)[1]
        bpp = accumulate_bytes_per_px(S, types)
        row = {"spp": S, "bytes_per_px": bpp, "film_major_ms": round(ms, 4), "film_major_GBs": round(bpp * W * H / ms / 1e6, 1),
               "film_major_frac_hbm": round(bpp * W * H / ms / 1e6 / HBM_PEAK_GBS, 4)}
        if W % 16 == 0:
            tf = _tile_fed_measure(W, H, dev, samples, types, S, reps=reps)
            row.update({"tile_fed_ms": tf["avg_ms"], "tile_fed_GBs": tf["achieved_GBs"], "tile_fed_frac_hbm": tf["frac_hbm"]})
        rows.append(row)
        del st2
    return {"film": "%dx%d" % (W, H), "launches": rows,
            "what": "one launch of S samples per pixel, all stat types (bytes = samples + read-modify-write of the state), %d launches back to back" % reps}


def accumulate_by_batch_4k(args, dev, types):
    """The same table on a 3840 x 2160 film (BASELINE configs[4]'s), whose 1.86 GB of state no cache holds between launches: 64
    samples per pixel generated for the leg, moments and arenas placed like the timed step's.  Secondary."""
    from statmc_amd import synthetic
    if (args.width, args.height) == (3840, 2160):
        return {"skipped": "the bench film is 3840x2160 already: see accumulate_by_batch"}
    if args.width * args.height < 1920 * 1080:
        return {"skipped": "a leg of the full-size lines (test-sized film)"}
    W, H, S = 3840, 2160, 64
    free_b = torch.cuda.mem_get_info(dev)[0]
    if free_b < 3 * 4 * args.channels * W * H * S:
        return {"skipped": "not enough free memory for a 4K / 64-spp pool (%.0f GiB free)" % (free_b / 2 ** 30)}
    scene = synthetic.Scene(W, H, n_regions=12, seed=1, device=dev)
    smp = new_arenas(S, H, W, dev, types)
    for s0 in range(0, S, 16):
        part = scene.samples(16, seed=77 + s0, features=types)
        for t in types:
            smp[t][s0:s0 + 16] = part[t]
        del part
    fs4 = new_film_stats(W, H, dev, types)
    out = accumulate_by_batch(fs4, smp, types, reps=6)
    del fs4
    try:        # BASELINE configs[4]'s film on ONE GPU, 64 resident spp: the step
        out["config4_step"] = config_step_leg(W, H, S, dev, types, args, "configs[4] film 3840x2160 on one GPU, 64 spp", reps=4, smp=smp)
    except Exception as e:      # noqa: BLE001
        out["config4_step"] = {"error": "%s: %s" % (type(e).__name__, str(e)[-300:])}
    del smp
    torch.cuda.empty_cache()
    return out


def _ev_ms(pairs):
    return sum(a.elapsed_time(b) for a, b in pairs)


def _step_legs(fs, samples, types, batches, reps, reset_per_step):
    """`reps` steps of accumulate(batch) -> pre-pass -> window filter per batch on `fs`; per-stage HIP-event sums and the wall
    time of the whole loop (device-synchronised on both sides)."""
    ev = lambda: torch.cuda.Event(enable_timing=True)
    stages = {"accumulate": [], "prepass": [], "filter": []}

    def timed(name, fn, *a):
        e0, e1 = ev(), ev()
        e0.record()
        fn(*a)
        e1.record()
        stages[name].append((e0, e1))

    def one(record):
        if reset_per_step:
            fs.reset()
        pos = 0
        for b in batches:
            part = samples if (pos == 0 and b == next(iter(samples.values())).shape[0]) else {t: v[pos:pos + b] for t, v in samples.items()}
            if record:
                timed("accumulate", fs.accumulate, part)
                timed("prepass", fs.prepass)
                timed("filter", fs.window_filter)
            else:
                fs.accumulate(part)
                fs.prepass()
                fs.window_filter()
            pos += b
    one(False)
    one(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        one(True)
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3 / reps
    return wall_ms, {k: _ev_ms(v) / reps for k, v in stages.items()}


def reference_schedule_leg(fs, samples, types, args, reps=4):
    """The step a StatMC user sees: the reference's progressive schedule -- iterations of 4, 4, 8, 16, ...
    samples up to --spp (statpath.cpp:272-279), the denoiser after EVERY iteration (statpath.cpp:406-418), statistics reset at the
    start of a render -- on the timed step's own film and sample pool.  Secondary; runs after the CPU baseline (it resets `fs`)."""
    from statmc_amd import api, synthetic
    S = next(iter(samples.values())).shape[0]
    batches = synthetic.sample_schedule(min(args.spp, S))
    wall_ms, st = _step_legs(fs, samples, types, batches, reps, reset_per_step=True)
    W, H = fs.width, fs.height
    acc_bytes = sum(accumulate_bytes_per_px(b, types) + (24 if fs.fused_prepass else 0) for b in batches) * W * H
    return {"film": "%dx%d" % (W, H), "spp": sum(batches), "iterations": len(batches), "batches": batches, "prepass_fused": bool(fs.fused_prepass),
            "ms_per_step": round(wall_ms, 4), "mpixels_per_s": round(W * H / wall_ms / 1e3, 2),
            "accumulate_ms": round(st["accumulate"], 4), "prepass_ms": round(st["prepass"], 4), "filter_ms": round(st["filter"], 4),
            "filter_share": round(st["filter"] / max(wall_ms, 1e-9), 4),
            "accumulate_frac_hbm": round(acc_bytes / (st["accumulate"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "filter_ms_per_iteration": round(st["filter"] / len(batches), 4), "filter_variant": api.last_filter_variant(),
            "what": "one 256-spp render in the reference's own schedule: %d iterations, pre-pass + window filter after each, statistics "
                    "reset per step (the reset's memsets are inside ms_per_step); %d steps" % (len(batches), reps)}


def config_step_leg(W, H, S, dev, types, args, name, reps, smp=None):
    """One BASELINE config that the timed step does not run, on this GPU: a WxH film, S resident samples per pixel, the same step
    (accumulate of all S, pre-pass, window filter).  Buffers placed like the timed step's.  Secondary."""
    from statmc_amd import api, synthetic
    free_b = torch.cuda.mem_get_info(dev)[0]
    need = 4 * args.channels * W * H * S
    if smp is None and free_b < 2 * need + (8 << 30):
        return {"skipped": "not enough free memory for a %dx%d / %d-spp pool (%.0f GiB free)" % (W, H, S, free_b / 2 ** 30)}
    own = smp is None
    if own:
        scene = synthetic.Scene(W, H, n_regions=12, seed=1, device=dev)
        smp = new_arenas(S, H, W, dev, types)
        for s0 in range(0, S, 16):
            part = scene.samples(min(16, S - s0), seed=77 + s0, features=types)
            for t in types:
                smp[t][s0:s0 + part[t].shape[0]] = part[t]
            del part
    fsx = new_film_stats(W, H, dev, types, filter_sd=args.filtersd, radius=args.radius, fused_prepass=args.fused_prepass)
    wall_ms, st = _step_legs(fsx, smp, types, [S], reps, reset_per_step=False)
    bpp = accumulate_bytes_per_px(S, types) + (24 if args.fused_prepass else 0)
    out = {"config": name, "film": "%dx%d" % (W, H), "spp": S, "ms_per_step": round(wall_ms, 4), "mpixels_per_s": round(W * H / wall_ms / 1e3, 2),
           "accumulate_ms": round(st["accumulate"], 4), "accumulate_frac_hbm": round(bpp * W * H / (st["accumulate"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "prepass_ms": round(st["prepass"], 4), "prepass_fused": bool(args.fused_prepass), "filter_ms": round(st["filter"], 4),
           "filter_variant": api.last_filter_variant(), "filter_parts": api.load().statmc_debug_last_filter_parts(), "steps": reps}
    del fsx
    if own:
        del smp
        torch.cuda.empty_cache()
    return out


def accumulate_unplaced_ab(fs, samples, types, reps=6):
    """In-process A/B of the placed allocator: the timed step's accumulation launch, back to back, on the
    step's own (placed) buffers and on copies of them that come from torch's allocator (hipMalloc).  Outside the timed region."""
    from statmc_amd import film
    if not PLACED["on"]:
        return {"skipped": "the step's buffers are not placed"}
    W, H, dev = fs.width, fs.height, fs.device
    need = sum(v.numel() * 4 for v in samples.values())
    free_b = torch.cuda.mem_get_info(dev)[0]
    if free_b < need + (8 << 30):
        return {"skipped": "not enough free memory for an unplaced copy of the sample pool (%.0f GiB free)" % (free_b / 2 ** 30)}
    S = next(iter(samples.values())).shape[0]
    bpp = accumulate_bytes_per_px(S, types)

    def b2b(f, smp):
        f.accumulate(smp)
        torch.cuda.synchronize()
        runs = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                f.accumulate(smp)
            e1.record()
            torch.cuda.synchronize()
            runs.append(e0.elapsed_time(e1) / reps)
        return sorted(runs)[1]
    placed_ms = b2b(fs, samples)
    smp_u = {t: torch.empty(v.shape, dtype=torch.float32, device=dev) for t, v in samples.items()}
    for t in types:
        smp_u[t].copy_(samples[t])
    fs_u = film.FilmStats(W, H, dev, types=types)
    unplaced_ms = b2b(fs_u, smp_u)
    del smp_u, fs_u
    torch.cuda.empty_cache()
    frac = lambda ms: round(bpp * W * H / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return {"spp": S, "placed_ms": round(placed_ms, 4), "placed_frac_hbm": frac(placed_ms), "unplaced_ms": round(unplaced_ms, 4),
            "unplaced_frac_hbm": frac(unplaced_ms), "what": "the step's accumulation launch back to back (%d launches, median of 3): statmc_malloc_placed "
            "buffers against torch-allocated copies of the same bytes, same process" % reps}


def placement_check(args, pipe, samples, types, layout, dev, make_pipe, reps=4):
    """Placement is an optimisation that can come out behind (a card whose slots mostly straddle classes): before the
    warm-up the step -- accumulate, pre-pass, filter, as in the timed region -- runs `reps` times on the placed buffers and on copies that
    come from torch's allocator, and the faster set (by the accumulation's median, > 1 %) goes on.  Returns (pipe, samples, report)."""
    need = sum(v.numel() * 4 for v in samples.values())
    free_b = torch.cuda.mem_get_info(dev)[0]
    forced = os.environ.get("STATMC_BENCH_CHECK_PICKS") == "allocator"      # (tests: the other branch)
    if need < (256 << 20) and not forced:
        return pipe, samples, {"skipped": "a sample pool of %.0f MiB: the accumulation is too short to tell two sets of buffers apart" % (need / 2 ** 20)}
    if free_b < need + (12 << 30):
        return pipe, samples, {"skipped": "not enough free memory for an unplaced copy of the sample pool (%.0f GiB free)" % (free_b / 2 ** 30)}

    def acc_ms(p, smp):
        p.fs.reset()
        runs = []
        for i in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            p.accumulate(smp)
            e1.record()
            p.prepass()
            p.window_filter()
            torch.cuda.synchronize()
            if i:
                runs.append(e0.elapsed_time(e1))
        p.fs.reset()
        return sorted(runs)[len(runs) // 2]
    placed_ms = acc_ms(pipe, samples)
    pipe_u = make_pipe(False)
    smp_u = {t: torch.empty(v.shape, dtype=torch.float32, device=dev) for t, v in samples.items()}
    for t in types:
        smp_u[t].copy_(samples[t])
    unplaced_ms = acc_ms(pipe_u, smp_u)
    report = {"placed_ms": round(placed_ms, 4), "allocator_ms": round(unplaced_ms, 4),
              "what": "the step (accumulate, pre-pass, filter) %d times on each set before the warm-up, median of the accumulation; the faster set (> 1 %%) runs the timed region" % reps}
    if unplaced_ms < 0.99 * placed_ms or forced:
        if forced:
            report["forced"] = "STATMC_BENCH_CHECK_PICKS=allocator"
        report["chosen"] = "torch's allocator"
        return pipe_u, smp_u, report
    report["chosen"] = "placed"
    del pipe_u, smp_u
    torch.cuda.empty_cache()
    return pipe, samples, report


def bind_to_gpu_numa(dev_index):
    """Keep the rank on the CPUs of the NUMA node its GPU hangs off (host-side issue latency, pinned staging buffers)."""
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip())
        if node < 0:
            return {"pci": bdf, "numa_node": node, "bound": False}
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return {"pci": bdf, "numa_node": node, "bound": False}
        os.sched_setaffinity(0, cpus)
        return {"pci": bdf, "numa_node": node, "bound": True, "cpus": len(cpus)}
    except Exception as e:      # noqa: BLE001  (diagnostic convenience only)
        return {"bound": False, "error": repr(e)[:120]}


def pool_slices(start, count, pool):
    """Samples [start, start + count) of a step as slices of the resident pool (sample s = pool sample s mod pool)."""
    out = []
    while count > 0:
        a = start % pool
        b = min(pool, a + count)
        out.append((a, b))
        count -= b - a
        start += b - a
    return out


def pcie_inclusive(fs, samples, types, args, chunk_spp=32, rounds=6):
    """What the step costs when the samples are NOT resident: the renderer's samples sit in page-locked host memory and
    cross PCIe on their way to the accumulation (statmc::Estimator's device-side accumulation, FlushSamples).  Two device
    chunks of `chunk_spp` samples per pixel in flight: chunk k + 1 is copied on a copy stream while chunk k is
    accumulated.  Reports the steady-state rate and what it makes of a whole step (spp / chunk copies + pre-pass +
    filter, copies hiding the kernels).  Secondary; `value` is defined with the inputs resident."""
    import ctypes as C
    from statmc_amd import api, film
    lib = api.load()
    W, H, dev = fs.width, fs.height, fs.device
    S = min(chunk_spp, next(iter(samples.values())).shape[0])
    nbytes = {t: samples[t][:S].numel() * 4 for t in types}
    host = {}
    for t in types:
        p = C.c_void_p()
        api.check(lib.statmc_malloc_host(C.byref(p), nbytes[t]))
        C.memset(p, 0, nbytes[t])                                   # touch every page before timing
        host[t] = p
    dchunks = [{t: torch.empty_like(samples[t][:S]) for t in types} for _ in range(2)]
    st2 = film.FilmStats(W, H, dev, types=types)
    copy_stream = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream()
    landed = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]

    def upload(k):
        copy_stream.wait_event(consumed[k])
        for t in types:
            api.check(lib.statmc_upload(C.c_void_p(dchunks[k][t].data_ptr()), host[t], nbytes[t], C.c_void_p(copy_stream.cuda_stream)))
        landed[k].record(copy_stream)

    def run(n):
        for k in range(2):
            consumed[k].record(main)
        upload(0)
        for i in range(n):
            k = i & 1
            if i + 1 < n:
                upload(k ^ 1)
            main.wait_event(landed[k])
            st2.accumulate(dchunks[k])
            consumed[k].record(main)
        torch.cuda.synchronize()

    try:
        run(2)
        t0 = time.perf_counter()
        run(rounds)
        dt = (time.perf_counter() - t0) / rounds
    finally:
        for p in host.values():
            lib.statmc_free_host(p)
    per_chunk_bytes = sum(nbytes.values())
    n_chunks = (args.spp + S - 1) // S
    step_ms = n_chunks * dt * 1e3 + 0.03 + 1.7          # + pre-pass + filter behind the last chunk
    return {"chunk_spp": S, "chunk_bytes": per_chunk_bytes, "ms_per_chunk": round(dt * 1e3, 3), "GBs": round(per_chunk_bytes / dt / 1e9, 1),
            "step_ms_if_samples_cross_pcie": round(step_ms, 1), "mpixels_per_s_if_samples_cross_pcie": round(W * H / step_ms / 1e3, 2),
            "what": "samples of a step streamed from page-locked host memory in %d-spp chunks, copy of chunk k + 1 beside the accumulation of "
                    "chunk k (the accumulation hides behind the copies: the link is the bound); step = %d chunks + pre-pass + filter" % (S, n_chunks)}


def eight_channel_filter(fs, args, reps=20):
    """The window filter with all four feature images of the 11-channel stream as G-buffers -- normal, albedo, depth,
    material id: eight feature channels (statpath.cpp:828-835, 1096-1130) -- on the statistics the timed loop left.
    Secondary: never part of `value`."""
    from statmc_amd import api
    if "depth" not in fs.state or "materialid" not in fs.state:
        return {"skipped": "needs the 11-channel stream"}
    rad = fs.state["radiance"]
    names, sds = ["normal", "albedo", "depth", "materialid"], [0.1, 0.02, 1.0, 0.5]
    out = torch.zeros_like(fs.film_f)
    a, keep = api.make_filter_args(n=[rad["n"]], mean=[rad["mean"]], m2=[rad["m2"]], m3=[rad["m3"]], film=[rad["film_mean"]],
                                   mean_corr=[fs.mean_corr], disc=[fs.disc], film_filtered=[out],
                                   g_buffers=[fs.g_buffer(g) for g in names], g_sds=sds, filter_sd=args.filtersd, radius=args.radius)
    for _ in range(3):
        api.window_filter(a, 3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        api.window_filter(a, 3)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return {"g_buffers": names, "g_sds": sds, "feature_channels": 8, "filter_variant": api.last_filter_variant(), "avg_ms": round(ms, 4),
            "mpixels_per_s": round(fs.width * fs.height / ms / 1e3, 1), "bytes_per_px": FILTER_BYTES_PER_PX + 8,
            "what": "back-to-back launches of the window filter (pre-pass not included) with normal, albedo, depth and material id as G-buffers"}


def float_buffer_filter(fs, args, reps=6):
    """filter<float> as ACRR and SMIS call it on every render iteration (estimator.cpp:434-460: 5 luminance / 12 win-rate buffers in
    ONE call sharing the G-buffers; statpath.cpp:306-313 reads the result back): the window filter over 1, 2, 5 and 12 one-channel
    buffers cut from the radiance statistics the timed loop left (two buffers per launch share the range weight).  Secondary."""
    from statmc_amd import api
    rad = fs.state["radiance"]
    gbs = [fs.g_buffer(g) for g in fs.g_names]
    W, H, dev = fs.width, fs.height, fs.device
    out, variants = {}, {}
    for nb in (1, 2, 5, 12):
        mc = [(fs.mean_corr[..., b % 3:b % 3 + 1] * (1.0 / (1 + b))).contiguous() for b in range(nb)]
        dc = [(fs.disc[..., b % 3:b % 3 + 1] * (1.0 / (1 + b)) ** 2).contiguous() for b in range(nb)]
        col = [(rad["film_mean"][..., b % 3:b % 3 + 1] * (1.0 / (1 + b))).contiguous() for b in range(nb)]
        outs = [torch.zeros(H, W, 1, device=dev) for _ in range(nb)]
        a, keep = api.make_filter_args(n=[], mean=[], m2=[], m3=[], film=col, mean_corr=mc, disc=dc, film_filtered=outs,
                                       g_buffers=gbs, g_sds=fs.g_sds, filter_sd=args.filtersd, radius=args.radius)
        for _ in range(2):
            api.window_filter(a, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            api.window_filter(a, 1)
        e1.record()
        torch.cuda.synchronize()
        out["%d_buffers_ms" % nb] = round(e0.elapsed_time(e1) / reps, 4)
        variants["%d" % nb] = api.last_filter_variant()
        del mc, dc, col, outs
    return dict(out, filter_variant=variants["12"], filter_variant_by_count=variants, acrr_5_buffers_ms=out["5_buffers_ms"], smis_12_buffers_ms=out["12_buffers_ms"],
                per_buffer_ms_at_12=round(out["12_buffers_ms"] / 12, 4),
                what="window filter of n one-channel buffers in one call, back to back (pre-pass not included): the pair-symmetric kernel takes two "
                     "buffers per launch (more behind one range weight do not fit the CU's LDS: DESIGN.md section 9 item 2); an odd count ends with three "
                     "buffers on the one-sided kernel, which shares the weight over three")


def _relay_child(cmd, env, timeout_s):
    """Runs a child to its end (or to the time limit: its whole process group is then ended), relays everything but the
    JSON line to stderr; returns (rc, line, last stderr lines, timed_out).  Nothing is exec'ed in this process."""
    import collections
    import signal
    import subprocess
    import threading
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    tail = collections.deque(maxlen=12)

    def pump_err():
        for l in proc.stderr:
            tail.append(l.rstrip()[-300:])
            sys.stderr.write(l)
    t = threading.Thread(target=pump_err, daemon=True)
    t.start()
    line = [None]

    def pump_out():
        for l in proc.stdout:
            if l.startswith("{") and '"metric"' in l:
                line[0] = l.strip()
            else:
                sys.stderr.write(l)
    t2 = threading.Thread(target=pump_out, daemon=True)
    t2.start()
    timed_out = False
    try:
        rc = proc.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        timed_out = True
        try:
            os.killpg(proc.pid, signal.SIGTERM)          # the group this call started, nothing else
            rc = proc.wait(timeout=20)
        except Exception:                                # noqa: BLE001
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except Exception:                            # noqa: BLE001
                pass
            rc = proc.wait()
    t.join(5)
    t2.join(5)
    return rc, line[0], list(tail), timed_out


def _child_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "LOCAL_WORLD_SIZE",
                                                           "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    # dmabuf IPC: the host driver of this pool supports no other kind -- without it RCCL (and any sharing of device
    # memory across processes) fails in hipIpcGetMemHandle.  The image exports it already; a launcher that builds its
    # own environment must keep it (the task's environment notes say so), hence setdefault rather than a guess.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def run_peer_child(args, why, error_lines, timeout_s=900):
    """The RCCL-independent leg as a FRESH process (this one may hold a half-initialised communicator): one process drives
    all N devices through the C ABI.  Returns (rc, line) with the fallback recorded in the line."""
    argv = [a for a in sys.argv[1:]]
    # drop a --backend given on the command line, then ask for the peer leg
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a == "--backend":
            skip = True
            continue
        if a.startswith("--backend="):
            continue
        out.append(a)
    cmd = [sys.executable, os.path.abspath(__file__)] + out + ["--backend", "peer"]
    rc, line, tail, timed_out = _relay_child(cmd, _child_env(), timeout_s)
    if line is not None:
        try:
            d = json.loads(line)
            d["fallback_from"] = args.backend
            d["nccl_error"] = {"why": why, "last_stderr_lines": error_lines[-8:]}
            line = json.dumps(d)
        except Exception:                                # noqa: BLE001
            pass
    return rc, line


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as FRESH child processes
    (python -m torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) before this process has made any GPU
    call -- it never does -- and relay rank 0's JSON line and the exit code.  If the nccl leg ends without a line (RCCL
    bring-up failed, a rank died, the time limit passed), a fresh child runs the peer leg instead and the line says so
    (`fallback_from`, `nccl_error`).  Nothing is exec'ed."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = _child_env()
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    env["STATMC_BENCH_NO_FALLBACK"] = "1"          # the ranks report a failure to this parent, which owns the fallback
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc, line, tail, timed_out = _relay_child(cmd, env, args.rank_timeout)
    if line is None and args.backend == "nccl" and not args.no_fallback:
        why = "time limit of %d s" % args.rank_timeout if timed_out else "exit code %d without a result line" % rc
        sys.stderr.write("bench.py: the nccl leg gave no result (%s); running the peer leg in a fresh process\n" % why)
        rc, line = run_peer_child(args, why, tail)
    if line is not None:
        print(line, flush=True)
    sys.exit(rc if rc else (0 if line is not None else 1))


def bring_up(args, rank, world, dev):
    """Initialises the process group and proves the primitive the halo exchange uses (grouped isend / irecv between strip
    neighbours) plus the all-reduce of the timing, under a watchdog: a bring-up that hangs ends the process instead of the
    run.  Returns None on success, a short description of the failure otherwise."""
    import datetime
    import threading
    done = threading.Event()
    state = {"stage": "init_process_group"}

    def watchdog():
        if not done.wait(args.bringup_timeout):
            sys.stderr.write("bench.py rank %d: %s bring-up stuck in %s for %d s\n" % (rank, args.backend, state["stage"], args.bringup_timeout))
            sys.stderr.flush()
            state["hung"] = True
            if rank == 0 and args.backend == "nccl" and not os.environ.get("STATMC_BENCH_NO_FALLBACK") and not args.no_fallback:
                rc, line = run_peer_child(args, "bring-up stuck in %s for %d s" % (state["stage"], args.bringup_timeout), [])
                if line is not None:
                    print(line, flush=True)
                os._exit(0 if line is not None else 1)
            os._exit(0 if (rank != 0 and not os.environ.get("STATMC_BENCH_NO_FALLBACK")) else 3)
    threading.Thread(target=watchdog, daemon=True).start()
    try:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        to = datetime.timedelta(seconds=max(60, args.bringup_timeout))
        if args.backend == "nccl":
            if os.environ.get("STATMC_BENCH_FAIL_NCCL") == "1":          # tests: the forced-failure path
                raise RuntimeError("STATMC_BENCH_FAIL_NCCL=1 (forced failure of the nccl bring-up)")
            if os.environ.get("STATMC_BENCH_FAIL_NCCL") == "hang":       # tests: a bring-up that never returns (the watchdog's case)
                time.sleep(10 ** 6)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=to)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=to)
        cdev = dev if args.backend == "nccl" else "cpu"
        state["stage"] = "all_reduce"
        t = torch.ones(1, device=cdev)
        dist.all_reduce(t)
        assert int(t.item()) == world
        state["stage"] = "batch_isend_irecv"
        ops, bufs = [], []
        for peer in (rank - 1, rank + 1):
            if 0 <= peer < world:
                sb, rb = torch.full((1024,), float(rank), device=cdev), torch.empty(1024, device=cdev)
                bufs.append((peer, rb))
                ops += [dist.P2POp(dist.isend, sb, peer), dist.P2POp(dist.irecv, rb, peer)]
        for req in (dist.batch_isend_irecv(ops) if ops else []):
            req.wait()
        if args.backend == "nccl":
            torch.cuda.synchronize()
        for peer, rb in bufs:
            assert float(rb[0].item()) == float(peer)
        return None
    except Exception as e:      # noqa: BLE001
        return "%s in %s: %s" % (type(e).__name__, state["stage"], str(e)[-400:])
    finally:
        done.set()


def main():
    global torch, dist
    t_process = time.perf_counter()
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.backend == "peer":
        if world > 1:           # started under a launcher: one process does it all, the others have nothing to do
            if rank != 0:
                return
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                os.environ.pop(k, None)
        import torch as _torch
        torch = _torch
        return main_peer(args)
    if world == 1 and args.gpus > 1:
        launch_ranks(args)                       # does not return
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d started with WORLD_SIZE=%d" % (args.gpus, world))
    import torch as _torch
    import torch.distributed as _dist
    torch, dist = _torch, _dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU (no CPU fallback exists for the product path)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    binding = {"bound": False} if args.no_bind else bind_to_gpu_numa(local_rank)
    if world > 1:
        err = bring_up(args, rank, world, dev)
        if err is not None:
            # every rank sees a failed bring-up (it is collective).  Under the self-launcher the parent owns the fallback;
            # under a foreign launcher (the driver's torch.distributed.run form) rank 0 does, in a fresh child process.
            sys.stderr.write("bench.py rank %d: %s bring-up failed: %s\n" % (rank, args.backend, err))
            own = args.backend == "nccl" and not os.environ.get("STATMC_BENCH_NO_FALLBACK") and not args.no_fallback
            if not own:
                sys.exit(3)
            if rank != 0:
                sys.exit(0)
            rc, line = run_peer_child(args, err, [])
            if line is not None:
                print(line, flush=True)
            sys.exit(rc if rc else (0 if line is not None else 1))
    n_ranks_seen = dist.get_world_size() if world > 1 else 1
    run_done = None
    stall_test = os.environ.get("STATMC_BENCH_FAIL_NCCL") == "stall"      # tests: a run that stalls behind a good bring-up (any backend)
    if world > 1 and (args.backend == "nccl" or stall_test) and not os.environ.get("STATMC_BENCH_NO_FALLBACK") and not args.no_fallback:
        # Under a foreign launcher nobody else ends a run that stalls after the bring-up (an exchange that never completes):
        # past --run-timeout every rank leaves, and rank 0 -- once the others have let go of their devices -- runs the peer leg
        # in a fresh process.  (Self-launched ranks: the parent's --rank-timeout does this.)
        import threading as _threading
        run_done = _threading.Event()

        def run_watchdog():
            if run_done.wait(args.run_timeout):
                return
            sys.stderr.write("bench.py rank %d: the nccl run has not finished %d s after the bring-up\n" % (rank, args.run_timeout))
            sys.stderr.flush()
            if rank != 0:
                os._exit(0)
            time.sleep(3.0)
            rc, line = run_peer_child(args, "the nccl run did not finish within %d s of the bring-up" % args.run_timeout, [])
            if line is not None:
                print(line, flush=True)
            os._exit(0 if line is not None else 1)
        _threading.Thread(target=run_watchdog, daemon=True).start()
        if stall_test:
            time.sleep(10 ** 6)

    if os.environ.get("STATMC_VARIANT"):      # experiments: a variant library (tools/experiments/build_variant.sh); the line says so
        from statmc_amd import build as _build
        os.environ.setdefault("STATMC_ALLOW_DIAGNOSTIC_BUILD", "1")
        _build.SO = os.path.abspath(os.environ["STATMC_VARIANT"])
    from statmc_amd import api, film, pipeline, sharding, synthetic
    api.setup(local_rank)
    if os.environ.get("STATMC_BENCH_ACC_RESIDENT") or os.environ.get("STATMC_BENCH_ACC_DMA"):      # experiments: launch knobs of the accumulation
        _lib = api.load()
        if os.environ.get("STATMC_BENCH_ACC_RESIDENT"):
            api.check(_lib.statmc_debug_accumulate_resident_blocks(int(os.environ["STATMC_BENCH_ACC_RESIDENT"])))
        if os.environ.get("STATMC_BENCH_ACC_DMA"):
            api.check(_lib.statmc_debug_accumulate_dma(int(os.environ["STATMC_BENCH_ACC_DMA"])))

    S, r = args.spp, args.radius
    types = list(synthetic.FEATURES) if args.channels == 11 else ["radiance", "normal", "albedo"]
    grid = sharding.row_strips(world) if args.grid == "rows" else sharding.grid_for(world)
    if args.scaling == "strong":        # one film, N blocks
        if args.width % grid[0] or args.height % grid[1]:
            raise SystemExit("film %dx%d does not split into a %dx%d grid of equal blocks" % (args.width, args.height, *grid))
        W, H = args.width // grid[0], args.height // grid[1]
    else:                               # one film-sized block per rank
        W, H = args.width, args.height
    layout = sharding.BlockLayout(rank, world, W, H, r, grid=grid)
    fw, fh = layout.film_size

    # ---- synthetic inputs, generated in place in HBM (seeded; same generator as the tests)
    # The running moments and the sample arenas come from statmc_malloc_placed (moments in one interference class of the
    # card's memory, arenas in another: + 6 - 17 % on the accumulation, the same bits); anything that goes wrong there is
    # reported in the line and the buffers come from torch's allocator instead.
    # (ranks that share a device -- test setups -- would each back their own slots on it: off there unless a test asks for it)
    PLACED["on"] = bool(args.placement) and (not args.share_device or os.environ.get("STATMC_BENCH_PLACED_ON_SHARED_DEVICE") == "1")
    _ballast = None
    if os.environ.get("STATMC_BENCH_BALLAST_GB"):      # experiment: occupy the first GiB of the card before anything is placed
        _ballast = torch.empty(int(float(os.environ["STATMC_BENCH_BALLAST_GB"]) * 2 ** 28), dtype=torch.float32, device=dev)
    try:
        pipe = pipeline.BlockPipeline(layout, dev, types, filter_sd=args.filtersd, radius=r,
                                      via_host=args.backend == "gloo", placed=PLACED["on"], fused_prepass=args.fused_prepass)
    except api.StatmcError as e:
        PLACED.update(on=False, error=str(e)[-300:])
        pipe = pipeline.BlockPipeline(layout, dev, types, filter_sd=args.filtersd, radius=r, via_host=args.backend == "gloo",
                                      fused_prepass=args.fused_prepass)
    samples, pool = block_samples(args, layout, dev, types, rank, world, share=world if args.share_device else 1)
    if PLACED["on"] and world == 1 and args.placement_check and pool == S:
        make_pipe = lambda placed: pipeline.BlockPipeline(layout, dev, types, filter_sd=args.filtersd, radius=r, via_host=False,
                                                          placed=placed, fused_prepass=args.fused_prepass)
        pipe_c, samples_c, PLACED["check"] = placement_check(args, pipe, samples, types, layout, dev, make_pipe)
        if pipe_c is not pipe:          # the allocator's memory measured faster here: the placed blocks go, and their slots with them
            pipe, samples = pipe_c, samples_c
            PLACED.update(on=False, error=None)
            try:
                PLACED["trimmed"] = api.placement_trim()
            except api.StatmcError as e:
                PLACED["trim_error"] = str(e)[-200:]
        del pipe_c, samples_c
    fs = pipe.fs
    batches = synthetic.sample_schedule(S) if args.schedule == "reference" else [S]

    ev = lambda: torch.cuda.Event(enable_timing=True)
    k_events = {k: [] for k in ("accumulate", "prepass", "halo", "filter", "gather", "border_chain", "interior_accumulate",
                                "interior_prepass", "interior_exposed", "exchange_exposed")}
    film_f = torch.empty(fh, fw, 3, dtype=torch.float32, device=dev) if (world > 1 and rank == 0) else None

    def timed(name, record, fn, *a):
        if not record:
            return fn(*a)
        e0, e1 = ev(), ev()
        e0.record()
        out = fn(*a)
        e1.record()
        k_events[name].append((e0, e1))
        return out

    def accumulate_range(start, count, rows=None):
        for a, b in pool_slices(start, count, pool):
            pipe.accumulate(samples if (a, b) == (0, pool) else {t: v[a:b] for t, v in samples.items()}, rows=rows)

    fed_by_tiles = args.feed == "tiles" and world == 1 and batches == [S] and pool == S and W % 16 == 0
    if fed_by_tiles:
        # --feed tiles: the timed step fed the way Render<T> hands samples over -- 16 x 16 tile blocks through statmc_accumulate_tiles
        # (the same samples, permuted once; `tile_fed_accumulate` measures this launch back to back): 3.61 against 3.68 - 3.74 ms in
        # the step at 1080p / 256 spp (a wave's consecutive sample rows are 3 KB apart instead of 25 MB)
        from statmc_amd import film as _film
        tiles = [(x, y, min(x + 16, W), min(y + 16, H)) for y in range(0, H, 16) for x in range(0, W, 16)]
        t_bounds = torch.tensor(tiles, dtype=torch.int32, device=dev)
        t_npx = torch.tensor([(x1 - x0) * (y1 - y0) for x0, y0, x1, y1 in tiles], dtype=torch.int64)
        t_offs = (torch.cumsum(t_npx * S, 0) - t_npx * S).to(dev)
        t_cnt = torch.full((len(tiles),), S, dtype=torch.int32, device=dev)
        t_sts, t_keep = [], []
        for t in types:
            c = _film.STAT_TYPES[t]["channels"]
            arena = new_arena((int((t_npx * S).sum()) * c,), dev)
            at = 0
            for y in range(0, H, 16):
                th = min(16, H - y)
                band = samples[t][:, y:y + th].reshape(S, th, W // 16, 16, c).permute(2, 0, 1, 3, 4).contiguous().reshape(-1)
                arena[at:at + band.numel()] = band
                at += band.numel()
            t_keep.append(arena)
            t_sts.append(api.make_stat_type_arena(arena, c, fs.state[t], _film.STAT_TYPES[t]["transform"], _film.STAT_TYPES[t]["max_moment"],
                                                  prepass_into=(fs.mean_corr, fs.disc) if (fs.fused_prepass and t == "radiance") else None))

        def accumulate_range(start, count, rows=None):      # noqa: F811
            api.accumulate_tiles(W, H, t_sts, t_bounds, t_offs, t_cnt)
            fs._prepass_current = fs._prepass_key() if fs.fused_prepass else None

    # Row-strip grids: the rows a neighbour needs are accumulated, pre-passed and sent first, the rest of the block is
    # accumulated while they travel (BlockPipeline.border_rows; same bits).  On one GPU the split costs 0.04 - 0.06 ms per
    # step (tools/experiments/halo_overlap_cost.py) and leaves the exchange 0.4 - 1.9 ms to hide in.
    border = pipe.border_rows() if (world > 1 and args.overlap_halo) else []

    def step(record, probe=None, overlapped=True):
        if args.schedule == "reference":
            fs.reset()                  # a progressive render starts from empty statistics (statpath.cpp:173-190)
        pos = 0
        block = None
        for b in batches:
            if border and overlapped:
                # border chain (both strips in one launch, pre-pass + pack, sends and receives issued) on this stream; the
                # rest of the block beside it on the pipeline's side stream.  Keys of the overlapped order: border_chain,
                # interior_accumulate / interior_prepass (side stream), interior_exposed (how long this stream still waits for
                # the side stream once the border chain is done), exchange_exposed (... and then for the receives).
                acc = lambda rows, pos=pos, b=b: accumulate_range(pos, b, rows)
                in_flight = timed("border_chain", record, pipe.border_first, acc)
                names = {"accumulate": "interior_accumulate", "prepass": "interior_prepass"}
                pipe.interior_beside(acc, timed=lambda name, fn, *a: timed(names[name], record, fn, *a))
                pos += b
                timed("interior_exposed", record, pipe.join_side)
                timed("exchange_exposed", record, in_flight.wait)
            else:
                timed("accumulate", record, accumulate_range, pos, b)
                pos += b
                if probe is not None:
                    probe("accumulate")
                timed("prepass", record, pipe.prepass)
                if world > 1:
                    timed("halo", record, pipe.exchange)
            if os.environ.get("STATMC_BENCH_SKIP_FILTER") == "1":      # experiment (tools/experiments: what follows the filter costs the accumulation)
                block = fs.film_f
                timed("filter", record, lambda: None)
            else:
                block = timed("filter", record, pipe.window_filter)
            if probe is not None:
                probe("filter")
            if world > 1 and args.gather:
                timed("gather", record, pipe.gather_film, block, film_f)
        return block

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    # every buffer of the step is dealt by now (the library's workspaces came with the warm-up's first filter): the slots of the
    # classes nobody asked for go back to the driver, so that the line's `placement` is what the step holds, not what the search met
    if PLACED["on"] and not os.environ.get("STATMC_BENCH_NO_TRIM"):
        try:
            PLACED["trimmed"] = api.placement_trim()
        except api.StatmcError as e:
            PLACED["trim_error"] = str(e)[-200:]
        barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    cdev = dev if args.backend == "nccl" else "cpu"
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    variant = api.last_filter_variant()
    n_iter = len(batches)
    # per-STEP totals of every stage (a reference-schedule step launches each stage n_iter times)
    ms = {k: (sum(a.elapsed_time(b) for a, b in v) / args.steps) for k, v in k_events.items()}

    # ---- outside the timed region: film-f assembled on rank 0 (SURVEY 8e "final gather", 12 B/px), and the shader
    # clock the chip holds right behind the two big kernels (one wave counting shader cycles against the 100 MHz clock)
    gather_ms = None
    if world > 1:
        block = pipe.window_filter()
        times = []
        for _ in range(5):
            barrier()
            g0 = time.perf_counter()
            pipe.gather_film(block, film_f)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - g0) * 1e3)
        t = torch.tensor([sorted(times)[len(times) // 2]], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        gather_ms = float(t.item())
    # The shader clock the chip HOLDS while a kernel runs: one wave on a side stream counts shader clocks (s_memtime) against
    # the constant 100 MHz clock (s_memrealtime) beside the kernel -- released together with it by an event recorded in
    # front of the kernel, it gets a CU as soon as one of the kernel's workgroups retires and counts for about the
    # kernel's duration.  (A probe enqueued BEHIND the kernel reads the idle clock: 2.3 - 2.4 GHz whatever ran before.)
    clocks = {}
    slots = torch.zeros(2 * 8, 2, dtype=torch.int64, device=dev)
    idx = {"accumulate": 0, "filter": 0}
    side = torch.cuda.Stream(device=dev)
    main_stream = torch.cuda.current_stream()
    n_acc_launches = sum(len(pool_slices(sum(batches[:i]), b, pool)) for i, b in enumerate(batches))
    acc_ms_total = ms["interior_accumulate"] if border else ms["accumulate"]
    est_ms = {"accumulate": acc_ms_total / max(1, n_acc_launches), "filter": ms["filter"] / n_iter}
    import ctypes
    C_void = ctypes.c_void_p

    def probed(name, fn, *a):
        i = idx[name]
        if rank != 0 or i >= 8:
            return fn(*a)
        idx[name] = i + 1
        go = torch.cuda.Event()
        go.record(main_stream)
        side.wait_event(go)
        cycles = int(min(max(est_ms[name] * 0.8 * 2.0e6, 5e4), 1.6e7))       # ~80 % of the kernel's time at 2 GHz
        api.clock_probe(slots[(0 if name == "accumulate" else 8) + i], cycles=cycles, stream=C_void(side.cuda_stream))
        return fn(*a)

    def probe_step():
        if args.schedule == "reference":
            fs.reset()
        pos = 0
        for b in batches:
            probed("accumulate", accumulate_range, pos, b)
            pos += b
            pipe.prepass()
            if world > 1:
                pipe.exchange()
            block = probed("filter", pipe.window_filter)
            if world > 1 and args.gather:
                pipe.gather_film(block, film_f)
    for _ in range(max(1, 8 // n_iter)):     # every rank steps (the halo exchange is collective); rank 0 probes
        probe_step()
    torch.cuda.synchronize()
    if rank == 0:
        clocks = read_clock_slots(slots, idx)
    if world > 1:
        barrier()

    # ---- the overlapped order against the plain one, on this run's own backend (the asynchronous nccl branch
    # -- border chain on this stream, interior on the side stream, sends and receives in flight until the join -- is
    # otherwise never compared with anything): from empty statistics, three steps each way, the filtered blocks must be
    # the same bits on every rank.
    self_check = None
    dump_block = None
    if world > 1:
        outs = []
        for overlapped in (True, False):
            fs.reset()
            blk = None
            for _ in range(1 if args.schedule == "reference" else 3):
                blk = step(False, overlapped=overlapped)
            torch.cuda.synchronize()
            outs.append(blk.clone())
            if overlapped:
                dump_block = outs[0]
        same = torch.tensor([1.0 if torch.equal(outs[0], outs[1]) else 0.0], device=cdev)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        self_check = {"overlapped_vs_plain_order": "bit-identical" if float(same.item()) == 1.0 else "DIFFERENT",
                      "ranks": world, "backend": args.backend, "overlapped_order_used": bool(border),
                      "steps_each": 1 if args.schedule == "reference" else 3}
    # secondary, N > 1: the block's window filter with all four feature types as G-buffers (eight feature planes: a
    # 17-channel block + halo image through the same pack / exchange / filter path), on 8 samples of the pool
    eight = None
    if world > 1 and args.channels == 11 and not args.no_host_legs:
        # (no collective inside the try: a rank that fails here must not leave the others waiting; the halo of the 17-channel
        # image stays zero -- the filter's time does not depend on what the halo holds)
        names = ("materialid", "depth", "normal", "albedo")
        local_ms, info, err8 = -1.0, {}, None
        try:
            pipe8 = pipeline.BlockPipeline(layout, dev, types, filter_sd=args.filtersd, radius=r, via_host=args.backend == "gloo", g_buffers=names)
            pipe8.accumulate({t: v[:min(8, pool)] for t, v in samples.items()})
            pipe8.prepass()
            for _ in range(2):
                pipe8.window_filter()
            torch.cuda.synchronize()
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(10):
                pipe8.window_filter()
            e1.record()
            torch.cuda.synchronize()
            local_ms = e0.elapsed_time(e1) / 10
            info = {"packed_channels": int(pipe8.packed.shape[2]), "filter_variant": api.last_filter_variant()}
            del pipe8
        except Exception as e:      # noqa: BLE001
            err8 = "%s: %s" % (type(e).__name__, str(e)[-300:])
        t8 = torch.tensor([local_ms], dtype=torch.float64, device=cdev)
        dist.all_reduce(t8, op=dist.ReduceOp.MAX)
        if err8 is not None:
            eight = {"error": err8}
        else:
            eight = dict({"g_buffers": list(names), "feature_channels": 8, "avg_ms": round(float(t8.item()), 4), "block": "%dx%d" % (W, H),
                          "what": "window filter of one block + halo image with eight feature planes (halo left empty), back to back; max over ranks"}, **info)
    if args.dump_film_f:
        if world == 1:
            fs.reset()
            blk = None
            for _ in range(1 if args.schedule == "reference" else 3):
                blk = step(False)
            torch.cuda.synchronize()
            import numpy as np
            np.save(args.dump_film_f, blk.cpu().numpy())
        else:
            whole = pipe.gather_film(dump_block, film_f)
            torch.cuda.synchronize()
            if rank == 0:
                import numpy as np
                np.save(args.dump_film_f, whole.cpu().numpy())

    result = None
    if rank == 0:
        ctx = dict(args=args, world=world, W=W, H=H, fw=fw, fh=fh, gx=layout.gx, gy=layout.gy, S=S, r=r, types=types,
                   batches=batches, pool=pool, n_acc_launches=n_acc_launches, ms=ms, elapsed=elapsed, variant=variant,
                   binding=binding, n_ranks_seen=n_ranks_seen, clocks=clocks, overlapped=bool(border), gather_ms=gather_ms,
                   backend=args.backend, border_rows=sum(y1 - y0 for y0, y1 in border), self_check=self_check, fed_by_tiles=fed_by_tiles,
                   fused_prepass=bool(fs.fused_prepass),
                   parallelism="film blocks x%d, one process per GPU, halo exchange over torch.distributed (%s%s)"
                               % (world, args.backend, " = RCCL" if args.backend == "nccl" else ", halos via the host") if world > 1 else "single GPU")
        result = build_result(ctx)
        if eight is not None:
            result["filter_8_feature_channels"] = eight
        # (before the host-side legs: their page-locked staging buffers and band pipelines keep threads and memory busy)
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(args, fs, samples, types)
            except Exception as e:          # noqa: BLE001
                result["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, str(e)[-300:])}
        if world == 1 and not args.no_host_legs:
            # secondary measurements, outside `value`: the reference's own `CUDA time` bracket through the C++ host
            # side, the tile-fed accumulation, and the raw host <-> device copy rates
            # (a failure in one of them must not cost the headline line: it is reported in place)
            def leg(fn, *a):
                try:
                    return fn(*a)
                except Exception as e:      # noqa: BLE001
                    return {"error": "%s: %s" % (type(e).__name__, str(e)[-300:])}
            ab = leg(accumulate_unplaced_ab, fs, samples, types)
            result["accumulate_placement_ab"] = ab
            if isinstance(ab, dict) and "unplaced_ms" in ab:      # the keys VERDICT r5 item 3 names
                result["kernels"]["accumulate"]["unplaced_ms"] = ab["unplaced_ms"]
                result["kernels"]["accumulate"]["unplaced_frac_hbm"] = ab["unplaced_frac_hbm"]
                result["kernels"]["accumulate"]["placed_back_to_back_ms"] = ab["placed_ms"]
            if args.schedule == "single" and not fed_by_tiles:
                result["reference_schedule"] = leg(reference_schedule_leg, fs, samples, types, args)
            if (args.width, args.height) == (1920, 1080):
                result["config1_1280x720_64spp"] = leg(config_step_leg, 1280, 720, 64, dev, types, args, "configs[1] 1280x720, 64 spp", 10)
            result["cuda_time_bracket"] = leg(host_bracket, fs, args)
            result["tile_fed_accumulate"] = leg(tile_fed_accumulate, fs, samples, types)
            result["accumulate_by_batch"] = leg(accumulate_by_batch, fs, samples, types)
            result["accumulate_by_batch_3840x2160"] = leg(accumulate_by_batch_4k, args, dev, types)
            if isinstance(result["accumulate_by_batch_3840x2160"], dict) and "config4_step" in result["accumulate_by_batch_3840x2160"]:
                result["config4_3840x2160_64spp_one_gpu"] = result["accumulate_by_batch_3840x2160"].pop("config4_step")
            result["host_copies"] = leg(host_copy_times, fs, dev)
            result["filter_8_feature_channels"] = leg(eight_channel_filter, fs, args)
            result["filter_float_buffers"] = leg(float_buffer_filter, fs, args)
            result["pcie_inclusive"] = leg(pcie_inclusive, fs, samples, types, args)
    if run_done is not None:
        run_done.set()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        result["bench_wall_s"] = round(time.perf_counter() - t_process, 1)      # this process, imports and every secondary leg included
        print(json.dumps(result), flush=True)


def block_samples(args, layout, dev, types, rank, world, share=1):
    """The resident sample pool of block `rank` on `dev`, generated in place (seeded: the same stream whichever backend
    drives the block).  share = blocks that share the device's memory."""
    from statmc_amd import api, synthetic
    W, H = layout.bw, layout.bh
    fw, fh = layout.film_size
    ox, oy = layout.origin
    scene = synthetic.Scene(W, H, n_regions=min(12 * world, 32), seed=1, device=dev, x_offset=ox, y_offset=oy,
                            full_width=fw, full_height=fh)
    chunk = 32
    bytes_per_spp = 4 * args.channels * W * H
    pool = args.pool_spp
    if pool <= 0:
        free_b = torch.cuda.mem_get_info(dev)[0] // max(1, share - rank if share > 1 else 1)
        # (placed arenas keep slots of the classes they cannot use backed and idle: up to half as much again)
        fit = int((0.4 if PLACED["on"] else 0.6) * free_b / bytes_per_spp)
        pool = args.spp if fit >= args.spp else max(chunk, fit // chunk * chunk)
    pool = min(pool, args.spp)
    samples = None
    if PLACED["on"]:
        try:
            # (the arenas' total, announced: ONE class is then searched for all of them, not arena by arena)
            api.placement_expect(api.MEM_STREAM, sum(4 * pool * H * W * synthetic.CHANNELS[t] for t in types), dev)
            samples = {t: new_arena((pool, H, W, synthetic.CHANNELS[t]), dev) for t in types}
        except Exception as e:      # noqa: BLE001  (placement is an optimisation: report and go on without)
            PLACED.update(on=False, error="%s: %s" % (type(e).__name__, str(e)[-300:]))
            samples = None
        finally:
            try:
                api.placement_expect(api.MEM_STREAM, 0, dev)      # (what was announced and not asked for is withdrawn)
            except Exception:      # noqa: BLE001
                pass
    if samples is None:
        samples = {t: torch.empty((pool, H, W, synthetic.CHANNELS[t]), dtype=torch.float32, device=dev) for t in types}
    for s0 in range(0, pool, chunk):
        part = scene.samples(min(chunk, pool - s0), seed=1000 * (rank + 1) + s0, features=types)
        for t in types:
            samples[t][s0:s0 + part[t].shape[0]] = part[t]
        del part
    return samples, pool


def read_clock_slots(slots, idx):
    clocks = {}
    try:
        sl = slots.cpu().numpy().astype("float64")
        for name, base in (("accumulate", 0), ("filter", 8)):
            rows = sl[base:base + idx[name]]
            rows = rows[rows[:, 1] > 0]
            if len(rows):
                ghz = rows[:, 0] / rows[:, 1] * 0.1       # cycles per 10 ns tick
                clocks["during_" + name + "_GHz"] = round(float(sorted(ghz)[len(ghz) // 2]), 3)
        clocks["how"] = "one wave on a side stream counting s_memtime against s_memrealtime beside the kernel (median of %d)" % idx["filter"]
    except Exception as e:      # noqa: BLE001
        clocks = {"error": repr(e)[:200]}
    return clocks


def placement_report():
    """Where the timed step's buffers came from (include/statmc.h: statmc_malloc_placed)."""
    out = {"requested": PLACED["on"] or PLACED["error"] is not None or "check" in PLACED, "error": PLACED["error"]}
    if "check" in PLACED:
        out["check"] = PLACED["check"]
        out["timed_region_ran_on"] = "placed blocks" if PLACED["on"] else "torch's allocator"
        if not PLACED["on"]:
            out["map_after_release"] = None
            try:
                from statmc_amd import api
                out["map_after_release"] = api.placement_info()["map"]
            except Exception:      # noqa: BLE001
                pass
    if PLACED["on"]:
        try:
            from statmc_amd import api
            info = api.placement_info()
            out.update(active=bool(info["active"]), slots=info["slots"], probes=info["probes"], slots_a=info["slots_a"], slots_b=info["slots_b"],
                       slots_c=info["slots_c"], slots_unclear=info["slots_unclear"], slots_idle=info["slots_idle"],
                       slots_as_they_came=info["slots_as_they_came"], probe_ms=[round(info["fast_probe_ms"], 4), round(info["slow_probe_ms"], 4)],
                       state_GiB=round(info["slab_bytes"][0] / 2 ** 30, 1), stream_GiB=round(info["slab_bytes"][1] / 2 ** 30, 1), map=info["map"],
                       slots_released=info["slots_released"], trimmed_before_timing=PLACED.get("trimmed"), trim_error=PLACED.get("trim_error"),
                       peak_slots=info["peak_slots"], rebased=bool(info["rebased"]),
                       state_live_GiB=round(info["live_bytes"][0] / 2 ** 30, 2), stream_live_GiB=round(info["live_bytes"][1] / 2 ** 30, 2),
                       budget="3 x the bytes asked for + 6 GiB (STATMC_PLACEMENT_MAX_GIB=%s)" % os.environ.get("STATMC_PLACEMENT_MAX_GIB", "unset"),
                       what="running moments in GiB slots of class A, sample arenas in ONE of the other two classes -- the one the card has at hand -- (a stream read beside writes into its own "
                            "class runs ~ 9 % slower on MI355X; the class travels with the physical memory -- most likely its HBM rank -- and is measured per GiB, 0.2 ms each)")
            # where the window filter's workspace (patch sums) lives: a block of the state role, or plain hipMalloc memory (role -1)
            import ctypes as C
            lib = api.load()
            wp, wb = C.c_void_p(), C.c_size_t()
            lib.statmc_debug_last_workspace.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
            lib.statmc_debug_placement_role.restype = C.c_int
            lib.statmc_debug_placement_role.argtypes = [C.c_void_p]
            if lib.statmc_debug_last_workspace(C.byref(wp), C.byref(wb)) == 0 and wp.value:
                out["filter_workspace"] = {"MiB": round(wb.value / 2 ** 20, 1), "role": int(lib.statmc_debug_placement_role(wp))}
        except Exception as e:      # noqa: BLE001
            out["info_error"] = repr(e)[:200]
    return out


def build_result(c):
    """The JSON line from what a leg measured (one process per GPU, or one process for all of them)."""
    args, world, W, H, fw, fh, S, r = c["args"], c["world"], c["W"], c["H"], c["fw"], c["fh"], c["S"], c["r"]
    types, batches, pool, ms, variant = c["types"], c["batches"], c["pool"], c["ms"], c["variant"]
    n_iter, n_acc_launches, overlapped = len(batches), c["n_acc_launches"], c["overlapped"]
    px_block = W * H
    ms_per_step = c["elapsed"] * 1e3 / args.steps
    value = fw * fh * args.steps / c["elapsed"] / 1e6       # whole job: every block's pixels per step time
    n_flt = n_iter
    flt_gbs = FILTER_BYTES_PER_PX * px_block * n_flt / (ms["filter"] * 1e-3) / 1e9
    # (fused pre-pass: every launch also writes the radiance type's mean-corr and discriminator, 24 B/px)
    fused = bool(c.get("fused_prepass"))
    acc_bytes_px = sum(accumulate_bytes_per_px(b, types) + (24 if fused else 0) for b in batches)     # every launch re-reads and re-writes the state
    acc_bpp = accumulate_bytes_per_px(S, types) + (24 if fused else 0)
    # the launch the roofline line describes: the whole block's accumulation, or -- in the overlapped order -- the
    # interior's (the rows that need no neighbour; the two border strips run beside its first moments)
    acc_px = W * (H - c["border_rows"]) if overlapped else px_block
    acc_ms = ms["interior_accumulate"] if overlapped else ms["accumulate"]
    acc_gbs = acc_bytes_px * acc_px / (max(acc_ms, 1e-9) * 1e-3) / 1e9
    pre_ms = ms["interior_prepass"] if overlapped else ms["prepass"]
    pre_gbs = (160 if world > 1 else PREPASS_BYTES_PER_PX) * acc_px * n_iter / (max(pre_ms, 1e-6) * 1e-3) / 1e9
    taps = (2 * r + 1) ** 2
    # fp32 VALU work of the window filter in lane-operations (a packed op = 2, v_exp_f32 = 4: quarter rate).
    # Pair-symmetric kernel: every unordered pair once, 29 for weight + gate and 4 + 4 for the two accumulations
    # = 37 per pair; one-sided kernels: 33 per directed tap.
    sym = variant.startswith("sym")
    lane_ops = (37.0 * (taps - 1) / 2 + 33.0) if sym else 33.0 * taps
    valu_rate = lane_ops * px_block * n_flt / (ms["filter"] * 1e-3) / 1e12
    traffic = {}
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    # the committed PMC figures were collected on the default workload only
    default_cfg = (W, H, S, args.channels, r, args.schedule) == (1920, 1080, 256, 11, 20, "single")
    if default_cfg and os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath))
        except Exception:
            traffic = {}
    traffic_source = None
    if traffic:
        traffic_source = ("profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on another "
                          "box (%s) -- a committed figure, NOT measured in this run" % str(traffic.get("_note", ""))[:160])
    # counter-based VALU figure (profiles/: SQ_INSTS_VALU per launch of the same command, and the instruction mix of
    # the compiled loop): SIMD issue cycles the instructions need at the 2.4 GHz peak clock / the measured time
    valu_counter = None
    vc = traffic.get("window_filter_valu") if isinstance(traffic, dict) else None
    if vc and sym:
        cyc = vc["SQ_INSTS_VALU"] * vc["cycles_per_inst"] / (256 * 4)
        bound_ms = cyc / 2.4e6
        valu_counter = {"SQ_INSTS_VALU_per_launch": vc["SQ_INSTS_VALU"], "issue_cycles_per_inst": vc["cycles_per_inst"],
                        "alu_pass_bound_ms": round(bound_ms, 4), "frac": round(bound_ms / (ms["filter"] / n_flt), 4),
                        "source": vc.get("source")}
    pred = None
    ppath = os.path.join(ROOT, "profiles", "block_step.json")
    if world > 1 and os.path.exists(ppath) and (fw, fh, S, args.channels, r, args.schedule, args.grid) == (1920, 1080, 256, 11, 20, "single", "rows"):
        try:
            bs = json.load(open(ppath))
            row = bs["per_rank_step_ms"].get(str(world))
            if row is not None:
                pred = {"per_rank_step_ms_without_exchange": row, "n1_step_ms": bs["per_rank_step_ms"]["1"],
                        "fraction_of_ideal": round(bs["per_rank_step_ms"]["1"] / world / row, 3),
                        "measured_over_predicted": round(ms_per_step / row, 3), "source": bs.get("source")}
        except Exception:
            pred = None
    kernels = {
        "filter": {"ms_per_step": round(ms["filter"], 4), "mpixels_per_s": round(px_block * n_flt / ms["filter"] / 1e3, 2)},
    }
    if overlapped:
        # the overlapped order (border rows first, the exchange behind the rest of the accumulation) has keys of its own:
        # nothing here shares a name with the plain order's stages
        kernels["border_chain"] = {"ms_per_step": round(ms["border_chain"], 4), "rows": c["border_rows"],
                                   "what": "accumulate + pre-pass/pack of the rows a neighbour needs (one launch each for both strips), "
                                           "the exchange issued"}
        kernels["interior"] = {"accumulate_ms_per_step": round(ms["interior_accumulate"], 4), "prepass_ms_per_step": round(ms["interior_prepass"], 4),
                               "rows": H - c["border_rows"], "achieved_GBs": round(acc_gbs, 1), "frac_hbm": round(acc_gbs / HBM_PEAK_GBS, 4),
                               "what": "the rest of the block on a side stream, beside the border chain and the exchange"}
        kernels["interior_exposed"] = {"ms_per_step": round(ms["interior_exposed"], 4),
                                       "what": "how long the main stream still waits for the interior once its own chain (border rows, exchange issued) is done"}
        kernels["exchange_exposed"] = {"ms_per_step": round(ms["exchange_exposed"], 4),
                                       "what": "how long it then still waits for the halo rows to land (0 = the exchange hid behind the interior)"}
    else:
        kernels["accumulate"] = {"ms_per_step": round(ms["accumulate"], 4), "bytes_per_px": acc_bytes_px,
                                 "achieved_GBs": round(acc_gbs, 1), "frac_hbm": round(acc_gbs / HBM_PEAK_GBS, 4)}
        if fused:
            kernels["prepass"] = {"ms_per_step": 0.0, "launches_per_step": 0, "fused_into": "accumulate_kernel's epilogue (statmc_stat_type::mean_corr / "
                                  "discriminator: + 24 B/px written, counted in the accumulation's bytes; the same bits as statmc_prepass)",
                                  "host_call_ms_per_step": round(ms["prepass"], 4)}
        else:
            kernels["prepass"] = {"ms_per_step": round(ms["prepass"], 4), "bytes_per_px": 160 if world > 1 else PREPASS_BYTES_PER_PX,
                                  "achieved_GBs": round(pre_gbs, 1), "frac_hbm": round(pre_gbs / HBM_PEAK_GBS, 4)}
        if world > 1:
            kernels["halo_exchange"] = {"ms_per_step": round(ms["halo"], 4), "order": "after the whole block's accumulation and pre-pass"}
    result = {
        "metric": "denoised_mpixels_per_s", "value": round(value, 3), "unit": "Mpixels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": args.scaling if world > 1 else "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "n_ranks_seen": c["n_ranks_seen"], "backend": c["backend"] if world > 1 else "single",
        "config": {
            # (the driver keeps 120 characters of this string: the facts that must survive come first and are keys of their own below)
            "workload": "StatMC acc+prepass+filter %dx%d %dspp %dch r=%d sd=%g %s%s"
                        % (fw, fh, S, args.channels, r, args.filtersd,
                           "configs[4]" if (fw, fh, S) == (3840, 2160, 1024) else "configs[1]" if (fw, fh, S) == (1280, 720, 64) else "configs[2]",
                           "" if args.schedule == "single" else " ref-schedule %dit" % n_iter),
            "workload_long": "StatMC accumulate+prepass+filter, %dx%d film in %d block(s) of %dx%d, %d spp, %d-channel samples, "
                             "filterradius %d, filtersd %g, G-buffers normal(sd 0.1)+albedo(sd 0.02) [synthetic stream]%s"
                             % (fw, fh, world, W, H, S, args.channels, r, args.filtersd,
                                "" if args.schedule == "single" else
                                "; reference schedule: %d iterations of %s samples, pre-pass + filter after each, statistics reset per step"
                                % (n_iter, ",".join(str(b) for b in batches))),
            "radius": r, "filter_sd": args.filtersd, "g_buffers": ["normal", "albedo"], "g_sds": [0.1, 0.02],
            "film": "%dx%d" % (fw, fh), "block_grid": "%dx%d" % (c["gx"], c["gy"]),
            "spp": S, "sample_channels": args.channels, "filter_variant": variant,
            "schedule": args.schedule, "iterations_per_step": n_iter,
            "resident_pool_spp": pool, "accumulate_launches_per_step": n_acc_launches,
            "gather_in_step": bool(args.gather and world > 1),
            "parallelism": c["parallelism"],
            "step_order": ("border rows first, exchange behind the interior's accumulation" if overlapped else
                           "accumulate, pre-pass, exchange, filter" if world > 1 else
                           "accumulate (+ pre-pass in its epilogue), filter" if fused else "accumulate, pre-pass, filter"),
            "rank0_binding": c["binding"],
            "feed": "16 x 16 tile blocks (statmc_accumulate_tiles)" if c.get("fed_by_tiles") else "film-major planes (statmc_accumulate)",
        },
        # the kernel that dominates the step: the sample-stream accumulation (HBM-bound)
        "roofline": {
            "kernel": "accumulate_tiles_kernel" if c.get("fed_by_tiles") else "accumulate_kernel", "bound": "hbm",
            "achieved": round(acc_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(acc_gbs / HBM_PEAK_GBS, 4),
            "traffic": None if c.get("fed_by_tiles") else traffic.get("accumulate_kernel"), "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": acc_bytes_px * acc_px // n_acc_launches, "bytes_per_px": acc_bpp,
            "avg_launch_ms": round(acc_ms / n_acc_launches, 4), "launches_per_step": n_acc_launches,
            "launch": ("the interior rows of the block (%d of %d)" % (H - c["border_rows"], H)) if overlapped else "the whole block",
        },
        # the kernel BASELINE.json's metric names.  It is a (2r+1)^2-tap fp32 stencil: three orders of magnitude above
        # the machine balance, bound by the fp32 VALU issue rate -- no MFMA, it is not a contraction -- so the
        # fraction that means something is the VALU one; the (necessarily tiny) HBM figures the metric asks for
        # ride along under "hbm".
        "roofline_filter": {
            "kernel": ("window_filter_sym + combine_sym_kernel (%s)" if sym else "window_filter_lds<%d> + combine_parts_kernel (%%s)" % r) % variant,
            "bound": "valu",
            "achieved": round(valu_rate, 2), "peak": VALU_PEAK_TOPS, "unit": "T fp32 lane-ops/s",
            "frac": round(valu_rate / VALU_PEAK_TOPS, 4), "lane_ops_per_px": round(lane_ops, 1),
            "note": "peak = 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz; a packed op counts 2, v_exp_f32 4",
            "counter": valu_counter,
            "avg_launch_ms": round(ms["filter"] / n_flt, 4), "launches_per_step": n_flt,
            "hbm": {"achieved": round(flt_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(flt_gbs / HBM_PEAK_GBS, 5),
                    "traffic": traffic.get("window_filter_sym" if sym else "window_filter_lds"), "traffic_source": traffic_source,
                    "algorithmic_bytes_per_launch": FILTER_BYTES_PER_PX * px_block, "bytes_per_px": FILTER_BYTES_PER_PX},
        },
        "shader_clock": c["clocks"],
        "kernels": kernels,
        "placement": placement_report(),
    }
    if os.environ.get("STATMC_VARIANT") or os.environ.get("STATMC_BENCH_ACC_RESIDENT") or os.environ.get("STATMC_BENCH_ACC_DMA"):
        result["experiment"] = {k: os.environ[k] for k in ("STATMC_VARIANT", "STATMC_BENCH_ACC_RESIDENT", "STATMC_BENCH_ACC_DMA") if k in os.environ}
    if world > 1:
        g = c["gather_ms"]
        result["gather_ms"] = round(g, 4)
        result["gather"] = {"ms": round(g, 4), "in_step": bool(args.gather), "bytes": 12 * fw * fh,
                            "what": "film-f blocks assembled on rank 0 / device 0 (median of 5)"}
        if args.gather:
            result["kernels"]["gather"] = {"ms_per_step": round(ms["gather"], 4)}
        result["block_step_prediction"] = pred
        result["overlap_self_check"] = c["self_check"]
    for k in ("host_enqueue_ms_per_step", "devices"):
        if k in c:
            result[k] = c[k]
    return result


def main_peer(args):
    """--backend peer: ONE process drives all N devices through the C ABI -- per block statmc_accumulate_row_ranges ->
    statmc_prepass_pack_rows -> statmc_halo_exchange (device-to-device copies, peer access over xGMI) ->
    statmc_window_filter, in the border-first order of the one-process-per-GPU leg.  No torch.distributed, no RCCL: the
    N > 1 measurement that cannot fail on RCCL bring-up, and the leg the nccl one falls back to."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU (no CPU fallback exists for the product path)")
    from statmc_amd import api, peer, sharding, synthetic
    n = args.gpus
    n_dev = torch.cuda.device_count()
    if args.share_device:
        devices = [0] * n
    else:
        if n_dev < n:
            raise SystemExit("bench.py --gpus %d --backend peer: %d device(s) visible (--share-device rehearses on one)" % (n, n_dev))
        devices = list(range(n))
    S, r = args.spp, args.radius
    types = list(synthetic.FEATURES) if args.channels == 11 else ["radiance", "normal", "albedo"]
    grid = sharding.row_strips(n) if args.grid == "rows" else sharding.grid_for(n)
    if args.scaling == "strong":
        if args.width % grid[0] or args.height % grid[1]:
            raise SystemExit("film %dx%d does not split into a %dx%d grid of equal blocks" % (args.width, args.height, *grid))
        W, H = args.width // grid[0], args.height // grid[1]
    else:
        W, H = args.width, args.height
    # (blocks that share a device -- test setups -- draw on one set of slots: off there unless a test asks for it)
    PLACED["on"] = bool(args.placement) and (not args.share_device or os.environ.get("STATMC_BENCH_PLACED_ON_SHARED_DEVICE") == "1")
    try:
        pf = peer.PeerFilm(n, W, H, r, devices, types, filter_sd=args.filtersd, grid=grid, overlap=args.overlap_halo, placed=PLACED["on"])
    except api.StatmcError as e:
        PLACED.update(on=False, error=str(e)[-300:])
        pf = peer.PeerFilm(n, W, H, r, devices, types, filter_sd=args.filtersd, grid=grid, overlap=args.overlap_halo)
    fw, fh = pf.film_size
    batches = synthetic.sample_schedule(S) if args.schedule == "reference" else [S]
    pools, per_block = [], []
    for b, blk in enumerate(pf.blocks):
        with torch.cuda.device(blk.dev):
            smp, pool = block_samples(args, blk.layout, blk.dev, types, b, n, share=n if args.share_device else 1)
            torch.cuda.synchronize(blk.dev)
        pools.append(pool)
        per_block.append(smp)
    pool = min(pools)
    # one prepared plan per iteration of the schedule (its accumulate launches: the slices of the resident pool)
    def make_plans(overlap=None):
        out, pos = [], 0
        for bsz in batches:
            sl = pool_slices(pos, bsz, pool)
            out.append(pf.prepare_step([[smp if (a, e) == (0, smp[types[0]].shape[0]) else {t: v[a:e] for t, v in smp.items()} for a, e in sl]
                                        for smp in per_block], overlap=overlap))
            pos += bsz
        return out
    plans = make_plans()
    n_acc_launches = sum(len(pool_slices(sum(batches[:i]), bsz, pool)) for i, bsz in enumerate(batches))
    marks = {}

    def timer(b, name, stream):
        if b != 0:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        marks.setdefault(name, []).append(e)

    host_s = [0.0]

    def step(record):
        if args.schedule == "reference":
            pf.reset()
        h0 = time.perf_counter()
        for plan in plans:
            pf.run(plan, timer if record else None)
        host_s[0] += time.perf_counter() - h0

    for _ in range(args.warmup):
        step(False)
    pf.synchronize()
    host_s[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    pf.synchronize()
    elapsed = time.perf_counter() - t0
    host_ms = host_s[0] * 1e3 / args.steps
    variant = api.last_filter_variant()

    def span(a, b):
        return sum(x.elapsed_time(y) for x, y in zip(marks[a], marks[b])) / args.steps
    ms = {k: 0.0 for k in ("accumulate", "prepass", "halo", "filter", "gather", "border_chain", "interior_accumulate", "interior_prepass",
                           "interior_exposed", "exchange_exposed")}
    border_rows = sum(y1 - y0 for y0, y1 in pf.blocks[0].border_rows()) if pf.overlap else 0
    if pf.overlap:
        ms["border_chain"] = span("start", "border")
        ms["exchange_exposed"] = span("border", "exchange")     # on the main stream: neighbours' packs awaited + the copies
        ms["interior_exposed"] = span("exchange", "joined")
        ms["interior_accumulate"] = span("interior_start", "interior_accumulated")
        ms["interior_prepass"] = span("interior_accumulated", "interior")
        ms["filter"] = span("joined", "filter")
    else:
        ms["accumulate"] = span("start", "accumulated")
        ms["prepass"] = span("accumulated", "border")
        ms["halo"] = span("border", "exchange")
        ms["filter"] = span("exchange", "filter")
    # film-f assembled on device 0 (SURVEY 8e "final gather")
    film_f = torch.empty(fh, fw, 3, dtype=torch.float32, device=pf.blocks[0].dev)
    times = []
    for _ in range(5):
        pf.synchronize()
        g0 = time.perf_counter()
        pf.gather(film_f)
        pf.synchronize()
        times.append((time.perf_counter() - g0) * 1e3)
    gather_ms = sorted(times)[len(times) // 2]
    # overlapped against plain order, from empty statistics (same bits)
    self_check, outs = None, []
    for pl in ([plans, make_plans(overlap=False)] if pf.overlap else [plans]):
        pf.reset()
        for _ in range(1 if args.schedule == "reference" else 3):
            if args.schedule == "reference":
                pf.reset()
            for plan in pl:
                pf.run(plan)
        pf.synchronize()
        outs.append(pf.gather().clone())
        pf.synchronize()
    if len(outs) == 2:
        self_check = {"overlapped_vs_plain_order": "bit-identical" if torch.equal(outs[0], outs[1]) else "DIFFERENT", "ranks": n,
                      "backend": "peer", "overlapped_order_used": True, "steps_each": 1 if args.schedule == "reference" else 3}
    if args.dump_film_f:
        import numpy as np
        np.save(args.dump_film_f, outs[0].cpu().numpy())
    eight = None
    if n > 1 and args.channels == 11 and not args.no_host_legs:
        try:
            names = ("materialid", "depth", "normal", "albedo")
            pf8 = peer.PeerFilm(n, W, H, r, devices, types, filter_sd=args.filtersd, grid=grid, overlap=False, g_buffers=names)
            plan8 = pf8.prepare_step([[{t: v[:min(8, pool)] for t, v in smp.items()}] for smp in per_block])
            pf8.run(plan8)
            pf8.synchronize()
            blk0 = pf8.blocks[0]
            c = plan8["filter"][0]
            api.check(pf8.lib.statmc_set_device(blk0.device_index))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(blk0.main)
            for _ in range(10):
                api.check(c.fn(*c.args))
            e1.record(blk0.main)
            pf8.synchronize()
            eight = {"g_buffers": list(names), "feature_channels": 8, "packed_channels": int(blk0.packed.shape[2]),
                     "filter_variant": api.last_filter_variant(), "avg_ms": round(e0.elapsed_time(e1) / 10, 4), "block": "%dx%d" % (W, H),
                     "what": "window filter of block 0's block + halo image with eight feature planes, back to back"}
            del pf8
        except Exception as e:      # noqa: BLE001
            eight = {"error": "%s: %s" % (type(e).__name__, str(e)[-300:])}
    ctx = dict(args=args, world=n, W=W, H=H, fw=fw, fh=fh, gx=pf.gx, gy=pf.gy, S=S, r=r, types=types, batches=batches, pool=pool,
               n_acc_launches=n_acc_launches, ms=ms, elapsed=elapsed, variant=variant,
               binding={"bound": False, "why": "one process drives every device"}, n_ranks_seen=len(pf.blocks), clocks=None,
               overlapped=pf.overlap, gather_ms=gather_ms, backend="peer", border_rows=border_rows, self_check=self_check,
               parallelism="film blocks x%d, ONE process, statmc_halo_exchange (device-to-device copies, peer access over xGMI%s)"
                           % (n, "; all blocks on device 0" if args.share_device else ""),
               host_enqueue_ms_per_step=round(host_ms, 4), devices=devices)
    result = build_result(ctx)
    result["n_gpus"] = n
    if eight is not None:
        result["filter_8_feature_channels"] = eight
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
