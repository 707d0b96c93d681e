"""statmc_malloc_placed (include/statmc.h, "Device memory dealt by interference class"): the blocks are ordinary device memory
-- same bits as torch's allocator through the accumulation and the filter --, free / reuse works, the report is consistent,
and the switch-off path is plain hipMalloc."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_placed_blocks_are_ordinary_memory(gpu):
    dev = torch.device("cuda:0")
    a = gpu.empty_placed((3, 5, 7), torch.float32, dev, gpu.MEM_STATE)
    b = gpu.empty_placed((1 << 20,), torch.int32, dev, gpu.MEM_STREAM)
    assert a.data_ptr() % (2 << 20) == 0 and b.data_ptr() % (2 << 20) == 0
    a.copy_(torch.arange(105, dtype=torch.float32, device=dev).view(3, 5, 7))
    b.fill_(7)
    assert float(a.sum().item()) == 105 * 104 / 2 and int(b.sum().item()) == 7 << 20
    info = gpu.placement_info()
    assert info["virtual_memory"] == 1 and info["slots"] >= 3
    assert info["live_bytes"][gpu.MEM_STATE] >= a.numel() * 4 and info["live_bytes"][gpu.MEM_STREAM] >= b.numel() * 4
    assert len(info["map"]) == info["slots"] + info["slots_released"] and info["map"][0] == "#"
    if info["active"]:
        # the state block sits in slot 0 itself, behind the probe's window (what every other slot is measured against), the stream
        # block in a slot of one of the other two classes
        lib = gpu.load()
        lib.statmc_debug_placement_role.restype = C.c_int
        lib.statmc_debug_placement_role.argtypes = [C.c_void_p]
        assert lib.statmc_debug_placement_role(C.c_void_p(a.data_ptr())) == gpu.MEM_STATE
        assert lib.statmc_debug_placement_role(C.c_void_p(b.data_ptr())) == gpu.MEM_STREAM
        assert any(ch in "BC" for ch in info["map"]) and "A" not in info["map"]
        assert info["slow_probe_ms"] > 1.05 * info["fast_probe_ms"]
    # free -> the space is reused by the next block of the same role
    p = a.data_ptr()
    del a
    a2 = gpu.empty_placed((3, 5, 7), torch.float32, dev, gpu.MEM_STATE)
    assert a2.data_ptr() == p
    lib = gpu.load()
    assert lib.statmc_malloc_placed(None, 16, 0) == gpu.ERR_INVALID
    q = C.c_void_p()
    assert lib.statmc_malloc_placed(C.byref(q), 16, 5) == gpu.ERR_INVALID


def test_accumulate_and_filter_on_placed_memory_give_the_same_bits(gpu):
    from statmc_amd import film, synthetic
    dev = torch.device("cuda:0")
    W, H, S = 256, 96, 12
    types = list(synthetic.FEATURES)
    scene = synthetic.Scene(W, H, seed=3, device=dev)
    smp = scene.samples(S, seed=11)
    placed_smp = {t: gpu.empty_placed(tuple(v.shape), torch.float32, dev, gpu.MEM_STREAM) for t, v in smp.items()}
    for t in types:
        placed_smp[t].copy_(smp[t])
    fs_t = film.FilmStats(W, H, dev, types=types, radius=6)
    fs_p = film.FilmStats(W, H, dev, types=types, radius=6, placed=True)
    for _ in range(2):
        fs_t.accumulate(smp)
        fs_p.accumulate(placed_smp)
    out_t, out_p = fs_t.denoise().clone(), fs_p.denoise().clone()
    torch.cuda.synchronize()
    for t in types:
        for k, v in fs_t.state[t].items():
            if v is not None:
                assert torch.equal(v.view(torch.int32), fs_p.state[t][k].view(torch.int32)), (t, k)
    assert torch.equal(out_t.view(torch.int32), out_p.view(torch.int32))
    assert np.isfinite(out_p.cpu().numpy()).all()


def test_random_alloc_free_sequences_keep_blocks_disjoint_and_intact(gpu):
    """300 random allocations and frees in both roles (2 MiB .. 1.4 GiB: blocks that span slots included): live blocks never
    overlap, every block keeps the pattern it was given, freed space is reused (what the roles hold does not grow without bound)."""
    rng = np.random.default_rng(2024)
    dev = torch.device("cuda:0")
    live = {}          # id -> (tensor, role, fill value)
    next_id = 0
    slots_seen = []
    for step in range(300):
        if live and (len(live) > 24 or rng.random() < 0.45):
            k = list(live)[int(rng.integers(len(live)))]
            t, role, val = live.pop(k)
            assert int(t[0].item()) == val and int(t[-1].item()) == val and int(t[t.numel() // 2].item()) == val, (step, k)
            del t
        else:
            role = int(rng.integers(2))
            mib = int(rng.choice([2, 3, 10, 64, 200, 700, 1400], p=[0.25, 0.2, 0.2, 0.15, 0.1, 0.07, 0.03]))
            t = gpu.empty_placed((mib << 18,), torch.int32, dev, role)
            t.fill_(next_id)
            live[next_id] = (t, role, next_id)
            next_id += 1
        if step % 50 == 49:
            spans = sorted((t.data_ptr(), t.data_ptr() + t.numel() * 4) for t, _, _ in live.values())
            assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), "live blocks overlap"
            info = gpu.placement_info()
            assert sum(info["live_bytes"]) >= sum(t.numel() * 4 for t, _, _ in live.values())
            slots_seen.append(sum(info["slab_bytes"]) >> 30)      # GiB dealt to the roles (how many slots had to be BACKED to find
                                                                  # them is the card's business: its classes come in runs)
    for k, (t, role, val) in live.items():
        assert int(t[0].item()) == val and int(t[-1].item()) == val
    # frees are reused: the run allocates ~ 20 GiB in all, holds ~ 3 GiB at any time (a 1.4-GiB block wants two slots side by side)
    assert slots_seen[-1] <= 14, slots_seen


def test_large_blocks_are_windows_over_scattered_slots(gpu):
    """A block above 2 GiB is a window: as many slots of the wanted class as it needs, wherever they lie, mapped side by side a
    second time.  Three 5-GiB stream blocks: ordinary memory (a pattern written through the window reads back, every GiB), in
    slots of ONE class each when the device tells classes apart; the accumulation treats samples in a window and moments in
    state blocks as known to lie apart; freed, the slots are idle again (same memory, same class) and serve the next window."""
    dev = torch.device("cuda:0")
    gib = 1 << 28                                            # int32 elements per GiB
    blocks = [gpu.empty_placed((5 * gib,), torch.int32, dev, gpu.MEM_STREAM) for _ in range(3)]
    for k, t in enumerate(blocks):
        for g in range(5):
            t[g * gib:(g + 1) * gib].fill_(100 * k + g)
    for k, t in enumerate(blocks):
        for g in range(5):
            assert int(t[g * gib].item()) == 100 * k + g and int(t[(g + 1) * gib - 1].item()) == 100 * k + g
    info = gpu.placement_info()
    assert info["live_bytes"][1] >= 3 * 5 * gib * 4
    if info["active"]:
        # in slots of the other classes -- unless the search's byte budget (3 x the bytes asked for, round 6) ran out on a card whose
        # first slots are nearly all of the state's class: then the rest comes as it comes ('T'), and says so
        came = info["slots_as_they_came"][1]
        assert info["map"].count("B") + info["map"].count("C") + info["map"].count("T") >= 15 and info["map"].count("T") == came
        lib = gpu.load()
        lib.statmc_debug_placement_role.restype = C.c_int
        lib.statmc_debug_placement_role.argtypes = [C.c_void_p]
        if came == 0:
            assert lib.statmc_debug_placement_role(C.c_void_p(blocks[1].data_ptr() + (3 << 30))) == gpu.MEM_STREAM     # inside a window
    before = info["slots"]
    del blocks, t
    torch.cuda.synchronize()
    after = gpu.placement_info()
    # (a block served by a run of slots that lay side by side goes back to its role's free list, a window's slots are idle again)
    assert after["slots"] == before and after["live_bytes"][1] <= info["live_bytes"][1] - 3 * 5 * gib * 4
    again = gpu.empty_placed((12 * gib,), torch.int32, dev, gpu.MEM_STREAM)          # 12 GiB: the idle slots first, new ones for the rest
    again[-1:].fill_(7)
    # (whole free slots are idle again and reused; how many MORE the card must back for twelve of one class is its business)
    assert int(again[-1].item()) == 7 and gpu.placement_info()["slots"] <= before + 12


def test_trim_releases_the_idle_slots_and_the_allocator_goes_on(gpu):
    dev = torch.device("cuda:0")
    keep = gpu.empty_placed((1 << 20,), torch.float32, dev, gpu.MEM_STREAM)
    keep.fill_(3.0)
    before = gpu.placement_info()
    n = gpu.placement_trim()
    after = gpu.placement_info()
    assert n == before["slots_idle"] and after["slots_idle"] == 0 and after["map"].count("_") >= n
    assert after["live_bytes"] == before["live_bytes"] and float(keep.sum().item()) == 3.0 * (1 << 20)
    # a block larger than what the stream role holds free needs fresh slots: the holes are backed and probed again first, the range
    # grows only behind them
    gib = (after["slab_bytes"][1] - after["live_bytes"][1]) // (1 << 30) + 2
    big = gpu.empty_placed((gib << 28,), torch.float32, dev, gpu.MEM_STREAM)
    big[: 1 << 20].fill_(1.0)
    assert float(big[: 1 << 20].sum().item()) == float(1 << 20)
    again = gpu.placement_info()
    assert again["slots"] >= after["slots"]
    if after["map"].count("_"):
        assert again["map"].count("_") < after["map"].count("_"), (after["map"], again["map"])
    assert float(keep.sum().item()) == 3.0 * (1 << 20)


def test_free_of_an_interior_pointer_is_an_error_and_role_needs_a_live_block(gpu):
    """statmc_free must not report OK for a pointer inside the allocator's range that is not a live block's start, and
    the role statmc_accumulate learns about a buffer is that of the LIVE block around the address (every slot it covers)."""
    dev = torch.device("cuda:0")
    lib = gpu.load()
    lib.statmc_debug_placement_role.restype = C.c_int
    lib.statmc_debug_placement_role.argtypes = [C.c_void_p]
    p = C.c_void_p()
    assert lib.statmc_malloc_placed(C.byref(p), 8 << 20, gpu.MEM_STREAM) == 0
    inside = C.c_void_p(p.value + (4 << 20))
    assert lib.statmc_free(inside) == gpu.ERR_INVALID            # an interior pointer
    if gpu.placement_info()["active"]:
        assert lib.statmc_debug_placement_role(inside) == gpu.MEM_STREAM
    assert lib.statmc_free(p) == 0
    assert lib.statmc_free(p) == gpu.ERR_INVALID                 # freed twice
    assert lib.statmc_debug_placement_role(inside) == -1         # no live block there any more
    info = gpu.placement_info()
    assert info["peer_devices"] >= 0 and info["peer_devices"] <= torch.cuda.device_count() - 1


def test_the_search_for_a_class_stays_inside_its_byte_budget():
    """the class search backs at most 3 x the bytes asked for (+ 6 GiB) -- not 60 % of the card --, an explicit
    STATMC_PLACEMENT_MAX_GIB is honoured, and trimming leaves no idle slot."""
    code = ("import torch, sys; sys.path.insert(0, %r)\n"
            "from statmc_amd import api\n"
            "api.setup(0)\n"
            "dev = torch.device('cuda:0')\n"
            "st = api.zeros_placed((1 << 24,), torch.float32, dev, api.MEM_STATE)\n"
            "c0 = api.placement_info()['slots']      # what the calibration had to back to see both probe levels (runs of one class can be long)\n"
            "blocks = [api.empty_placed((6 << 28,), torch.float32, dev, api.MEM_STREAM) for _ in range(3)]\n"
            "i = api.placement_info()\n"
            "cap = max(int(sys.argv[1]), c0)\n"
            "assert i['slots'] <= cap, (i['slots'], cap, i['map'])\n"
            "assert i['live_bytes'][1] >= 3 * (6 << 30)\n"
            "n = api.placement_trim(); j = api.placement_info()\n"
            "assert j['slots_idle'] == 0 and j['slots'] == i['slots'] - n and j['slots_released'] == n, (i, j)\n"
            "for b in blocks: b[-1:].fill_(1.0)\n"
            "assert all(float(b[-1].item()) == 1.0 for b in blocks)\n"
            "print('ok', i['slots'], n, i['map'])\n" % ROOT)
    # default budget: 3 x (18 GiB + 64 MiB) + 6 = 61 slots (the calibration may have backed a few more before any class was known)
    out = subprocess.run([sys.executable, "-c", code, "64"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr[-2000:]
    # an explicit budget below the request: the class search backs nothing beyond it, the last resort still serves the request
    out = subprocess.run([sys.executable, "-c", code, "26"], capture_output=True, text=True, timeout=300, env=dict(os.environ, STATMC_PLACEMENT_MAX_GIB="20"))
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr[-2000:]


def test_the_reference_slot_can_trade_places_with_a_slot_of_another_class():
    """Round 6: on a card whose first GiB are a long run of the reference slot's class the allocator maps the first slot apart from it at
    the reference's address (and vice versa) and probes everything again, so that the moments' home is the class the card has least of
    there.  Forced here (STATMC_PLACEMENT_FORCE_REBASE=1: the box at hand may not call for it): the report says so, classes are still
    told apart, a stream block and a state block land in different classes, and accumulation + filter on such memory leave the bits
    they leave on torch's allocator."""
    code = ("import torch, sys, ctypes as C; sys.path.insert(0, %r)\n"
            "from statmc_amd import api, film, synthetic\n"
            "api.setup(0)\n"
            "dev = torch.device('cuda:0')\n"
            "W, H, S = 256, 96, 6\n"
            "types = list(synthetic.FEATURES)\n"
            "smp = synthetic.Scene(W, H, seed=3, device=dev).samples(S, seed=11)\n"
            "fs_p = film.FilmStats(W, H, dev, types=types, radius=6, placed=True)\n"
            "psmp = {t: api.empty_placed(tuple(v.shape), torch.float32, dev, api.MEM_STREAM) for t, v in smp.items()}\n"
            "[psmp[t].copy_(smp[t]) for t in types]\n"
            "big = api.empty_placed((3 << 28,), torch.float32, dev, api.MEM_STREAM)\n"
            "big[-1:].fill_(2.0)\n"
            "i = api.placement_info()\n"
            "fs_t = film.FilmStats(W, H, dev, types=types, radius=6)\n"
            "fs_t.accumulate(smp); fs_p.accumulate(psmp)\n"
            "a, b = fs_t.denoise().clone(), fs_p.denoise().clone()\n"
            "torch.cuda.synchronize()\n"
            "assert torch.equal(a.view(torch.int32), b.view(torch.int32)) and float(big[-1].item()) == 2.0\n"
            "for t in types:\n"
            "    for k, v in fs_t.state[t].items():\n"
            "        assert v is None or torch.equal(v.view(torch.int32), fs_p.state[t][k].view(torch.int32)), (t, k)\n"
            "lib = api.load(); lib.statmc_debug_placement_role.restype = C.c_int; lib.statmc_debug_placement_role.argtypes = [C.c_void_p]\n"
            "if i['active']:\n"
            "    assert i['rebased'] == 1 and i['map'][0] == '#', i\n"
            "    assert lib.statmc_debug_placement_role(C.c_void_p(fs_p.state['radiance']['mean'].data_ptr())) == api.MEM_STATE\n"
            "    if i['slots_as_they_came'][1] == 0:\n"
            "        assert lib.statmc_debug_placement_role(C.c_void_p(big.data_ptr())) == api.MEM_STREAM\n"
            "print('ok', i['active'], i['rebased'], i['map'])\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, STATMC_PLACEMENT_FORCE_REBASE="1"))
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr[-2000:]


def test_switch_off_is_plain_hipmalloc():
    code = ("import torch, sys; sys.path.insert(0, %r)\n"
            "from statmc_amd import api\n"
            "api.setup(0)\n"
            "t = api.zeros_placed((1000,), torch.float32, torch.device('cuda:0'), api.MEM_STATE)\n"
            "i = api.placement_info()\n"
            "assert i['virtual_memory'] == 0 and i['slots'] == 0 and i['active'] == 0, i\n"
            "assert float(t.sum().item()) == 0.0\n"
            "print('ok')\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, STATMC_PLACEMENT="0"))
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-800:]


def test_a_window_at_addresses_another_window_left_reaches_its_own_memory():
    """ROCm 7.2 / gfx950: after hipMemUnmap the shaders keep the old translation of the range until the driver rewrites the page tables
    the ordinary way (tools/microbench/vmm_remap.hip) -- a window block mapped at the addresses a freed window left used to read and
    write the FREED window's slots, which by then belong to somebody else.  The allocator now forces that rewrite after every batch of
    unmaps.  Here: a three-slot window over slots that do not lie side by side is freed, a one-slot block takes its first slot, a second
    window lands at the first one's addresses over other slots -- and the two live blocks must not share a byte
    (tools/experiments/window_reuse_check.py, profiles/r06_window_reuse.log: the library before the fix leaves 268 M wrong values)."""
    code = ("import torch, sys; sys.path.insert(0, %r)\n"
            "from statmc_amd import api\n"
            "api.setup(0)\n"
            "dev = torch.device('cuda:0')\n"
            "G = 1 << 28\n"
            "x = [api.empty_placed((G,), torch.float32, dev, api.MEM_STREAM) for _ in range(6)]\n"
            "for k in (0, 2, 4): x[k] = None\n"
            "a = api.empty_placed((3 * G,), torch.float32, dev, api.MEM_STREAM)\n"
            "a.fill_(1.0); pa = a.data_ptr()\n"
            "torch.cuda.synchronize()\n"
            "del a\n"
            "b = api.empty_placed((G,), torch.float32, dev, api.MEM_STREAM)\n"
            "c = api.empty_placed((3 * G,), torch.float32, dev, api.MEM_STREAM)\n"
            "same_addresses = c.data_ptr() == pa\n"
            "b.fill_(2.0); c.fill_(3.0)\n"
            "torch.cuda.synchronize()\n"
            "for t, v in ((b, 2.0), (c, 3.0)):\n"
            "    for k in range(0, t.numel(), G // 2):\n"
            "        part = t[k:k + G // 2]\n"
            "        assert float(part.min().item()) == v and float(part.max().item()) == v, (v, k, float(part.min().item()), float(part.max().item()))\n"
            "n = api.load().statmc_placement_trim()\n"
            "d = api.empty_placed((3 * G,), torch.float32, dev, api.MEM_STREAM)\n"
            "d.fill_(4.0)\n"
            "torch.cuda.synchronize()\n"
            "for t, v in ((b, 2.0), (c, 3.0), (d, 4.0)):\n"
            "    assert float(t[::4099].min().item()) == v and float(t[::4099].max().item()) == v, v\n"
            "print('ok', same_addresses, n, api.placement_info()['map'])\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr[-2000:]


def test_placement_expect_announces_a_total(gpu):
    """statmc_placement_expect (include/statmc.h): argument checks; an announced total is counted down by the role's allocations and
    changes nothing about the blocks themselves."""
    lib = gpu.load()
    assert lib.statmc_placement_expect(7, 1 << 30) == gpu.ERR_INVALID
    dev = torch.device("cuda:0")
    gpu.placement_expect(gpu.MEM_STREAM, 3 << 30, dev)
    a = gpu.empty_placed((1 << 28,), torch.float32, dev, gpu.MEM_STREAM)
    b = gpu.empty_placed((1 << 28,), torch.float32, dev, gpu.MEM_STREAM)
    a.fill_(1.0)
    b.fill_(2.0)
    torch.cuda.synchronize()
    assert float(a[::4097].sum().item()) == float(a[::4097].numel()) and float(b[::4097].min().item()) == 2.0
    gpu.placement_expect(gpu.MEM_STREAM, 0, dev)
    info = gpu.placement_info()
    assert info["live_bytes"][gpu.MEM_STREAM] >= 2 << 30


def test_random_allocations_and_frees_never_share_memory():
    """A soak of the placed allocator's address handling (windows used again, blocks carved from slots, trims in between): every live
    block is filled with its own number when it is made, and after every operation every live block still holds only that number
    (tools/experiments/alloc_soak.py; 200 steps, 44 windows: profiles/r06_alloc_soak.log)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "experiments", "alloc_soak.py"), "48"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok 48" in out.stdout, out.stdout + out.stderr[-2000:]


def test_headline_shape_takes_the_resident_grid_and_leaves_the_same_bits(gpu):
    """1080p, 256 samples per launch, samples in a STREAM block and moments in STATE blocks: the accumulation runs as one workgroup per
    compute unit (launch_accumulate, round 6) -- the same bits as the large grid, which every other shape keeps."""
    from statmc_amd import film, synthetic
    lib = gpu.load()
    dev = torch.device("cuda:0")
    W, H, S = 1920, 1080, 256
    types = ["radiance"]
    smp = {"radiance": gpu.empty_placed((S, H, W, 3), torch.float32, dev, gpu.MEM_STREAM)}
    scene = synthetic.Scene(W, H, seed=5, device=dev)
    for s0 in range(0, S, 32):
        smp["radiance"][s0:s0 + 32] = scene.samples(32, seed=100 + s0, features=types)["radiance"]
    fs_a = film.FilmStats(W, H, dev, types=types, placed=True)
    fs_b = film.FilmStats(W, H, dev, types=types, placed=True)
    fs_a.accumulate(smp)
    grid_auto = lib.statmc_debug_last_accumulate_grid()
    gpu.accumulate_resident_blocks(-1)
    try:
        fs_b.accumulate(smp)
        grid_large = lib.statmc_debug_last_accumulate_grid()
    finally:
        gpu.accumulate_resident_blocks(0)
    fs_a.accumulate({"radiance": smp["radiance"][:64]})      # a shorter batch keeps the large grid
    grid_short = lib.statmc_debug_last_accumulate_grid()
    fs_b.accumulate({"radiance": smp["radiance"][:64]})
    torch.cuda.synchronize()
    if gpu.placement_info()["active"] and lib.statmc_debug_placement_role(C.c_void_p(smp["radiance"].data_ptr())) == gpu.MEM_STREAM:
        assert grid_auto == lib.statmc_device_cus() and grid_large > 4 * grid_auto and grid_short > 4 * grid_auto, (grid_auto, grid_large, grid_short)
    for k, v in fs_a.state["radiance"].items():
        if v is not None:
            assert torch.equal(v.view(torch.int32), fs_b.state["radiance"][k].view(torch.int32)), k
