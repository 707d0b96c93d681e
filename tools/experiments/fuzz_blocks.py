"""Randomised differential run of the film-block path (statmc_amd/peer.py: accumulate_row_ranges -> prepass_pack_rows -> halo
exchange -> window filter per block, all through the C ABI) against the same film filtered as ONE block: random grids, block
sizes, radii, G-buffer sets (15- / 16- / 17- / 18-channel block + halo images), filter specs (gates, channel rules, Welch
degrees of freedom, clamped borders), two steps each; the assembled film must equal the whole film bit for bit (the
window-sweep split is pinned on both sides).  All blocks on cuda:0.
usage: fuzz_blocks.py [seconds] [first_case]"""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from statmc_amd import api as gpu, peer, pipeline, sharding, synthetic
DEV = torch.device("cuda:0")
gpu.setup(0)
SETS = [(("radiance", "normal", "albedo"), ("normal", "albedo")),
        (("radiance", "normal", "albedo", "depth", "materialid"), ("materialid", "depth", "normal", "albedo")),
        (("radiance", "normal", "albedo", "depth"), ("depth", "normal", "albedo")),
        (("radiance", "normal", "albedo"), ("albedo",)),
        (("radiance", "normal", "albedo", "depth"), ("normal", "depth"))]


def case(k):
    rng = np.random.default_rng(1000003 * 97 + k)
    radius = int(rng.choice([3, 5, 6, 9, 12, 20, 20]))
    gx, gy = [(1, 2), (2, 1), (2, 2), (1, 3), (3, 1), (1, 4)][int(rng.integers(0, 6))]
    bw = int(rng.integers(max(radius, 8), 40)) * 4
    bh = int(rng.integers(max(radius, 8), 72))
    types, g_buffers = SETS[int(rng.integers(0, len(SETS)))]
    spec = {}
    if rng.random() < 0.7:
        spec = dict(gate=int(rng.integers(0, 2)), channel_rule=int(rng.integers(0, 2)), border=int(rng.integers(0, 2)), dof=int(rng.random() < 0.6))
    split = int(rng.choice([1, 2, 3]))
    world = gx * gy
    scene = synthetic.Scene(gx * bw, gy * bh, n_regions=7, seed=k)
    batches = [scene.samples(int(rng.integers(2, 6)), seed=k + 1, features=types), scene.samples(3, seed=k + 2, features=types)]
    desc = dict(case=k, grid=(gx, gy), bw=bw, bh=bh, radius=radius, g_buffers=g_buffers, spec=spec, split=split)
    gpu.set_filter_split(split)
    gpu.set_filter_spec(**spec)
    try:
        try:
            one = pipeline.BlockPipeline(sharding.BlockLayout(0, 1, gx * bw, gy * bh, radius), DEV, types, radius=radius, filter_sd=radius / 2.0,
                                         g_buffers=g_buffers)
            pf = peer.PeerFilm(world, bw, bh, radius, [0] * world, types, filter_sd=radius / 2.0, grid=(gx, gy), g_buffers=g_buffers)
        except (gpu.StatmcError, ValueError) as e:
            return None, dict(desc, refused=str(e)[:160])
        desc["channels"] = int(pf.blocks[0].packed.shape[2])
        ok = True
        for smp in batches:
            one.accumulate({t: v.to(DEV) for t, v in smp.items()})
            try:
                ref = one.denoise().clone()
                per_block = []
                for blk in pf.blocks:
                    ox, oy = blk.layout.origin
                    per_block.append([{t: v[:, oy:oy + bh, ox:ox + bw].contiguous().to(DEV) for t, v in smp.items()}])
                pf.run(pf.prepare_step(per_block))
                pf.synchronize()
            except gpu.StatmcError as e:
                return None, dict(desc, refused=str(e)[:160])
            desc["variant"] = gpu.last_filter_variant()
            got = pf.gather()
            pf.synchronize()
            ok = ok and torch.equal(got, ref)
        return ok, desc
    finally:
        gpu.set_filter_split(0)
        gpu.set_filter_spec()


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
k = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t_end, last = time.time() + budget, time.time()
n = fails = refused = 0
seen = collections.Counter()
while time.time() < t_end:
    ok, d = case(k)
    k += 1
    n += 1
    if ok is None:
        refused += 1
        print("REFUSED", d, flush=True)
    else:
        seen[(d["channels"], d["variant"])] += 1
        if not ok:
            fails += 1
            print("FAIL", d, flush=True)
    if time.time() - last > 60:
        print("... %d cases, %d failures, %d refused" % (n, fails, refused), flush=True)
        last = time.time()
print("cases %d (next %d), failures %d, refused %d" % (n, k, fails, refused))
print("(channels, variant) exercised:", dict(sorted(seen.items())))
