import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import make_case, rel_l2
from oracle import oracle
from statmc_amd import api
api.setup(0)
DEV = torch.device("cuda:0")
G_DR = [-50.0, -1250.0]
def to_dev(a): return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
for (W, H, spp, seed) in ((300, 41, 8, 320), (64, 9, 4, 1), (1000, 70, 4, 2), (257, 5, 4, 3)):
    _, smp, st = make_case(W, H, spp, seed=seed)
    rad = st["radiance"]
    mc, dc = oracle.prepass(rad["n"], rad["mean"], rad["m2"], rad["m3"])
    gbs = [st["normal"]["mean"], st["albedo"]["mean"]]
    ref = oracle.filter_image(mc, dc, rad["film_mean"], gbs, G_DR, -0.005, 20)
    for parts in (0, 1, 2, 3, 21):
        out = torch.zeros(H, W, 3, device=DEV)
        a, keep = api.make_filter_args([], [], [], [], [to_dev(rad["film_mean"])], [to_dev(mc)], [to_dev(dc)], [out],
                                       [to_dev(g) for g in gbs], g_dr=G_DR, filter_sd=10.0, radius=20)
        api.force_filter_parts(parts)
        api.window_filter(a, 3)
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        print(W, H, "parts", parts, api.last_filter_variant(), [rel_l2(o[..., c], ref[..., c]) for c in range(3)], flush=True)
    api.force_filter_parts(0)
