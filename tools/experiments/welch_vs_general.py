import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, film, synthetic
dev = torch.device("cuda:0")
api.setup(0)
for W in (96, 98, 200):
    for r in (7, 20):
        for dof in (0, 1):
            H, spp = 40, 8
            scene = synthetic.Scene(W, H, seed=3, device=dev, n_regions=4)
            fs = film.FilmStats(W, H, dev, filter_sd=4.0, radius=r)
            fs.accumulate(scene.samples(spp, seed=5, features=("radiance", "normal", "albedo")))
            api.set_filter_spec(dof=dof)
            outs = []
            for force in (0, 1):
                fs.prepass()
                a, keep = fs.filter_args()
                api.force_filter_variant(force)
                api.window_filter(a, 3)
                torch.cuda.synchronize()
                api.force_filter_variant(0)
                outs.append((fs.film_f.clone(), api.last_filter_variant()))
            d = (outs[0][0] - outs[1][0]).abs()
            rel = float(((outs[0][0] - outs[1][0]) ** 2).sum().sqrt() / (outs[1][0] ** 2).sum().sqrt())
            print("W %3d r %2d dof %d  %-12s vs %-8s rel_l2 %.3e  px > 1e-5: %d" % (W, r, dof, outs[0][1], outs[1][1], rel, int((d.amax(dim=2) > 1e-5).sum())), flush=True)
api.set_filter_spec()
