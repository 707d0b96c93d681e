"""Writes the 1080p statistics dump the bracket experiments feed to tools/bin/statmc_denoise: make_dump.py <dir> -> <dir>/scene-32-*.pfm"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from statmc_amd import api, build, film, pfm, synthetic
W, H, spp = 1920, 1080, 32
dev = torch.device("cuda:0"); api.setup(0); build.build_tools()
scene = synthetic.Scene(W, H, seed=1, device=dev)
fs = film.FilmStats(W, H, dev)
fs.accumulate(scene.samples(spp, seed=2, features=("radiance", "normal", "albedo"))); torch.cuda.synchronize()
rad = fs.state["radiance"]
os.makedirs(sys.argv[1], exist_ok=True)
stem = os.path.join(sys.argv[1], "scene")
for name, img in {"film": rad["film_mean"], "t0-b0-n": rad["n"], "t0-b0-mean": rad["mean"], "t0-b0-m2": rad["m2"], "t0-b0-m3": rad["m3"],
                  "t1-b0-film-mean": fs.g_buffer("normal"), "t2-b0-film-mean": fs.g_buffer("albedo")}.items():
    pfm.write_pfm("%s-%d-%s.pfm" % (stem, spp, name), img.cpu().numpy())
print(stem)
