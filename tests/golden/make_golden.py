#!/usr/bin/env python3
"""Writes tests/golden/case_*.npz: small seeded inputs and the outputs of the CPU oracle.

These are SELF-ORACLE vectors (the reference cannot be built or imported in this image, see
DESIGN.md "Oracle"): they freeze the oracle's behaviour so that a later change to oracle/ or to
the HIP kernels shows up as a diff against committed data, and they give the GPU tests a
checker that does not depend on recomputing anything.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from conftest import make_case  # noqa: E402
from oracle import oracle  # noqa: E402

CASES = [
    # name, width, height, spp, radius, filter sd, seed
    ("default_r20", 40, 28, 8, 20, 10.0, 1),
    ("caustics_r6", 37, 21, 16, 6, 3.0, 2),      # scenes/render-denoise-glass-caustics.pbrt:19-20
]


def main():
    for name, w, h, spp, r, sd, seed in CASES:
        _, smp, st = make_case(w, h, spp, seed=seed)
        rad = st["radiance"]
        mc, disc = oracle.prepass(rad["n"], rad["mean"], rad["m2"], rad["m3"])
        out = oracle.filter_image(mc, disc, rad["film_mean"], [st["normal"]["mean"], st["albedo"]["mean"]],
                                  [-0.5 / 0.1 ** 2, -0.5 / 0.02 ** 2], -0.5 / sd ** 2, r)
        np.savez_compressed(
            os.path.join(HERE, "case_%s.npz" % name),
            radius=r, filter_sd=sd, spp=spp,
            samples_radiance=smp["radiance"], samples_normal=smp["normal"], samples_albedo=smp["albedo"],
            n=rad["n"], mean=rad["mean"], m2=rad["m2"], m3=rad["m3"], film_mean=rad["film_mean"],
            film_m2=rad["film_m2"], normal_mean=st["normal"]["mean"], albedo_mean=st["albedo"]["mean"],
            mean_corr=mc, discriminator=disc, film_f=out)
        print(name, out.shape, float(out.mean()))


if __name__ == "__main__":
    main()
