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

from conftest import edge_case_stream, make_case  # noqa: E402
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


def accumulate_edge_cases():
    """SURVEY 8c golden set (1): the accumulation's edge cases -- zeros (Box-Cox -> -2), constants (m2 = m3 = 0), one firefly,
    n = 1, ragged counts -- through all twelve (T, transform, maxMoment) variants, planar, in both contraction modes of the
    oracle ("fma_" = what clang -O3 -ffp-contract=on makes of StatTile<Float>, oracle_set_fp_contract)."""
    count, smp = edge_case_stream()
    S, H, W, _ = smp.shape
    out = dict(count=count, samples=smp)
    for contract, prefix in ((False, ""), (True, "fma_")):
        oracle.set_fp_contract(contract)
        for c in (1, 3):
            for transform in (0, 1):
                for moment in (1, 2, 3):
                    arr = {k: np.zeros((H, W, c), np.float32) for k in ("mean", "m2", "m3", "film_mean", "film_m2")}
                    n = np.zeros((H, W), np.int32)
                    for y in range(H):
                        for x in range(W):
                            px = oracle.add_samples_to_pixel(smp[:count[y, x], y, x, :c], c, transform, moment)
                            n[y, x] = int(px["n"])
                            for k in arr:
                                arr[k][y, x] = px[k]
                    key = "%sc%d_t%d_m%d_" % (prefix, c, transform, moment)
                    out[key + "n"] = n
                    for k, v in arr.items():
                        out[key + k] = v
    oracle.set_fp_contract(False)
    np.savez_compressed(os.path.join(HERE, "accumulate_edge_cases.npz"), **out)
    print("accumulate_edge_cases", len(out), "arrays")


def mean_vars_row_quirk():
    """SURVEY 8c golden set (2): CalculateMeanVars on a ProDen-shaped case whose rows do NOT hold one count each -- the case that
    tells the reference's per-row read of n (estimator.cpp:540,558) from the per-pixel one."""
    rng = np.random.default_rng(11)
    H, W = 6, 10
    n = rng.integers(2, 40, (H, W)).astype(np.int32)
    n[2] = 17                                  # one uniform row: both readings agree there
    m2 = (rng.random((H, W, 3)) * 5).astype(np.float32)
    out = dict(n=n, film_m2=m2, film_var_row_quirk=oracle.mean_vars(n, m2, row_n_quirk=True), film_var_per_pixel=oracle.mean_vars(n, m2, row_n_quirk=False))
    np.savez_compressed(os.path.join(HERE, "mean_vars_row_quirk.npz"), **out)
    print("mean_vars_row_quirk", float(np.abs(out["film_var_row_quirk"] - out["film_var_per_pixel"]).max()))


def buffer_catalogue():
    """SURVEY 8c golden set (3): names / types of the registered images and the upload / download sets of the shipped
    configurations, as the C++ host side's AllocateBuffers produces them (tools/bin/statmc_denoise --catalogue; SURVEY App. C
    and tests/test_host_cpu.py say what they must be)."""
    import json
    import subprocess
    from statmc_amd import build
    exe = build.build_tools()
    cat = {}
    for cfg in ("denoise", "acrr", "smis", "proden", "ours"):
        cat[cfg] = subprocess.check_output([exe, "--catalogue", "--config", cfg, "--width", "32", "--height", "16"], text=True).splitlines()
    json.dump(cat, open(os.path.join(HERE, "buffer_catalogue.json"), "w"), indent=1)
    print("buffer_catalogue", {k: len(v) for k, v in cat.items()})


if __name__ == "__main__":
    main()
    accumulate_edge_cases()
    mean_vars_row_quirk()
    buffer_catalogue()
