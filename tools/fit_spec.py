#!/usr/bin/env python3
"""fit_spec.py -- which of this build's filter specs reproduces a set of dumps of the CUDA StatMC?

The reference's filter arithmetic is not in its tree (stat_denoiser.cu lives in an un-vendored submodule), so every
choice the tree leaves open is a run-time option of this build (statmc_filter_spec: gate form, channel rule, quantile
sides, per-pixel / Welch dof, border policy, n < 2).  Given dumps written by the CUDA build --
`<stem>-<spp>-{film,t0-b0-n,t0-b0-mean,t0-b0-m2,t0-b0-m3,t1-b0-film-mean,t2-b0-film-mean}.pfm` (inputs,
scenes/render-for-ours.pbrt:24) and `<ref>-<spp>-{film-f,t0-b0-mean-corr,t0-b0-discriminator}.pfm` (its outputs) --
this runs every spec x significance level through tools/bin/statmc_denoise --sweep --compare (one process) and prints the per-channel
relative L2 table, best first.  BASELINE.json's bound is 1e-5 per channel.  Every candidate runs on the general
kernel (--kernel general), whose sums are formed in one order for every spec: the differences in the table are the
specs', not the kernels' (the fast kernels agree with it within 1e-6, which is more than rounding-twin specs differ by).

    python tools/fit_spec.py --stem dumps/scene --ref cuda/scene --spp 4,8,16 [--filtersd 10 --filterradius 20]
                             [--tquantiles table.txt]  [--quick]   (--quick: default dof / border / n<2 only)
"""
import argparse
import itertools
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = (("gate", ("sym", "asym", "centre")), ("channels", ("and", "joint")), ("sides", ("two", "one")),
          ("dof", ("pixel", "welch")), ("border", ("clip", "clamp")), ("small_n", ("accept", "exclude")))


def variants(quick):
    free = FIELDS[:3] if quick else FIELDS
    for combo in itertools.product(*[v for _, v in free]):
        yield ",".join("%s=%s" % (k, c) for (k, _), c in zip(free, combo))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stem", required=True)
    ap.add_argument("--ref", required=True, help="stem of the reference outputs (film-f, mean-corr, discriminator dumps)")
    ap.add_argument("--spp", required=True)
    ap.add_argument("--filtersd", default="10")
    ap.add_argument("--filterradius", default="20")
    ap.add_argument("--tquantiles", default=None)
    ap.add_argument("--significance", default="0,1,2")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--json", default=None, help="also write the table as JSON (tools/pin_from_dumps.py reads it)")
    ap.add_argument("--exe", default=os.path.join(ROOT, "tools", "bin", "statmc_denoise"))
    args = ap.parse_args()
    rows = []
    # ONE process for the whole grid (statmc_denoise --sweep): a process per candidate spent 0.35 s each on bringing the
    # device up -- 2 min for 192 candidates of a small dump
    cmd = [args.exe, "--stem", args.stem, "--spp", args.spp, "--filtersd", args.filtersd, "--filterradius", args.filterradius,
           "--sweep", "quick" if args.quick else "all", "--sweep-significance", args.significance, "--compare", args.ref, "--no-write",
           "--kernel", "general", "--output", "film-f,t0-b0-mean-corr,t0-b0-discriminator"]
    if args.tquantiles:
        cmd += ["--tquantiles", args.tquantiles]
    out = subprocess.run(cmd, capture_output=True, text=True)
    if out.returncode != 0:
        print("FAILED", out.stderr.strip()[-400:], file=sys.stderr)
        return 1
    want = set(variants(args.quick))
    for block in out.stdout.split("==== sweep significance ")[1:]:
        head, _, body = block.partition("\n")
        sig, _, spec = head.partition(" spec ")
        spec = spec.strip()
        assert spec in want, spec
        errs = {}
        for m in re.finditer(r"compare (\S+) ch(\d) rel_l2 (\S+)", body):
            errs.setdefault(m.group(1), []).append(float(m.group(3)))     # worst over channels and iterations
        worst = {k: max(v) for k, v in errs.items()}
        rows.append((worst.get("film-f", float("inf")), sig.strip(), spec, worst))
    rows.sort(key=lambda r: r[0])
    print("%-12s %-3s %-80s %s" % ("film-f", "sig", "spec", "mean-corr / discriminator"))
    for w, sig, spec, worst in rows:
        print("%-12.3e %-3s %-80s %.3e / %.3e" % (w, sig, spec, worst.get("t0-b0-mean-corr", float("nan")),
                                                  worst.get("t0-b0-discriminator", float("nan"))))
    if args.json:
        import json
        json.dump([{"significance": sig, "spec": spec, "worst": worst} for _, sig, spec, worst in rows], open(args.json, "w"))
    if rows:
        print("\nbest: significance %s, --spec %s  (film-f worst channel %.3e; bound 1e-5)" % (rows[0][1], rows[0][2], rows[0][0]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
