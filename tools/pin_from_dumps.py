#!/usr/bin/env python3
"""pin_from_dumps.py -- pin this build's filter to dumps of the CUDA StatMC, in one command.

    tools/pin_from_dumps.sh <dir> [--ref <dir>] [--filtersd 10] [--filterradius 20] [--tquantiles table.txt]
                                  [--header include/statmc_pinned_spec.h] [--no-rebuild] [--quick]

<dir> holds dumps written by the CUDA build (`pbrt --writeimages`, scenes/render-for-ours.pbrt:24 for the inputs, a
denoising run for the outputs): for every <stem> and sample count
    inputs   <stem>-<spp>-{film, t0-b0-n, t0-b0-mean, t0-b0-m2, t0-b0-m3, t1-b0-film-mean, t2-b0-film-mean}.pfm
    outputs  <stem>-<spp>-{film-f, t0-b0-mean-corr, t0-b0-discriminator}.pfm         (in <dir> or in --ref <dir>)
What it does:
  1. finds the (stem, spp) sets that are complete,
  2. runs tools/fit_spec.py over them: all 96 filter specs (three gate forms x 2^5) x 3 significance levels through the HIP library
     (tools/bin/statmc_denoise --compare), per-channel relative L2 of film-f / mean-corr / discriminator -> <dir>/pin_table.txt,
  3. writes the winner as the NEW DEFAULT of library and oracle: include/statmc_pinned_spec.h,
  4. unless --no-rebuild: rebuilds libstatmc_hip.so, the host tools and the oracle, regenerates tests/golden/ with the
     pinned oracle (tests/golden/make_golden.py) and runs the golden tests.
A winner above BASELINE.json's 1e-5 is reported as such and NOT written (exit 3)."""
import argparse
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INPUTS = ("film", "t0-b0-n", "t0-b0-mean", "t0-b0-m2", "t0-b0-m3", "t1-b0-film-mean", "t2-b0-film-mean")
OUTPUTS = ("film-f", "t0-b0-mean-corr", "t0-b0-discriminator")
FIELDS = (("gate", ("sym", "asym", "centre")), ("channels", ("and", "joint")), ("sides", ("two", "one")),
          ("dof", ("pixel", "welch")), ("border", ("clip", "clamp")), ("small_n", ("accept", "exclude")))
BOUND = 1e-5


def discover(d, refdir):
    sets = {}
    for f in sorted(glob.glob(os.path.join(d, "*-film.pfm"))):
        m = re.match(r"(.*)-(\d+)-film\.pfm$", os.path.basename(f))
        if not m:
            continue
        stem, spp = m.group(1), int(m.group(2))
        ok = all(os.path.exists(os.path.join(d, "%s-%d-%s.pfm" % (stem, spp, n))) for n in INPUTS) and \
            all(os.path.exists(os.path.join(refdir, "%s-%d-%s.pfm" % (stem, spp, n))) for n in OUTPUTS)
        if ok:
            sets.setdefault(stem, []).append(spp)
    return sets


def spec_ints(spec):
    kv = dict(p.split("=") for p in spec.split(","))
    return [vals.index(kv.get(k, vals[0])) for k, vals in FIELDS]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--ref", default=None, help="directory of the CUDA build's output dumps (default: <dir>)")
    ap.add_argument("--filtersd", default="10")
    ap.add_argument("--filterradius", default="20")
    ap.add_argument("--tquantiles", default=None)
    ap.add_argument("--header", default=os.path.join(ROOT, "include", "statmc_pinned_spec.h"))
    ap.add_argument("--no-rebuild", action="store_true")
    ap.add_argument("--quick", action="store_true", help="gate / channel rule / sides only (12 specs x 3 levels)")
    args = ap.parse_args()
    refdir = args.ref or args.dir
    sets = discover(args.dir, refdir)
    if not sets:
        print("no complete (stem, spp) set of input and output dumps in %s / %s" % (args.dir, refdir), file=sys.stderr)
        return 2
    # ---- the table: per stem, every spec x level; the winner must win on every stem (worst film-f channel over all of them)
    merged = {}
    with open(os.path.join(args.dir, "pin_table.txt"), "w") as table:
        for stem, spps in sorted(sets.items()):
            cmd = [sys.executable, os.path.join(ROOT, "tools", "fit_spec.py"), "--stem", os.path.join(args.dir, stem),
                   "--ref", os.path.join(refdir, stem), "--spp", ",".join(str(s) for s in sorted(spps)), "--filtersd", args.filtersd,
                   "--filterradius", args.filterradius, "--json", os.path.join(args.dir, "pin_%s.json" % stem)]
            if args.tquantiles:
                cmd += ["--tquantiles", args.tquantiles]
            if args.quick:
                cmd.append("--quick")
            out = subprocess.run(cmd, capture_output=True, text=True)
            if out.returncode != 0:
                print(out.stdout + out.stderr, file=sys.stderr)
                return out.returncode
            table.write("==== %s, spp %s\n%s\n" % (stem, sorted(spps), out.stdout))
            for row in json.load(open(os.path.join(args.dir, "pin_%s.json" % stem))):
                key = (row["significance"], row["spec"])
                w = merged.setdefault(key, {"film-f": 0.0, "t0-b0-mean-corr": 0.0, "t0-b0-discriminator": 0.0})
                for k in w:
                    w[k] = max(w[k], row["worst"].get(k, float("inf")))
    ranked = sorted(merged.items(), key=lambda kv: (kv[1]["film-f"], kv[1]["t0-b0-discriminator"], kv[1]["t0-b0-mean-corr"]))
    print("%-12s %-3s %-80s %s" % ("film-f", "sig", "spec", "mean-corr / discriminator   (worst channel over %d stem(s))" % len(sets)))
    for (sig, spec), w in ranked[:12]:
        print("%-12.3e %-3s %-80s %.3e / %.3e" % (w["film-f"], sig, spec, w["t0-b0-mean-corr"], w["t0-b0-discriminator"]))
    (sig, spec), w = ranked[0]
    print("\nwinner: significance %s, spec %s: film-f %.3e (bound %.0e); table in %s" % (sig, spec, w["film-f"], BOUND, os.path.join(args.dir, "pin_table.txt")))
    if not w["film-f"] <= BOUND:
        print("no spec of this build reproduces the dumps within %.0e: NOT pinned.  The per-channel table says where the "
              "arithmetic differs (mean-corr / discriminator columns: pre-pass; film-f only: window filter)." % BOUND)
        return 3
    ints = spec_ints(spec)
    # the description ends up inside a C string literal: backslash and double quote escaped, anything else outside
    # printable ASCII replaced (a dump stem is a file name of the user's)
    def c_string(text):
        return "".join(("\\" + ch) if ch in '\\"' else (ch if 32 <= ord(ch) < 127 else "?") for ch in text)
    # written next to the target and renamed into place only when everything after it has succeeded (or --no-rebuild):
    # a rebuild that fails halfway leaves the tracked header as it was
    final_header, args.header = args.header, args.header + ".new"
    with open(args.header, "w") as h:
        h.write("/* statmc_pinned_spec.h -- the filter spec and significance level a freshly set-up device (and the CPU oracle) start\n"
                " * with.  WRITTEN by tools/pin_from_dumps.sh: the spec that reproduces the dumps named below is the default of\n"
                " * library and oracle alike.  Field order = statmc_filter_spec: gate, channel_rule, sides, dof, border, small_n. */\n"
                "#ifndef STATMC_PINNED_SPEC_H\n#define STATMC_PINNED_SPEC_H\n"
                "#define STATMC_PINNED_SPEC {%s}\n#define STATMC_PINNED_SIGNIFICANCE %d\n"
                "#define STATMC_PINNED_FROM \"%s\"\n#endif\n"
                % (", ".join(str(i) for i in ints), int(sig),
                   c_string("%s, significance %s: film-f within %.2e of %d stem(s) of dumps (%s)" % (spec, sig, w["film-f"], len(sets), ",".join(sorted(sets))))))
    if args.no_rebuild:
        os.replace(args.header, final_header)
        print("wrote", final_header)
        return 0
    backup = None
    if os.path.exists(final_header):
        backup = open(final_header).read()
    os.replace(args.header, final_header)
    print("wrote", final_header)

    def restore():
        if backup is not None:
            open(final_header, "w").write(backup)
            print("rebuild failed: %s restored (rebuild once more to get the library back in step with it)" % final_header)
    env = dict(os.environ, PYTHONPATH=ROOT)
    for cmd in ([sys.executable, os.path.join(ROOT, "__graft_entry__.py")],
                [sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden.py")],
                [sys.executable, "-m", "pytest", "-q", os.path.join(ROOT, "tests", "test_golden.py")]):
        print("+", " ".join(cmd), flush=True)
        rc = subprocess.call(cmd, cwd=ROOT, env=env)
        if rc != 0:
            restore()
            return rc
    print("pinned: library, oracle and tests/golden/ now follow %s" % spec)
    return 0


if __name__ == "__main__":
    sys.exit(main())
