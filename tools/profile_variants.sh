#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the window-filter builds round 4 added -- Welch degrees of
# freedom, eight feature planes at runtime radius, the centre-interval gate -- as timed by tools/experiments/time_welch.py,
# time_g8_radii.py and time_specs.py.  The stats CSVs are condensed into profiles/<tag>_variants_kernel_stats.csv.
set -u
TAG=${1:-r04b}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_variants
mkdir -p $OUT
cd $ROOT
for s in time_welch time_float_welch time_g8_radii time_specs; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$s -- python3 $ROOT/tools/experiments/$s.py > $OUT/$s.log 2>&1
  tail -3 $OUT/$s.log
done
# instruction counters of the default and the Welch build (one short run each counter set: QUICK=1 tools/experiments/time_welch.py)
export QUICK=1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_welch -- python3 $ROOT/tools/experiments/time_welch.py > $OUT/pmc_welch.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_welch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "window_filter_sym" in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$ROOT/gpurun_out/${TAG}_welch_pmc_per_launch.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "launches", "mean_per_launch"])
    for k, d in sorted(agg.items()):
        for c, v in sorted(d.items()):
            w.writerow([k, c, len(v), "%.6g" % (sum(v) / len(v))])
print(len(agg), "kernels with counters")
PY
python3 - <<PY
import csv, glob, os
rows = []
for p in sorted(glob.glob("$OUT/*/*/*kernel_stats.csv")):
    script = p.split("/")[-3]
    for r in csv.DictReader(open(p)):
        if "window_filter" in r["Name"] or "combine_sym" in r["Name"] or "border_virtual" in r["Name"]:
            rows.append([script, r["Name"], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"]])
with open("$ROOT/gpurun_out/${TAG}_variants_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["script", "kernel", "calls", "average_ns", "min_ns", "max_ns"])
    w.writerows(rows)
print(len(rows), "rows")
PY
