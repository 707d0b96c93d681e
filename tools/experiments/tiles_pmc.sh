#!/bin/bash
# Round 6: what the tile walk of accumulate_tiles_kernel pays at short batches -- HBM bytes, L2 requests / hits,
# wave cycles -- beside accumulate_kernel on the same samples (tiles_pmc_case.py).  One counter group per pass; the program directly
# after `--`.  Runs on the GPU box: S=4 bash tools/experiments/tiles_pmc.sh; outputs under gpurun_out/tiles_pmc_S<S>/.
set -u
export TMPDIR=/tmp
export S=${S:-4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/tiles_pmc_S$S
mkdir -p $OUT
cd $ROOT
pass() {
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/tools/experiments/tiles_pmc_case.py > $OUT/$name.log 2>&1 || echo "pass $name failed"
  echo "pass $name done"
}
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE
pass l2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
pass ea TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
python3 - <<PY
import csv, glob, collections, os
out = "$OUT"
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "accumulate" not in k: continue
        rows["tiles" if "tiles" in k else "film"][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/fetch/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "accumulate" in k: dur["tiles" if "tiles" in k else "film"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("S = $S  counter means per launch (FETCH_SIZE / WRITE_SIZE in KiB as reported; FETCH_SIZE x2 on gfx950 = bytes / 1024)")
names = sorted(set(rows["film"]) | set(rows["tiles"]))
print("%-28s %16s %16s %8s" % ("counter", "film-major", "tile-fed", "ratio"))
for n in names:
    a = sum(rows["film"][n]) / max(1, len(rows["film"][n])); b = sum(rows["tiles"][n]) / max(1, len(rows["tiles"][n]))
    print("%-28s %16.4g %16.4g %8.3f" % (n, a, b, b / a if a else 0))
for k in ("film", "tiles"):
    if dur[k]: print("duration us (%s, under the profiler): %s" % (k, ", ".join("%.1f" % d for d in dur[k])))
PY
