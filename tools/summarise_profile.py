#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (rocprofv3 CSVs written by tools/profile_gpu.sh) into the
small tracked files under profiles/: kernel-trace stats, PMC averages per kernel, and
hbm_traffic.json (HBM bytes per launch per kernel, FETCH_SIZE doubled as
MI355X_MICROARCH.md's HBM section prescribes for gfx950 wide streaming reads)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag):
    if tag.startswith("-"):
        raise SystemExit("usage: summarise_profile.py TAG   (condenses gpurun_out/prof_TAG/ into profiles/TAG_*)")
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    # 1. kernel stats
    for p in glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")):
        rows = list(csv.DictReader(open(p)))
        with open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    # 2. PMC averages per kernel
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in sorted(glob.glob(os.path.join(src, "pmc*", "*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(p)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(os.path.join(dst, "%s_pmc_per_launch.csv" % tag), "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Counter", "MeanPerLaunch", "Launches"])
        for k in sorted(agg):
            for c in sorted(agg[k]):
                v = agg[k][c]
                w.writerow([k, c, "%.6g" % (sum(v) / len(v)), len(v)])
    # 3. HBM traffic per launch (FETCH_SIZE / WRITE_SIZE are in KiB)
    traffic = {}
    for k, cs in agg.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            name = k.split("(")[0].replace("void ", "").replace("statmc::", "").replace("sym::", "").split("<")[0]
            fetch = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024.0
            write = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024.0
            traffic[name] = int(2.0 * fetch + write)
            traffic[name + "__detail"] = {"FETCH_SIZE_bytes_raw": int(fetch), "fetch_corrected_x2": int(2 * fetch),
                                          "WRITE_SIZE_bytes": int(write)}
    # VALU issue work of the window filter from the counter: wave instructions per launch, and the issue cycles an
    # instruction of the compiled sweep costs on average (per read group 240 packed at 4 cycles, 48 single-rate at 2,
    # 16 v_exp_f32 at 8: the mix of statmc_filter_sym.hip's loop body) -- bench.py turns this into the
    # ALU-pass bound next to its measured launch time
    for k, cs in agg.items():
        if "window_filter_sym" in k and "SQ_INSTS_VALU" in cs:
            traffic["window_filter_valu"] = {
                "SQ_INSTS_VALU": int(sum(cs["SQ_INSTS_VALU"]) / len(cs["SQ_INSTS_VALU"])),
                "cycles_per_inst": round((240 * 4 + 48 * 2 + 16 * 8) / 304.0, 3),
                "GRBM_GUI_ACTIVE": int(sum(cs["GRBM_GUI_ACTIVE"]) / len(cs["GRBM_GUI_ACTIVE"])) if "GRBM_GUI_ACTIVE" in cs else None,
                "source": "profiles/%s_pmc_per_launch.csv (SQ_INSTS_VALU per launch of window_filter_sym); mix 240 packed : 48 single : 16 exp per read group" % tag}
    traffic["_note"] = ("bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), "
                        "profile tag %s; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B "
                        "requests at 64 B); Infinity-Cache hits are included in these fabric-side counters" % tag)
    json.dump(traffic, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(traffic, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
