"""Per-shape table of the counters acc_pmc.sh collected: accumulate_kernel dispatches in launch order are
3 x (1080p / 256 spp), 3 x (1080p / 64), 3 x (4K / 64), 3 x (4K / 16); the mean of each group of three."""
import csv
import glob
import os
import sys

root = sys.argv[1]
shapes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["1080p/256", "1080p/64", "4K/64", "4K/16"]
table = {}
for f in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "accumulate_kernel" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    order = {d: i for i, d in enumerate(ids)}
    for r in rows:
        i = order[int(r["Dispatch_Id"])]
        if i >= 3 * len(shapes):
            continue
        table.setdefault(r["Counter_Name"], [0.0] * len(shapes))[i // 3] += float(r["Counter_Value"]) / 3
print("%-46s" % "counter (mean per launch)" + "".join("%16s" % s for s in shapes))
for k in sorted(table):
    print("%-46s" % k + "".join("%16.4g" % v for v in table[k]))
