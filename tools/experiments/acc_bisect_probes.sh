for i in 1 2 3 4 5 6 7 8; do
  STATMC_PLACEMENT_DEBUG=1 V=a timeout -k 10 100 python tools/experiments/acc_bisect.py > gpurun_out/bis_$i.out 2> gpurun_out/bis_$i.err || exit 1
  python - <<PY
import re
err=open("gpurun_out/bis_$i.err").read()
v0=[float(m.group(1)) for m in re.finditer(r"against target 0: ([0-9.]+) ms", err)]
f=min(v0)
line=open("gpurun_out/bis_$i.out").read().strip().splitlines()[-1]
m=re.search(r"grid 1 [0-9.]+ ms ([0-9.]+)", line)
mid=sum(1 for x in v0 if 1.035<=x/f<1.07)
print("run $i: grid1", m.group(1), "probes", len(v0), "fast<1.035:", sum(1 for x in v0 if x/f<1.035), "mid:", mid, "slow>=1.07:", sum(1 for x in v0 if x/f>=1.07), "| first 40 ratios:", " ".join("%.3f" % (x/f) for x in v0[:40]), "|", line.split("(state vs state")[1][8:] if "(state vs state" in line else "")
PY
done
