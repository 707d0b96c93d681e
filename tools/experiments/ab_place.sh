#!/bin/bash
# two processes on one box: default, and without the rebase; each prints placed / unplaced back to back and the step
cd $GRAFT_REPO_ROOT
run() {
  tag=$1; shift
  env "$@" STATMC_PLACEMENT_DEBUG=1 python bench.py --no-cpu-baseline --steps 100 > gpurun_out/r06_bench_$tag.json 2> gpurun_out/r06_bench_$tag.err || return 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_$tag.json").read().strip().splitlines()[-1])
p=d["placement"]; ab=d.get("accumulate_placement_ab",{})
print("$tag", d["value"], d["roofline"]["frac"], "ab placed/unplaced", ab.get("placed_frac_hbm"), ab.get("unplaced_frac_hbm"), p.get("map"), p.get("probes"), p.get("peak_slots"), p.get("rebased"), p.get("timed_region_ran_on"), p.get("filter_workspace"), p.get("state_live_GiB"), p.get("info_error"), flush=True)
PY
}
run ${TAG:-z}1 ${ENV1:-A=1} && run ${TAG:-z}2 ${ENV2:-A=1} && run ${TAG:-z}3 ${ENV3:-A=1} && { [ -z "$ENV4" ] || run ${TAG:-z}4 $ENV4; }
