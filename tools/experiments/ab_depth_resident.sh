#!/bin/bash
# ring depth of the film-major accumulation under the resident grid (four waves per CU: half the rows in flight of the large grids), in the step
cd $GRAFT_REPO_ROOT
export STATMC_VARIANT=tools/experiments/variants/depths.so
for round in 1 2; do
for d in 3 4 5 6; do
  STATMC_BENCH_ACC_DMA=$d python bench.py --no-cpu-baseline --no-host-legs --steps 200 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('depth', $d, 'step', d['ms_per_step'], 'acc', d['kernels']['accumulate']['ms_per_step'], 'filter', d['kernels']['filter']['ms_per_step'], d['placement'].get('map'), flush=True)
"
done; done
