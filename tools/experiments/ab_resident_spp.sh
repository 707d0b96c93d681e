#!/bin/bash
# the resident grid (one workgroup per CU) against the large grids at shorter batches and other films, in the step
cd $GRAFT_REPO_ROOT
for cfg in "1920x1080 128" "1920x1080 64" "1280x720 256" "3840x2160 64" "2560x1440 256"; do
  set -- $cfg
  for n in -1 256 -1 256; do
    STATMC_BENCH_ACC_RESIDENT=$n python bench.py --film $1 --spp $2 --no-cpu-baseline --no-host-legs --steps 100 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', $2, 'resident', $n, 'step', d['ms_per_step'], 'acc', d['kernels']['accumulate']['ms_per_step'], 'filter', d['kernels']['filter']['ms_per_step'], flush=True)
"
  done
done
