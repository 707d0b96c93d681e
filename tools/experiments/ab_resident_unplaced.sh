#!/bin/bash
# the resident grid on buffers of torch's allocator (no placement), in the step; two rounds
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
for n in -1 256; do
  STATMC_BENCH_ACC_RESIDENT=$n python bench.py --no-placement --no-cpu-baseline --no-host-legs --steps 200 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('unplaced, resident', $n, 'step', d['ms_per_step'], 'acc', d['kernels']['accumulate']['ms_per_step'], 'filter', d['kernels']['filter']['ms_per_step'], flush=True)
"
done; done
