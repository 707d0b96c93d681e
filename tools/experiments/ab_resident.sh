#!/bin/bash
# the step with the accumulation on a resident grid of N workgroups (statmc_debug_accumulate_resident_blocks) against the default launch:
# does a leaner accumulation leave the filter a higher clock?  one box, every variant its own process, two rounds
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for n in ${NS:-0 256}; do
  STATMC_BENCH_ACC_RESIDENT=$n python bench.py --no-cpu-baseline --no-host-legs --steps 200 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('resident', $n, 'step', d['ms_per_step'], 'acc', d['kernels']['accumulate']['ms_per_step'], 'filter', d['kernels']['filter']['ms_per_step'], d['shader_clock']['during_accumulate_GHz'], d['shader_clock']['during_filter_GHz'], d['placement'].get('map'), flush=True)
"
done; done
