#!/bin/bash
# the default step with the product library against a variant library (tools/experiments/variants/$1.so), alternating, three rounds;
# extra bench arguments after the name
cd $GRAFT_REPO_ROOT
V=$1; shift
for round in 1 2 3; do
for lib in product $V; do
  if [ $lib = product ]; then unset STATMC_VARIANT; else export STATMC_VARIANT=tools/experiments/variants/$lib.so; fi
  python bench.py --no-cpu-baseline --no-host-legs --steps 200 "$@" 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'step', d['ms_per_step'], 'acc', d['kernels']['accumulate']['ms_per_step'], 'filter', d['kernels']['filter']['ms_per_step'], d['shader_clock']['during_filter_GHz'], d['config']['filter_variant'], flush=True)
"
done; done
