#!/bin/bash
# the default step with the product library against several variant libraries (tools/experiments/variants/NAME.so), round robin
# usage: ab_variants_step.sh ROUNDS NAME...
cd $GRAFT_REPO_ROOT
R=$1; shift
for round in $(seq $R); do
for lib in product "$@"; do
  if [ $lib = product ]; then unset STATMC_VARIANT; else export STATMC_VARIANT=tools/experiments/variants/$lib.so; fi
  python bench.py --no-cpu-baseline --no-host-legs --steps 200 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'step', d['ms_per_step'], 'acc', d['kernels']['accumulate']['ms_per_step'], 'filter', d['kernels']['filter']['ms_per_step'], d['placement'].get('map'), flush=True)
"
done; done
