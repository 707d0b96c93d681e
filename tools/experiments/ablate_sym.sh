#!/bin/bash
# Where does the pair-symmetric filter spend its time?  Timing-only builds with parts of the kernel removed
# (STATMC_SYM_ABLATE bits: 1 no q side, 2 no row staging, 4 no flush, 8 no sweep arithmetic, 16 no barrier).
# Build here (container), run on the GPU box:  tools/experiments/ablate_sym.sh build | run
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SETS=${SETS:-"0 1 2 4 6 8 16 24"}
if [ "$1" = build ]; then
  for a in $SETS; do $ROOT/tools/experiments/build_variant.sh ablate_$a -DSTATMC_SYM_ABLATE=$a > /dev/null & done; wait
  ls $ROOT/tools/experiments/variants/
else
  cd $ROOT
  for a in $SETS; do python3 tools/experiments/time_filter.py --lib tools/experiments/variants/ablate_$a.so 2>&1 | grep "parts"; done
fi
