#!/bin/bash
# build_variant.sh NAME [-DFLAG ...]: libstatmc_hip.so with extra compile flags -> tools/experiments/variants/NAME.so
# SYM_ONLY=1: only statmc_filter_sym.hip is compiled with the flags; the other objects are the product build's (statmc_amd/csrc/*.o)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; shift
OUT=$ROOT/tools/experiments/variants
mkdir -p $OUT/obj_$NAME
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function ${STATMC_VARIANT_BASE_FLAGS--mllvm -enable-misched=0}"
if [ -n "$SYM_ONLY" ]; then
  for f in statmc_pointwise statmc_filter statmc_placement statmc_abi statmc_rccl; do cp $ROOT/statmc_amd/csrc/$f.o $OUT/obj_$NAME/$f.o; done
  hipcc $FLAGS "$@" -c $ROOT/statmc_amd/csrc/statmc_filter_sym.hip -o $OUT/obj_$NAME/statmc_filter_sym.o
else
  for f in statmc_pointwise statmc_filter statmc_filter_sym statmc_placement statmc_abi statmc_rccl; do
    hipcc $FLAGS "$@" -c $ROOT/statmc_amd/csrc/$f.hip -o $OUT/obj_$NAME/$f.o &
  done
  wait
fi
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$NAME.so $OUT/obj_$NAME/*.o
rm -rf $OUT/obj_$NAME
echo $OUT/$NAME.so
