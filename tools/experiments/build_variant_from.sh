#!/bin/bash
# build_variant_from.sh NAME GIT_REV [-DFLAG ...]: libstatmc_hip.so from the csrc/ of a git revision
# (A/B runs of two source versions on one box) -> tools/experiments/variants/NAME.so
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; REV=$2; shift; shift
OUT=$ROOT/tools/experiments/variants
SRC=$OUT/src_$NAME
rm -rf $SRC; mkdir -p $SRC/statmc_amd/csrc $SRC/include
for f in statmc_pointwise.hip statmc_filter.hip statmc_abi.hip statmc_device.h t_quantiles.h; do
  git -C $ROOT show $REV:statmc_amd/csrc/$f > $SRC/statmc_amd/csrc/$f
done
git -C $ROOT show $REV:include/statmc.h > $SRC/include/statmc.h
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function"
for f in statmc_pointwise statmc_filter statmc_abi; do
  hipcc $FLAGS "$@" -c $SRC/statmc_amd/csrc/$f.hip -o $SRC/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$NAME.so $SRC/*.o
rm -rf $SRC
echo $OUT/$NAME.so
