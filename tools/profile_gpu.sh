#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes of the bench workload.
# Outputs under gpurun_out/prof_<tag>/ ; summaries are copied into profiles/ by tools/summarise_profile.py.
set -u
TAG=${1:-r04a}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-legs"
cd $ROOT
# kernel-trace stats of the bench's default command (300 timed steps + 5 warm-up), so that the average
# kernel durations are the ones bench.py's own HIP events report; the PMC passes use a short run
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-host-legs > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -- $BENCH > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VALU --output-format csv -d $OUT/pmc2 -- $BENCH > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc3 -- $BENCH > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- $BENCH > $OUT/pmc4.log 2>&1
find $OUT -name "*.csv" | head -40
du -sh $OUT
