#!/bin/bash
# Round 5: address-translation, L2 and fabric counters of accumulate_kernel at 1080p / 256 spp against
# 4K / 64 spp (the same sample bytes, 14 % apart in time).  One counter group per pass; program directly after `--`.
# Runs on the GPU box: bash tools/experiments/acc_pmc.sh [tag]; outputs under gpurun_out/acc_pmc_<tag>/.
set -u
TAG=${1:-r05a}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/acc_pmc_$TAG
mkdir -p $OUT
cd $ROOT
CASE="python3 $ROOT/tools/experiments/acc_pmc_case.py"
pass() {  # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- $CASE > $OUT/$name.log 2>&1 || echo "pass $name failed (see $OUT/$name.log)"
  echo "pass $name done"
}
pass tlb1 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum
pass tlb2 TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_MULTI_MISS_sum
pass l2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
pass ea TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
pass sq SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass tag TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUBBLE_sum
python3 $ROOT/tools/experiments/acc_pmc_table.py $OUT > $OUT/table.txt 2>&1
cat $OUT/table.txt
