#!/bin/bash
# Round 5: fabric-side read latency and address-translation counters of the SAME launch (4K, 64 spp, all stat types) on a pool
# allocated first (contiguous: the slow placement) and on one allocated out of 16 MiB holes (acc_frag.py, PMC=1 mode).
set -u
TAG=${1:-r05b}
export TMPDIR=/tmp PMC=1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/acc_frag_pmc_$TAG
mkdir -p $OUT
cd $ROOT
pass() {
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/tools/experiments/acc_frag.py > $OUT/$name.log 2>&1 || echo "pass $name failed"
  echo "pass $name done"
}
pass ea TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum
pass tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum
pass l2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum
pass stall TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY
python3 $ROOT/tools/experiments/acc_pmc_table.py $OUT "poolA,poolB,poolA2,poolC,poolB2" > $OUT/table.txt 2>&1
cat $OUT/table.txt
