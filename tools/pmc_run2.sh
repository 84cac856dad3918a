#!/bin/bash
# tools/pmc_run2.sh <tag> [bench args...] -- second counter set (instruction fetch, LDS queueing, VMEM issue, VALU mix),
# one rocprofv3 --pmc pass per group.  Output: gpurun_out/<tag>/
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
ARGS="--no-cpu-baseline --steps 3 --warmup 1 $*"
i=0
while read -r GROUP; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $GROUP -d $OUT/q$i -o q$i -- python $R/bench.py $ARGS > $OUT/q$i.log 2>&1
done <<'GROUPS'
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES
SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_BANK_CONFLICT
SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL
SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM
GROUPS
python $R/tools/pmc_summary.py $OUT k_loglik
