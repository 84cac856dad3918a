#!/bin/bash
# tools/pmc_any.sh <kernel-substring> <python script + args...> : wave-cycle breakdown of the matching kernels (rocprofv3 --pmc,
# two passes), printed; nothing is left under gpurun_out/ (the rocpd databases are large).
K=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pmc_any; mkdir -p /tmp/pmc_any
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/pmc_any/p1 -o p1 -- python "$@" > /tmp/pmc_any/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d /tmp/pmc_any/p2 -o p2 -- python "$@" > /tmp/pmc_any/p2.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_any "$K"
