#!/bin/bash
# tools/icache_probe.sh <lanes> <B> [B ...]: instruction-cache counters of the fused gradient kernels (are the unrolled loop bodies
# of several wavefronts per CU fighting over the instruction cache?)
R=${GRAFT_REPO_ROOT:-/root/repo}
L=$1; shift
cd /tmp; export TMPDIR=/tmp
for B in "$@"; do
  rm -rf /tmp/icp; mkdir -p /tmp/icp
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQC_ICACHE_BUSY_CYCLES -d /tmp/icp/p0 -o p0 -- python $R/tools/lanes_any_run.py $L 4096 $B 4 > /tmp/icp/p0.log 2>&1
  echo "== lanes $L, B = $B"
  python $R/tools/pmc_summary.py /tmp/icp k_loglik k_q4_fwd k_q4_rev k_k2 | grep -v "false, false, 1, false"
done
