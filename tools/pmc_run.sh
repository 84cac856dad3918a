#!/bin/bash
# tools/pmc_run.sh <tag> [bench args...] -- run on the GPU box (via gpurun): kernel stats + PMC passes of bench.py,
# each counter group in its own rocprofv3 run (no trace domains mixed with --pmc).  Output: gpurun_out/<tag>/
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
ARGS="--no-cpu-baseline --steps 3 --warmup 1 $*"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/pmc1 -o pmc1 -- python $R/bench.py $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT -d $OUT/pmc2 -o pmc2 -- python $R/bench.py $ARGS > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- python $R/bench.py $ARGS > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- python $R/bench.py $ARGS > $OUT/pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc5 -o pmc5 -- python $R/bench.py $ARGS > $OUT/pmc5.log 2>&1
ls -la $OUT/*/ | head -30
tail -2 $OUT/pmc1.log
