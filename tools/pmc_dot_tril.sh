#!/bin/bash
# PMC passes of the long-series dot_tril (tools/mfma_time.py: N = 1e7, J = 16, nrhs = 32) -- what bounds k_mm_mfma?
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-pmc_dt}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python $R/tools/mfma_time.py > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/pmc1 -o pmc1 -- python $R/tools/mfma_time.py > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -d $OUT/pmc2 -o pmc2 -- python $R/tools/mfma_time.py > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- python $R/tools/mfma_time.py > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- python $R/tools/mfma_time.py > $OUT/pmc4.log 2>&1
python $R/tools/pmc_summary.py $OUT k_mm_mfma
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    if "at::" in r["Name"] or "rocclr" in r["Name"]: continue
    print("%-80s calls %4s avg %9.1f us" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
