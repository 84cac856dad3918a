#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dt -o out --output-format csv -- python $R/tools/mfma_time.py > /dev/null 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/prof_dt/out_kernel_stats.csv")))
for r in rows[:10]:
    if "at::" in r["Name"] or "rocclr" in r["Name"]: continue
    print("%-80s calls %4s avg %9.1f us" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
