#!/bin/bash
# per-kernel durations of BASELINE configs[4] (2-D, collapsed method)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/kron_stats
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kron_stats -o out --output-format csv -- python $R/tools/bench_configs.py 5 > /dev/null 2>&1
python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$R/gpurun_out/kron_stats/**/*kernel_stats.csv", recursive=True)[0])))
for r in rows[:30]:
    print("%-100s calls %4s avg %9.1f us total %8.2f ms" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
