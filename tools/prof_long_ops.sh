#!/bin/bash
# per-kernel durations of the drop-in ops on ONE long series: prof_long_ops.sh ROWS op-substring [op-substring ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=$1; shift
for op in "$@"; do
  rm -rf $R/gpurun_out/long_stats
  C2_BENCH_N=$N timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/long_stats -o out --output-format csv -- python $R/tools/bench_ops.py 1 "$op" > /dev/null 2>&1
  echo "== one series of $N rows, J = 8: ops matching '$op' (7 calls each: 2 warm-up + 5 timed)"
  python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$R/gpurun_out/long_stats/**/*kernel_stats.csv", recursive=True)[0])))
for r in rows[:16]:
    print("%-100s calls %4s avg %9.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
