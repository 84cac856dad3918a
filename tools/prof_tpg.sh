#!/bin/bash
# per-kernel durations of the time-parallel gradient: prof_tpg.sh B,N,J [B,N,J ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in "$@"; do
  rm -rf $R/gpurun_out/tpg_stats
  C2_TIMEPAR_GRAD=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/tpg_stats -o out --output-format csv -- python $R/tools/timepar_grad_time.py $shape > /dev/null 2>&1
  echo "== $shape"
  python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$R/gpurun_out/tpg_stats/**/*kernel_stats.csv", recursive=True)[0])))
for r in rows[:14]:
    print("%-90s calls %4s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
