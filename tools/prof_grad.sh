#!/bin/bash
# per-kernel durations of the fused gradient at the bench shape, for the lane mapping in $C2_LANES (rocprofv3 --stats)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-run}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o out --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_$tag.json 2> $R/gpurun_out/prof_$tag.err
cat $R/gpurun_out/prof_$tag.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:8]:
    print("%-70s calls %5s avg %10.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
