#!/bin/bash
# tools/b8_rocprof.sh [series per GPU]: kernel trace of the bench step on the group mappings, reverse sweep by replay
# (C2_LOGLIK_BACK=0) and by the backward recursion (default)
R=${GRAFT_REPO_ROOT:-/root/repo}
BP=${1:-8192}
cd /tmp && export TMPDIR=/tmp
for back in 0 1; do
  rm -rf /tmp/b8fr
  C2_LOGLIK_BACK=$back timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/b8fr -o out --output-format csv -- python $R/bench.py --batch-per-gpu $BP --steps 10 --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy > /tmp/b8_bench.json 2>/tmp/b8.err
  python - "$back" <<'PY'
import csv, glob, json, sys
f = glob.glob("/tmp/b8fr/**/*kernel_stats.csv", recursive=True)[0]
print("C2_LOGLIK_BACK=%s" % sys.argv[1])
for r in csv.DictReader(open(f)):
    if "k_loglik" in r["Name"]:
        print("| `%s` | %s | %.3f | %.3f | %.3f |" % (r["Name"].split("(")[0].replace("void ", ""), r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
d = json.loads(open("/tmp/b8_bench.json").read().strip().splitlines()[-1])
print("bench line: ms_per_step %.3f, frac %.4f" % (d["ms_per_step"], d["roofline"]["frac"]))
PY
done
