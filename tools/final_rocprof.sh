#!/bin/bash
# tools/final_rocprof.sh: the bench step under rocprofv3 --kernel-trace --stats and the bench line of the same process (summary on stdout)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/fr; rocprofv3 --kernel-trace --stats -d /tmp/fr -o out --output-format csv -- python $R/bench.py --steps 10 --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy > /tmp/fr_bench.json 2>/tmp/fr.err
python - <<'PY'
import csv, glob, json
f = glob.glob("/tmp/fr/**/*kernel_stats.csv", recursive=True)[0]
print("| kernel | calls | avg ms | min ms | max ms |"); print("|---|---|---|---|---|")
tot = 0.0
for r in csv.DictReader(open(f)):
    if "k_loglik_t_" in r["Name"]:
        print("| `%s` | %s | %.3f | %.3f | %.3f |" % (r["Name"].split("(")[0].replace("void ", ""), r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
        tot += float(r["AverageNs"]) / 1e6
d = json.load(open("/tmp/fr_bench.json"))
print("sum of the averages %.2f ms; bench line of the same process: ms_per_step %.3f, kernel_ms_avg %.3f, frac %.4f" % (tot, d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"]))
PY
