#!/bin/bash
# tools/final_rocprof.sh [series per GPU]: the bench step under rocprofv3 --kernel-trace --stats and the bench line of the same
# process (markdown summary on stdout; default: the bench's own default, 65536 series on one GPU)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
BP=${1:+--batch-per-gpu $1}
rm -rf /tmp/fr; rocprofv3 --kernel-trace --stats -d /tmp/fr -o out --output-format csv -- python $R/bench.py $BP --steps 10 --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy > /tmp/fr_bench.json 2>/tmp/fr.err
python - <<'PY'
import csv, glob, json
f = glob.glob("/tmp/fr/**/*kernel_stats.csv", recursive=True)[0]
d = json.loads(open("/tmp/fr_bench.json").read().strip().splitlines()[-1])
print("%d series per GPU (`python bench.py` %s, rocprofv3 --kernel-trace --stats):\n" % (d["config"]["batch_per_gpu"], "default" if d["config"]["batch_per_gpu"] == 65536 else "--batch-per-gpu"))
print("| kernel | calls | avg ms | min ms | max ms |"); print("|---|---|---|---|---|")
tot = 0.0
for r in csv.DictReader(open(f)):
    # (k_loglik_fwd<..., 2, ...> = FACTOR mode: the untimed c2_condition call behind the timed region, 16 slices of 4096 series)
    if any(k in r["Name"] for k in ("k_loglik", "k_q4_", "k_k2_", "k_anchor")) and "16, 8, 2," not in r["Name"] and int(r["Calls"]) >= 10 and float(r["AverageNs"]) > 2e4:
        print("| `%s` | %s | %.3f | %.3f | %.3f |" % (r["Name"].split("(")[0].replace("void ", ""), r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
        tot += float(r["AverageNs"]) / 1e6
print("\nsum of the averages %.3f ms; bench line of the same process: ms_per_step %.3f, kernel_ms_avg %.3f, frac %.4f\n" % (tot, d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"]))
PY
