#!/bin/bash
# tools/k2_rocprof.sh <outdir>: the bench step at the 2-GPU shard of configs[2] (32768 series per GPU: the two-lanes-per-series
# kernels) under rocprofv3 --kernel-trace --stats, and its HBM traffic from the PMC passes (FETCH_SIZE, WRITE_SIZE separately).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$(realpath -m ${1:-$R/gpurun_out/k2prof}); mkdir -p $O
BP=${K2_BATCH:-32768}
cd /tmp && export TMPDIR=/tmp
ARGS="--batch-per-gpu $BP --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy"
rm -rf /tmp/k2fr
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/k2fr -o out --output-format csv -- python $R/bench.py --steps 10 $ARGS > $O/bench.json 2> $O/bench.err
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/k2tr_$ctr; mkdir -p /tmp/k2tr_$ctr
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/k2tr_$ctr -o tr -- python $R/bench.py --steps 3 --warmup 1 $ARGS > /tmp/k2tr_$ctr/log 2>&1
done
python - "$O" "$BP" <<'PY'
import csv, glob, json, sqlite3, sys
O, BP = sys.argv[1], int(sys.argv[2])
out = open(O + "/summary.md", "w")
def P(*a):
    print(*a); print(*a, file=out)
f = glob.glob("/tmp/k2fr/**/*kernel_stats.csv", recursive=True)[0]
P("| kernel | calls | avg ms | min ms | max ms |"); P("|---|---|---|---|---|")
tot = 0.0
for r in csv.DictReader(open(f)):
    if "k_k2_" in r["Name"] or "k_loglik" in r["Name"]:
        P("| `%s` | %s | %.3f | %.3f | %.3f |" % (r["Name"].split("(")[0].replace("void ", ""), r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
        if "k_k2_" in r["Name"]: tot += float(r["AverageNs"]) / 1e6
d = json.loads(open(O + "/bench.json").read().strip().splitlines()[-1])
P("sum of the k_k2 averages %.2f ms; bench line of the same process: ms_per_step %.3f, kernel_ms_avg %.3f, frac %.4f" % (tot, d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"]))
vals = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    db = sqlite3.connect(glob.glob("/tmp/k2tr_%s/**/*_results.db" % ctr, recursive=True)[0])
    q = ("select kernel_name, avg(v) from (select dispatch_id, kernel_name, sum(value) as v from counters_collection "
         "where counter_name = ? group by dispatch_id, kernel_name) group by kernel_name")
    for name, v in db.execute(q, (ctr,)):
        if "k_k2_" in name:
            vals.setdefault(name.split("(")[0].replace("void ", "").replace("c2k2::", ""), {})[ctr] = v
kern = {k: int((2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024) for k, v in vals.items()}
for k, v in vals.items():
    P("%-28s FETCH_SIZE %.4g KiB  WRITE_SIZE %.4g KiB  -> %.2f GB" % (k, v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0), kern[k] / 1e9))
total = sum(kern.values())
P("total %.2f GB per step = %.3f x the %.2f GB algorithmic" % (total / 1e9, total / (BP * 1245320), BP * 1245320 / 1e9))
json.dump({"mode": "grad", "batch_per_gpu": BP, "N": 4096, "J": 8, "traffic_bytes_per_step": total, "kernels": kern}, open(O + "/traffic_workload.json", "w"), indent=1)
PY
