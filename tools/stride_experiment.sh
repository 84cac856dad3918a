#!/bin/bash
# Does the run-to-run spread of the one-lane reverse sweep (16 - 22 ms at the bench shape) come from the power-of-two series
# stride of the API arrays (N J 8 = 2^18 bytes at N = 4096, J = 8: the 64 rows a wavefront touches per instruction differ in
# address bits >= 18 only)?  Same workload at N = 4094 / 4098 (series stride an ODD number of 128-byte lines), three fresh
# processes each, per-kernel averages from rocprofv3.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for N in 4096 4094 4098 4096 4094; do
  rm -rf /tmp/se_prof
  rocprofv3 --kernel-trace --stats -d /tmp/se_prof -o out --output-format csv -- python $R/bench.py --N $N --steps 5 --warmup 2 --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy > /tmp/se.json 2>/dev/null
  f=$(find /tmp/se_prof -name "*kernel_stats.csv" | head -1)
  python - "$f" $N <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
d = json.loads(open("/tmp/se.json").read().strip().splitlines()[-1])
k = {r["Name"].split("(")[0]: float(r["AverageNs"]) / 1e6 for r in rows}
rev = [v for n, v in k.items() if "k_loglik_t_rev" in n]; fwd = [v for n, v in k.items() if "k_loglik_t_fwd" in n]
print("N %s  step %.2f ms  fwd %.2f ms  rev %.2f ms  (%.3f M GP/s)" % (sys.argv[2], d["ms_per_step"], fwd[0] if fwd else -1, rev[0] if rev else -1, d["value"] / 1e6), flush=True)
PY
done
