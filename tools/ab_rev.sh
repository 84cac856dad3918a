#!/bin/bash
# tools/ab_rev.sh lib1.so lib2.so ... : per-kernel averages (rocprofv3) of the bench step for each library, alternating, 3 rounds
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for round in 1 2 3; do
  for lib in "$@"; do
    rm -rf /tmp/abp
    C2_LIB_PATH=$R/$lib rocprofv3 --kernel-trace --stats -d /tmp/abp -o out --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy > /tmp/ab.json 2>/dev/null
    f=$(find /tmp/abp -name "*kernel_stats.csv" | head -1)
    python - "$f" "$lib" <<'PY'
import csv, sys
k = {r["Name"].split("(")[0]: float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(sys.argv[1]))}
rev = [v for n, v in k.items() if "k_loglik_t_rev" in n]; fwd = [v for n, v in k.items() if "k_loglik_t_fwd" in n]
print("%-40s fwd %.2f ms  rev %.2f ms  sum %.2f" % (sys.argv[2], fwd[0], rev[0], fwd[0] + rev[0]), flush=True)
PY
  done
done
