#!/bin/bash
# tools/k2_pmc.sh [series per GPU]: SQ counters of the two-lanes-per-series pair (two PMC passes, every profiler call guarded)
R=${GRAFT_REPO_ROOT:-/root/repo}
BP=${1:-16384}
cd /tmp; export TMPDIR=/tmp
ARGS="--batch-per-gpu $BP --steps 3 --warmup 1 --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy"
for grp in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  rm -rf /tmp/pk; mkdir -p /tmp/pk
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pk -o pk -- python $R/bench.py $ARGS > /tmp/pk/log 2>&1 || { echo "group [$grp]: profiler failed / timed out"; tail -3 /tmp/pk/log; continue; }
  python - <<'PY'
import glob, sqlite3
f = glob.glob("/tmp/pk/**/*_results.db", recursive=True)
if not f:
    print("no db"); raise SystemExit
cur = sqlite3.connect(f[0]).cursor()
for kern in ("k_k2_rev", "k_k2_fwd"):
    d = cur.execute("select avg(duration)/1e6 from kernels where name like ?", ("%" + kern + "%",)).fetchone()
    q = ("select counter_name, avg(v) from (select dispatch_id, counter_name, sum(value) as v from counters_collection "
         "where kernel_name like ? group by dispatch_id, counter_name) group by counter_name")
    c = dict(cur.execute(q, ("%" + kern + "%",)).fetchall())
    print("%-10s %6.2f ms  " % (kern, d[0] or 0.0) + "  ".join("%s %.4g" % kv for kv in sorted(c.items())), flush=True)
PY
done
