#!/bin/bash
# tools/latency_pmc.sh [runs]: TCP->TCC read / write latency counters of the one-lane pair in fresh processes, next to the
# kernels' durations (one counter group, every profiler call guarded).  Prints; leaves nothing under gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy"
for i in $(seq 1 ${1:-4}); do
  rm -rf /tmp/pp; mkdir -p /tmp/pp
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum -d /tmp/pp -o pp -- python $R/bench.py $ARGS > /tmp/pp/log 2>&1 || { echo "run $i: profiler failed / timed out"; continue; }
  python - <<'PY'
import glob, sqlite3
f = glob.glob("/tmp/pp/**/*_results.db", recursive=True)
if not f:
    print("no db"); raise SystemExit
cur = sqlite3.connect(f[0]).cursor()
for kern in ("k_loglik_t_rev", "k_loglik_t_fwd"):
    d = cur.execute("select avg(duration)/1e6 from kernels where name like ?", ("%" + kern + "%",)).fetchone()
    q = ("select counter_name, avg(v) from (select dispatch_id, counter_name, sum(value) as v from counters_collection "
         "where kernel_name like ? group by dispatch_id, counter_name) group by counter_name")
    c = dict(cur.execute(q, ("%" + kern + "%",)).fetchall())
    g = lambda k: c.get(k, 0.0)
    rl = g("TCP_TCC_READ_REQ_LATENCY_sum") / max(g("TCP_TCC_READ_REQ_sum"), 1.0)
    wl = g("TCP_TCC_WRITE_REQ_LATENCY_sum") / max(g("TCP_TCC_WRITE_REQ_sum"), 1.0)
    print("%-15s %6.2f ms   read latency %5.0f cycles (%.3g requests)   write latency %5.0f cycles (%.3g requests)" % (kern, d[0], rl, g("TCP_TCC_READ_REQ_sum"), wl, g("TCP_TCC_WRITE_REQ_sum")), flush=True)
PY
done
