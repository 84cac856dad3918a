#!/bin/bash
# tools/placement_pmc.sh -- what differs between a fast (16 ms) and a slow (20 ms) run of the one-lane reverse sweep?  Several
# fresh processes (each gets its own physical placement of the nine streams), one counter group each; per run: the average
# duration of k_loglik_t_rev next to its counters.  Prints one line per run; leaves nothing under gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy"
run() {  # tag, counters...
  tag=$1; shift
  for i in 1 2 3 4 5; do
    rm -rf /tmp/pp; mkdir -p /tmp/pp
    rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pp -o pp -- python $R/bench.py $ARGS > /tmp/pp/log 2>&1
    python - "$tag" <<'PY'
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob("/tmp/pp/**/*_results.db", recursive=True)[0])
cur = db.cursor()
for kern in ("k_loglik_t_rev", "k_loglik_t_fwd"):
    d = cur.execute("select avg(duration)/1e6, count(*) from kernels where name like ?", ("%" + kern + "%",)).fetchone()
    q = ("select counter_name, avg(v) from (select dispatch_id, counter_name, sum(value) as v from counters_collection "
         "where kernel_name like ? group by dispatch_id, counter_name) group by counter_name")
    c = dict(cur.execute(q, ("%" + kern + "%",)).fetchall())
    print("%s %-15s %6.2f ms  %s" % (sys.argv[1], kern, d[0], "  ".join("%s=%.4g" % (k.replace("_sum", ""), v) for k, v in sorted(c.items()))), flush=True)
PY
  done
}
run A TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
run B TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum
run C TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
