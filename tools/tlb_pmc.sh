#!/bin/bash
# tools/tlb_pmc.sh [runs]: translation counters of the one-lane pair in several fresh processes next to the kernels' durations
# (does the slow mode of a process come with more UTCL1 / UTCL2 misses?).  Prints; leaves nothing under gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy"
rocprofv3 -L 2>/dev/null | grep -o "UTCL[0-9A-Za-z_]*\|TCP_UTCL1[A-Z0-9_]*\|TCC_[A-Z0-9_]*PROBE[A-Z0-9_]*\|[A-Z0-9_]*TLB[A-Z0-9_]*\|MALL[A-Z0-9_]*\|[A-Z0-9_]*_MALL_[A-Z0-9_]*" | sort -u | tr '\n' ' '; echo
run() {
  tag=$1; shift
  for i in $(seq 1 ${RUNS:-6}); do
    rm -rf /tmp/pp; mkdir -p /tmp/pp
    rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pp -o pp -- python $R/bench.py $ARGS > /tmp/pp/log 2>&1
    python - "$tag" <<'PY'
import glob, sqlite3, sys
f = glob.glob("/tmp/pp/**/*_results.db", recursive=True)
if not f:
    print(sys.argv[1], "no db:", open("/tmp/pp/log").read()[-300:]); sys.exit(0)
db = sqlite3.connect(f[0]); cur = db.cursor()
for kern in ("k_loglik_t_rev", "k_loglik_t_fwd"):
    d = cur.execute("select avg(duration)/1e6, count(*) from kernels where name like ?", ("%" + kern + "%",)).fetchone()
    q = ("select counter_name, avg(v) from (select dispatch_id, counter_name, sum(value) as v from counters_collection "
         "where kernel_name like ? group by dispatch_id, counter_name) group by counter_name")
    c = dict(cur.execute(q, ("%" + kern + "%",)).fetchall())
    print("%s %-15s %6.2f ms  %s" % (sys.argv[1], kern, d[0], "  ".join("%s=%.4g" % (k.replace("_sum", ""), v) for k, v in sorted(c.items()))), flush=True)
PY
  done
}
RUNS=${1:-6}
run A TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_PERMISSION_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_LRU_INFLIGHT_sum TCP_UTCL1_STALL_MULTI_MISS_sum
