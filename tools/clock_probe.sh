#!/bin/bash
# tools/clock_probe.sh <lanes> <B> [B ...]: effective shader clock (GRBM_GUI_ACTIVE / duration) and the wave-cycle breakdown of the
# fused gradient kernels at several batch sizes -- tells power-limited clocks from contention (MI355X_MICROARCH.md, DVFS give-back)
R=${GRAFT_REPO_ROOT:-/root/repo}
L=$1; shift
cd /tmp; export TMPDIR=/tmp
for B in "$@"; do
  rm -rf /tmp/clk; mkdir -p /tmp/clk
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d /tmp/clk/p0 -o p0 -- python $R/tools/lanes_any_run.py $L 4096 $B 4 > /tmp/clk/p0.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/clk/p1 -o p1 -- python $R/tools/lanes_any_run.py $L 4096 $B 4 > /tmp/clk/p1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS -d /tmp/clk/p2 -o p2 -- python $R/tools/lanes_any_run.py $L 4096 $B 4 > /tmp/clk/p2.log 2>&1
  echo "== lanes $L, B = $B"
  python - <<'PY'
import glob, sqlite3
for path in sorted(glob.glob("/tmp/clk/*/*_results.db")):
    db = sqlite3.connect(path); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    dur = {n: d for n, d in cur.execute("select name, avg(duration) from kernels group by name") if "k_loglik" in n or "k_k2" in n or "k_q4_fwd" in n or "k_q4_rev" in n}
    if "counters_collection" not in tabs: continue
    q = ("select kernel_name, counter_name, avg(v) from (select dispatch_id, kernel_name, counter_name, sum(value) as v "
         "from counters_collection group by dispatch_id, kernel_name, counter_name) group by kernel_name, counter_name")
    for n, cn, v in cur.execute(q):
        if n in dur:
            extra = ""
            if cn == "GRBM_GUI_ACTIVE": extra = "  -> %.3f GHz (8 XCDs summed / 8) over %.3f ms" % (v / 8 / dur[n], dur[n] / 1e6)
            print("%-60s %-22s %.5g%s" % (n.split("(")[0].replace("void c2::", "")[:60], cn, v, extra))
PY
done
