#!/bin/bash
# tools/kernel_breakdown.sh <op> <B> <N> <J> [env VAR=val ...]: every kernel of one op with its average duration and launches per call
# (rocprofv3 --kernel-trace; tools/op_run.py runs the op 2 + 5 times)
R=${GRAFT_REPO_ROOT:-/root/repo}
OP=$1; B=$2; N=$3; J=$4; shift 4
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kb; mkdir -p /tmp/kb
env "$@" timeout 300 rocprofv3 --kernel-trace -d /tmp/kb -o kb -- python $R/tools/op_run.py $OP $B $N $J 5 > /tmp/kb/log 2>&1
grep "ms per call" /tmp/kb/log
python - "$OP $B $N $J $*" <<'PY'
import glob, sqlite3, sys
for path in glob.glob("/tmp/kb/*_results.db") + glob.glob("/tmp/kb/*/*_results.db"):
    cur = sqlite3.connect(path).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
    rows = cur.execute("select name, count(*), avg(duration)/1e3, sum(duration)/1e3, min(duration)/1e3, max(duration)/1e3 from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[3] for r in rows)
    print("## %s  (7 calls; total kernel time per call %.1f us)" % (sys.argv[1], tot / 7))
    for n, c, a, s, mn, mx in rows[:24]:
        print("  %-90s x%-4.1f avg %8.1f us (min %.1f, max %.1f)   per call %8.1f us" % (n.split("(")[0].replace("void ", "")[:90], c / 7.0, a, mn, mx, s / 7))
PY
