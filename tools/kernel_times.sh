#!/bin/bash
# tools/kernel_times.sh <lanes> "<B B ...>" [lib.so ...]: average duration of the fused gradient kernels (rocprofv3 --kernel-trace)
# per batch size and per build of the library (C2_LIB_PATH; default: the regular build)
R=${GRAFT_REPO_ROOT:-/root/repo}
L=$1; BS=$2; shift 2
LIBS=${@:-celerite2_amd/libcelerite2_amd.so}
cd /tmp; export TMPDIR=/tmp
for lib in $LIBS; do
  for B in $BS; do
    rm -rf /tmp/kt; mkdir -p /tmp/kt
    C2_LIB_PATH=$R/$lib rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/lanes_any_run.py $L 4096 $B 6 > /tmp/kt/log 2>&1
    python - "$lib" "$B" <<'PY'
import glob, sqlite3, sys
for path in glob.glob("/tmp/kt/*_results.db") + glob.glob("/tmp/kt/*/*_results.db"):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), avg(duration)/1e6, min(duration)/1e6 from kernels group by name").fetchall()
    out = []
    for n, c, a, m in rows:
        if ("k_loglik" in n or "k_q4" in n or "k_k2" in n) and a > 0.05:
            out.append("%s avg %.3f min %.3f ms (%d)" % (n.split("(")[0].replace("void c2::", "").replace("c2::", "")[:44], a, m, c))
    print("%-44s B=%-6s %s" % (sys.argv[1].split("/")[-1], sys.argv[2], " | ".join(sorted(out))))
PY
  done
done
