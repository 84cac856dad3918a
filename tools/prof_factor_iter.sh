#!/bin/bash
# per-launch durations of the chunk passes: prof_factor_iter.sh B,N,J
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/fi_stats
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/fi_stats -o out --output-format csv -- python $R/tools/factor_iter_time.py $1 > /dev/null 2>&1
python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$R/gpurun_out/fi_stats/**/*kernel_trace.csv", recursive=True)[0])))
rows = [r for r in rows if "k_newton" in r["Kernel_Name"] or "k_loglik_fwd" in r["Kernel_Name"]]
last = rows[-24:]
print(" ".join("%s%.0f" % ("P" if "newton_pass" in r["Kernel_Name"] else ("C" if "newton_chain" in r["Kernel_Name"] else "F"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in last))
PY
