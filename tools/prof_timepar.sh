#!/bin/bash
# per-kernel durations of the time-parallel forward pass at configs[1] (B = 1024, N = 4096, J = 4)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/tp.py <<PY
import os, sys; sys.path.insert(0, "$R")
import torch
from celerite2_amd import ops, synth
os.environ["C2_TIMEPAR"] = "1"
t, c, a, U, V, y = synth.device_batch_fast(0, 1024, 4096, 4, torch.device("cuda:0"))
for _ in range(6): ll, f = ops.loglik(t, c, a, U, V, y)
torch.cuda.synchronize()
PY
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/tp_stats -o out --output-format csv -- python /tmp/tp.py > /dev/null 2>&1
python - <<PY
import csv, glob
for r in list(csv.DictReader(open(glob.glob("$R/gpurun_out/tp_stats/**/*kernel_stats.csv", recursive=True)[0]))):
    if "k_tp" in r["Name"] or "k_loglik" in r["Name"]:
        print("%-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
