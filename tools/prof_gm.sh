#!/bin/bash
# per-kernel durations + resource summary of general_matmul_lower at B = 8192, N = M = 4096, J = 8, nrhs = 1 and 8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/gm.py <<PY
import sys; sys.path.insert(0, "$R")
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
B, N, J = 8192, 4096, 8
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
t2 = (t + 0.013).contiguous()
for nrhs in (1, 8):
    Y = torch.randn((B, N, nrhs), dtype=torch.float64, device=dev)
    for _ in range(4): Z = ops.general_matmul_lower(t2, t, c, U, V, Y)
torch.cuda.synchronize()
PY
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/gm_stats -o out --output-format csv -- python /tmp/gm.py > /dev/null 2>&1
python - <<PY
import csv, glob
for r in list(csv.DictReader(open(glob.glob("$R/gpurun_out/gm_stats/**/*kernel_stats.csv", recursive=True)[0])))[:8]:
    print("%-80s calls %4s avg %9.1f us" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
