#!/bin/bash
# counters of k_matrices at B = 8192, N = 4096, Jc = 4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/mat.py <<PY
import sys; sys.path.insert(0, "$R")
import torch
from celerite2_amd import ops, synth
t, diag, y, ac, bc, cc, dc = synth.device_coeffs_fast(0, 8192, 4096, 8, torch.device("cuda:0"))
e = torch.zeros((8192, 0), dtype=torch.float64, device="cuda")
for _ in range(4): a, U, V = ops.get_celerite_matrices(e, ac, bc, dc, t, diag)
torch.cuda.synchronize()
PY
for set in "FETCH_SIZE WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/mat_$name -o out --output-format csv -- python /tmp/mat.py > /dev/null 2>&1
done
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/mat_stats -o out --output-format csv -- python /tmp/mat.py > /dev/null 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/mat_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_matrices" in row["Kernel_Name"]: agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for c, v in sorted(agg.items()): print("%-24s %.5g" % (c, sum(v) / len(v)))
for r in csv.DictReader(open(glob.glob("$R/gpurun_out/mat_stats/**/*kernel_stats.csv", recursive=True)[0])):
    if "k_matrices" in r["Name"]: print("avg us", float(r["AverageNs"]) / 1e3)
PY
