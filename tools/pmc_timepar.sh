#!/bin/bash
# SQ counters of the time-parallel kernels at configs[1]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/tp.py <<PY
import os, sys; sys.path.insert(0, "$R")
import torch
from celerite2_amd import ops, synth
os.environ["C2_TIMEPAR"] = "1"
t, c, a, U, V, y = synth.device_batch_fast(0, 1024, 4096, 4, torch.device("cuda:0"))
for _ in range(4): ll, f = ops.loglik(t, c, a, U, V, y)
torch.cuda.synchronize()
PY
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/tp_pmc -o out --output-format csv -- python /tmp/tp.py > /dev/null 2>&1
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/tp_stats -o out --output-format csv -- python /tmp/tp.py > /dev/null 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/tp_pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_tp" in row["Kernel_Name"]: agg[row["Kernel_Name"][:28]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print(k, " ".join("%s=%.3g" % (c.replace("SQ_", ""), sum(v) / len(v)) for c, v in sorted(d.items())))
for r in csv.DictReader(open(glob.glob("/tmp/tp_stats/**/*kernel_stats.csv", recursive=True)[0])):
    if "k_tp" in r["Name"]: print("%-40s avg %8.1f us" % (r["Name"][:40], float(r["AverageNs"]) / 1e3))
PY
