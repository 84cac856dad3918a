#!/bin/bash
# HBM bytes of the forward-only kernels at B=65536 (separate PMC passes, as the guide prescribes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace -d $R/gpurun_out/pmc_t_$ctr -o out -- python $R/tools/lanes_fwd3.py > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob("$R/gpurun_out/pmc_t_%s/**/*counter_collection.csv" % ctr, recursive=True):
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"][:60]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        if "loglik" in k: print(ctr, k, "n=%d" % len(v), "avg per dispatch (KB as reported) %.4g" % (sum(v) / len(v)))
PY
