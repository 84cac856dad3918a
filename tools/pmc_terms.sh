#!/bin/bash
# per-kernel durations + SQ wait/issue counters of the coefficient-level (fused) kernels, B = 65536, N = 1000
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-terms}
CMD="python $R/tools/terms_time.py 1000 65536"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_stats -o out --output-format csv -- $CMD > /dev/null 2>&1
f=$(find $R/gpurun_out/${tag}_stats -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:12]:
    print("%-70s calls %5s avg %10.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_IFETCH" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/${tag}_$name -o out --output-format csv -- $CMD > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/${tag}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "loglik_tt" in k:
            agg[k[:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-22s %.4g" % (c, sum(v) / len(v)))
PY
