#!/bin/bash
# Round-2 measurement set: per-op table, BASELINE configs, bench line, rocprofv3 stats + PMC traffic of the bench command.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
python tools/bench_ops.py 8192 > $O/per_op_B8192.md 2> $O/per_op.err
python tools/bench_configs.py > $O/bench_configs.jsonl 2> $O/bench_configs.err
python tools/lanes_sweep.py > $O/lanes_sweep.txt 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --mode fwd --no-cpu-baseline > $O/bench_fwd.json 2>> $O/bench_default.err
python bench.py --batch-per-gpu 8192 --no-cpu-baseline > $O/bench_B8192.json 2>> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_bench -o out --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/prof_bench.json 2> $O/prof_bench.err
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $O/pmc_$name -o out --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, json
O = "$O"
rows = list(csv.DictReader(open(O + "/prof_bench/out_kernel_stats.csv")))
with open(O + "/prof_bench_summary.md", "w") as f:
    f.write("| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n")
    for r in rows[:12]:
        f.write("| %s | %s | %.1f | %.2f | %s |\n" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(O + "/pmc_*/*counter_collection.csv"):
    for row in csv.DictReader(open(fn)):
        k = row["Kernel_Name"]
        if "loglik" in k: agg[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(O + "/pmc_summary.md", "w") as f:
    for k, d in agg.items():
        f.write("## %s\n" % k)
        for c, v in sorted(d.items()): f.write("- %s: %.5g (mean of %d dispatches)\n" % (c, sum(v) / len(v), len(v)))
print(open(O + "/prof_bench_summary.md").read()); print(open(O + "/pmc_summary.md").read())
PY
cat $O/bench_default.json | cut -c1-300
