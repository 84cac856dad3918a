#!/bin/bash
# tools/r04_measure.sh <outdir>: the round's measurement set on one box -- shard table of configs[2], per-op table, bench
# configs, the bench step under rocprofv3 (kernel trace) and its HBM traffic (PMC passes).  Every profiler call is guarded.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=${1:-$R/gpurun_out/r4f}; mkdir -p $O
cd $R
for B in 65536 32768 16384 8192; do
  python bench.py --global-batch $B --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shard $B', round(d['ms_per_step'],3), round(d['value']), round(d['roofline']['frac'],4))"
done > $O/shards.txt 2>&1
python tools/bench_ops.py 8192 > $O/per_op_B8192.md 2> $O/per_op.err
python tools/bench_configs.py > $O/bench_configs.jsonl 2> $O/bench_configs.err
timeout -k 5 300 bash tools/final_rocprof.sh > $O/final_bench_rocprof.md 2> $O/final_rocprof.err
cp /tmp/fr_bench.json $O/final_bench.json 2>/dev/null
timeout -k 5 400 bash tools/traffic.sh $O/traffic.json > $O/traffic.txt 2>&1
tail -3 $O/shards.txt $O/traffic.txt $O/final_bench_rocprof.md
