mkdir -p gpurun_out/r5p
tools/shard_ab.sh > gpurun_out/r5p/shard_ab.txt 2>&1
for b in 8192 16384 32768; do tools/final_rocprof.sh $b > gpurun_out/r5p/rocprof_$b.md 2>&1; done
tools/final_rocprof.sh > gpurun_out/r5p/rocprof_65536.md 2>&1
tools/traffic.sh gpurun_out/r5p/traffic.json 65536 32768 16384 8192 > gpurun_out/r5p/traffic.txt 2>&1
python bench.py > gpurun_out/r5p/bench.json 2> gpurun_out/r5p/bench.err
python tools/host_call_cost.py > gpurun_out/r5p/host_call_cost.txt 2>&1
tail -3 gpurun_out/r5p/shard_ab.txt; tail -2 gpurun_out/r5p/traffic.txt; python -c "
import json; d=json.loads(open('gpurun_out/r5p/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['parity_sample'].get('worst'), d['parity_sample'].get('within_1e-10'), d['cpu_baseline']['value'])"
# the element kernels of c2_timepar.hip (configs[1] and the drop-in shapes): profiles/r05_onepass.md
mkdir -p gpurun_out/r5q; Q=$PWD/gpurun_out/r5q
python tools/bench_configs.py 2 4 5 terms 2>&1 | grep "^{" > $Q/bench_configs.jsonl
python tools/onepass_grid.py 2>&1 | grep "^J" > $Q/grid_loglik.txt
python tools/onepass_grid.py factor 2>&1 | grep "^J" > $Q/grid_factor.txt
bash tools/traffic_any.sh k_tp_ $PWD/tools/bench_configs.py 2 2>&1 | tail -3 > $Q/traffic_cfg1.txt
bash tools/pmc_any.sh k_tp_onepass $PWD/tools/bench_configs.py 2 2>&1 | tail -20 > $Q/pmc_cfg1.txt
(cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_configs.py 2 > /tmp/kt.log 2>&1; python - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/r5q/kernel_trace_cfg1.md
import csv, glob
f = glob.glob("/tmp/kt/**/*kernel_stats.csv", recursive=True)[0]
print("| kernel | calls | avg us | min us | max us |"); print("|---|---|---|---|---|")
for r in csv.DictReader(open(f)):
    if "k_tp_" in r["Name"]:
        print("| `%s` | %s | %.1f | %.1f | %.1f |" % (r["Name"].split("(")[0].replace("void ", ""), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
)
python tools/host_breakdown.py 2>&1 | grep -v amdgpu > $Q/host_breakdown.txt
cat $Q/bench_configs.jsonl | cut -c1-160; cat $Q/traffic_cfg1.txt $Q/kernel_trace_cfg1.md
