mkdir -p gpurun_out/r5p
tools/shard_ab.sh > gpurun_out/r5p/shard_ab.txt 2>&1
for b in 8192 16384 32768; do tools/final_rocprof.sh $b > gpurun_out/r5p/rocprof_$b.md 2>&1; done
tools/final_rocprof.sh > gpurun_out/r5p/rocprof_65536.md 2>&1
tools/traffic.sh gpurun_out/r5p/traffic.json 65536 32768 16384 8192 > gpurun_out/r5p/traffic.txt 2>&1
python bench.py > gpurun_out/r5p/bench.json 2> gpurun_out/r5p/bench.err
python tools/host_call_cost.py > gpurun_out/r5p/host_call_cost.txt 2>&1
tail -3 gpurun_out/r5p/shard_ab.txt; tail -2 gpurun_out/r5p/traffic.txt; python -c "
import json; d=json.loads(open('gpurun_out/r5p/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['parity_sample'].get('worst'), d['parity_sample'].get('within_1e-10'), d['cpu_baseline']['value'])"
