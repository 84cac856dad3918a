# tools/r06_collect.sh -- the round's profile collection in one gpurun call (outputs under gpurun_out/r6p; copied into profiles/r06_*)
mkdir -p gpurun_out/r6p; P=$PWD/gpurun_out/r6p
python bench.py > $P/bench.json 2> $P/bench.err
python bench.py --parity-sample all --no-gappy --no-long-series --no-coefficient-level > $P/bench_parity_all.json 2> $P/bench_parity_all.err
tools/final_rocprof.sh > $P/rocprof_65536.md 2>&1
tools/final_rocprof.sh 8192 > $P/rocprof_8192.md 2>&1
tools/traffic.sh $P/traffic.json 65536 8192 > $P/traffic.txt 2>&1
tools/shard_ab.sh > $P/shard_ab.txt 2>&1
python tools/bench_ops.py > $P/bench_ops.txt 2>&1
python tools/bench_configs.py 2 4 5 terms 2>&1 | grep "^{" > $P/bench_configs.jsonl
python tools/host_call_cost.py 2>&1 | grep -v amdgpu > $P/host_call_cost.txt
python tools/host_breakdown.py 2>&1 | grep -v amdgpu > $P/host_breakdown.txt
python tools/large_nrhs.py 2>&1 | grep -v amdgpu > $P/large_nrhs.txt
python -c "
import json; d=json.loads(open('$P/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['parity_sample'].get('worst'), d['parity_sample'].get('within_1e-10'), d['cpu_baseline']['value'], d['config']['kappa'])
d=json.loads(open('$P/bench_parity_all.json').read().strip().splitlines()[-1]); print(json.dumps(d['parity_sample'].get('step_batch_all'))[:600])"
tail -4 $P/shard_ab.txt; tail -2 $P/traffic.txt
