mkdir -p gpurun_out/r5s3g
for n in 4096 4112 4128 4096 4112; do
for b in 8192 65536; do
python bench.py --batch-per-gpu $b --N $n --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('N', $n, 'B', d['config']['batch_per_gpu'], 'ms', round(d['ms_per_step'],3), 'us/row/1k series', round(d['ms_per_step']*1e3/$n/($b/1000.0),5))"
done; done > gpurun_out/r5s3g/n.txt 2>&1
cat gpurun_out/r5s3g/n.txt
