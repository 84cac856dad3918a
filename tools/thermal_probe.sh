R=$GRAFT_REPO_ROOT
q() { python $R/bench.py --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), 'ms; kernel median', round(d['roofline']['kernel_ms_median'],2))"; }
rocm-smi --showtemp --showclocks --showpower 2>/dev/null | grep -i "temp\|sclk\|mclk\|power" | head -8
q cold
q again
python $R/tools/bench_ops.py 8192 > /dev/null 2>&1; python $R/tools/bench_ops.py 8192 > /dev/null 2>&1
rocm-smi --showtemp --showclocks --showpower 2>/dev/null | grep -i "temp\|sclk\|mclk\|power" | head -8
q after-load
sleep 45
q after-45s-idle
rocm-smi --showtemp --showpower 2>/dev/null | grep -i "temp\|power" | head -6
