#!/bin/bash
# N fresh-process default bench runs (informational objects off): the process-to-process spread of the headline step.
# usage: tools/headline_spread.sh <outdir> [runs]
out=${1:-gpurun_out/spread}; n=${2:-10}
mkdir -p $out
for i in $(seq 1 $n); do
  python bench.py --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy > $out/run_$i.json 2> $out/run_$i.err
  python - $out/run_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%s ms_per_step %.3f frac %.4f kernel_ms %s" % (sys.argv[1], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("kernel_ms_avg")))
PY
done
