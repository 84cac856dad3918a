#!/bin/bash
# bench.py over a few series lengths around the headline N (checks sensitivity to the power-of-two batch stride)
for N in "$@"; do
  timeout 400 python bench.py --steps 5 --warmup 2 --N $N --no-cpu-baseline 2>&1 | tail -1 | N=$N python -c "
import sys,json,os
d=json.loads(sys.stdin.read()); N=int(os.environ['N'])
print('N=%d  %.0f GP/s  %.2f ms/step  (%.2f ms scaled to N=4096)' % (N, d['value'], d['ms_per_step'], d['ms_per_step']/N*4096))"
done
