#!/bin/bash
# tools/alloc_probe.sh: the headline step from fresh processes, torch's default allocator against expandable segments, alternating
R=$GRAFT_REPO_ROOT
q() { python $R/bench.py --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2))"; }
for i in 1 2 3 4 5; do
  q default
  PYTORCH_HIP_ALLOC_CONF=expandable_segments:True q expandable
done
