#!/bin/bash
# tools/spacer_probe.sh: does WHERE in HBM the step's arrays land decide the mode of a process?  Fresh processes, each after
# 25 s of idling (the predecessor's memory is back: the state of a fresh box), with and without a spacer held during allocation.
R=${GRAFT_REPO_ROOT:-/root/repo}
q() { C2_BENCH_SPACER_GB=$1 python $R/bench.py --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy --steps 10 2>/dev/null \
  | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spacer $1 GB:', round(d['ms_per_step'],2))"; }
for rep in 1 2 3; do
  for gb in ${GBS:-0 60 120 150}; do sleep ${SLEEP:-20}; q $gb; done
done
