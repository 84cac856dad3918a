#!/bin/bash
# tools/ab_headline.sh <tag> [rounds]: the headline step from fresh processes, regular library against a variant, alternating
R=$GRAFT_REPO_ROOT
q() { python $R/bench.py --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2))"; }
for i in $(seq 1 ${2:-6}); do
  q main
  C2_LIB_PATH=$R/celerite2_amd/libcelerite2_amd_$1.so q $1
done
