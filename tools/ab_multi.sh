#!/bin/bash
# tools/ab_multi.sh <rounds> <tag...>: the headline step from fresh processes, the regular library ("main") and variants
# (libcelerite2_amd_<tag>.so, tools/build_variant.sh) alternating; prints ms per step and the kernels' own sum per process
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
rounds=$1; shift
q() {
  if [ "$1" != main ]; then export C2_LIB_PATH=$R/celerite2_amd/libcelerite2_amd_$1.so; else unset C2_LIB_PATH; fi
  python $R/bench.py --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy --steps 10 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2))"
}
for i in $(seq 1 $rounds); do
  for t in main "$@"; do q $t; done
done
