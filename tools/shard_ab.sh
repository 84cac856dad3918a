#!/bin/bash
# tools/shard_ab.sh: bench step at the shards of configs[2] in fresh processes on ONE box, three processes each -- the current
# library and, with SHARD_AB_BASE=<path to another build of libcelerite2_amd.so>, that build in alternation (A/B on one box:
# process-to-process and box-to-box spreads are as large as a round's gains).  SHARD_AB_OLD=1 adds the previous rounds' kernels
# forced through the dispatch options.
R=${GRAFT_REPO_ROOT:-/root/repo}
one() {  # label, batch, env...
  local label=$1 b=$2; shift 2
  env "$@" python $R/bench.py --batch-per-gpu $b --no-gappy --no-cpu-baseline --no-long-series --no-coefficient-level 2>/dev/null | tail -1 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['config']['batch_per_gpu'], round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
}
run() {  # label, batch, env...
  local label=$1 b=$2; shift 2
  for i in 1 2 3; do
    one "$label" $b "$@"
    [ -n "$SHARD_AB_BASE" ] && one "  (base build)            " $b C2_LIB_PATH=$SHARD_AB_BASE "$@"
  done
}
run "two lanes (default)      " 32768 C2_NOP=1
[ -n "$SHARD_AB_OLD" ] && run "one lane (round 3)       " 32768 C2_LANES=1
run "four lanes (default)     " 16384 C2_NOP=1
[ -n "$SHARD_AB_OLD" ] && run "8 lanes (round 4)        " 16384 C2_LANES=8 C2_LOGLIK_SCALED=0
run "8 lanes (default)        " 8192 C2_NOP=1
[ -n "$SHARD_AB_OLD" ] && run "8 lanes, plain frame (round 4)" 8192 C2_LOGLIK_SCALED=0
exit 0
