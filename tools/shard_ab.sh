#!/bin/bash
# tools/shard_ab.sh: bench step at the shards of configs[2] in fresh processes on ONE box -- the dispatch's choice against the
# previous round's kernels forced, three processes each
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {  # label, batch, env...
  local label=$1 b=$2; shift 2
  for i in 1 2 3; do
    env "$@" python $R/bench.py --batch-per-gpu $b --no-gappy --no-cpu-baseline --no-long-series --no-coefficient-level 2>/dev/null | tail -1 \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['config']['batch_per_gpu'], round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
  done
}
run "two lanes (default)      " 32768 C2_NOP=1
run "one lane (round 3)       " 32768 C2_LANES=1
run "four lanes (default)      " 16384 C2_NOP=1
run "8 lanes (round 4)         " 16384 C2_LANES=8 C2_LOGLIK_SCALED=0
run "8 lanes, scaled frame (default)" 8192 C2_NOP=1
run "8 lanes, plain frame (round 4)" 8192 C2_LOGLIK_SCALED=0
