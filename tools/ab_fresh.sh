#!/bin/bash
# tools/ab_fresh.sh: the bench step in FRESH processes, the current library alternating with A/B builds (tools/build_variant.sh) on one box --
# process-to-process and box-to-box spreads are as large as a round's gains, so every change of round 6's second session was judged this way.
#   AB_BATCH=<series per GPU, default 65536> AB_REPS=<pairs, default 3> AB_VARIANTS="<tag> ..." bash tools/ab_fresh.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${AB_BATCH:-65536}
one() { local label=$1; shift
  env "$@" python $R/bench.py --batch-per-gpu $B --no-gappy --no-cpu-baseline --no-long-series --no-coefficient-level 2>/dev/null | tail -1 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['config']['batch_per_gpu'], round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"; }
for i in $(seq 1 ${AB_REPS:-3}); do
  one "default        " C2_NOP=1
  for v in $AB_VARIANTS; do one "  variant $v" C2_LIB_PATH=$R/celerite2_amd/libcelerite2_amd_$v.so; done
done
