#!/bin/bash
# same-box A/B of the bench line: default library vs the variants given as arguments (tags of tools/build_variant.sh)
R=$GRAFT_REPO_ROOT
run() { python $R/bench.py --no-cpu-baseline --steps 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],2))"; }
for rep in 1 2 3; do
  run default
  for tag in "$@"; do C2_LIB_PATH=$R/celerite2_amd/libcelerite2_amd_$tag.so run $tag; done
done
