#!/bin/bash
# tools/ab_lanes8.sh <B> [tag ...]: the 8-lane gradient pair at B series (N = 4096, J = 8) for A/B builds, alternating, 3 rounds
R=$GRAFT_REPO_ROOT
B=$1; shift
for round in 1 2 3; do
  for tag in main "$@"; do
    if [ "$tag" = main ]; then unset C2_LIB_PATH; else export C2_LIB_PATH=$R/celerite2_amd/libcelerite2_amd_$tag.so; fi
    echo -n "$tag: "; python $R/tools/lanes8_run.py 4096 $B 5 2>&1 | grep -v amdgpu
  done
done
