R=$GRAFT_REPO_ROOT
for tag in "$@"; do echo "== $tag"; C2_LIB_PATH=$R/celerite2_amd/libcelerite2_amd_$tag.so python $R/tools/sweepk_rev_quick.py 8 2>&1 | grep -v amdgpu; done
