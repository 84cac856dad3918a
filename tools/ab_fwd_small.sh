R=$GRAFT_REPO_ROOT
for n in 2 3 4 5 7; do for op in solve_lower matmul_upper solve_lower_ws; do python $R/tools/ab_option.py sweept $op $n 2>&1 | grep -v amdgpu; done; done
