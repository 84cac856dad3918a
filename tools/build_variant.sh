#!/bin/bash
# tools/build_variant.sh <tag> <source.hip> <extra hipcc flags...>: A/B build -- libcelerite2_amd_<tag>.so with ONE source
# recompiled under extra flags, the other objects taken from the regular build.  Select it with C2_LIB_PATH.
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
obj=celerite2_amd/build/$(basename ${src%.hip})_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm --amdgpu-sched-strategy=max-ilp "$@" -c celerite2_amd/csrc/$src -o $obj
objs=""
stem=$(basename ${src%.hip})
for o in celerite2_amd/build/*.o; do
  b=$(basename $o .o)
  # every REGULAR object (one with a source file of its name) except the one being replaced; variant objects of earlier
  # A/B builds (<stem>_<tag>.o) have no source of that name
  [ -f celerite2_amd/csrc/$b.hip ] || continue
  [ "$b" = "$stem" ] && continue
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $obj -o celerite2_amd/libcelerite2_amd_$tag.so
echo built celerite2_amd/libcelerite2_amd_$tag.so
