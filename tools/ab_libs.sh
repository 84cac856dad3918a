#!/bin/bash
# tools/ab_libs.sh "<bench args>" lib1.so lib2.so ... -- A/B a set of experimental builds of the library with bench.py
# (each picked up through C2_LIB_PATH) on the SAME GPU box; prints GP/s, ms/step and the HIP-event kernel time per build.
ARGS=$1; shift
for lib in "$@"; do
  C2_LIB_PATH=$PWD/$lib timeout 400 python bench.py --no-cpu-baseline $ARGS 2>&1 | tail -1 | LIB=$lib python -c '
import sys, json, os
d = json.loads(sys.stdin.read())
r = d["roofline"]
print("%-36s %.0f GP/s  %.3f ms/step  kernel avg %.3f median %.3f ms" % (os.environ["LIB"], d["value"], d["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_median"]))'
done
