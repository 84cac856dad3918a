#!/bin/bash
# tools/lanes8_probe.sh [tag ...]: per-kernel time and dynamic instruction counts of the 8-lanes-per-series pair at the shard of
# the literal configs[2] on eight GPUs (B = 8192, N = 4096, J = 8).  rocprofv3 writes to /tmp; only the summary is printed.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in "${@:-main}"; do
  if [ "$tag" = main ]; then unset C2_LIB_PATH; else export C2_LIB_PATH=$R/celerite2_amd/libcelerite2_amd_$tag.so; fi
  echo "=== $tag"
  CMD="python $R/tools/lanes8_run.py 4096 8192"
  D=/tmp/l8_${tag}; rm -rf $D; mkdir -p $D
  $CMD
  rocprofv3 --kernel-trace --stats -d $D/stats -o out --output-format csv -- $CMD > $D.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $D/pmc -o out --output-format csv -- $CMD >> $D.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("$D/stats/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "loglik" in r["Name"]:
        print("  %-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$D/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "loglik" in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    waves = 8192 / 8
    print("  ", k)
    print("     per wavefront and row: " + "  ".join("%s %.1f" % (c.replace("SQ_", ""), sum(v) / len(v) / waves / 4095) for c, v in sorted(d.items())))
PY
done
