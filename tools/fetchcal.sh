#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
$R/tools/ubench/fetchcal
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace -d $R/gpurun_out/fetchcal_$ctr -o out --output-format csv -- $R/tools/ubench/fetchcal > /dev/null 2>&1
done
python - <<PY
import csv, glob
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$R/gpurun_out/fetchcal_%s/*counter_collection.csv" % ctr):
        for row in csv.DictReader(open(f)):
            if "k_cal" in row["Kernel_Name"]:
                print("%-11s %-40s %.4f GB as reported (KiB x 1024)" % (ctr, row["Kernel_Name"][:40], float(row["Counter_Value"]) * 1024 / 1e9))
PY
