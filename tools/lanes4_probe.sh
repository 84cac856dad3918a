cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for l in 4 8; do rm -rf /tmp/l4p; rocprofv3 --kernel-trace --stats -d /tmp/l4p -o out --output-format csv -- python $R/tools/lanes_any_run.py $l 4096 16384 > /dev/null 2>&1
python - $l <<'PY'
import csv, glob, sys
for r in csv.DictReader(open(glob.glob("/tmp/l4p/**/*kernel_stats.csv", recursive=True)[0])):
    if "loglik" in r["Name"]: print(sys.argv[1], "lanes:", r["Name"][:60], r["Calls"], "avg %.2f ms" % (float(r["AverageNs"]) / 1e6))
PY
done
