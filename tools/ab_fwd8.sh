R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for tag in main "$@"; do
  if [ "$tag" = main ]; then unset C2_LIB_PATH; else export C2_LIB_PATH=$R/celerite2_amd/libcelerite2_amd_$tag.so; fi
  rm -rf /tmp/ab8p; rocprofv3 --kernel-trace --stats -d /tmp/ab8p -o out --output-format csv -- python $R/tools/bench_ops.py 8192 "fused log-lik" > /tmp/ab8.log 2>&1
  f=$(find /tmp/ab8p -name "*kernel_stats.csv" | head -1)
  python - "$f" "$tag" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_loglik_" in r["Name"]: print("%-8s %-50s calls %4s avg %8.1f us" % (sys.argv[2], r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
