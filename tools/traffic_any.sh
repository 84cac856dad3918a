#!/bin/bash
# tools/traffic_any.sh <kernel-substring> <python script + args...>: HBM bytes per launch of the matching kernels (rocprofv3 PMC,
# FETCH_SIZE and WRITE_SIZE in separate passes: 2 x FETCH_SIZE + WRITE_SIZE, KiB x 1024) next to their average duration.
K=$1; shift
cd /tmp; export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ta_$ctr; mkdir -p /tmp/ta_$ctr
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/ta_$ctr -o tr -- python "$@" > /tmp/ta_$ctr/log 2>&1
done
python - "$K" <<'PY'
import glob, sqlite3, sys
vals = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    db = sqlite3.connect(glob.glob("/tmp/ta_%s/**/*_results.db" % ctr, recursive=True)[0])
    q = ("select kernel_name, avg(v), count(*) from (select dispatch_id, kernel_name, sum(value) as v from counters_collection "
         "where counter_name = ? group by dispatch_id, kernel_name) group by kernel_name")
    for name, v, n in db.execute(q, (ctr,)):
        if sys.argv[1] in name:
            vals.setdefault(name.split("(")[0][-60:], {})[ctr] = v
    if ctr == "FETCH_SIZE":
        tq = "select name, avg(end - start), count(*) from kernels group by name"
        try:
            for name, d, n in db.execute(tq):
                if sys.argv[1] in name: vals.setdefault(name.split("(")[0][-60:], {})["ns"] = d
        except Exception as e:
            pass
for k, v in vals.items():
    tot = (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024
    ns = v.get("ns")
    print("%-62s read %.2f GB  written %.2f GB  total %.2f GB%s" % (k, 2 * v.get("FETCH_SIZE", 0) * 1024 / 1e9, v.get("WRITE_SIZE", 0) * 1024 / 1e9, tot / 1e9, ("  %.2f ms -> %.2f TB/s" % (ns / 1e6, tot / ns / 1e3)) if ns else ""))
PY
