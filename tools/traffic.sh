#!/bin/bash
# tools/traffic.sh <out.json> -- HBM bytes per bench step from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate
# runs, as MI355X_MICROARCH.md prescribes): 2 x FETCH_SIZE + WRITE_SIZE, KiB x 1024 (calibration:
# profiles/r02_fetch_size_calibration.md).  Prints the per-kernel figures and writes the profiles/r*_traffic.json layout.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$(realpath -m ${1:-$R/gpurun_out/traffic.json})
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_$ctr; mkdir -p /tmp/tr_$ctr
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/tr_$ctr -o tr -- python $R/bench.py $ARGS > /tmp/tr_$ctr/log 2>&1
done
python - "$OUT" <<'PY'
import glob, json, sqlite3, sys
vals = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    db = sqlite3.connect(glob.glob("/tmp/tr_%s/**/*_results.db" % ctr, recursive=True)[0])
    q = ("select kernel_name, avg(v) from (select dispatch_id, kernel_name, sum(value) as v from counters_collection "
         "where counter_name = ? group by dispatch_id, kernel_name) group by kernel_name")
    for name, v in db.execute(q, (ctr,)):
        if "k_loglik_t_" in name:
            short = name.split("(")[0].replace("void ", "").replace("c2t_j8::", "").replace("c2t::", "")
            vals.setdefault(short, {})[ctr] = v
kern = {k: int((2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024) for k, v in vals.items()}
for k, v in vals.items():
    print("%-28s FETCH_SIZE %.4g KiB  WRITE_SIZE %.4g KiB  -> %.2f GB" % (k, v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0), kern[k] / 1e9))
total = sum(kern.values())
print("total %.2f GB per step = %.3f x the 81.61 GB algorithmic" % (total / 1e9, total / (65536 * 1245320)))
json.dump({"note": "HBM bytes per bench step from rocprofv3 PMC passes: 2 x FETCH_SIZE + WRITE_SIZE, KiB x 1024 (tools/traffic.sh; "
                   "calibration profiles/r02_fetch_size_calibration.md)",
           "workloads": [{"mode": "grad", "batch_per_gpu": 65536, "N": 4096, "J": 8, "traffic_bytes_per_step": total,
                          "kernels": kern}]}, open(sys.argv[1], "w"), indent=1)
PY
