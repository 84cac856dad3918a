#!/bin/bash
# tools/traffic.sh <out.json> [series per GPU ...] -- HBM bytes per bench step from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
# in separate runs, as MI355X_MICROARCH.md prescribes): 2 x FETCH_SIZE + WRITE_SIZE, KiB x 1024 (calibration:
# profiles/r02_fetch_size_calibration.md).  Prints the per-kernel figures and writes the profiles/r*_traffic.json layout
# (one workload per batch size; default: the bench's 65536).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$(realpath -m ${1:-$R/gpurun_out/traffic.json}); shift
BS=${@:-65536}
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-long-series --no-coefficient-level --no-gappy"
for B in $BS; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/tr_${B}_$ctr; mkdir -p /tmp/tr_${B}_$ctr
    rocprofv3 --kernel-trace --pmc $ctr -d /tmp/tr_${B}_$ctr -o tr -- python $R/bench.py --batch-per-gpu $B $ARGS > /tmp/tr_${B}_$ctr/log 2>&1
  done
done
python - "$OUT" $BS <<'PY'
import glob, json, sqlite3, sys
work = []
for B in [int(x) for x in sys.argv[2:]]:
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        db = sqlite3.connect(glob.glob("/tmp/tr_%d_%s/**/*_results.db" % (B, ctr), recursive=True)[0])
        q = ("select kernel_name, avg(v), count(*) from (select dispatch_id, kernel_name, sum(value) as v from counters_collection "
             "where counter_name = ? group by dispatch_id, kernel_name) group by kernel_name")
        for name, v, n in db.execute(q, (ctr,)):
            # ("16, 8, 2,": the FACTOR-mode kernel of the untimed c2_condition call behind the timed region -- not the step)
            if any(k in name for k in ("k_loglik", "k_q4_fwd", "k_q4_rev", "k_k2_", "k_anchor")) and "16, 8, 2," not in name and n >= 3:
                short = name.split("(")[0].replace("void ", "").replace("c2t_j8::", "").replace("c2t::", "").replace("c2::", "")
                vals.setdefault(short, {})[ctr] = v
    kern = {k: int((2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024) for k, v in vals.items()}
    kern = {k: v for k, v in kern.items() if v > 1e7}
    print("== %d series per GPU" % B)
    for k in kern:
        v = vals[k]
        print("%-44s FETCH_SIZE %.4g KiB  WRITE_SIZE %.4g KiB  -> %.2f GB" % (k, v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0), kern[k] / 1e9))
    total = sum(kern.values())
    print("total %.2f GB per step = %.3f x the %.2f GB algorithmic" % (total / 1e9, total / (B * 1245320), B * 1245320 / 1e9))
    work.append({"mode": "grad", "batch_per_gpu": B, "N": 4096, "J": 8, "traffic_bytes_per_step": total, "kernels": kern})
json.dump({"note": "HBM bytes per bench step from rocprofv3 PMC passes: 2 x FETCH_SIZE + WRITE_SIZE, KiB x 1024 (tools/traffic.sh; "
                   "calibration profiles/r02_fetch_size_calibration.md)", "workloads": work}, open(sys.argv[1], "w"), indent=1)
PY
