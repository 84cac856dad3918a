import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from celerite2_amd import _lib, ops, synth
dev = torch.device("cuda:0")
for B in [int(x) for x in (sys.argv[1:] or ["128", "256", "512", "1024", "2048"])]:
    t, c, a, U, V, y = synth.device_batch_fast(0, B, 4096, 8, dev)
    fn = lambda: ops.loglik_grad(t, c, a, U, V, y)
    r = {}
    _lib.set_option("timepar_grad", 0); r["rows"] = synth.timed_steady(fn, reps=5)
    for rows in (16, 32, 64):
        _lib.set_option("timepar_grad", 1); _lib.set_option("tpg_rows", rows)
        try: r[rows] = synth.timed_steady(fn, reps=5)
        except Exception as e: r[rows] = float("nan")
    _lib.set_option("tpg_rows", None); _lib.set_option("timepar_grad", None)
    r["default"] = synth.timed_steady(fn, reps=5)
    print("B %5d: rows %.3f | chunks of 16: %.3f  32: %.3f  64: %.3f | default %.3f" % (B, r["rows"], r[16], r[32], r[64], r["default"]), flush=True)
