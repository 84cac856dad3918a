# -*- coding: utf-8 -*-
"""Time-parallel forward log-likelihood (c2_timepar.hip) against the row-by-row kernels: agreement and timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
def timed(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for J in (4, 2):
    for B, N in ((1024, 4096), (256, 4096), (4096, 4096), (16384, 4096), (32, 50000), (1024, 700)):
        t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
        os.environ["C2_TIMEPAR"] = "0"
        ll0, f0 = ops.loglik(t, c, a, U, V, y); ms0 = timed(lambda: ops.loglik(t, c, a, U, V, y))
        os.environ["C2_TIMEPAR"] = "1"
        ll1, f1 = ops.loglik(t, c, a, U, V, y); ms1 = timed(lambda: ops.loglik(t, c, a, U, V, y))
        rel = float(((ll1 - ll0).abs() / ll0.abs()).max())
        nbytes = B * N * 8 * (3 + 2 * J)
        print("J %d B %6d N %6d: row-by-row %.3f ms, time-parallel %.3f ms (%.2fx, %.3f of 8 TB/s)  max rel diff of ll %.1e  flags %d/%d"
              % (J, B, N, ms0, ms1, ms0 / ms1, nbytes / ms1 / 1e6 / 8000, rel, int((f0 != 0).sum()), int((f1 != 0).sum())), flush=True)
        del t, c, a, U, V, y
