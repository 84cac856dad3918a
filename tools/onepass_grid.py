# -*- coding: utf-8 -*-
"""Forward log-likelihood (default) or `factor` (argument "factor") at widths 4 and 2: row by row (C2_TIMEPAR=0) against the
time-parallel form built on chunk elements (C2_TIMEPAR=1) over a grid of batch sizes and lengths, and whether the default
dispatch picks the faster one ([default ...] marks a miss)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
FACTOR = len(sys.argv) > 1 and sys.argv[1] == "factor"
WIDTHS = (8,) if len(sys.argv) > 1 and sys.argv[1] == "8" else (4, 2)   # "8": forward log-likelihood at width 8 (rows = C2_TIMEPAR=0 C2_FACTOR_ITER=0)
def timed(fn, reps=8, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for J in WIDTHS:
    for B in (1, 64, 512, 1024, 2048, 4096, 8192, 12288, 16384, 20480):
        row = []
        for N in (128, 192, 256, 384, 512, 768, 1024, 4096) + ((100000,) if B <= 64 else ()):
            t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
            fn = (lambda: ops.factor(t, c, a, U, V)) if FACTOR else (lambda: ops.loglik(t, c, a, U, V, y))
            ms = []
            for tp in ("0", "1", None):
                if tp is None: os.environ.pop("C2_TIMEPAR", None)
                else: os.environ["C2_TIMEPAR"] = tp
                if tp == "0" and J == 8: os.environ["C2_FACTOR_ITER"] = "0"
                else: os.environ.pop("C2_FACTOR_ITER", None)
                fn()
                ms.append(timed(fn))
            row.append("%d: %.3f/%.3f%s" % (N, ms[0], ms[1], "" if ms[2] <= 1.08 * min(ms[0], ms[1]) else " [default %.3f]" % ms[2]))
            del t, c, a, U, V, y
        print("J %d B %5d  rows/time-parallel ms  " % (J, B) + "  ".join(row), flush=True)
