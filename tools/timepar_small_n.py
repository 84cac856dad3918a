"""Where the time-parallel kernels start to pay for very small batches: B in (1, 8, 64), N from 128 to 4096, widths 2 / 4
(8 for the solves): forward log-likelihood, factor, solve_lower, row by row (C2_TIMEPAR=0) against time-parallel (=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
def timed(fn, reps=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for J in (2, 4, 8):
    for B in (1, 8, 64):
        for N in (128, 256, 512, 1000, 2048, 4096):
            t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
            d = torch.empty_like(a); W = torch.empty_like(V); Z = torch.empty((B, N, 1), dtype=torch.float64, device=dev)
            Y = y.unsqueeze(-1).contiguous()
            row = []
            for tp in ("0", "1"):
                os.environ["C2_TIMEPAR"] = tp
                r = []
                if J != 8:
                    r.append(timed(lambda: ops.loglik(t, c, a, U, V, y)))
                    r.append(timed(lambda: ops.factor(t, c, a, U, V, d=d, W=W)))
                else:
                    ops.factor(t, c, a, U, V, d=d, W=W); r += [0.0, 0.0]
                r.append(timed(lambda: ops.solve_lower(t, c, U, W, Y, Z=Z)))
                row.append(r)
            print("J %d B %3d N %5d  loglik %6.1f -> %6.1f us   factor %6.1f -> %6.1f us   solve_lower %6.1f -> %6.1f us" % (
                J, B, N, row[0][0], row[1][0], row[0][1], row[1][1], row[0][2], row[1][2]), flush=True)
