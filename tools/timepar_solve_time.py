# -*- coding: utf-8 -*-
"""solve_lower / solve_upper (one right-hand side) on small batches of long series: time-parallel against row by row, and
the drop-in GP chain on host arrays (driver.factor + driver.solve_lower)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celerite2_amd import ops, synth, driver
dev = torch.device("cuda:0")
def timed(fn, reps=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for J in (8, 4, 2):
    for B, N in ((1, 100000), (1, 1000000), (32, 50000), (1024 if J < 8 else 512, 4096)):
        t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
        os.environ["C2_TIMEPAR"] = "0"
        d, W, fl = ops.factor(t, c, a, U, V)
        Y = y[:, :, None].contiguous(); Z = torch.empty_like(Y)
        row = ["J %d B %5d N %8d:" % (J, B, N)]
        for name in ("solve_lower", "solve_upper"):
            os.environ["C2_TIMEPAR"] = "0"; ms0 = timed(lambda: getattr(ops, name)(t, c, U, W, Y, Z=Z)); z0 = Z.clone()
            os.environ["C2_TIMEPAR"] = "1"; ms1 = timed(lambda: getattr(ops, name)(t, c, U, W, Y, Z=Z))
            row.append("%s %.3f -> %.3f ms (%.0fx, diff %.0e)" % (name, ms0, ms1, ms0 / ms1, float((Z - z0).abs().max() / z0.abs().max())))
        print("  ".join(row), flush=True)
    if J < 8:
        N = 100000
        t, c, a, U, V, y = [x[0].cpu().numpy() for x in synth.device_batch_fast(0, 1, N, J, dev)]
        dh, Wh, Zh = np.empty_like(a), np.empty_like(V), np.empty((N, 1))
        Yh = np.ascontiguousarray(y[:, None])
        for tp in ("0", "1"):
            os.environ["C2_TIMEPAR"] = tp
            driver.factor(t, c, a, U, V, dh, Wh); driver.solve_lower(t, c, U, Wh, Yh, Zh)
            t0 = time.perf_counter()
            for _ in range(5):
                driver.factor(t, c, a, U, V, dh, Wh); driver.solve_lower(t, c, U, Wh, Yh, Zh)
            print("   host drop-in, one series N = %d J = %d: driver.factor + driver.solve_lower, C2_TIMEPAR=%s: %.2f ms" % (N, J, tp, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
