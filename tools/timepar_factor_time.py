# -*- coding: utf-8 -*-
"""factor (d, W) on small batches of long series: time-parallel (C2_TIMEPAR=1) against row by row (=0), and the B = 1
host drop-in (driver.factor)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celerite2_amd import ops, synth, driver
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
    for B, N in ((1, 100000), (1, 1000000), (32, 50000), (1024, 4096)):
        t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
        d = torch.empty_like(a); W = torch.empty_like(V)
        os.environ["C2_TIMEPAR"] = "0"; ms0 = timed(lambda: ops.factor(t, c, a, U, V, d=d, W=W), reps=3, warm=1)
        d0 = d.clone()
        os.environ["C2_TIMEPAR"] = "1"; ms1 = timed(lambda: ops.factor(t, c, a, U, V, d=d, W=W), reps=3, warm=1)
        print("J %d B %5d N %8d: factor row by row %.3f ms, time-parallel %.3f ms (%.1fx)  max rel diff of d %.1e"
              % (J, B, N, ms0, ms1, ms0 / ms1, float(((d - d0).abs() / d0.abs()).max())), flush=True)
    # host drop-in, one series
    N = 100000
    t, c, a, U, V, y = [x[0].cpu().numpy() for x in synth.device_batch_fast(0, 1, N, J, dev)]
    dh, Wh = np.empty_like(a), np.empty_like(V)
    for tp in ("0", "1"):
        os.environ["C2_TIMEPAR"] = tp
        driver.factor(t, c, a, U, V, dh, Wh)
        t0 = time.perf_counter()
        for _ in range(5): driver.factor(t, c, a, U, V, dh, Wh)
        print("   driver.factor (host arrays, N = %d, J = %d) C2_TIMEPAR=%s: %.2f ms per call" % (N, J, tp, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
