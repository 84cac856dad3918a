import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
N, J, nrhs = 10_000_000, 16, 32
t, c, a, U, V, y = synth.device_batch_fast(0, 1, N, J, dev)
d, W, flag = ops.factor(t, c, a, U, V)
Yd = torch.randn((1, N, nrhs), dtype=torch.float64, device=dev)
for _ in range(6): ops.dot_tril(t, c, U, W, d, Yd)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
times = []
for _ in range(10):
    e0.record(); ops.dot_tril(t, c, U, W, d, Yd); e1.record(); torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1))
times.sort()
nbytes = N * 8 * (2 + 2 * J + 2 * nrhs)
print("dot_tril N=%d J=%d nrhs=%d: median %.3f ms  min %.3f ms  (%.2f TB/s of %.2f GB algorithmic, frac %.3f of 8 TB/s)"
      % (N, J, nrhs, times[5], times[0], nbytes / times[5] / 1e9, nbytes / 1e9, nbytes / times[5] / 1e9 / 8.0))
