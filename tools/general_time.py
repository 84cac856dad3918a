"""general_matmul_* with several right-hand sides: lanes-over-rhs merge kernel vs the first-round kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
B, N, J = 8192, 4096, 8
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
t1 = (t + 0.013).contiguous()
for nrhs in (3, 8, 16):
    Y = torch.randn((B, N, nrhs), dtype=torch.float64, device=dev)
    for name in ("general_matmul_lower", "general_matmul_upper"):
        res = {}
        for mode in ("0", "1"):
            os.environ["C2_GENERALK"] = mode
            f = getattr(ops, name)
            for _ in range(2): Z = f(t1, t, c, U, V, Y)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): Z = f(t1, t, c, U, V, Y)
            torch.cuda.synchronize(); res[mode] = ((time.perf_counter() - t0) / 3, Z)
        alg = B * 8.0 * (1 + J + nrhs) * 2 * N
        print("%s nrhs=%d: first-round %.2f ms, lanes-over-rhs %.2f ms (frac %.3f), max |diff| %.1e" % (name, nrhs, res["0"][0] * 1e3, res["1"][0] * 1e3, alg / res["1"][0] / 8e12, float((res["0"][1] - res["1"][1]).abs().max())), flush=True)
