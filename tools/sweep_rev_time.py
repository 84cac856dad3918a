"""Multi-rhs reverse sweeps: lanes-over-rhs kernel vs the first-round kernel (B=8192, N=4096, J=8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
B, N, J = 8192, 4096, 8
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
d, W, flag = ops.factor(t, c, a, U, V)
for nrhs in (8, 3, 16):
    Y = torch.randn((B, N, nrhs), dtype=torch.float64, device=dev)
    bZ = torch.randn((B, N, nrhs), dtype=torch.float64, device=dev)
    for name, sec in (("solve_lower", W), ("solve_upper", W), ("matmul_lower", V), ("matmul_upper", V)):
        kw = dict(workspace=True) if name.startswith("solve") else dict(workspace=True, zero_z=True)
        Z, F = getattr(ops, name)(t, c, U, sec, Y, **kw)
        res = {}
        for mode in ("0", "1"):
            os.environ["C2_SWEEPK_REV"] = mode
            f = getattr(ops, name + "_rev")
            for _ in range(2): out = f(t, c, U, sec, Y, Z, F, bZ)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): out = f(t, c, U, sec, Y, Z, F, bZ)
            torch.cuda.synchronize(); res[mode] = ((time.perf_counter() - t0) / 3, out)
        alg = B * N * 8.0 * (1 + 2 * J + 3 * nrhs + J * nrhs + 2 * J + nrhs + 1)
        err = max(float((x - y_).abs().max() / (1e-300 + y_.abs().max())) for x, y_ in zip(res["1"][1], res["0"][1]))
        print("%s_rev nrhs=%d: first-round %.2f ms, lanes-over-rhs %.2f ms (frac %.3f), max rel diff %.1e" % (name, nrhs, res["0"][0] * 1e3, res["1"][0] * 1e3, alg / res["1"][0] / 8e12, err), flush=True)
        del Z, F, out, res
