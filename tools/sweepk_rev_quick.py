"""Multi-rhs reverse sweeps, lanes over the right-hand sides: time at B = 8192, N = 4096, J = 8 (A/B builds through C2_LIB_PATH)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
B, N, J = int(os.environ.get("QB", 8192)), 4096, 8
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
d, W, flag = ops.factor(t, c, a, U, V)
for nrhs in [int(v) for v in sys.argv[1:]] or [8]:
    Y = torch.randn((B, N, nrhs), dtype=torch.float64, device=dev)
    bZ = torch.randn((B, N, nrhs), dtype=torch.float64, device=dev)
    for name, sec in (("solve_lower", W), ("matmul_upper", V)):
        kw = dict(workspace=True) if name.startswith("solve") else dict(workspace=True, zero_z=True)
        Z, F = getattr(ops, name)(t, c, U, sec, Y, **kw)
        f = getattr(ops, name + "_rev")
        for _ in range(2): out = f(t, c, U, sec, Y, Z, F, bZ)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): out = f(t, c, U, sec, Y, Z, F, bZ)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        alg = B * N * 8.0 * (1 + 2 * J + 3 * nrhs + J * nrhs + 2 * J + nrhs + 1)
        chk = sum(float(x.double().abs().sum()) for x in out)
        print("%s_rev nrhs=%d B=%d: %.2f ms (frac %.3f) checksum %.12e" % (name, nrhs, B, ms, alg / ms / 8e9, chk), flush=True)
        del Z, F, out
