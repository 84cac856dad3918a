# -*- coding: utf-8 -*-
"""Distance of the time-parallel gradient (c2_timepar_grad.hip) from the CPU oracle as a function of the conditioning
kappa = max a_n / d_n: the synthetic series of the bench with their white noise scaled down (kappa from ~1e2 to ~1e5).
The row-by-row kernels share the oracle's operation order and agree with it to ~1e-13 whatever kappa; any other order of
the same arithmetic moves the result by ~eps * kappa^2 -- as does the oracle itself against a long-double evaluation.

    python tools/kappa_sweep.py [N] [J]
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from celerite2_amd import _lib, ops  # noqa: E402
from oracle import cpu as orc  # noqa: E402
from oracle import dense  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    J = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    B = 6
    orc.build()
    lib = _lib.load()
    sink = torch.zeros(64, dtype=torch.float64, device="cuda")
    lib.c2_internal_set_debug_sink.argtypes = [ctypes.c_void_p]
    lib.c2_internal_set_debug_sink.restype = None
    lib.c2_internal_set_debug_sink(ctypes.c_void_p(sink.data_ptr()))
    os.environ["C2_VERIFY_FALLBACK"] = "0"
    for scale in (10.0, 1.0, 0.3, 0.1, 0.03, 0.01, 3e-3, 1e-3, 1e-4):
        t = np.empty((B, N)); c = np.empty((B, J)); a = np.empty((B, N))
        U = np.empty((B, N, J)); V = np.empty((B, N, J)); y = np.empty((B, N))
        for b in range(B):
            rng = np.random.default_rng(4242 + b)
            t[b] = np.sort(rng.uniform(0, N / 10.0, N))
            diag = scale * rng.uniform(0.1, 0.3, N)
            xi = rng.uniform(-1, 1)
            y[b] = np.sin(t[b]) + 0.1 * rng.standard_normal(N)
            c[b], a[b], U[b], V[b] = dense.celerite_matrices(dense.sho_sum_coeffs(J, xi), t[b], diag)
        llo, go, flo = orc.loglik_grad_batched(t, c, a, U, V, y, nthreads=4)
        args = [torch.from_numpy(x).cuda() for x in (t, c, a, U, V, y)]
        res = {}
        for name, tp in (("rows", "0"), ("timepar", "1")):
            os.environ["C2_TIMEPAR_GRAD"] = tp; os.environ["C2_FACTOR_ITER"] = tp
            sink.zero_()
            ll, grads, flag = ops.loglik_grad(*args)
            torch.cuda.synchronize()
            e = 0.0
            for g, w in zip(grads, go):
                gn = g.cpu().numpy()
                for b in range(B):
                    if flo[b] == 0:
                        e = max(e, float(np.abs(gn[b] - w[b]).max() / np.abs(w[b]).max()))
            res[name] = (e, sink.cpu().numpy().copy())
        w = res["timepar"][1]
        print("diag x %.0e  kappa %.2e  failed %d | row-by-row vs oracle %.2e | time-parallel vs oracle %.2e = %.2f eps kappa^2 | gate %.2e es %.1e eb %.1e ef %.1e ez %.1e newton %s"
              % (scale, w[3], int((flo != 0).sum()), res["rows"][0], res["timepar"][0], res["timepar"][0] / (1.1e-16 * w[3] ** 2),
                 w[0], w[1], w[2], w[4], w[5], " ".join("%.1e" % x for x in w[9:14])), flush=True)


if __name__ == "__main__":
    main()
