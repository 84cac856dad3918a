"""Print, per gradient, the largest error of the two device methods of the 2-D row against the CPU oracle's 1-D
recursion on the interleaved series: (max |diff| / max |ref|) and the worst element-relative error above a floor."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celerite2_amd import ops
from oracle import cpu, dense
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_kron import oracle_interleaved, totals, dev

NAMES7 = ("bt", "bc", "bU+baV", "bV+baU", "balpha", "bdiag", "by")
for (B, N, M, J) in [(5, 300, 16, 6), (3, 1000, 5, 8), (2, 50000, 16, 6)]:
    t, c, a, U, V, alpha, diag, y, _ = dense.kron_synthetic(B, N, M, J)
    llo, go = oracle_interleaved(cpu, t, c, a, U, V, alpha, diag, y)
    args = dev(t, c, a, U, V, alpha, diag, y)
    for method in ("collapsed", "interleaved"):
        ll, g, flag = ops.kron_loglik_grad(*args, method=method)
        print(B, N, M, J, method, "ll rel", float(np.abs(ll.cpu().numpy() - llo).max() / np.abs(llo).max()))
        for nm, x, e in zip(NAMES7, totals(g, args[3], args[4]), totals(go, U, V)):
            x = x.cpu().numpy(); dlt = np.abs(x - e); mx = np.abs(e).max()
            big = np.abs(e) > 1e-3 * mx
            print("   %-8s max|d|/max|ref| %.2e   worst elementwise (|ref| > 1e-3 max) %.2e" % (nm, dlt.max() / mx, (dlt[big] / np.abs(e[big])).max()))
