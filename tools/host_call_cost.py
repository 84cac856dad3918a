"""Per-call cost of the host-pointer drop-in modules (celerite2_amd.driver / backprop over c2h_*): one PCIe round trip,
one hipMalloc / hipMemcpy / hipFree per argument and a device sync per call (c2_host.hip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from celerite2_amd import driver
from oracle import dense
for (N, J) in [(1000, 2), (4096, 8), (100000, 8)]:
    co = dense.sho_sum_coeffs(J)
    rng = np.random.default_rng(1)
    t = np.sort(rng.uniform(0, N / 10.0, N)); diag = rng.uniform(0.1, 0.3, N)
    c, a, U, V = dense.celerite_matrices(co, t, diag)
    Y = rng.standard_normal((N, 1))
    d, W = np.empty_like(a), np.empty_like(V)
    Z = np.empty_like(Y)
    for _ in range(3): driver.factor(t, c, a, U, V, d, W); driver.solve_lower(t, c, U, W, Y, Z)
    n = 20
    t0 = time.perf_counter()
    for _ in range(n): driver.factor(t, c, a, U, V, d, W)
    tf = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n): driver.solve_lower(t, c, U, W, Y, Z)
    ts = (time.perf_counter() - t0) / n
    print("N=%d J=%d: driver.factor %.0f us per call (%.1f KB each way), driver.solve_lower %.0f us" % (N, J, tf * 1e6, 8 * N * (3 + 3 * J) / 1e3, ts * 1e6), flush=True)


# the reverse-mode chain of the drop-in (celerite2_amd.backprop, same signatures as the reference's backprop module) on
# ONE series -- factor_fwd, solve_lower_fwd, solve_lower_rev, factor_rev -- beside the CPU restatement (oracle/, one thread)
from celerite2_amd import backprop
from oracle import cpu as ocpu
ocpu.build()
for (N, J) in [(1000, 2), (4096, 8), (20000, 4), (100000, 8)]:
    co = dense.sho_sum_coeffs(J)
    rng = np.random.default_rng(2)
    t = np.sort(rng.uniform(0, N / 10.0, N)); diag = rng.uniform(0.1, 0.3, N)
    c, a, U, V = dense.celerite_matrices(co, t, diag)
    Y = rng.standard_normal((N, 1)); bZ = rng.standard_normal((N, 1))
    d, W, S = np.empty_like(a), np.empty_like(V), np.empty((N, J, J))
    Z, F = np.empty_like(Y), np.empty((N, J, 1))
    bt, bc, ba, bU, bV, bY = np.zeros(N), np.zeros(J), np.zeros(N), np.zeros((N, J)), np.zeros((N, J)), np.zeros((N, 1))
    bd, bW = rng.standard_normal(N), rng.standard_normal((N, J))

    def chain(m):
        m.factor_fwd(t, c, a, U, V, d, W, S)
        m.solve_lower_fwd(t, c, U, W, Y, Z, F)
        m.solve_lower_rev(t, c, U, W, Y, Z, F, bZ, bt, bc, bU, bV, bY)
        m.factor_rev(t, c, a, U, V, d, W, S, bd, bW, bt, bc, ba, bU, bV)

    res = {}
    for name, m in (("drop-in", backprop), ("cpu", ocpu)):
        for _ in range(2): chain(m)
        n = 10
        t0 = time.perf_counter()
        for _ in range(n): chain(m)
        res[name] = (time.perf_counter() - t0) / n
    print("N=%d J=%d: factor_fwd + solve_lower_fwd + solve_lower_rev + factor_rev: drop-in (host arrays, copies included) %.2f ms, "
          "CPU restatement (1 thread) %.2f ms" % (N, J, res["drop-in"] * 1e3, res["cpu"] * 1e3), flush=True)
