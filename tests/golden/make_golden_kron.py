# -*- coding: utf-8 -*-
"""Generate tests/golden/kron.npz -- run in the dev container:  python tests/golden/make_golden_kron.py

Fixtures of the 2-D (multi-band) extension, rank-1 band covariance (BASELINE configs[4]).  The reference has NO
2-D code, so nothing of the reference is involved: the expectations come from the dense Kronecker matrix
K = T (x) alpha alpha^T + diag (oracle/dense.py: kron_dense) -- Cholesky log-likelihood and central finite
differences of it w.r.t. alpha, diag, y and the diagonal of T -- cross-checked here against the CPU oracle's 1-D
recursion on the interleaved series.  Inputs and expected outputs only.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cpu, dense  # noqa: E402

CASES = [(16, 3, 2), (40, 4, 4), (64, 2, 6), (7, 1, 2)]
B = 2


def dense_ll(co, t, a_diag, alpha, diag, y):
    N, M = diag.shape
    T = dense.kernel_value(co, t[:, None] - t[None, :])
    T[np.diag_indices(N)] = a_diag
    K = np.kron(T, np.outer(alpha, alpha))
    K[np.diag_indices(N * M)] += diag.ravel()
    return dense.dense_loglik(K, y.ravel())


def central(f, x, h=1e-6):
    g = np.empty_like(x)
    it = np.nditer(x, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        xp = x.copy(); xp[i] += h
        xm = x.copy(); xm[i] -= h
        g[i] = (f(xp) - f(xm)) / (2 * h)
    return g


def main():
    out = {}
    for N, M, J in CASES:
        t, c, a, U, V, alpha, diag, y, cos = dense.kron_synthetic(B, N, M, J, seed0=4242)
        key = "N%d_M%d_J%d_" % (N, M, J)
        for nm, x in zip(("t", "c", "a", "U", "V", "alpha", "diag", "y"), (t, c, a, U, V, alpha, diag, y)):
            out[key + nm] = x
        ll = np.empty(B); fa = np.empty((B, N)); fal = np.empty((B, M)); fd = np.empty((B, N, M)); fy = np.empty((B, N, M))
        for b in range(B):
            co = cos[b]
            K = dense.kron_dense(co, t[b], alpha[b], diag[b])
            ll[b] = dense.dense_loglik(K, y[b].ravel())
            assert abs(ll[b] - dense_ll(co, t[b], a[b], alpha[b], diag[b], y[b])) < 1e-9 * abs(ll[b])
            # the SURVEY construction: 1-D recursion on the interleaved series
            t2, c2, a2, U2, V2 = dense.kron_interleaved(c[b], a[b], U[b], V[b], t[b], alpha[b], diag[b])
            ll1, flag = cpu.loglik(t2, c2, a2, U2, V2, np.ascontiguousarray(y[b].ravel()))
            assert flag == 0 and abs(ll1 - ll[b]) < 1e-10 * abs(ll[b]), (ll1, ll[b])
            fa[b] = central(lambda v: dense_ll(co, t[b], v, alpha[b], diag[b], y[b]), a[b])
            fal[b] = central(lambda v: dense_ll(co, t[b], a[b], v, diag[b], y[b]), alpha[b])
            fd[b] = central(lambda v: dense_ll(co, t[b], a[b], alpha[b], v, y[b]), diag[b])
            fy[b] = central(lambda v: dense_ll(co, t[b], a[b], alpha[b], diag[b], v), y[b])
        out[key + "loglik_dense"] = ll
        out[key + "fd_ba"] = fa; out[key + "fd_balpha"] = fal; out[key + "fd_bdiag"] = fd; out[key + "fd_by"] = fy
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kron.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
