# -*- coding: utf-8 -*-
"""Generate tests/golden/golden.npz -- run in the dev container:  python tests/golden/make_golden.py

The reference stores no golden vectors (every upstream test recomputes its
expectation from dense math), cannot be compiled (Eigen submodule empty) and
cannot be imported here, so the fixtures are produced by the DENSE oracle
(oracle/dense.py: closed-form kernel matrix -> numpy Cholesky / triangular
products, i.e. the oracle of c++/test/test_factor.cpp:16-38 and
python/test/test_driver.py:26-135) on the reference's deterministic inputs:

  cpp_<kernel>_*   c++/test/helpers.hpp:14-62   (N=50, Nrhs=5, 8 kernels)
  py_*             python/celerite2/testing.py:10-49 (default_rng(721), N=100, SHOTerm(5, 0.1, 3.45))
  cfg1_*           BASELINE.json configs[0]: N=1000, J=2 (one SHOTerm), log-likelihood

Gradients (`*_grad_*`) come from the CPU restatement (oracle/c2_oracle.cpp), after this
script has checked them against central finite differences of the dense log-likelihood.
Only inputs and expected outputs are stored -- no reference source text.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cpu, dense  # noqa: E402


def dense_expectations(prefix, x, diag, co, Y, out):
    c, a, U, V = dense.celerite_matrices(co, x, diag)
    K = dense.dense_matrix(co, x, diag)
    L = np.linalg.cholesky(K)
    d = np.diag(L) ** 2
    out[prefix + "x"] = x; out[prefix + "diag"] = diag; out[prefix + "Y"] = Y
    out[prefix + "c"] = c; out[prefix + "a"] = a; out[prefix + "U"] = U; out[prefix + "V"] = V
    out[prefix + "d"] = d
    out[prefix + "Lunit"] = L / np.diag(L)[None, :]              # I + tril(U W^T)
    out[prefix + "solve_lower"] = np.linalg.solve(L, Y)         # = L^-1 Y  (celerite: solve_lower / sqrt(d))
    out[prefix + "solve_upper"] = np.linalg.solve(L.T, Y)       # = L^-T Y  (celerite: solve_upper(Y / sqrt(d)))
    out[prefix + "matmul_lower"] = np.tril(K, -1) @ Y
    out[prefix + "matmul_upper"] = np.triu(K, 1) @ Y
    out[prefix + "dot_tril"] = L @ Y
    y = Y[:, 0]
    out[prefix + "loglik"] = np.array(dense.dense_loglik(K, y))
    return c, a, U, V, K


def fd_check(x, diag, co, y, grads):
    """Central finite differences of the DENSE log-likelihood w.r.t. a and y (inputs the dense model exposes)."""
    bt, bc, ba, bU, bV, by = grads
    K = dense.dense_matrix(co, x, diag)
    for idx in (0, len(x) // 2, len(x) - 1):
        h = 1e-6
        Kp = K.copy(); Kp[idx, idx] += h
        Km = K.copy(); Km[idx, idx] -= h
        fd = (dense.dense_loglik(Kp, y) - dense.dense_loglik(Km, y)) / (2 * h)
        assert abs(fd - ba[idx]) < 1e-6 * (1 + abs(fd)), ("ba", idx, fd, ba[idx])
        yp = y.copy(); yp[idx] += h
        ym = y.copy(); ym[idx] -= h
        fd = (dense.dense_loglik(K, yp) - dense.dense_loglik(K, ym)) / (2 * h)
        assert abs(fd - by[idx]) < 1e-6 * (1 + abs(fd)), ("by", idx, fd, by[idx])


def main():
    out = {}
    # --- C++ test recipe, 8 kernels ------------------------------------------------------
    x, diag, Y = dense.cpp_test_data(50, 5)
    for name, co in dense.cpp_test_kernels().items():
        dense_expectations("cpp_%s_" % name, x, diag, co, Y, out)
    # --- python test recipe ---------------------------------------------------------------
    m = dense.get_matrices(include_dense=True, conditional=True)
    c, a, U, V, K = dense_expectations("py_", m["x"], m["diag"], m["kernel"], m["Y"], out)
    out["py_t"] = m["t"]; out["py_U2"] = m["U2"]; out["py_V2"] = m["V2"]
    out["py_general_matmul"] = m["K_star"] @ m["Y"]
    y = np.ascontiguousarray(m["Y"][:, 0])
    ll, grads, flag = cpu.loglik_grad(m["x"], c, a, U, V, y)
    assert flag == 0 and abs(ll - out["py_loglik"]) < 1e-10 * abs(ll)
    fd_check(m["x"], m["diag"], m["kernel"], y, grads)
    for nme, g in zip(("bt", "bc", "ba", "bU", "bV", "by"), grads):
        out["py_grad_" + nme] = g
    # --- BASELINE config 1: N=1000, J=2 ------------------------------------------------------
    rng = np.random.default_rng(721)
    N = 1000
    t = np.sort(rng.uniform(0, N / 10.0, N))
    dg = rng.uniform(0.1, 0.3, N)
    yy = np.sin(t) + 0.1 * rng.standard_normal(N)
    co = dense.sho_term(5.0, 0.1, 3.45)
    c1, a1, U1, V1 = dense.celerite_matrices(co, t, dg)
    K1 = dense.dense_matrix(co, t, dg)
    out["cfg1_t"] = t; out["cfg1_diag"] = dg; out["cfg1_y"] = yy
    out["cfg1_loglik"] = np.array(dense.dense_loglik(K1, yy))
    ll1, flag1 = cpu.loglik(t, c1, a1, U1, V1, yy)
    assert flag1 == 0 and abs(ll1 - out["cfg1_loglik"]) < 1e-10 * abs(ll1), (ll1, out["cfg1_loglik"])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
