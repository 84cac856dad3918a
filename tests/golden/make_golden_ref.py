# -*- coding: utf-8 -*-
"""Generate tests/golden/ref_golden.npz FROM THE REFERENCE'S OWN PYTHON -- build container only:

    python tests/golden/make_golden_ref.py

The reference's compiled pieces cannot be built here (Eigen submodule empty), but its pure-Python modules import once
`celerite2.driver` exists (oracle/ref_shim.py registers it; the reference files are read where they lie under
/root/reference, nothing is copied, nothing travels).  Three kinds of fixtures, all DATA (inputs + expected outputs):

(ii) REFERENCE-PURE.  Every expectation below is computed by reference code + numpy dense algebra, exactly as the
     reference's own tests do (python/test/test_driver.py:26-135, c++/test/test_factor.cpp:16-38); the CPU restatement
     is not involved in any expected value:
       K            = <reference Term>.to_dense(x, diag)            terms.py:58-79, 106-115
       coefficients = <reference Term>.get_coefficients()           terms.py:515-521, 554-569, 658-691, 729-745, 791-812
       d, Lunit, solve_*, matmul_*, dot_tril, loglik, K_star @ Y    numpy on K (Cholesky / triangular products)
     for the 8 kernels of c++/test/helpers.hpp:27-62 on helpers.hpp:14-24's data (cpp_*), testing.get_matrices()
     incl. conditional=True (py_*: `testing.py` itself runs here and supplies x, Y, K, t, K_star) and BASELINE
     configs[0] (cfg1_*).  The inputs (c, a, U, V) stored beside them come from the reference's
     `Term.get_celerite_matrices` (terms.py:117-177: the interleaved `c` is reference code, the (a, U, V) fill is the
     one compiled call it makes, served by the restatement of driver.cpp:456-474) -- and are VERIFIED here against the
     reference's K: K[n, m] = sum_j U[n, j] V[m, j] exp(-c_j (x_n - x_m)) for n > m, K[n, n] = a[n].
     The script asserts that all of this equals the round-1 fixtures (golden.npz, from oracle/dense.py): inputs and
     dense expectations to 1e-15 (bit-identical but for the last bit of some U, V entries), gradients to 1e-12.
(iii) REFERENCE CALLERS.  The reference's numpy `GaussianProcess` / `ConditionalDistribution`
     (numpy.py:66-121, core.py:9-150, 262-501) executed over the shim: compute, log_likelihood, apply_inverse,
     dot_tril, predict (mean at the data and at new times, return_var, return_cov, kernel=component,
     include_mean=False) -- pins the log-likelihood assembly, the interleaved layout and the conditional formulas AS
     EXECUTED BY REFERENCE CODE.  Each is also checked here against dense algebra on the reference's K.
(iv) `*_grad_*`: reverse-mode gradients from the restatement, after central finite differences of the DENSE
     log-likelihood (reference K) agree.

Only inputs and expected outputs are stored -- no reference source text.
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True   # never write __pycache__ into /root/reference
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cpu, ref_shim  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def dense_loglik(K, y):
    L = np.linalg.cholesky(K)
    alpha = np.linalg.solve(L, y)
    return -0.5 * (alpha @ alpha) - np.sum(np.log(np.diag(L))) - 0.5 * len(y) * np.log(2 * np.pi)


def check_matrices_reproduce_K(x, c, a, U, V, K):
    """(c, a, U, V) are a valid semiseparable representation of the REFERENCE's dense K."""
    dt = x[:, None] - x[None, :]
    Kl = np.einsum("nj,mj,nmj->nm", U, V, np.exp(-c[None, None, :] * np.abs(dt)[:, :, None]))
    err = max(np.max(np.abs(np.tril(Kl, -1) - np.tril(K, -1))), np.max(np.abs(a - np.diag(K))))
    assert err < 2e-14 * max(1.0, np.max(np.abs(K))), err
    assert np.max(np.abs(K - K.T)) == 0.0 or np.allclose(K, K.T, rtol=0, atol=1e-15)
    return err


def dense_expectations(prefix, term, x, diag, Y, out):
    c, a, U, V = term.get_celerite_matrices(x, diag)
    K = term.to_dense(x, diag)
    check_matrices_reproduce_K(x, c, a, U, V, K)
    L = np.linalg.cholesky(K)
    d = np.diag(L) ** 2
    out[prefix + "x"] = x; out[prefix + "diag"] = diag; out[prefix + "Y"] = Y
    out[prefix + "c"] = c; out[prefix + "a"] = a; out[prefix + "U"] = U; out[prefix + "V"] = V
    for name, v in zip(("ar", "cr", "ac", "bc", "cc", "dc"), term.get_coefficients()):
        out[prefix + "coef_" + name] = np.asarray(v, dtype=np.float64)
    out[prefix + "K"] = K
    out[prefix + "d"] = d
    out[prefix + "Lunit"] = L / np.diag(L)[None, :]
    out[prefix + "solve_lower"] = np.linalg.solve(L, Y)
    out[prefix + "solve_upper"] = np.linalg.solve(L.T, Y)
    out[prefix + "matmul_lower"] = np.tril(K, -1) @ Y
    out[prefix + "matmul_upper"] = np.triu(K, 1) @ Y
    out[prefix + "dot_tril"] = L @ Y
    out[prefix + "apply_inverse"] = np.linalg.solve(K, Y)
    out[prefix + "loglik"] = np.array(dense_loglik(K, Y[:, 0]))
    return c, a, U, V, K


def fd_check(K, y, grads):
    bt, bc, ba, bU, bV, by = grads
    for idx in (0, len(y) // 2, len(y) - 1):
        h = 1e-6
        Kp = K.copy(); Kp[idx, idx] += h
        Km = K.copy(); Km[idx, idx] -= h
        fd = (dense_loglik(Kp, y) - dense_loglik(Km, y)) / (2 * h)
        assert abs(fd - ba[idx]) < 1e-6 * (1 + abs(fd)), ("ba", idx, fd, ba[idx])
        yp = y.copy(); yp[idx] += h
        ym = y.copy(); ym[idx] -= h
        fd = (dense_loglik(K, yp) - dense_loglik(K, ym)) / (2 * h)
        assert abs(fd - by[idx]) < 1e-6 * (1 + abs(fd)), ("by", idx, fd, by[idx])


def gp_case(ref, prefix, kernel, x, diag, y, ts, mean, component, out):
    """The reference's GaussianProcess on one series; every stored value also checked against dense algebra."""
    GP = ref.numpy.GaussianProcess
    gp = GP(kernel, mean=mean)
    gp.compute(x, diag=diag)
    K = kernel.to_dense(x, diag)
    Ks = kernel.get_value(ts[:, None] - x[None, :])
    Kss = kernel.get_value(ts[:, None] - ts[None, :])
    r = y - mean
    o = {}
    o["x"], o["diag"], o["y"], o["ts"], o["mean"] = x, diag, y, ts, np.array(mean)
    # (numpy.py:66-76 factors in place: after compute() the reference's gp._a IS d -- take the inputs from a fresh call)
    o["c"], o["a"], o["U"], o["V"] = kernel.get_celerite_matrices(x, diag)
    assert np.array_equal(o["c"], gp._c) and np.array_equal(o["U"], gp._U) and np.array_equal(o["V"], gp._V)
    assert np.array_equal(gp._a, gp._d)
    o["d"], o["W"] = gp._d.copy(), gp._W.copy()
    o["loglik"] = np.array(gp.log_likelihood(y))
    o["apply_inverse"] = gp.apply_inverse(y)
    Y3 = np.ascontiguousarray(np.vstack([np.sin(x), np.cos(x), x ** 2]).T)
    o["Y3"] = Y3
    o["apply_inverse3"] = gp.apply_inverse(Y3)
    o["dot_tril"] = gp.dot_tril(y)
    o["dot_tril3"] = gp.dot_tril(Y3)
    o["mu_self"] = gp.predict(y)
    o["mu_star"] = gp.predict(y, ts)
    mu, var = gp.predict(y, ts, return_var=True)
    mu2, cov = gp.predict(y, ts, return_cov=True)
    assert np.array_equal(mu, mu2)
    o["var_star"], o["cov_star"] = var, cov
    mu0, var0 = gp.predict(y, return_var=True)
    _, cov0 = gp.predict(y, return_cov=True)
    o["var_self"], o["cov_self"] = var0, cov0
    o["mu_star_nomean"] = gp.predict(y, ts, include_mean=False)
    if component is not None:
        muk, vark = gp.predict(y, ts, return_var=True, kernel=component)
        _, covk = gp.predict(y, ts, return_cov=True, kernel=component, include_mean=False)
        o["mu_star_comp"], o["var_star_comp"], o["cov_star_comp"] = muk, vark, covk
        Kc = component.get_value(ts[:, None] - x[None, :])
        Kcc = component.get_value(ts[:, None] - ts[None, :])
        np.testing.assert_allclose(muk, Kc @ np.linalg.solve(K, r) + mean, rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(covk, Kcc - Kc @ np.linalg.solve(K, Kc.T), rtol=1e-8, atol=1e-10)
    # the reference callers against dense algebra on the reference's K
    want = dense_loglik(K, r)
    assert abs(o["loglik"] - want) <= 1e-11 * abs(want), (o["loglik"], want)
    np.testing.assert_allclose(o["apply_inverse"], np.linalg.solve(K, y), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(o["dot_tril"], np.linalg.cholesky(K) @ y, rtol=1e-10, atol=1e-11)
    a_r = np.linalg.solve(K, r)
    np.testing.assert_allclose(o["mu_star"], Ks @ a_r + mean, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(o["mu_self"], y - diag * a_r, rtol=1e-9, atol=1e-10)
    want_cov = Kss - Ks @ np.linalg.solve(K, Ks.T)
    np.testing.assert_allclose(cov, want_cov, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(var, np.diag(want_cov), rtol=1e-8, atol=1e-10)
    K0 = K - np.diag(diag)
    np.testing.assert_allclose(cov0, K0 - K0 @ np.linalg.solve(K, K0), rtol=1e-8, atol=1e-10)
    for k, v in o.items():
        out[prefix + k] = np.asarray(v, dtype=np.float64)


def main():
    ref = ref_shim.install()
    T = ref.terms
    out = {}
    # --- (ii) c++/test/helpers.hpp:14-62: data recipe + 8 kernels, built from the reference's Python term classes -----
    N, Nrhs = 50, 5
    delta = np.arange(N, dtype=np.float64) / (N - 1)
    x = 10 * delta + delta * delta
    diag = np.full(N, 0.5)
    Y = np.ascontiguousarray(np.sin(x[:, None] + np.arange(Nrhs, dtype=np.float64)[None, :] / Nrhs))
    real = lambda: T.RealTerm(a=1.0, c=0.1)
    cplx = lambda: T.ComplexTerm(a=0.8, b=0.03, c=1.0, d=0.1)
    sho1 = lambda: T.SHOTerm(S0=1.2, w0=0.3, Q=0.1)
    sho2 = lambda: T.SHOTerm(S0=0.1, w0=1.3, Q=5.3)
    kernels = {"real": real(), "complex": cplx(), "sho1": sho1(), "sho2": sho2(), "sum1": real() + cplx(),
               "sum2": real() + cplx() + sho1(), "sum3": real() + cplx() + sho1() + sho2(), "sum4": sho1() + sho2()}
    for name, term in kernels.items():
        dense_expectations("cpp_%s_" % name, term, x, diag, Y, out)
    # --- (ii) python/celerite2/testing.py:10-49, run as written upstream ---------------------------------------------
    x, c, a, U, V, K, Yp, t, U2, V2, K_star = ref.testing.get_matrices(conditional=True, include_dense=True)
    rng = np.random.default_rng(721); rng.uniform(0, 10, 100); dg = rng.uniform(0.1, 0.3, 100)   # testing.py:18-28
    term = T.SHOTerm(S0=5.0, w0=0.1, Q=3.45)                                                      # testing.py:30
    assert np.array_equal(term.to_dense(x, dg), K)
    c_, a_, U_, V_, K_ = dense_expectations("py_", term, x, dg, Yp, out)
    assert all(np.array_equal(p, q) for p, q in ((c, c_), (a, a_), (U, U_), (V, V_), (K, K_)))
    out["py_t"] = t; out["py_U2"] = U2; out["py_V2"] = V2; out["py_K_star"] = K_star
    out["py_general_matmul"] = K_star @ Yp
    # test_general_matmul_fallback (test_driver.py:117-135): no diagonal, both grids the data grid
    xf, cf, af, Uf, Vf, Kf, Yf = ref.testing.get_matrices(include_dense=True, no_diag=True)
    out["py_nodiag_a"] = af; out["py_nodiag_general_matmul"] = Kf @ Yf
    y = np.ascontiguousarray(Yp[:, 0])
    ll, grads, flag = cpu.loglik_grad(x, c, a, U, V, y)
    assert flag == 0 and abs(ll - out["py_loglik"]) < 1e-11 * abs(ll)
    fd_check(K, y, grads)
    for nme, g in zip(("bt", "bc", "ba", "bU", "bV", "by"), grads):
        out["py_grad_" + nme] = g
    # --- (ii) BASELINE configs[0]: N = 1000, J = 2, through the reference's GaussianProcess AND dense ------------------
    rng = np.random.default_rng(721)
    N = 1000
    t1 = np.sort(rng.uniform(0, N / 10.0, N))
    dg1 = rng.uniform(0.1, 0.3, N)
    y1 = np.sin(t1) + 0.1 * rng.standard_normal(N)
    term = T.SHOTerm(S0=5.0, w0=0.1, Q=3.45)
    K1 = term.to_dense(t1, dg1)
    out["cfg1_t"] = t1; out["cfg1_diag"] = dg1; out["cfg1_y"] = y1
    out["cfg1_loglik"] = np.array(dense_loglik(K1, y1))
    gp = ref.numpy.GaussianProcess(term)
    gp.compute(t1, diag=dg1)
    out["cfg1_loglik_ref_gp"] = np.array(gp.log_likelihood(y1))      # configs[0] as BASELINE.json words it
    assert abs(out["cfg1_loglik_ref_gp"] - out["cfg1_loglik"]) < 1e-11 * abs(out["cfg1_loglik"])
    # --- reference coefficients of every term class (both SHO regimes and the boundary, alt. parameterisations) --------
    coef_cases = {
        "real": T.RealTerm(a=1.3, c=0.4), "complex": T.ComplexTerm(a=0.8, b=0.03, c=1.0, d=0.1),
        "sho_under": T.SHOTerm(S0=5.0, w0=0.1, Q=3.45), "sho_over": T.SHOTerm(S0=1.2, w0=0.3, Q=0.1),
        "sho_near_half_lo": T.SHOTerm(S0=1.0, w0=1.0, Q=0.5 - 1e-9), "sho_near_half_hi": T.SHOTerm(S0=1.0, w0=1.0, Q=0.5 + 1e-9),
        "sho_sigma_rho_tau": T.SHOTerm(sigma=1.5, rho=3.0, tau=2.0), "sho_sigma_rho_Q": T.SHOTerm(sigma=0.7, rho=1.1, Q=0.3),
        "matern32": T.Matern32Term(sigma=0.5, rho=2.0), "matern32_eps": T.Matern32Term(sigma=1.5, rho=0.7, eps=1e-3),
        "rotation": T.RotationTerm(sigma=1.5, period=3.45, Q0=1.3, dQ=1.05, f=0.5),
        "sum": T.SHOTerm(S0=5.0, w0=0.1, Q=3.45) + T.RealTerm(a=1.0, c=0.1) + T.Matern32Term(sigma=0.5, rho=2.0),
    }
    tau = np.concatenate([[0.0], np.logspace(-3, 1.5, 40), -np.logspace(-2, 1, 10)])
    out["coef_tau"] = tau
    xg = np.sort(np.random.default_rng(40582).uniform(0, 10, 40))
    dgg = np.random.default_rng(40583).uniform(0.1, 0.3, 40)
    out["coef_x"], out["coef_diag"] = xg, dgg
    for name, term in coef_cases.items():
        for cn, v in zip(("ar", "cr", "ac", "bc", "cc", "dc"), term.get_coefficients()):
            out["coef_%s_%s" % (name, cn)] = np.asarray(v, dtype=np.float64)
        out["coef_%s_value" % name] = term.get_value(tau)
        cc, aa, UU, VV = term.get_celerite_matrices(xg, dgg)
        check_matrices_reproduce_K(xg, cc, aa, UU, VV, term.to_dense(xg, dgg))
        out["coef_%s_c" % name] = cc; out["coef_%s_a" % name] = aa
        out["coef_%s_U" % name] = UU; out["coef_%s_V" % name] = VV
    # --- (iii) the reference's GaussianProcess / ConditionalDistribution over the shim ---------------------------------
    rng = np.random.default_rng(40582)                       # python/test/test_celerite2.py:11-19 recipe
    Ng, M = 50, 100
    for b in range(3):
        xg = np.sort(rng.uniform(0, 10, Ng)); ts = np.sort(rng.uniform(-1, 12, M)); dgg = rng.uniform(0.1, 0.3, Ng)
        comp = T.SHOTerm(S0=5.0 - b, w0=0.1, Q=3.45)
        kernel = comp + T.RealTerm(a=1.0, c=0.1) + T.Matern32Term(sigma=0.5, rho=2.0)
        gp_case(ref, "gp%d_" % b, kernel, xg, dgg, np.sin(xg), ts, 0.3, comp, out)
    xg = np.sort(rng.uniform(0, 10, 120)); ts = np.sort(rng.uniform(-1, 12, 300)); dgg = rng.uniform(0.1, 0.3, 120)
    gp_case(ref, "gprot_", T.RotationTerm(sigma=1.5, period=3.45, Q0=1.3, dQ=1.05, f=0.5), xg, dgg, np.sin(xg), ts, 0.0, None, out)
    # --- width 8 and width 4 from the COEFFICIENTS (what c2_loglik_terms serves: SURVEY.md section 8f-1): the reference's
    # get_coefficients() next to its GaussianProcess log-likelihood -- four complex terms; two real + three complex terms;
    # the rotation term above (two complex terms)
    xg = np.sort(rng.uniform(0, 20, 200)); ts = np.sort(rng.uniform(-1, 22, 40)); dgg = rng.uniform(0.1, 0.3, 200)
    k8a = (T.SHOTerm(S0=1.0, w0=0.3, Q=2.0) + T.SHOTerm(S0=0.5, w0=1.1, Q=5.0) + T.SHOTerm(S0=0.3, w0=2.3, Q=1.3)
           + T.SHOTerm(S0=0.2, w0=3.1, Q=8.0))
    k8b = (T.RealTerm(a=0.7, c=0.15) + T.RealTerm(a=0.4, c=0.6) + T.SHOTerm(S0=1.0, w0=0.4, Q=3.0)
           + T.SHOTerm(S0=0.4, w0=1.7, Q=0.9) + T.SHOTerm(S0=0.25, w0=2.9, Q=6.0))
    for nm, kern in (("gp8a_", k8a), ("gp8b_", k8b)):
        gp_case(ref, nm, kern, xg, dgg, np.sin(xg) + 0.1 * rng.standard_normal(200), ts, 0.0, None, out)
        for cn, v in zip(("ar", "cr", "ac", "bc", "cc", "dc"), kern.get_coefficients()):
            out[nm + "coef_" + cn] = np.asarray(v, dtype=np.float64)
        assert out[nm + "coef_ar"].size + 2 * out[nm + "coef_ac"].size == 8
    for cn, v in zip(("ar", "cr", "ac", "bc", "cc", "dc"),
                     T.RotationTerm(sigma=1.5, period=3.45, Q0=1.3, dQ=1.05, f=0.5).get_coefficients()):
        out["gprot_coef_" + cn] = np.asarray(v, dtype=np.float64)
    # --- agreement with the round-1 fixtures (oracle/dense.py): same numbers, now produced by reference code -----------
    old = dict(np.load(os.path.join(HERE, "golden.npz")))
    worst = 0.0
    differs = {}
    for k, v in old.items():
        assert k in out, k
        e = np.max(np.abs(out[k] - v) / np.maximum(1.0, np.abs(v))) if v.size else 0.0
        worst = max(worst, float(e))
        differs.setdefault(k.split("_", 2)[-1] if k.startswith("cpp_") else k, 0.0)
        kk = k.split("_", 2)[-1] if k.startswith("cpp_") else k
        differs[kk] = max(differs[kk], float(e))
        # inputs and dense expectations: the same numbers to the last bit or two (U, V: the restatement's sincos vs numpy's);
        # the gradients are the restatement's on inputs that differ in the last bit (conditioning ~1e3)
        assert e <= (1e-12 if "_grad_" in k else 1e-15), (k, e)
    path = os.path.join(HERE, "ref_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d bytes, %d arrays; worst relative difference from golden.npz over its %d keys: %.2e"
          % (path, os.path.getsize(path), len(out), len(old), worst))
    print("   keys not bit-identical to golden.npz:", {k: "%.1e" % v for k, v in sorted(differs.items()) if v > 0})
    ref_shim.uninstall()


if __name__ == "__main__":
    main()
