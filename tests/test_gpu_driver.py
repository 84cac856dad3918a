# -*- coding: utf-8 -*-
"""GPU parity tests of the drop-in modules `celerite2_amd.driver` / `celerite2_amd.backprop`.

They read like the reference's python/test/test_driver.py and test_backprop.py (same inputs from the
testing.py:10-49 recipe, same dense expectations, same in-place assertions), but call the HIP backend
through the host C-ABI (c2h_*) and additionally compare against the CPU oracle at 1e-10."""
import numpy as np
import pytest

from oracle import dense

pytestmark = pytest.mark.gpu

RTOL = 1e-10  # north_star: <= 1e-10 relative vs the CPU reference path


@pytest.fixture(scope="module")
def mods():
    from celerite2_amd import backprop, driver
    return driver, backprop


def _mat(vector=False, **kw):
    m = dense.get_matrices(vector=vector, include_dense=True, **kw)
    Y = m["Y"][:, None].copy() if vector else m["Y"]
    return m["x"], m["c"], m["a"], m["U"], m["V"], m["K"], np.ascontiguousarray(Y), m


def _close(a, b, tol=RTOL):
    np.testing.assert_allclose(a, b, rtol=tol, atol=tol * max(1.0, np.abs(b).max()))


def test_factor(mods, oracle):
    driver, _ = mods
    x, c, a, U, V, K, Y, _ = _mat()
    d0 = np.empty_like(a); W0 = np.empty_like(V)
    oracle.factor(x, c, a, U, V, d0, W0)
    d, W = driver.factor(x, c, a, U, V, a, V)       # in place
    assert d is a and W is V                          # no copy made (test_driver.py:13-15)
    _close(d, d0); _close(W, W0)
    _close(d, np.diag(np.linalg.cholesky(K)) ** 2, 1e-9)


def test_factor_not_positive_definite(mods):
    driver, backprop = mods
    x, c, a, U, V, K, Y, _ = _mat()
    a = a.copy(); a[11] = -1.0
    with pytest.raises(driver.LinAlgError, match="failed to factorize or solve matrix"):
        driver.factor(x, c, a, U, V, np.empty_like(a), np.empty_like(V))
    with pytest.raises(backprop.LinAlgError):
        backprop.factor_fwd(x, c, a, U, V, np.empty_like(a), np.empty_like(V), np.empty((len(x), 2, 2)))


@pytest.mark.parametrize("vector", [True, False])
def test_solve_lower_upper(mods, vector):
    driver, _ = mods
    x, c, a, U, V, K, Y, _ = _mat(vector)
    L = np.linalg.cholesky(K)
    d, W = driver.factor(x, c, a, U, V, a, V)
    Y1 = Y.copy()
    value = driver.solve_lower(x, c, U, W, Y1, Y1)
    assert value is Y1
    np.testing.assert_allclose(value / np.sqrt(d)[:, None], np.linalg.solve(L, Y), rtol=1e-7)
    Y2 = np.ascontiguousarray(Y / np.sqrt(d)[:, None])
    value = driver.solve_upper(x, c, U, W, Y2, Y2)
    assert value is Y2
    np.testing.assert_allclose(value, np.linalg.solve(L.T, Y), rtol=1e-7)


@pytest.mark.parametrize("vector", [True, False])
def test_matmul_lower_upper(mods, oracle, vector):
    driver, _ = mods
    x, c, a, U, V, K, Y, _ = _mat(vector)
    value = driver.matmul_lower(x, c, U, V, Y, np.zeros_like(Y))
    np.testing.assert_allclose(value, np.tril(K, -1) @ Y, rtol=1e-7)
    _close(value, oracle.matmul_lower(x, c, U, V, Y, np.zeros_like(Y)))
    value = driver.matmul_upper(x, c, U, V, Y, np.zeros_like(Y))
    np.testing.assert_allclose(value, np.triu(K, 1) @ Y, rtol=1e-7)
    # accumulates into the caller's Z
    Z = np.ones_like(Y)
    driver.matmul_upper(x, c, U, V, Y, Z)
    _close(Z, 1.0 + np.triu(K, 1) @ Y, 1e-9)


@pytest.mark.parametrize("vector", [True, False])
def test_general_matmul(mods, vector):
    driver, _ = mods
    m = dense.get_matrices(conditional=True, include_dense=True, vector=vector)
    Y = m["Y"][:, None].copy() if vector else m["Y"]
    Z = np.zeros((len(m["t"]), Y.shape[1]))
    Z = driver.general_matmul_lower(m["t"], m["x"], m["c"], m["U2"], m["V"], Y, Z)
    Z = driver.general_matmul_upper(m["t"], m["x"], m["c"], m["V2"], m["U"], Y, Z)
    np.testing.assert_allclose(Z, m["K_star"] @ Y, rtol=1e-7)
    # fallback (test_driver.py:117-135)
    m = dense.get_matrices(include_dense=True, vector=vector, no_diag=True)
    Y = m["Y"][:, None].copy() if vector else m["Y"]
    Z = np.zeros_like(Y)
    Z = driver.general_matmul_lower(m["x"], m["x"], m["c"], m["U"], m["V"], Y, Z)
    Z = driver.general_matmul_upper(m["x"], m["x"], m["c"], m["V"], m["U"], Y, Z)
    np.testing.assert_allclose(Z, m["K"] @ Y, rtol=1e-7)


def test_get_celerite_matrices(mods, golden):
    driver, _ = mods
    co = dense.cpp_test_kernels()["sum3"]
    x, diag = golden["cpp_sum3_x"], golden["cpp_sum3_diag"]
    J = co.J
    a = np.empty(len(x)); U = np.empty((len(x), J)); V = np.empty((len(x), J))
    a2, U2, V2 = driver.get_celerite_matrices(co.ar, co.ac, co.bc, co.dc, x, diag, a, U, V)
    assert a2 is a and U2 is U and V2 is V
    np.testing.assert_allclose(a, golden["cpp_sum3_a"], rtol=1e-14)
    np.testing.assert_allclose(U, golden["cpp_sum3_U"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(V, golden["cpp_sum3_V"], rtol=1e-12, atol=1e-14)


def test_non_contiguous_argument_is_copied(mods):
    """pybind11 forcecast: a non C-contiguous argument is copied; the result lands in the copy."""
    driver, _ = mods
    x, c, a, U, V, K, Y, _ = _mat()
    Yf = np.asfortranarray(Y)
    Z = driver.matmul_lower(x, c, U, V, Yf, np.zeros_like(Y))
    np.testing.assert_allclose(Z, np.tril(K, -1) @ Y, rtol=1e-7)


# ----------------------------------------------------------------------------- backprop
def test_factor_fwd_rev(mods, oracle):
    driver, backprop = mods
    x, c, a, U, V, K, Y, _ = _mat()
    N, J = U.shape
    d = np.empty_like(a); W = np.empty_like(V); S = np.empty((N, J, J))
    d0, W0 = driver.factor(x, c, a, U, V, np.copy(a), np.copy(V))
    d, W, S = backprop.factor_fwd(x, c, a, U, V, d, W, S)
    np.testing.assert_allclose(d, d0); np.testing.assert_allclose(W, W0)
    So = np.empty_like(S)
    oracle.factor(x, c, a, U, V, np.empty_like(a), np.empty_like(V), So)
    _close(S, So)
    rng = np.random.default_rng(5)
    bd = rng.standard_normal(N); bW = rng.standard_normal((N, J))
    outs = [np.empty(N), np.empty(J), np.empty(N), np.empty((N, J)), np.empty((N, J))]
    res = backprop.factor_rev(x, c, a, U, V, d, W, S, bd, bW, *outs)
    ref = oracle.factor_rev(x, c, a, U, V, d, W, S, bd, bW, *[np.empty_like(o) for o in outs])
    for r, e, o in zip(res, ref, outs):
        assert r is o
        _close(r, e)


@pytest.mark.parametrize("vector", [True, False])
@pytest.mark.parametrize("op", ["solve_lower", "solve_upper", "matmul_lower", "matmul_upper"])
def test_sweep_fwd_rev(mods, oracle, op, vector):
    driver, backprop = mods
    x, c, a, U, V, K, Y, _ = _mat(vector)
    N, J = U.shape
    nrhs = Y.shape[1]
    if op.startswith("solve"):
        d, W = driver.factor(x, c, a, U, V, a, V)
        V = W
        Z0 = getattr(driver, op)(x, c, U, V, Y, np.copy(Y))
    else:
        Z0 = getattr(driver, op)(x, c, U, V, Y, np.zeros_like(Y))
    Z = np.full_like(Y, 7.0); F = np.empty((N, J, nrhs))   # *_fwd zeroes Z first (backprop.cpp:505)
    Z, F = getattr(backprop, op + "_fwd")(x, c, U, V, Y, Z, F)
    np.testing.assert_allclose(Z0, Z)
    Zo = np.empty_like(Y); Fo = np.empty_like(F)
    getattr(oracle, op + "_fwd")(x, c, U, V, Y, Zo, Fo)
    _close(Z, Zo); _close(F, Fo)
    rng = np.random.default_rng(11)
    bZ = rng.standard_normal(Y.shape)
    outs = [np.empty(N), np.empty(J), np.empty((N, J)), np.empty((N, J)), np.empty_like(Y)]
    res = getattr(backprop, op + "_rev")(x, c, U, V, Y, Z, F, bZ, *outs)
    ref = getattr(oracle, op + "_rev")(x, c, U, V, Y, Zo, Fo, bZ, *[np.empty_like(o) for o in outs])
    for r, e in zip(res, ref):
        _close(r, e)


def test_general_matmul_fwd(mods, oracle):
    _, backprop = mods
    m = dense.get_matrices(conditional=True)
    Y = m["Y"]
    for name, U_, V_ in (("general_matmul_lower", m["U2"], m["V"]), ("general_matmul_upper", m["V2"], m["U"])):
        Z = np.full((len(m["t"]), 3), 3.0); F = np.full((len(m["x"]), len(m["c"]), 3), -1.0)
        Zo = Z.copy(); Fo = F.copy()
        getattr(backprop, name + "_fwd")(m["t"], m["x"], m["c"], U_, V_, Y, Z, F)
        getattr(oracle, name + "_fwd")(m["t"], m["x"], m["c"], U_, V_, Y, Zo, Fo)
        _close(Z, Zo); _close(F, Fo)


def test_wide_model_through_the_drop_in(mods, oracle):
    """A model wider than the tuned kernels (J = 40 > 32; the reference's dynamic path takes any width, driver.hpp:98-99)
    through the pybind11 drop-ins: factor, the solves, the product and the reverse-mode chain against the dense matrix and
    the oracle (csrc/c2_wide.hip underneath)."""
    driver, backprop = mods
    rng = np.random.default_rng(77)
    N, J = 60, 40
    x = np.sort(rng.uniform(0, 6, N))
    diag = rng.uniform(0.1, 0.3, N)
    co = dense.term_sum(*[dense.sho_term(1.0 / (k + 1), 0.05 * 100.0 ** rng.uniform(), rng.uniform(1.0, 8.0)) for k in range(J // 2)])
    c, a, U, V = dense.celerite_matrices(co, x, diag)
    K = dense.dense_matrix(co, x, diag)
    Y = np.ascontiguousarray(np.stack([np.sin(x), np.cos(x), x * x], axis=1))
    d, W = driver.factor(x, c, a, U, V, np.empty_like(a), np.empty_like(V))
    L = np.linalg.cholesky(K)
    _close(d, np.diag(L) ** 2, 1e-9)
    Z = driver.solve_lower(x, c, U, W, Y, np.empty_like(Y))
    _close(Z / np.sqrt(d)[:, None], np.linalg.solve(L, Y), 1e-8)
    Z2 = driver.matmul_lower(x, c, U, V, Y, np.zeros_like(Y))
    _close(Z2, np.tril(K, -1) @ Y, 1e-9)
    for op, ref in (("matmul_lower", np.tril(K - np.diag(diag), -1) @ Y), ("matmul_upper", np.triu(K - np.diag(diag), 1) @ Y)):
        # *_fwd zeroes Z itself (backprop.cpp: Z.setZero()): a NaN-filled output buffer must come back clean, first row included
        Zf, Ff = getattr(backprop, op + "_fwd")(x, c, U, V, Y, np.full_like(Y, np.nan), np.full((N, J, 3), np.nan))
        _close(Zf, ref, 1e-9)
        Zo = np.empty_like(Y); Fo = np.empty((N, J, 3))
        getattr(oracle, op + "_fwd")(x, c, U, V, Y, Zo, Fo)
        _close(Zf, Zo); _close(Ff, Fo)
    S = np.empty((N, J, J)); F = np.empty((N, J, 3))
    d2, W2, S2 = backprop.factor_fwd(x, c, a, U, V, np.empty_like(a), np.empty_like(V), S)
    do = np.empty(N); Wo = np.empty((N, J)); So = np.empty((N, J, J))
    oracle.factor(x, c, a, U, V, do, Wo, So)
    _close(d2, do); _close(W2, Wo); _close(S2, So)
    bd = rng.standard_normal(N); bW = rng.standard_normal((N, J))
    outs = [np.empty(N), np.empty(J), np.empty(N), np.empty((N, J)), np.empty((N, J))]
    got = backprop.factor_rev(x, c, a, U, V, d2, W2, S2, bd, bW, *[np.empty_like(o) for o in outs])
    oracle.factor_rev(x, c, a, U, V, do, Wo, So, bd, bW, *outs)
    for g, e in zip(got, outs):
        _close(g, e)


def test_two_threads_are_reentrant(mods, oracle):
    """The host entry points keep their staging arena, bounce buffer and stream per calling THREAD (c2_host.hip): two threads
    hammering driver.factor / driver.solve_lower / backprop.factor_rev on different problems at once (the bindings release
    the GIL) get the answers they get alone -- as with the reference, whose functions are re-entrant."""
    import threading
    driver, backprop = mods
    problems = []
    for seed, (N, J) in enumerate([(300, 2), (1200, 8)]):
        co = dense.sho_sum_coeffs(J)
        rng = np.random.default_rng(50 + seed)
        t = np.sort(rng.uniform(0, N / 10.0, N)); diag = rng.uniform(0.1, 0.3, N)
        c, a, U, V = dense.celerite_matrices(co, t, diag)
        Y = rng.standard_normal((N, 3))
        d0, W0, S0 = np.empty_like(a), np.empty_like(V), np.empty((N, J, J))
        oracle.factor(t, c, a, U, V, d0, W0, S0)
        Z0 = Y.copy(); oracle.solve_lower(t, c, U, W0, Y, Z0)
        bd, bW = rng.standard_normal(N), rng.standard_normal((N, J))
        g0 = [np.zeros(N), np.zeros(J), np.zeros(N), np.zeros((N, J)), np.zeros((N, J))]
        oracle.factor_rev(t, c, a, U, V, d0, W0, S0, bd, bW, *g0)
        problems.append(dict(t=t, c=c, a=a, U=U, V=V, Y=Y, d0=d0, W0=W0, S0=S0, Z0=Z0, bd=bd, bW=bW, g0=g0))
    errors = []

    def work(p):
        try:
            for _ in range(40):
                d, W = driver.factor(p["t"], p["c"], p["a"], p["U"], p["V"], np.empty_like(p["a"]), np.empty_like(p["V"]))
                _close(d, p["d0"]); _close(W, p["W0"])
                Z = driver.solve_lower(p["t"], p["c"], p["U"], W, p["Y"], p["Y"].copy())
                _close(Z, p["Z0"])
                g = [np.zeros_like(x) for x in p["g0"]]
                backprop.factor_rev(p["t"], p["c"], p["a"], p["U"], p["V"], d, W, p["S0"], p["bd"], p["bW"], *g)
                for x, e in zip(g, p["g0"]):
                    _close(x, e)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(p,)) for p in problems for _ in range(2)]
    for th in threads: th.start()
    for th in threads: th.join()
    assert not errors, errors[:3]
