# -*- coding: utf-8 -*-
"""Parity against fixtures produced BY THE REFERENCE'S OWN PYTHON (tests/golden/ref_golden.npz; generator
tests/golden/make_golden_ref.py, which imports /root/reference/python/celerite2/{terms,core,numpy,testing}.py in the
build container over oracle/ref_shim.py):

  * coef_*   the reference term classes' coefficients, k(tau) and celerite matrices      terms.py:58-79, 117-177, 515-812
  * cpp_* / py_* / cfg1_*   dense expectations on the reference's K = term.to_dense()    test_driver.py:26-135
  * gp*_     the reference's numpy GaussianProcess / ConditionalDistribution, executed    numpy.py:66-121, core.py:9-150

CPU part: the host-side mirror (celerite2_amd/terms.py coefficients) and the CPU restatement against them, the
agreement of the reference-generated file with the round-1 fixtures, and -- where /root/reference is mounted (build
container only) -- the reference's acceptance suites over the shim.  GPU part (`-m gpu`): the device path through the
same fixtures.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

COEF_CASES = {
    "real": lambda T: T.RealTerm(a=1.3, c=0.4),
    "complex": lambda T: T.ComplexTerm(a=0.8, b=0.03, c=1.0, d=0.1),
    "sho_under": lambda T: T.SHOTerm(S0=5.0, w0=0.1, Q=3.45),
    "sho_over": lambda T: T.SHOTerm(S0=1.2, w0=0.3, Q=0.1),
    "sho_near_half_lo": lambda T: T.SHOTerm(S0=1.0, w0=1.0, Q=0.5 - 1e-9),
    "sho_near_half_hi": lambda T: T.SHOTerm(S0=1.0, w0=1.0, Q=0.5 + 1e-9),
    "sho_sigma_rho_tau": lambda T: T.SHOTerm(sigma=1.5, rho=3.0, tau=2.0),
    "sho_sigma_rho_Q": lambda T: T.SHOTerm(sigma=0.7, rho=1.1, Q=0.3),
    "matern32": lambda T: T.Matern32Term(sigma=0.5, rho=2.0),
    "matern32_eps": lambda T: T.Matern32Term(sigma=1.5, rho=0.7, eps=1e-3),
    "rotation": lambda T: T.RotationTerm(sigma=1.5, period=3.45, Q0=1.3, dQ=1.05, f=0.5),
    "sum": lambda T: T.SHOTerm(S0=5.0, w0=0.1, Q=3.45) + T.RealTerm(a=1.0, c=0.1) + T.Matern32Term(sigma=0.5, rho=2.0),
}
COEF_NAMES = ("ar", "cr", "ac", "bc", "cc", "dc")


def _close(a, b, tol=1e-10, floor=1e-12):
    a = a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=tol, atol=floor * max(1.0, float(np.abs(b).max())))


# ---------------------------------------------------------------------------------------------------------------------
# CPU: host logic + restatement + the files themselves
# ---------------------------------------------------------------------------------------------------------------------
def test_reference_file_carries_the_round1_numbers(golden, golden_r1):
    """ref_golden.npz (reference code) is a superset of golden.npz (oracle/dense.py) with the same values: inputs and
    dense expectations to 1e-15, the restatement's gradients to 1e-12 (inputs differ in the last bit)."""
    for k, v in golden_r1.items():
        assert k in golden, k
        e = float(np.max(np.abs(golden[k] - v) / np.maximum(1.0, np.abs(v)))) if v.size else 0.0
        assert e <= (1e-12 if "_grad_" in k else 1e-15), (k, e)


@pytest.mark.parametrize("name", sorted(COEF_CASES))
def test_term_coefficients_vs_reference_classes(golden, name):
    """celerite2_amd/terms.py (host side) against the reference classes' get_coefficients() and get_value()."""
    from celerite2_amd import terms

    term = COEF_CASES[name](terms)
    got = term.get_coefficients()
    for cn, v in zip(COEF_NAMES, got):
        want = golden["coef_%s_%s" % (name, cn)]
        assert np.shape(v) == want.shape, (cn, np.shape(v), want.shape)
        np.testing.assert_allclose(v, want, rtol=1e-14, atol=0, err_msg=cn)
    np.testing.assert_allclose(term.get_value(golden["coef_tau"]), golden["coef_%s_value" % name], rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize("name", sorted(COEF_CASES))
def test_restatement_matrices_vs_reference_layout(oracle, golden, name):
    """c2o_get_celerite_matrices + the interleaved c against what the reference's Term.get_celerite_matrices returned
    (and the generator verified against the reference's dense K)."""
    co = [golden["coef_%s_%s" % (name, cn)] for cn in COEF_NAMES]
    ar, cr, ac, bc, cc, dc = co
    x, diag = golden["coef_x"], golden["coef_diag"]
    J = len(ar) + 2 * len(ac)
    a = np.empty(len(x)); U = np.empty((len(x), J)); V = np.empty((len(x), J))
    oracle.get_celerite_matrices(ar, ac, bc, dc, x, diag, a, U, V)
    assert np.array_equal(a, golden["coef_%s_a" % name]) and np.array_equal(U, golden["coef_%s_U" % name])
    assert np.array_equal(V, golden["coef_%s_V" % name])
    c = golden["coef_%s_c" % name]
    assert np.array_equal(c[:len(ar)], cr) and np.array_equal(c[len(ar)::2], cc) and np.array_equal(c[len(ar) + 1::2], cc)


@pytest.mark.parametrize("case", ["gp0_", "gp1_", "gp2_", "gprot_"])
def test_restatement_vs_reference_callers(oracle, golden, case):
    """The log-likelihood / solve / dot_tril assembly written out here (numpy.py:84-109) over the restatement reproduces
    what the reference's GaussianProcess returned, and d, W what its compute() stored."""
    g = {k[len(case):]: v for k, v in golden.items() if k.startswith(case)}
    x, c, a, U, V, y = (np.ascontiguousarray(g[k]) for k in ("x", "c", "a", "U", "V", "y"))
    d = np.empty_like(a); W = np.empty_like(V)
    oracle.factor(x, c, a, U, V, d, W)
    assert np.array_equal(d, g["d"]) and np.array_equal(W, g["W"])
    r = (y - float(g["mean"]))[:, None].copy()
    z = oracle.solve_lower(x, c, U, W, r, r.copy())[:, 0]
    ll = -0.5 * np.sum(z * z / d) - 0.5 * (np.sum(np.log(d)) + len(x) * np.log(2 * np.pi))
    assert abs(ll - float(g["loglik"])) <= 1e-13 * abs(ll)
    ll2, flag = oracle.loglik(x, c, a, U, V, np.ascontiguousarray(r[:, 0]))
    assert flag == 0 and abs(ll2 - float(g["loglik"])) <= 1e-12 * abs(ll2)


def test_cfg1_reference_gp_equals_dense(golden):
    """BASELINE configs[0] as worded (celerite2.GaussianProcess on the CPU path): the reference's own numpy
    GaussianProcess (over the restatement) and dense algebra on the reference's K agree to 1e-11."""
    assert abs(golden["cfg1_loglik_ref_gp"] - golden["cfg1_loglik"]) <= 1e-11 * abs(golden["cfg1_loglik"])


@pytest.mark.skipif(not os.path.isfile("/root/reference/python/celerite2/terms.py"),
                    reason="the reference is mounted in the build container only")
def test_reference_acceptance_suites_over_the_shim():
    """/root/reference/python/test/test_driver.py + test_backprop.py, unmodified, with celerite2.driver / .backprop = the
    CPU restatement (tools/ref_acceptance.py)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_acceptance.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "27 passed" in r.stdout, r.stdout[-500:]


@pytest.mark.skipif(not os.path.isfile("/root/reference/python/celerite2/terms.py"),
                    reason="the reference is mounted in the build container only")
def test_fixture_regenerates_from_the_reference(golden, tmp_path):
    """The committed file is what the generator produces from the reference today (same numpy): spot keys, bit for bit."""
    sys.dont_write_bytecode = True
    from oracle import ref_shim

    ref = ref_shim.install()
    try:
        x, c, a, U, V, K, Y = ref.testing.get_matrices(include_dense=True)
        assert np.array_equal(K, golden["py_K"]) and np.array_equal(U, golden["py_U"]) and np.array_equal(c, golden["py_c"])
        term = ref.terms.RotationTerm(sigma=1.5, period=3.45, Q0=1.3, dQ=1.05, f=0.5)
        for cn, v in zip(COEF_NAMES, term.get_coefficients()):
            assert np.array_equal(v, golden["coef_rotation_" + cn])
        gp = ref.numpy.GaussianProcess(ref.terms.SHOTerm(S0=5.0, w0=0.1, Q=3.45))
        gp.compute(golden["cfg1_t"], diag=golden["cfg1_diag"])
        assert gp.log_likelihood(golden["cfg1_y"]) == float(golden["cfg1_loglik_ref_gp"])
    finally:
        ref_shim.uninstall()


# ---------------------------------------------------------------------------------------------------------------------
# GPU: the device path through the reference's fixtures
# ---------------------------------------------------------------------------------------------------------------------
def _dev(*xs):
    import torch
    return [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in xs]


@pytest.mark.gpu
@pytest.mark.parametrize("tile", ["1", "0"])
@pytest.mark.parametrize("name", sorted(COEF_CASES))
def test_device_matrices_and_kernel_values_vs_reference_classes(golden, monkeypatch, name, tile):
    """Row M + (f)2 on the device: (c, a, U, V) of every reference term class and k(t1 - t2) on a grid -- by 32 x 64 tiles
    with angle addition (C2_KERNEL_VALUES_TILE, the default) and with a thread per entry (= 0); plus a 150 x 130 grid of the
    fixture's own times against the reference's get_value formula evaluated in numpy on the reference's coefficients."""
    from celerite2_amd import terms
    monkeypatch.setenv("C2_KERNEL_VALUES_TILE", tile)

    term = COEF_CASES[name](terms)
    x, diag = golden["coef_x"], golden["coef_diag"]
    xd, dd = _dev(x[None], diag[None])
    c, a, U, V = term.get_celerite_matrices(xd, dd)
    _close(c.reshape(-1), golden["coef_%s_c" % name], 1e-15, 0)
    _close(a[0], golden["coef_%s_a" % name], 1e-14)
    _close(U[0], golden["coef_%s_U" % name], 1e-12, 1e-14)
    _close(V[0], golden["coef_%s_V" % name], 1e-12, 1e-14)
    tau = golden["coef_tau"]
    t1, t2 = _dev(np.sort(tau)[None], np.zeros((1, 1)))
    order = np.argsort(tau)
    k = term.get_value_grid(t1, t2, B=1)
    _close(k[0, :, 0], golden["coef_%s_value" % name][order], 1e-12, 1e-14)
    rng = np.random.default_rng(5)
    g1, g2 = np.sort(rng.uniform(0, 40, 150)), rng.uniform(-3, 43, 130)     # (an unsorted second grid: nothing assumes order)
    kk = term.get_value_grid(*_dev(g1[None], g2[None]), B=1)
    ar, cr, ac, bc, cc, dc = (golden["coef_%s_%s" % (name, cn)] for cn in COEF_NAMES)
    tau = np.abs(g1[:, None] - g2[None, :])[..., None]                       # terms.py:58-79 on the reference's coefficients
    want = np.sum(ar * np.exp(-cr * tau), axis=-1) + np.sum(np.exp(-cc * tau) * (ac * np.cos(dc * tau) + bc * np.sin(dc * tau)), axis=-1)
    _close(kk[0], want, 1e-12, 1e-14)


@pytest.mark.gpu
def test_device_gp_vs_reference_gaussian_process(golden):
    """celerite2_amd.gp (batched, on the device) against what the REFERENCE's numpy GaussianProcess and
    ConditionalDistribution returned for the same three series (per-series S0) -- log-likelihood, apply_inverse (vector
    and matrix), dot_tril, conditional mean at the data / at new times / without the mean / of one component, predictive
    variance and covariance on both grids.  Tolerances: 1e-10 per element with a floor of 1e-11 of the largest entry for
    quantities that pass through K^-1 (cond K ~ 1e3: the reference's own value is that far from dense algebra)."""
    import torch
    from celerite2_amd import gp as gpmod, terms

    B = 3
    g = [{k[4:]: v for k, v in golden.items() if k.startswith("gp%d_" % b)} for b in range(B)]
    st = lambda k: np.stack([g[b][k] for b in range(B)])
    comp = terms.SHOTerm(S0=np.array([5.0, 4.0, 3.0]), w0=0.1, Q=3.45)
    kernel = comp + terms.RealTerm(a=1.0, c=0.1) + terms.Matern32Term(sigma=0.5, rho=2.0)
    xd, dd, yd, tsd, Y3 = _dev(st("x"), st("diag"), st("y"), st("ts"), st("Y3"))
    gp = gpmod.GaussianProcess(kernel, mean=0.3)
    gp.compute(xd, diag=dd)
    _close(gp._c, st("c"), 1e-15, 0); _close(gp._a, st("a"), 1e-14); _close(gp._U, st("U"), 1e-12, 1e-14)
    _close(gp._d, st("d")); _close(gp._W, st("W"), 1e-10, 1e-11)
    _close(gp.log_likelihood(yd), st("loglik")); _close(gp.log_likelihood_fused(yd), st("loglik"))
    _close(gp.apply_inverse(yd), st("apply_inverse"), 1e-10, 1e-11)
    _close(gp.apply_inverse(Y3), st("apply_inverse3"), 1e-10, 1e-11)
    _close(gp.dot_tril(yd), st("dot_tril")); _close(gp.dot_tril(Y3), st("dot_tril3"))
    _close(gp.predict(yd), st("mu_self"), 1e-10, 1e-11)
    _close(gp.predict(yd, tsd), st("mu_star"), 1e-10, 1e-11)
    _close(gp.predict(yd, tsd, include_mean=False), st("mu_star_nomean"), 1e-10, 1e-11)
    mu, var = gp.predict(yd, tsd, return_var=True)
    mu2, cov = gp.predict(yd, tsd, return_cov=True)
    assert torch.equal(mu, mu2)
    _close(var, st("var_star"), 1e-9, 1e-11); _close(cov, st("cov_star"), 1e-9, 1e-11)
    _, var0 = gp.predict(yd, return_var=True)
    _, cov0 = gp.predict(yd, return_cov=True)
    _close(var0, st("var_self"), 1e-9, 1e-11); _close(cov0, st("cov_self"), 1e-9, 1e-11)
    muk, vark = gp.predict(yd, tsd, return_var=True, kernel=comp)
    _, covk = gp.predict(yd, tsd, return_cov=True, kernel=comp, include_mean=False)
    _close(muk, st("mu_star_comp"), 1e-10, 1e-11)
    _close(vark, st("var_star_comp"), 1e-9, 1e-11); _close(covk, st("cov_star_comp"), 1e-9, 1e-11)


@pytest.mark.gpu
def test_device_gp_vs_reference_rotation_term(golden):
    """Same through RotationTerm (two SHO terms, J = 4), N = 120, M = 300 prediction points (300 right-hand sides)."""
    from celerite2_amd import gp as gpmod, terms

    g = {k[6:]: v for k, v in golden.items() if k.startswith("gprot_")}
    kernel = terms.RotationTerm(sigma=1.5, period=3.45, Q0=1.3, dQ=1.05, f=0.5)
    xd, dd, yd, tsd = _dev(g["x"][None], g["diag"][None], g["y"][None], g["ts"][None])
    gp = gpmod.GaussianProcess(kernel, mean=0.0)
    gp.compute(xd, diag=dd)
    _close(gp._d[0], g["d"]); _close(gp._W[0], g["W"], 1e-10, 1e-11)
    _close(gp.log_likelihood(yd)[0], g["loglik"])
    _close(gp.apply_inverse(yd)[0], g["apply_inverse"], 1e-10, 1e-11)
    _close(gp.dot_tril(yd)[0], g["dot_tril"])
    mu, cov = gp.predict(yd, tsd, return_cov=True)
    _, var = gp.predict(yd, tsd, return_var=True)
    _close(mu[0], g["mu_star"], 1e-10, 1e-11)
    _close(cov[0], g["cov_star"], 1e-9, 1e-11); _close(var[0], g["var_star"], 1e-9, 1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", ["composed", "one", "two", "four", "eight"])
@pytest.mark.parametrize("case", ["gp8a", "gp8b", "gprot"])
def test_device_loglik_terms_vs_reference_gaussian_process(golden, monkeypatch, case, lanes):
    """The coefficient-level entry point (c2_loglik_terms[_grad]) on the REFERENCE's coefficients
    (`get_coefficients()` of sums of its SHOTerm / RealTerm / RotationTerm: four complex terms; two real + three complex; two
    complex) against the log-likelihood the reference's numpy GaussianProcess returned for the same series -- through the composed
    chain and through every kernel that forms the rows in the lanes (one / two / four / eight lanes per series; width 4: one lane
    and the group of four), 70 copies of the series so that whole and partial groups of 64 are exercised.  The gradient call must
    return the same log-likelihood; its gradients are pinned against the oracle chain in tests/test_gpu_terms.py."""
    import torch
    from celerite2_amd import ops
    if case == "gprot" and lanes in ("two", "four"):
        pytest.skip("two / four lanes per series are width-8 mappings")
    monkeypatch.setenv("C2_TERMS_FUSED", "1" if lanes == "one" else "0")
    monkeypatch.setenv("C2_TERMS_TWO_LANES", "1" if lanes == "two" else "0")
    monkeypatch.setenv("C2_TERMS_FOUR_LANES", "1" if lanes == "four" else "0")
    monkeypatch.setenv("C2_TERMS_EIGHT_LANES", "1" if lanes == "eight" else "0")
    g = {k[len(case) + 1:]: v for k, v in golden.items() if k.startswith(case + "_")}
    B = 70
    rep = lambda v: np.ascontiguousarray(np.tile(np.atleast_1d(v)[None], (B,) + (1,) * np.atleast_1d(v).ndim))
    args = _dev(*[rep(g["coef_" + cn]) for cn in COEF_NAMES], rep(g["x"]), rep(g["diag"]), rep(g["y"] - g["mean"]))
    ll, flag = ops.loglik_terms(*args)
    ll2, grads, flag2 = ops.loglik_terms_grad(*args)
    assert int(flag.abs().sum()) == 0 and int(flag2.abs().sum()) == 0
    want = np.full(B, float(g["loglik"]))
    _close(ll, want); _close(ll2, want)
    assert all(bool(torch.isfinite(v).all()) for v in grads)


@pytest.mark.gpu
def test_device_general_matmul_vs_reference_k_star(golden):
    """test_driver.py:96-135 on the device: K_star @ Y with K_star = reference kernel.get_value(t - x), and the
    no-diagonal fallback (both grids the data grid, ties everywhere)."""
    from celerite2_amd import ops

    x, c, U, V, Y = (golden["py_" + k] for k in ("x", "c", "U", "V", "Y"))
    t, U2, V2 = golden["py_t"], golden["py_U2"], golden["py_V2"]
    xd, cd, Ud, Vd, Yd, td, U2d, V2d = _dev(x[None], c[None], U[None], V[None], Y[None], t[None], U2[None], V2[None])
    Z = ops.general_matmul_lower(td, xd, cd, U2d, Vd, Yd)
    Z = ops.general_matmul_upper(td, xd, cd, V2d, Ud, Yd, Z=Z)
    _close(Z[0], golden["py_general_matmul"])
    Z = ops.general_matmul_lower(xd, xd, cd, Ud, Vd, Yd)
    Z = ops.general_matmul_upper(xd, xd, cd, Vd, Ud, Yd, Z=Z)
    _close(Z[0], golden["py_nodiag_general_matmul"])
