# -*- coding: utf-8 -*-
"""oracle/dense.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Dense-linear-algebra oracle and deterministic input recipes.  This is exactly
the ground truth the reference's own tests use (there are no stored golden
vectors upstream): build the dense kernel matrix from the closed-form kernel,
then compare against numpy Cholesky / triangular products.

Reference citations (paths relative to /root/reference):
  * kernel value k(tau)             python/celerite2/terms.py:58-79, c++/test/test_to_dense.cpp:27-38
  * celerite matrices (a,U,V,c)     python/celerite2/driver.cpp:456-474, python/celerite2/terms.py:171-173
  * SHOTerm coefficients            python/celerite2/terms.py:658-691
  * RealTerm / ComplexTerm          python/celerite2/terms.py:515-521, 554-569
  * Matern32Term                    python/celerite2/terms.py:729-745
  * python test inputs              python/celerite2/testing.py:10-49
  * C++ test inputs and kernels     c++/test/helpers.hpp:14-62
"""
import numpy as np

__all__ = [
    "Coeffs", "real_term", "complex_term", "sho_term", "matern32_term", "term_sum",
    "kernel_value", "dense_matrix", "celerite_matrices", "get_matrices", "cpp_test_data",
    "cpp_test_kernels", "dense_loglik", "synthetic_batch", "sho_sum_coeffs",
]


class Coeffs:
    """(ar, cr, ac, bc, cc, dc) coefficient lists of a celerite term."""

    def __init__(self, ar=(), cr=(), ac=(), bc=(), cc=(), dc=()):
        f = lambda x: np.atleast_1d(np.asarray(x, dtype=np.float64)).reshape(-1)
        self.ar, self.cr, self.ac, self.bc, self.cc, self.dc = map(f, (ar, cr, ac, bc, cc, dc))

    @property
    def J(self):
        return len(self.ar) + 2 * len(self.ac)

    def __add__(self, other):  # TermSum concatenates coefficient lists (terms.py:233-235)
        return Coeffs(*[np.concatenate([x, y]) for x, y in zip(self.tuple(), other.tuple())])

    def tuple(self):
        return (self.ar, self.cr, self.ac, self.bc, self.cc, self.dc)


def real_term(a, c):
    return Coeffs(ar=[a], cr=[c])


def complex_term(a, b, c, d):
    return Coeffs(ac=[a], bc=[b], cc=[c], dc=[d])


def sho_term(S0, w0, Q, eps=1e-5):
    """terms.py:658-691 (overdamped -> 2 real terms, underdamped -> 1 complex term)."""
    if Q < 0.5:
        f = np.sqrt(max(1.0 - 4.0 * Q**2, eps))
        return Coeffs(
            ar=0.5 * S0 * w0 * Q * np.array([1.0 + 1.0 / f, 1.0 - 1.0 / f]),
            cr=0.5 * w0 / Q * np.array([1.0 - f, 1.0 + f]),
        )
    f = np.sqrt(max(4.0 * Q**2 - 1.0, eps))
    a = S0 * w0 * Q
    c = 0.5 * w0 / Q
    return Coeffs(ac=[a], bc=[a / f], cc=[c], dc=[c * f])


def matern32_term(sigma, rho, eps=0.01):
    w0 = np.sqrt(3.0) / rho
    S0 = sigma**2 / w0
    return Coeffs(ac=[w0 * S0], bc=[w0 * w0 * S0 / eps], cc=[w0], dc=[eps])


def term_sum(*terms):
    out = terms[0]
    for t in terms[1:]:
        out = out + t
    return out


def kernel_value(co, tau):
    """k(tau) = sum_r ar e^{-cr|tau|} + sum_c e^{-cc|tau|}(ac cos(dc|tau|) + bc sin(dc|tau|))."""
    tau = np.abs(np.asarray(tau, dtype=np.float64))[..., None]
    k = np.sum(co.ar * np.exp(-co.cr * tau), axis=-1)
    k = k + np.sum(np.exp(-co.cc * tau) * (co.ac * np.cos(co.dc * tau) + co.bc * np.sin(co.dc * tau)), axis=-1)
    return k


def dense_matrix(co, x, diag):
    K = kernel_value(co, x[:, None] - x[None, :])
    K[np.diag_indices_from(K)] += diag
    return K


def celerite_matrices(co, x, diag):
    """numpy restatement of driver.cpp:456-474 + the c layout of terms.py:171-173."""
    x = np.asarray(x, dtype=np.float64)
    N, Jr, Jc = len(x), len(co.ar), len(co.ac)
    J = Jr + 2 * Jc
    c = np.empty(J)
    c[:Jr] = co.cr
    c[Jr::2] = co.cc
    c[Jr + 1::2] = co.cc
    a = diag + np.sum(co.ar) + np.sum(co.ac)
    U = np.empty((N, J))
    V = np.empty((N, J))
    U[:, :Jr] = co.ar
    V[:, :Jr] = 1.0
    arg = co.dc[None, :] * x[:, None]
    cs, sn = np.cos(arg), np.sin(arg)
    V[:, Jr::2] = cs
    V[:, Jr + 1::2] = sn
    U[:, Jr::2] = co.ac * cs + co.bc * sn
    U[:, Jr + 1::2] = co.ac * sn - co.bc * cs
    return c, np.ascontiguousarray(a), np.ascontiguousarray(U), np.ascontiguousarray(V)


def get_matrices(size=100, kernel=None, vector=False, conditional=False, include_dense=False, no_diag=False):
    """The reference's python test-input recipe (python/celerite2/testing.py:10-49)."""
    random = np.random.default_rng(721)
    x = np.sort(random.uniform(0, 10, size))
    if vector:
        Y = np.sin(x)
    else:
        Y = np.ascontiguousarray(np.vstack([np.sin(x), np.cos(x), x**2]).T, dtype=np.float64)
    if no_diag:
        diag = np.zeros_like(x)
    else:
        diag = random.uniform(0.1, 0.3, len(x))
    kernel = kernel if kernel is not None else sho_term(S0=5.0, w0=0.1, Q=3.45)
    c, a, U, V = celerite_matrices(kernel, x, diag)
    out = dict(x=x, c=c, a=a, U=U, V=V, Y=Y, diag=diag, kernel=kernel)
    if include_dense:
        out["K"] = dense_matrix(kernel, x, diag)
    if conditional:
        t = np.sort(random.uniform(-1, 12, 200))
        _, _, U2, V2 = celerite_matrices(kernel, t, np.zeros_like(t))
        out.update(t=t, U2=U2, V2=V2)
        if include_dense:
            out["K_star"] = kernel_value(kernel, t[:, None] - x[None, :])
    return out


def cpp_test_data(N=50, Nrhs=5):
    """c++/test/helpers.hpp:14-24."""
    delta = np.arange(N, dtype=np.float64) / (N - 1)
    x = 10 * delta + delta * delta
    diag = np.full(N, 0.5)
    Y = np.sin(x[:, None] + np.arange(Nrhs, dtype=np.float64)[None, :] / Nrhs)
    return x, diag, np.ascontiguousarray(Y)


def cpp_test_kernels():
    """c++/test/helpers.hpp:27-62 (true sums; not the operator+ slip at terms.hpp:160-162)."""
    real = real_term(1.0, 0.1)
    cplx = complex_term(0.8, 0.03, 1.0, 0.1)
    sho1 = sho_term(1.2, 0.3, 0.1)
    sho2 = sho_term(0.1, 1.3, 5.3)
    return {
        "real": real, "complex": cplx, "sho1": sho1, "sho2": sho2,
        "sum1": real + cplx, "sum2": real + cplx + sho1, "sum3": real + cplx + sho1 + sho2, "sum4": sho1 + sho2,
    }


def dense_loglik(K, y):
    """-0.5 (y^T K^-1 y + log det K + N log 2 pi)."""
    L = np.linalg.cholesky(K)
    alpha = np.linalg.solve(L, y)
    return -0.5 * (alpha @ alpha) - np.sum(np.log(np.diag(L))) - 0.5 * len(y) * np.log(2 * np.pi)


def sho_sum_coeffs(J, xi=0.0):
    """Sum of J/2 underdamped SHO terms: S0=5*0.7^k, w0=0.1*3^k*(1+0.05 xi), Q=3.45+k.
    k=0, xi=0 is exactly the reference test kernel SHOTerm(S0=5, w0=0.1, Q=3.45) (testing.py:30)."""
    assert J % 2 == 0
    terms = [sho_term(5.0 * 0.7**k, 0.1 * 3.0**k * (1.0 + 0.05 * xi), 3.45 + k) for k in range(J // 2)]
    return term_sum(*terms)


def synthetic_batch(B, N, J, seed0=721):
    """Synthetic batch of independent GPs: per-series default_rng(seed0+b),
    t = sort(U(0, N/10)), diag ~ U(0.1, 0.3), y = sin t + 0.1 eps, kernel = sho_sum_coeffs(J, xi_b)."""
    t = np.empty((B, N)); c = np.empty((B, J)); a = np.empty((B, N))
    U = np.empty((B, N, J)); V = np.empty((B, N, J)); y = np.empty((B, N))
    for b in range(B):
        rng = np.random.default_rng(seed0 + b)
        t[b] = np.sort(rng.uniform(0, N / 10.0, N))
        diag = rng.uniform(0.1, 0.3, N)
        xi = rng.uniform(-1, 1)
        y[b] = np.sin(t[b]) + 0.1 * rng.standard_normal(N)
        c[b], a[b], U[b], V[b] = celerite_matrices(sho_sum_coeffs(J, xi), t[b], diag)
    return t, c, a, U, V, y


# ---- 2-D (multi-band) extension, rank-1 band covariance -------------------------------
# No reference code exists for this row (no core2.hpp): the dense Kronecker matrix below IS the definition the
# device path is checked against; kron_interleaved gives the 1-D view of the same model for the CPU oracle.
def kron_dense(co, x, alpha, diag):
    """K = T (x) alpha alpha^T + diag, T_nn' = k(|x_n - x_n'|); rows ordered epoch-major (n*M + m); diag (N, M)."""
    alpha = np.asarray(alpha, dtype=np.float64)
    T = kernel_value(co, x[:, None] - x[None, :])
    K = np.kron(T, np.outer(alpha, alpha))
    K[np.diag_indices_from(K)] += np.asarray(diag, dtype=np.float64).ravel()
    return K


def kron_interleaved(c, a, U, V, x, alpha, diag):
    """(t', c, a', U', V') of the length N*M series: U' = U (x) alpha, V' = V (x) alpha, a' = diag + alpha^2 k(0),
    dt = 0 between the bands of one epoch.  (c, a, U, V) are the epoch grid's celerite matrices with ZERO diag."""
    alpha = np.asarray(alpha, dtype=np.float64)
    N, J = U.shape
    M = len(alpha)
    t2 = np.repeat(x, M)
    a2 = (np.asarray(diag) + alpha[None, :] ** 2 * a[:, None]).ravel()
    U2 = (U[:, None, :] * alpha[None, :, None]).reshape(N * M, J)
    V2 = (V[:, None, :] * alpha[None, :, None]).reshape(N * M, J)
    return t2, c, np.ascontiguousarray(a2), np.ascontiguousarray(U2), np.ascontiguousarray(V2)


def kron_fold_gradients(grads2, a, U, V, alpha):
    """Gradients of the interleaved series (bt', bc, ba', bU', bV', by') folded back onto the 2-D model's inputs:
    returns (bt, bc, ba, bU, bV, balpha, bdiag, by)."""
    bt2, bc, ba2, bU2, bV2, by2 = grads2
    alpha = np.asarray(alpha, dtype=np.float64)
    N, J = U.shape
    M = len(alpha)
    ba2 = ba2.reshape(N, M); bU2 = bU2.reshape(N, M, J); bV2 = bV2.reshape(N, M, J)
    bt = bt2.reshape(N, M).sum(1)
    ba = (ba2 * alpha[None, :] ** 2).sum(1)
    bU = (bU2 * alpha[None, :, None]).sum(1)
    bV = (bV2 * alpha[None, :, None]).sum(1)
    balpha = (2.0 * alpha[None, :] * a[:, None] * ba2).sum(0) + (bU2 * U[:, None, :] + bV2 * V[:, None, :]).sum((0, 2))
    return bt, bc, ba, bU, bV, balpha, ba2.copy(), by2.reshape(N, M).copy()


def kron_synthetic(B, N, M, J, seed0=1721):
    """Synthetic multi-band batch: the section-8d kernel / epoch grid per series, alpha ~ U(0.5, 1.5)^M,
    diag ~ U(0.1, 0.3)^(N x M), y_nm = alpha_m sin(t_n) + sqrt(diag) eps."""
    t = np.empty((B, N)); c = np.empty((B, J)); a = np.empty((B, N)); U = np.empty((B, N, J)); V = np.empty((B, N, J))
    alpha = np.empty((B, M)); diag = np.empty((B, N, M)); y = np.empty((B, N, M)); cos = []
    for b in range(B):
        rng = np.random.default_rng(seed0 + b)
        t[b] = np.sort(rng.uniform(0, N / 10.0, N))
        xi = rng.uniform(-1, 1)
        alpha[b] = rng.uniform(0.5, 1.5, M)
        diag[b] = rng.uniform(0.1, 0.3, (N, M))
        y[b] = alpha[b][None, :] * np.sin(t[b])[:, None] + np.sqrt(diag[b]) * rng.standard_normal((N, M))
        co = sho_sum_coeffs(J - J % 2, xi) if J >= 2 else None   # odd widths: one real term in front
        if J % 2:
            co = real_term(1.3, 0.4 * (1.0 + 0.05 * xi)) if co is None else real_term(1.3, 0.4 * (1.0 + 0.05 * xi)) + co
        c[b], a[b], U[b], V[b] = celerite_matrices(co, t[b], np.zeros(N))
        cos.append(co)
    return t, c, a, U, V, alpha, diag, y, cos


def coefficient_chain(cpu, ar, cr, ac, bc, cc, dc, x, diag, y):
    """Log-likelihood and its nine gradients w.r.t. the celerite coefficients and (x, diag, y) for ONE series: the CPU
    restatement's gradients w.r.t. (t, c, a, U, V, y) pushed through the reverse of the matrix recipe
    (python/celerite2/driver.cpp:456-474) in numpy.  `cpu`: the oracle.cpu module.  Returns ll, (bar, bcr, bac, bbc, bcc,
    bdc, bx, bdiag, by), flag."""
    co = Coeffs(ar=ar, cr=cr, ac=ac, bc=bc, cc=cc, dc=dc)
    c, a, U, V = celerite_matrices(co, x, diag)
    ll, (bt, bcv, ba, bU, bV, by), flag = cpu.loglik_grad(x, c, a, U, V, y)
    Jr = len(ar)
    bar = ba.sum() + bU[:, :Jr].sum(0)
    bcr = bcv[:Jr]
    arg = dc[None, :] * x[:, None]
    co_, s_ = np.cos(arg), np.sin(arg)
    U0, U1 = U[:, Jr::2], U[:, Jr + 1::2]
    bU0, bU1, bV0, bV1 = bU[:, Jr::2], bU[:, Jr + 1::2], bV[:, Jr::2], bV[:, Jr + 1::2]
    bac = ba.sum() + (bU0 * co_ + bU1 * s_).sum(0)
    bbc = (bU0 * s_ - bU1 * co_).sum(0)
    bcc = bcv[Jr::2] + bcv[Jr + 1::2]
    g = -bU0 * U1 + bU1 * U0 - bV0 * s_ + bV1 * co_
    bdc = (g * x[:, None]).sum(0)
    bx = bt + (g * dc[None, :]).sum(1)
    return ll, (bar, bcr, bac, bbc, bcc, bdc, bx, ba.copy(), by), flag

