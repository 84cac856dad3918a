# -*- coding: utf-8 -*-
"""2-D (multi-band) extension, rank-1 band covariance (BASELINE configs[4]).

PARITY UNPINNED BY THE REFERENCE: the reference has no 2-D code, so what pins this row is (1) the dense Kronecker
matrix K = T (x) alpha alpha^T + diag at tiny sizes (Cholesky log-likelihood, central finite differences), (2) the
CPU oracle's 1-D recursions on the interleaved N*M series (the construction of section 8a-2D) at moderate sizes, and
(3) agreement of the two device methods with each other at the full BASELINE shape."""
import numpy as np
import pytest

from oracle import dense

pytestmark = pytest.mark.gpu
NAMES = ("bt", "bc", "ba", "bU", "bV", "balpha", "bdiag", "by")


@pytest.fixture(scope="module")
def ops():
    import torch
    from celerite2_amd import ops as o
    assert torch.cuda.is_available()
    return o


def dev(*xs):
    import torch
    return [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in xs]


def close(a, b, tol=1e-10, floor=1e-12):
    a = a.cpu().numpy() if hasattr(a, "cpu") else a
    np.testing.assert_allclose(a, b, rtol=tol, atol=floor * max(1.0, float(np.abs(b).max())))


# The collapsed method is a different (mathematically equivalent) formulation from the oracle's interleaved recursion:
# the two agree to the conditioning of the problem, k(0) / (effective per-epoch noise 1/A) ~ 1e3..1e4 here, i.e.
# 1e-12..1e-10 of the largest element (measured in round 2; the interleaved method, which shares the oracle's
# arithmetic, agrees to 1e-12).  Log-likelihoods agree to 1e-13.
COLLAPSED_FLOOR = 5e-10


def oracle_interleaved(oracle, t, c, a, U, V, alpha, diag, y):
    """Per series: 1-D CPU oracle on the interleaved series, gradients folded back (oracle/dense.py)."""
    lls, folded = [], []
    for b in range(len(y)):
        t2, c2, a2, U2, V2 = dense.kron_interleaved(c[b], a[b], U[b], V[b], t[b], alpha[b], diag[b])
        ll, g2, flag = oracle.loglik_grad(t2, c2, a2, U2, V2, np.ascontiguousarray(y[b].ravel()))
        assert flag == 0
        lls.append(ll)
        folded.append(dense.kron_fold_gradients(g2, a[b], U[b], V[b], alpha[b]))
    return np.array(lls), [np.stack([f[i] for f in folded]) for i in range(8)]


def totals(g, U, V):
    """(ba, bU, bV) -> the parametrisation-independent total derivatives along a = U.V."""
    bt, bc, ba, bU, bV, bal, bdiag, by = g
    return bt, bc, bU + ba[..., None] * V, bV + ba[..., None] * U, bal, bdiag, by


@pytest.mark.parametrize("N,M,J", [(16, 3, 2), (40, 4, 4), (64, 2, 6), (7, 1, 2)])
def test_kron_vs_dense_kronecker(ops, golden_kron, N, M, J):
    """Both device methods against the dense Kronecker matrix (committed fixture, tests/golden/kron.npz)."""
    key = "N%d_M%d_J%d_" % (N, M, J)
    g = {k[len(key):]: v for k, v in golden_kron.items() if k.startswith(key)}
    args = dev(g["t"], g["c"], g["a"], g["U"], g["V"], g["alpha"], g["diag"], g["y"])
    for method in ("collapsed", "interleaved"):
        ll, flag = ops.kron_loglik(*args, method=method)
        assert int(flag.abs().sum()) == 0
        close(ll, g["loglik_dense"])
        ll2, grads, flag2 = ops.kron_loglik_grad(*args, method=method)
        close(ll2, g["loglik_dense"])
        got = dict(zip(NAMES, [x.cpu().numpy() for x in grads]))
        # finite differences of the DENSE log-likelihood w.r.t. alpha, diag, y (and T_nn = a for the collapsed
        # parametrisation, which is the literal Kronecker definition): tolerance of the differences themselves
        np.testing.assert_allclose(got["balpha"], g["fd_balpha"], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(got["bdiag"], g["fd_bdiag"], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(got["by"], g["fd_by"], rtol=2e-6, atol=2e-6)
        if method == "collapsed":
            np.testing.assert_allclose(got["ba"], g["fd_ba"], rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("banded", ["1", "0"])
@pytest.mark.parametrize("B,N,M,J", [(5, 300, 16, 6), (3, 1000, 5, 8), (9, 33, 7, 3), (2, 1, 4, 2), (70, 50, 2, 4), (2, 1500, 8, 4)])
def test_kron_vs_interleaved_oracle(ops, oracle, monkeypatch, B, N, M, J, banded):
    monkeypatch.setenv("C2_KRON_BANDED", banded)   # "1": lanes over the bands where M is a power of two, "0": thread per epoch
    """Value and all eight gradients against the CPU oracle on the interleaved series; shared and per-series alpha."""
    t, c, a, U, V, alpha, diag, y, _ = dense.kron_synthetic(B, N, M, J)   # a = k(0) = U_n . V_n (celerite matrices)
    llo, go = oracle_interleaved(oracle, t, c, a, U, V, alpha, diag, y)
    args = dev(t, c, a, U, V, alpha, diag, y)
    ll_i, g_i, flag = ops.kron_loglik_grad(*args, method="interleaved")
    assert int(flag.abs().sum()) == 0
    close(ll_i, llo)
    for nm, g, e in zip(NAMES, g_i, go):   # same parametrisation as the oracle: every partial
        close(g, e)
    ll_c, g_c, flag = ops.kron_loglik_grad(*args, method="collapsed")
    assert int(flag.abs().sum()) == 0
    close(ll_c, llo)
    Ud, Vd = args[3], args[4]
    for g, e in zip(totals(g_c, Ud, Vd), totals(go, U, V)):
        close(g, e, 1e-10, COLLAPSED_FLOOR)
    ll_f, _ = ops.kron_loglik(*args, method="collapsed")
    close(ll_f, llo)
    # shared alpha (M,): same numbers as the batched alpha
    alpha_s = np.tile(alpha[0], (B, 1))
    llo_s, _ = oracle_interleaved(oracle, t, c, a, U, V, alpha_s, diag, y)
    args_s = dev(t, c, a, U, V, alpha[0], diag, y)
    close(ops.kron_loglik(*args_s, method="collapsed")[0], llo_s)
    close(ops.kron_loglik(*args_s, method="interleaved")[0], llo_s)


def test_kron_failures_and_invalid_diag(ops):
    """A non positive definite series gives -inf / flag / NaN gradients; a non-positive band variance is refused by
    the collapsed method (flag -1) and the other series are unaffected."""
    import torch
    B, N, M, J = 4, 60, 3, 4
    t, c, a, U, V, alpha, diag, y, _ = dense.kron_synthetic(B, N, M, J)
    a[1, 20] = -50.0        # epoch 20 of series 1: negative latent variance
    diag[2, 7, 1] = 0.0     # a zero band variance
    args = dev(t, c, a, U, V, alpha, diag, y)
    ll, grads, flag = ops.kron_loglik_grad(*args, method="collapsed")
    assert int(flag[0]) == 0 and int(flag[3]) == 0 and int(flag[1]) != 0 and int(flag[2]) == -1
    assert np.isneginf(float(ll[1])) and np.isneginf(float(ll[2])) and bool(torch.isfinite(ll[[0, 3]]).all())
    for g in grads:
        assert bool(torch.isnan(g[1]).all()) and bool(torch.isfinite(g[0]).all()) and bool(torch.isfinite(g[3]).all())
    for g in (grads[5], grads[6], grads[7]):
        assert bool(torch.isnan(g[2]).all())
    with pytest.raises(ValueError, match="Invalid shape: alpha"):
        ops.kron_loglik(*dev(t, c, a, U, V, alpha[:, :2], diag, y))
    with pytest.raises(ValueError, match="Invalid shape: y"):
        ops.kron_loglik(*dev(t, c, a, U, V, alpha, diag, y[:, :, :2]))


def test_config4_full_shape(ops, oracle):
    """BASELINE configs[4] at the per-GPU shape: 32 series x (N = 50000 epochs x M = 16 bands), J = 6, forward +
    gradient.  The interleaved method walks 800000 rows per series, the collapsed one 50000: they must agree (value
    and total derivatives), replicas must be bit-identical, and the distinct series must match the CPU oracle's
    1-D recursion on the interleaved series."""
    import torch
    B, N, M, J, nb = 32, 50000, 16, 6, 2
    t, c, a, U, V, alpha, diag, y, _ = dense.kron_synthetic(nb, N, M, J)
    llo, go = oracle_interleaved(oracle, t, c, a, U, V, alpha, diag, y)
    rep = B // nb
    args = [x.repeat((rep,) + (1,) * (x.dim() - 1)).contiguous() for x in dev(t, c, a, U, V, alpha, diag, y)]
    ll_c, g_c, flag = ops.kron_loglik_grad(*args, method="collapsed")
    assert int(flag.abs().sum()) == 0
    close(ll_c[:nb], llo)
    assert bool((ll_c.view(rep, nb) == ll_c[:nb]).all())
    for g, e in zip(totals([x[:nb] for x in g_c], args[3][:nb], args[4][:nb]), totals(go, U, V)):
        close(g, e, 1e-10, COLLAPSED_FLOOR)
    for g in g_c:
        assert bool((g.view((rep, nb) + tuple(g.shape[1:])) == g[:nb]).all())
    ll_i, g_i, flag = ops.kron_loglik_grad(*args, method="interleaved")
    assert int(flag.abs().sum()) == 0
    close(ll_i[:nb], llo)
    for g, e in zip(g_i, go):
        close(g[:nb], e)
    assert float((ll_i - ll_c).abs().max()) <= 1e-10 * float(ll_c.abs().max())


@pytest.mark.parametrize("M", [2, 4, 8, 16, 32])
def test_kron_banded_collapse_matches_thread_per_epoch(ops, monkeypatch, M):
    """The lanes-over-bands collapse kernels (M a power of two up to 32) against the thread-per-epoch ones: value and all
    eight gradients, a ragged last block (N not a multiple of 1024 epochs), shared alpha, an invalid band variance."""
    B, N, J = 3, 2500, 4
    t, c, a, U, V, alpha, diag, y, _ = dense.kron_synthetic(B, N, M, J)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("C2_KRON_BANDED", mode)
        ll, g, flag = ops.kron_loglik_grad(*dev(t, c, a, U, V, alpha, diag, y), method="collapsed")
        lls, gs, _ = ops.kron_loglik_grad(*dev(t, c, a, U, V, alpha[0].copy(), diag, y), method="collapsed")
        d2 = diag.copy(); d2[1, N // 2, M - 1] = -1.0
        llb, gb, flagb = ops.kron_loglik_grad(*dev(t, c, a, U, V, alpha, d2, y), method="collapsed")
        assert int(flag.abs().sum()) == 0 and flagb.cpu().tolist() == [0, -1, 0]
        res[mode] = [ll, *g, lls, *gs, llb, *gb]
        assert np.isneginf(llb.cpu().numpy()[1]) and np.isnan(gb[-1].cpu().numpy()[1]).all()   # by of the invalid series
    for x0, x1 in zip(res["0"], res["1"]):
        x0, x1 = x0.cpu().numpy(), x1.cpu().numpy()
        fin = np.isfinite(x0)
        assert np.array_equal(fin, np.isfinite(x1))
        # (two orders of the band sums in front of the SAME 1-D solver, whose problem has kappa = k(0) A ~ 1e3 .. 1e4:
        # rounding differences of the collapsed inputs come back multiplied by ~eps kappa^2)
        np.testing.assert_allclose(x1[fin], x0[fin], rtol=1e-10, atol=1e-11 * max(1.0, float(np.abs(x0[fin]).max())))
