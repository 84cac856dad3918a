# -*- coding: utf-8 -*-
"""Randomised shape sweep of the device C-ABI against the CPU oracle (fixed seeds): batch sizes that do not fill a
wavefront, every width 1..32, series lengths around the block / checkpoint / ring sizes, 1..70 right-hand sides,
shared and per-series time grids.  Complements the targeted cases of test_gpu_ops.py; tolerance 1e-10 relative per element
with an absolute floor of 1e-12 of the largest element."""
import numpy as np
import pytest

from oracle import dense

pytestmark = pytest.mark.gpu


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
    """|a - b| <= tol |b| + floor max(1, max|b|) per element (same rule as tests/test_gpu_ops.py)."""
    a = a.cpu().numpy() if hasattr(a, "cpu") else a
    np.testing.assert_allclose(a, b, rtol=tol, atol=floor * max(1.0, float(np.abs(b).max())))


def problem(rng, B, N, J):
    """A positive-definite batch of width J (columns dropped from an even-width SHO sum, diagonal lifted)."""
    Je = J if J % 2 == 0 else J + 1
    t, c, a, U, V, y = dense.synthetic_batch(B, max(N, 2), Je)
    t = np.ascontiguousarray(t[:, :N]); a = np.ascontiguousarray(a[:, :N]) + 1.0
    U = np.ascontiguousarray(U[:, :N, :J]); V = np.ascontiguousarray(V[:, :N, :J])
    c = np.ascontiguousarray(c[:, :J]); y = np.ascontiguousarray(y[:, :N])
    return t, c, a, U, V, y


CASES = [(int(s),) for s in range(24)]


@pytest.mark.parametrize("seed", [s for (s,) in CASES])
def test_fuzz_loglik_grad(ops, oracle, seed):
    rng = np.random.default_rng(5000 + seed)
    B = int(rng.integers(1, 40)); J = int(rng.integers(1, 33))
    N = int(rng.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 24, 25, 31, 33, 63, 64, 65, 100, 257]))
    t, c, a, U, V, y = problem(rng, B, N, J)
    td, cd, ad, Ud, Vd, yd = dev(t, c, a, U, V, y)
    llo, go, flago = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=2)
    ll, grads, flag = ops.loglik_grad(td, cd, ad, Ud, Vd, yd)
    assert flag.cpu().tolist() == list(flago)
    ok = np.asarray(flago) == 0
    close(ll[ok], llo[ok])
    for g, e in zip(grads, go):
        close(g[ok], e[ok])
    ll0, flag0 = ops.loglik(td, cd, ad, Ud, Vd, yd)
    close(ll0[ok], llo[ok])
    d, W, flagf = ops.factor(td, cd, ad, Ud, Vd)
    d2, W2, S2, _ = ops.factor(td, cd, ad, Ud, Vd, workspace=True)
    bd = rng.standard_normal((B, N)); bW = rng.standard_normal((B, N, J))
    bdd, bWd = dev(bd, bW)
    res = ops.factor_rev(td, cd, ad, Ud, Vd, d2, W2, S2, bdd, bWd)   # segment replay from every 8th / 4th / 2nd S row
    for b in range(min(B, 3)):
        do = np.empty(N); Wo = np.empty((N, J)); So = np.empty((N, J, J))
        fo = oracle.factor_flag(t[b], c[b], a[b], U[b], V[b], do, Wo, So)
        if fo == 0:
            close(d[b], do); close(W[b], Wo); close(d2[b], do); close(W2[b], Wo); close(S2[b], So)
            outs = [np.empty(N), np.empty(J), np.empty(N), np.empty((N, J)), np.empty((N, J))]
            oracle.factor_rev(t[b], c[b], a[b], U[b], V[b], do, Wo, So, bd[b], bW[b], *outs)
            for r_, e_ in zip(res, outs):
                close(r_[b], e_)


@pytest.mark.parametrize("seed", [s for (s,) in CASES])
def test_fuzz_sweeps(ops, oracle, seed):
    rng = np.random.default_rng(7000 + seed)
    B = int(rng.integers(1, 20)); J = int(rng.integers(1, 33)); nrhs = int(rng.choice([1, 1, 2, 3, 4, 5, 7, 8, 9, 16, 31, 33, 64, 70]))
    N = int(rng.choice([1, 2, 5, 8, 9, 10, 16, 17, 25, 26, 40, 129]))
    t, c, a, U, V, y = problem(rng, B, N, J)
    W = (0.3 / J) * rng.standard_normal((B, N, J))
    Y = rng.standard_normal((B, N, nrhs))
    shared_t = bool(rng.integers(0, 2))
    if shared_t:
        t = np.repeat(t[:1], B, axis=0)
    td, cd, Ud, Vd, Wd, Yd = dev(t[0] if shared_t else t, c, U, V, W, Y)
    for name in ("solve_lower", "solve_upper", "matmul_lower", "matmul_upper"):
        solve = name.startswith("solve")
        sec, secd = (W, Wd) if solve else (V, Vd)
        Zo = np.empty_like(Y); Fo = np.empty((B, N, J, nrhs))
        for b in range(B):
            getattr(oracle, name + "_fwd")(t[b], c[b], U[b], sec[b], Y[b], Zo[b], Fo[b])
        Z1 = getattr(ops, name)(td, cd, Ud, secd, Yd, zero_z=True)
        close(Z1, Zo)
        Z2, F2 = getattr(ops, name)(td, cd, Ud, secd, Yd, workspace=True, zero_z=True)
        close(Z2, Zo); close(F2, Fo)
        bZ = rng.standard_normal((B, N, nrhs))
        (bZd,) = dev(bZ)
        res = getattr(ops, name + "_rev")(td, cd, Ud, secd, Yd, Z2, F2, bZd)
        for b in range(min(B, 2)):
            outs = [np.empty(N), np.empty(J), np.empty((N, J)), np.empty((N, J)), np.empty((N, nrhs))]
            getattr(oracle, name + "_rev")(t[b], c[b], U[b], sec[b], Y[b], Zo[b], Fo[b], bZ[b], *outs)
            if shared_t:   # bt is summed over the batch for a shared grid: compare the per-series outputs only
                outs, got = outs[1:], [r[b] for r in res[1:]]
            else:
                got = [r[b] for r in res]
            for r_, e_ in zip(got, outs):
                close(r_, e_)


@pytest.mark.parametrize("seed", [s for (s,) in CASES[:12]])
def test_fuzz_general_matmul(ops, oracle, seed):
    rng = np.random.default_rng(9000 + seed)
    B = int(rng.integers(1, 12)); J = int(rng.integers(1, 33)); nrhs = int(rng.choice([1, 2, 3, 5, 8]))
    N = int(rng.choice([1, 3, 17, 64, 130])); M = int(rng.choice([1, 2, 9, 33, 100]))
    t2, c, a, U2, V, y = problem(rng, B, M, J)
    lo, hi = t2[:, :1], t2[:, -1:]
    t1 = np.sort(lo - 0.5 + (hi - lo + 1.0) * rng.random((B, N)), axis=1)
    U = rng.standard_normal((B, N, J)); Y = rng.standard_normal((B, M, nrhs))
    t1d, t2d, cd, Ud, Vd, Yd = dev(t1, t2, c, U, V, Y)
    for name in ("general_matmul_lower", "general_matmul_upper"):
        Z0 = rng.standard_normal((B, N, nrhs)); Zo = Z0.copy(); Fo = np.full((B, M, J, nrhs), 3.0)
        for b in range(B):
            getattr(oracle, name)(t1[b], t2[b], c[b], U[b], V[b], Y[b], Zo[b], Fo[b])
        (Zd,) = dev(Z0); (Fd,) = dev(np.full((B, M, J, nrhs), 3.0))
        Zd, Fd = getattr(ops, name)(t1d, t2d, cd, Ud, Vd, Yd, Z=Zd, F=Fd)
        close(Zd, Zo); close(Fd, Fo)
        (Zd2,) = dev(Z0)
        close(getattr(ops, name)(t1d, t2d, cd, Ud, Vd, Yd, Z=Zd2), Zo)
