# -*- coding: utf-8 -*-
"""GPU parity tests of the batched device C-ABI (celerite2_amd.ops -> c2_*), against the CPU oracle
on identical inputs (tolerance 1e-10 relative, north_star) and against the committed golden vectors;
plus size-independent properties at the BASELINE sizes (N=4096, J=4/8)."""
import numpy as np
import pytest

from oracle import dense

pytestmark = pytest.mark.gpu
TOL = 1e-10

CPP_KERNELS = ["real", "complex", "sho1", "sho2", "sum1", "sum2", "sum3", "sum4"]


@pytest.fixture(scope="module")
def ops():
    import torch
    from celerite2_amd import ops as o
    assert torch.cuda.is_available()
    return o


def dev(*xs):
    import torch
    return [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in xs]


FLOOR = 1e-12  # absolute floor, as a fraction of the array's largest magnitude (elements that are themselves
                # the result of cancellation cannot be held to a relative bound of their own size)


def close(a, b, tol=TOL, floor=FLOOR):
    """Element-wise |a - b| <= tol |b| + floor max(1, max|b|): 1e-10 relative per element (north_star), with an
    explicit absolute floor two orders below it."""
    a = a.cpu().numpy() if hasattr(a, "cpu") else a
    np.testing.assert_allclose(a, b, rtol=tol, atol=floor * max(1.0, float(np.abs(b).max())))


def batch_from_golden(golden, names):
    """Stack golden cases with the same J into a batch."""
    return [np.stack([golden["cpp_%s_%s" % (n, k)] for n in names]) for k in ("x", "c", "a", "U", "V", "Y")]


@pytest.mark.parametrize("names", [["real"], ["complex", "sho1"], ["sho2"], ["sum1"], ["sum2"], ["sum3"], ["sum4"]])
def test_golden_cpp_kernels(ops, golden, names):
    """J = 1, 2, 2, 3, 5, 7, 4 -- every group size up to 8, padded lanes included."""
    x, c, a, U, V, Y = batch_from_golden(golden, names)
    xd, cd, ad, Ud, Vd, Yd = dev(x, c, a, U, V, Y)
    d, W, flag = ops.factor(xd, cd, ad, Ud, Vd)
    assert int(flag.abs().sum()) == 0
    for b, n in enumerate(names):
        p = "cpp_%s_" % n
        close(d[b], golden[p + "d"])
    sd = d.sqrt()[:, :, None]
    # expectations: dense Cholesky on the REFERENCE's K = term.to_dense() (tests/golden/make_golden_ref.py); the CPU
    # restatement is 1e-15 ... 2e-14 of the largest entry from them, so the north_star criterion applies unchanged
    close(ops.solve_lower(xd, cd, Ud, W, Yd) / sd, np.stack([golden["cpp_%s_solve_lower" % n] for n in names]))
    close(ops.solve_upper(xd, cd, Ud, W, (Yd / sd).contiguous()),
          np.stack([golden["cpp_%s_solve_upper" % n] for n in names]))
    close(ops.solve_upper(xd, cd, Ud, W, (ops.solve_lower(xd, cd, Ud, W, Yd) / d[:, :, None]).contiguous()),
          np.stack([golden["cpp_%s_apply_inverse" % n] for n in names]))      # K^-1 Y (numpy.py:94-98)
    close(ops.matmul_lower(xd, cd, Ud, Vd, Yd), np.stack([golden["cpp_%s_matmul_lower" % n] for n in names]))
    close(ops.matmul_upper(xd, cd, Ud, Vd, Yd), np.stack([golden["cpp_%s_matmul_upper" % n] for n in names]))
    close(ops.dot_tril(xd, cd, Ud, W, d, Yd), np.stack([golden["cpp_%s_dot_tril" % n] for n in names]))
    ll, flag = ops.loglik(xd, cd, ad, Ud, Vd, Yd[:, :, 0].contiguous())
    close(ll, np.array([golden["cpp_%s_loglik" % n] for n in names]))


def test_config1_loglik(ops, golden):
    """BASELINE configs[0] through the HIP path: N=1000, J=2, rel err <= 1e-10 vs dense."""
    t, diag, y = golden["cfg1_t"], golden["cfg1_diag"], golden["cfg1_y"]
    c, a, U, V = dense.celerite_matrices(dense.sho_term(5.0, 0.1, 3.45), t, diag)
    td, cd, ad, Ud, Vd, yd = dev(t[None], c[None], a[None], U[None], V[None], y[None])
    ll, flag = ops.loglik(td, cd, ad, Ud, Vd, yd)
    assert int(flag[0]) == 0
    assert abs(float(ll[0]) - golden["cfg1_loglik"]) <= 1e-10 * abs(golden["cfg1_loglik"])
    # ... and against the reference's own numpy GaussianProcess.log_likelihood on these inputs (configs[0] as worded)
    assert abs(float(ll[0]) - golden["cfg1_loglik_ref_gp"]) <= 1e-10 * abs(golden["cfg1_loglik_ref_gp"])


@pytest.mark.parametrize("J", [2, 4, 6, 8, 16, 32])
def test_ops_vs_oracle_batched(ops, oracle, J):
    import torch
    B, N, nrhs = 5, 257, 3
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    rng = np.random.default_rng(3)
    Y = rng.standard_normal((B, N, nrhs))
    td, cd, ad, Ud, Vd, yd, Yd = dev(t, c, a, U, V, y, Y)
    d, W, S, flag = ops.factor(td, cd, ad, Ud, Vd, workspace=True)
    do = np.empty_like(a); Wo = np.empty_like(V); So = np.empty((B, N, J, J))
    for b in range(B):
        oracle.factor(t[b], c[b], a[b], U[b], V[b], do[b], Wo[b], So[b])
    assert int(flag.abs().sum()) == 0
    close(d, do); close(W, Wo); close(S, So)
    # in-place factor: d aliases a, W aliases V
    a2, V2 = ad.clone(), Vd.clone()
    d2, W2, _ = ops.factor(td, cd, a2, Ud, V2, d=a2, W=V2)
    d3, W3, _ = ops.factor(td, cd, ad, Ud, Vd)            # workspace-free, out of place (widths 4, 2: parallel along time here)
    assert d2.data_ptr() == a2.data_ptr()
    if J in (2, 4):
        close(d2, d3.cpu().numpy(), 1e-12); close(W2, W3.cpu().numpy(), 1e-10)
    else:
        assert torch.equal(d2, d3) and torch.equal(W2, W3)
    close(d2, do); close(W2, Wo); close(d3, do); close(W3, Wo)
    for name in ("solve_lower", "solve_upper", "matmul_lower", "matmul_upper"):
        second = W if name.startswith("solve") else Vd
        second_o = Wo if name.startswith("solve") else V
        Z, F = getattr(ops, name)(td, cd, Ud, second, Yd, workspace=True, zero_z=True) if "matmul" in name else \
            getattr(ops, name)(td, cd, Ud, second, Yd, workspace=True)
        Zo = np.empty_like(Y); Fo = np.empty((B, N, J, nrhs))
        for b in range(B):
            getattr(oracle, name + "_fwd")(t[b], c[b], U[b], second_o[b], Y[b], Zo[b], Fo[b])
        close(Z, Zo); close(F, Fo)
        # in place (Y is Z)
        Yc = Yd.clone()
        Zi = getattr(ops, name)(td, cd, Ud, second, Yc, Z=Yc)
        ref = Zo if name.startswith("solve") else Zo + Y
        close(Zi, ref)
        # reverse
        bZ = rng.standard_normal((B, N, nrhs))
        (bZd,) = dev(bZ)
        res = getattr(ops, name + "_rev")(td, cd, Ud, second, Yd, Z, F, bZd)
        for b in range(B):
            outs = [np.empty(N), np.empty(J), np.empty((N, J)), np.empty((N, J)), np.empty((N, nrhs))]
            getattr(oracle, name + "_rev")(t[b], c[b], U[b], second_o[b], Y[b], Zo[b], Fo[b], bZ[b], *outs)
            for r, e in zip(res, outs):
                close(r[b], e)
    # factor_rev
    bd = rng.standard_normal((B, N)); bW = rng.standard_normal((B, N, J))
    bdd, bWd = dev(bd, bW)
    res = ops.factor_rev(td, cd, ad, Ud, Vd, d, W, S, bdd, bWd)
    for b in range(B):
        outs = [np.empty(N), np.empty(J), np.empty(N), np.empty((N, J)), np.empty((N, J))]
        oracle.factor_rev(t[b], c[b], a[b], U[b], V[b], do[b], Wo[b], So[b], bd[b], bW[b], *outs)
        for r, e in zip(res, outs):
            close(r[b], e)


def wide_batch(B, N, J, seed=11):
    """A batch of WIDE models: J // 2 underdamped SHO terms (+ one real term for odd J) with frequencies spread over two
    decades and amplitudes 1 / (k + 1) -- a valid (positive definite) celerite kernel of any width."""
    t = np.empty((B, N)); c = np.empty((B, J)); a = np.empty((B, N)); U = np.empty((B, N, J)); V = np.empty((B, N, J))
    y = np.empty((B, N))
    for b in range(B):
        rng = np.random.default_rng(seed + b)
        t[b] = np.sort(rng.uniform(0, N / 10.0, N))
        diag = rng.uniform(0.1, 0.3, N)
        terms = [dense.sho_term(1.0 / (k + 1), 0.05 * 100.0 ** rng.uniform(), rng.uniform(1.0, 8.0)) for k in range(J // 2)]
        if J % 2:
            terms.append(dense.real_term(0.7, 0.3))
        c[b], a[b], U[b], V[b] = dense.celerite_matrices(dense.term_sum(*terms), t[b], diag)
        y[b] = np.sin(t[b]) + 0.1 * rng.standard_normal(N)
    return t, c, a, U, V, y


@pytest.mark.parametrize("J,nrhs", [(33, 3), (40, 17), (64, 1), (97, 2), (128, 3)])
def test_wide_models(ops, oracle, J, nrhs):
    """Widths beyond the 32 of the tuned kernels (the reference's dynamic path takes any J, driver.hpp:98-99): every op
    of the drop-in surface on the workgroup-per-series kernels of c2_wide.hip (state in LDS, J <= 128) against the oracle
    -- factor with and without S and in place, the four sweeps with F, in place and with zero_z, their reverses,
    factor_rev, general_matmul_* with F, dot_tril, the composed log-likelihood and its gradient, a failed series."""
    import torch
    B, N = 3, 45
    t, c, a, U, V, y = wide_batch(B, N, J)
    rng = np.random.default_rng(5)
    Y = rng.standard_normal((B, N, nrhs))
    td, cd, ad, Ud, Vd, yd, Yd = dev(t, c, a, U, V, y, Y)
    d, W, S, flag = ops.factor(td, cd, ad, Ud, Vd, workspace=True)
    do = np.empty_like(a); Wo = np.empty_like(V); So = np.empty((B, N, J, J))
    for b in range(B):
        oracle.factor(t[b], c[b], a[b], U[b], V[b], do[b], Wo[b], So[b])
    assert int(flag.abs().sum()) == 0
    close(d, do); close(W, Wo); close(S, So)
    a2, V2 = ad.clone(), Vd.clone()
    d2, W2, _ = ops.factor(td, cd, a2, Ud, V2, d=a2, W=V2)
    assert d2.data_ptr() == a2.data_ptr()
    close(d2, do); close(W2, Wo)
    for name in ("solve_lower", "solve_upper", "matmul_lower", "matmul_upper"):
        second = W if name.startswith("solve") else Vd
        second_o = Wo if name.startswith("solve") else V
        Z, F = getattr(ops, name)(td, cd, Ud, second, Yd, workspace=True, zero_z=True) if "matmul" in name else \
            getattr(ops, name)(td, cd, Ud, second, Yd, workspace=True)
        Zo = np.empty_like(Y); Fo = np.empty((B, N, J, nrhs))
        for b in range(B):
            getattr(oracle, name + "_fwd")(t[b], c[b], U[b], second_o[b], Y[b], Zo[b], Fo[b])
        close(Z, Zo); close(F, Fo)
        Yc = Yd.clone()
        Zi = getattr(ops, name)(td, cd, Ud, second, Yc, Z=Yc)
        close(Zi, Zo if name.startswith("solve") else Zo + Y)
        if "matmul" in name:
            # zero_z with a caller buffer full of NaN: the *_fwd variants zero Z first (backprop.cpp Z.setZero()), the
            # first visited row included (it receives no product term)
            Zn = torch.full_like(Yd, float("nan"))
            close(getattr(ops, name)(td, cd, Ud, second, Yd, Z=Zn, zero_z=True), Zo)
        bZ = rng.standard_normal((B, N, nrhs))
        (bZd,) = dev(bZ)
        res = getattr(ops, name + "_rev")(td, cd, Ud, second, Yd, Z, F, bZd)
        for b in range(B):
            outs = [np.empty(N), np.empty(J), np.empty((N, J)), np.empty((N, J)), np.empty((N, nrhs))]
            getattr(oracle, name + "_rev")(t[b], c[b], U[b], second_o[b], Y[b], Zo[b], Fo[b], bZ[b], *outs)
            for r, e in zip(res, outs):
                close(r[b], e)
    bd = rng.standard_normal((B, N)); bW = rng.standard_normal((B, N, J))
    bdd, bWd = dev(bd, bW)
    res = ops.factor_rev(td, cd, ad, Ud, Vd, d, W, S, bdd, bWd)
    for b in range(B):
        outs = [np.empty(N), np.empty(J), np.empty(N), np.empty((N, J)), np.empty((N, J))]
        oracle.factor_rev(t[b], c[b], a[b], U[b], V[b], do[b], Wo[b], So[b], bd[b], bW[b], *outs)
        for r, e in zip(res, outs):
            close(r[b], e)
    # prediction products on another grid (ties included), with the F workspace
    M = N
    t1 = np.sort(np.concatenate([t[:, ::3] + 0.013, t[:, 1::7], t[:, :1] - 1.0, t[:, -1:] + 1.0], axis=1), axis=1)
    N1 = t1.shape[1]
    U1 = rng.standard_normal((B, N1, J))
    (t1d, U1d) = dev(t1, U1)
    for name in ("general_matmul_lower", "general_matmul_upper"):
        Zg, Fg = getattr(ops, name)(t1d, td, cd, U1d, Vd, Yd, workspace=True)
        Zo = np.zeros((B, N1, nrhs)); Fo = np.full((B, M, J * nrhs), np.nan)
        for b in range(B):
            getattr(oracle, name + "_fwd")(t1[b], t[b], c[b], U1[b], V[b], Y[b], Zo[b], Fo[b])
        close(Zg, Zo)
        Fg = Fg.cpu().numpy().reshape(B, M, J * nrhs)
        seen = ~np.isnan(Fo)
        close(Fg[seen], Fo[seen])
    # dot_tril, the composed log-likelihood and its gradient, a failed series
    Zd = ops.dot_tril(td, cd, Ud, W, d, Yd)
    Zo = np.empty_like(Y)
    for b in range(B):
        z = Y[b] * np.sqrt(do[b])[:, None]
        oracle.matmul_lower(t[b], c[b], U[b], Wo[b], z.copy(), z)
        Zo[b] = z
    close(Zd, Zo)
    llo, go, flago = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=2)
    ll, fl = ops.loglik(td, cd, ad, Ud, Vd, yd)
    close(ll, llo)
    ll2, grads, fl2 = ops.loglik_grad(td, cd, ad, Ud, Vd, yd)
    assert int(fl.abs().sum()) == 0 and int(fl2.abs().sum()) == 0
    close(ll2, llo)
    for g, e in zip(grads, go):
        close(g, e)
    a3 = a.copy(); a3[1, N // 2] = -5.0
    (a3d,) = dev(a3)
    ll3, grads3, fl3 = ops.loglik_grad(td, cd, a3d, Ud, Vd, yd)
    assert int(fl3[1]) == N // 2 and int(fl3[0]) == 0 and np.isneginf(float(ll3[1]))
    for g, e in zip(grads3, go):
        assert bool(torch.isnan(g[1]).all())
        close(g[[0, 2]], e[[0, 2]])


@pytest.mark.parametrize("J", [2, 4, 8])
def test_loglik_and_grad_vs_oracle(ops, oracle, J):
    B, N = 9, 300
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    td, cd, ad, Ud, Vd, yd = dev(t, c, a, U, V, y)
    llo, go, flago = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=2)
    ll, flag = ops.loglik(td, cd, ad, Ud, Vd, yd)
    close(ll, llo)
    ll2, grads, flag2 = ops.loglik_grad(td, cd, ad, Ud, Vd, yd)
    assert int(flag.abs().sum()) == 0 and int(flag2.abs().sum()) == 0
    close(ll2, llo)
    for g, e in zip(grads, go):
        close(g, e)


@pytest.mark.parametrize("B,N", [(1, 1), (3, 2), (17, 5), (16, 9), (33, 300), (70, 1031)])
def test_two_columns_per_lane_variant(ops, oracle, monkeypatch, B, N):
    """J = 8 has a second lane mapping (c2_loglik4.hip, 4 lanes x 2 columns per series, checkpoint interval 4):
    same results as the oracle, incl. shared t/c, ragged last wavefront and non-positive-definite flags."""
    monkeypatch.setenv("C2_LANES", "4")
    J = 8
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    td, cd, ad, Ud, Vd, yd = dev(t, c, a, U, V, y)
    llo, go, flago = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=2)
    ll, flag = ops.loglik(td, cd, ad, Ud, Vd, yd)
    close(ll, llo)
    ll2, grads, flag2 = ops.loglik_grad(td, cd, ad, Ud, Vd, yd)
    assert int(flag.abs().sum()) == 0 and int(flag2.abs().sum()) == 0
    close(ll2, llo)
    for g, e in zip(grads, go):
        close(g, e)
    # against the one-column-per-lane kernels on the same inputs
    monkeypatch.setenv("C2_LANES", "8")
    ll8, grads8, _ = ops.loglik_grad(td, cd, ad, Ud, Vd, yd)
    close(ll2, ll8.cpu().numpy())
    for g, e in zip(grads, grads8):
        close(g, e.cpu().numpy())
    monkeypatch.setenv("C2_LANES", "4")
    if N > 2 and B > 2:
        a2 = a.copy(); a2[1, N // 2] = -5.0
        (a2d,) = dev(a2)
        ll3, grads3, flag3 = ops.loglik_grad(td, cd, a2d, Ud, Vd, yd)
        assert int(flag3[1]) == N // 2 and int(flag3[0]) == 0 and np.isneginf(float(ll3[1]))
        close(ll3[0:1], llo[0:1])
        close(grads3[2][0], go[2][0])
    # shared time grid and decay rates
    t0, c0 = t[0].copy(), c[0].copy()
    t0d, c0d = dev(t0, c0)
    ll4, flag4 = ops.loglik(t0d, c0d, ad, Ud, Vd, yd)
    for b in range(min(B, 3)):
        e, f = oracle.loglik(t0, c0, a[b], U[b], V[b], y[b])
        if f == 0:
            close(ll4[b:b + 1], np.array([e]))
        else:
            assert int(flag4[b]) == f


@pytest.mark.parametrize("J", [8, 6, 4, 2])
@pytest.mark.parametrize("B,N", [(1, 1), (3, 2), (5, 3), (64, 16), (65, 17), (70, 33), (130, 100), (7, 1031), (200, 257)])
def test_one_lane_per_series_variant(ops, oracle, monkeypatch, B, N, J):
    """Widths 8, 6 (computed as 8 with two empty columns), 4, 2 have a third lane mapping for chip-filling batches (c2_loglik_t.hip: one lane per series, rows through LDS
    transposes, reverse sweep by the backward recursion between checkpoints every 32 rows): same results as the oracle
    on ragged wavefronts, around the tile (8 rows) and checkpoint (32 rows) edges, with unpaired rates, with a failed
    series, with shared t / c -- and the stability guard hands a batch with long gaps to the replay kernels."""
    monkeypatch.setenv("C2_LANES", "1")
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    td, cd, ad, Ud, Vd, yd = dev(t, c, a, U, V, y)
    llo, go, _ = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=2)
    ll, flag = ops.loglik(td, cd, ad, Ud, Vd, yd)
    close(ll, llo)
    ll2, grads, flag2 = ops.loglik_grad(td, cd, ad, Ud, Vd, yd)
    assert int(flag.abs().sum()) == 0 and int(flag2.abs().sum()) == 0
    close(ll2, llo)
    for g, e in zip(grads, go):
        close(g, e)
    # rates that are not pairwise equal take the general exponential path
    c2 = c.copy(); c2[:, 1] *= 1.01; c2[:, J - 2] *= 0.97
    (c2d,) = dev(c2)
    llo2, go2, _ = oracle.loglik_grad_batched(t, c2, a, U, V, y, nthreads=2)
    ll3, grads3, _ = ops.loglik_grad(td, c2d, ad, Ud, Vd, yd)
    close(ll3, llo2)
    for g, e in zip(grads3, go2):
        close(g, e)
    close(ops.loglik(td, c2d, ad, Ud, Vd, yd)[0], llo2)
    if N > 2 and B > 2:   # a failed series: flag, -inf, NaN gradients; its neighbours in the wavefront untouched
        a2 = a.copy(); a2[1, N // 2] = -5.0
        (a2d,) = dev(a2)
        ll4, grads4, flag4 = ops.loglik_grad(td, cd, a2d, Ud, Vd, yd)
        assert int(flag4[1]) == N // 2 and int(flag4[0]) == 0 and np.isneginf(float(ll4[1]))
        good = [b for b in range(B) if b != 1]
        close(ll4[good], llo[good])
        for g, e in zip(grads4, go):
            assert bool(np.isnan(g[1].cpu().numpy()).all())
            close(g[good], e[good])
    # shared time grid and decay rates
    t0d, c0d = dev(t[0].copy(), c[0].copy())
    ts, cs = np.tile(t[0], (B, 1)), np.tile(c[0], (B, 1))
    llo5, go5, flo5 = oracle.loglik_grad_batched(ts, cs, a, U, V, y, nthreads=2)
    ll5, grads5, flag5 = ops.loglik_grad(t0d, c0d, ad, Ud, Vd, yd)
    ok = flo5 == 0
    assert np.array_equal(flag5.cpu().numpy() != 0, ~ok)
    close(ll5[ok], llo5[ok])
    for g, e in zip(grads5, go5):
        close(g[ok], e[ok])
    # gaps that make the backward recursion unsafe (c * span >> guard): the device-side gate routes the batch to the
    # replay kernels; the results are the oracle's either way
    if N >= 4:
        tg = t.copy(); tg[:, N // 2:] += 300.0
        (tgd,) = dev(tg)
        llo6, go6, flo6 = oracle.loglik_grad_batched(tg, c, a, U, V, y, nthreads=2)
        ll6, grads6, flag6 = ops.loglik_grad(tgd, cd, ad, Ud, Vd, yd)
        ok = flo6 == 0
        close(ll6[ok], llo6[ok])
        for g, e in zip(grads6, go6):
            close(g[ok], e[ok])


@pytest.mark.parametrize("B,N", [(1, 1), (33, 1), (64, 2), (3, 3), (64, 9), (31, 16), (128, 33), (64, 64), (192, 65), (100, 66), (64, 130), (150, 257)])
def test_two_lanes_per_series_variant(ops, oracle, monkeypatch, B, N):
    """Width 8 has a fourth lane mapping for batches that give the one-lane kernels half a chip (c2_loglik_k2.hip: a PAIR of
    lanes per series, the packed states split between them by a rotation of the odd lane's vectors, 32 series per
    wavefront): same results as the oracle on ragged wavefronts, around the tile (2 / 8 / 16 rows) and checkpoint (32 rows) edges, with unpaired
    rates, with a failed series, with shared t / c -- and a group of 64 series whose gaps make the backward recursion
    unsafe goes to the replay kernels, the other groups stay."""
    J = 8
    monkeypatch.setenv("C2_LANES", "2")
    t, c, a, U, V, y = dense.synthetic_batch(B, max(N, 2), J)
    t, a, U, V, y = (np.ascontiguousarray(v[:, :N]) for v in (t, a, U, V, y))
    td, cd, ad, Ud, Vd, yd = dev(t, c, a, U, V, y)
    llo, go, _ = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=2)
    ll, flag = ops.loglik(td, cd, ad, Ud, Vd, yd)
    close(ll, llo)
    ll2, grads, flag2 = ops.loglik_grad(td, cd, ad, Ud, Vd, yd)
    assert int(flag.abs().sum()) == 0 and int(flag2.abs().sum()) == 0
    close(ll2, llo)
    for g, e in zip(grads, go):
        close(g, e)
    c2 = c.copy(); c2[:, 1] *= 1.01; c2[:, J - 2] *= 0.97
    (c2d,) = dev(c2)
    llo2, go2, _ = oracle.loglik_grad_batched(t, c2, a, U, V, y, nthreads=2)
    ll3, grads3, _ = ops.loglik_grad(td, c2d, ad, Ud, Vd, yd)
    close(ll3, llo2)
    for g, e in zip(grads3, go2):
        close(g, e)
    close(ops.loglik(td, c2d, ad, Ud, Vd, yd)[0], llo2)
    if N > 2 and B > 2:   # a failed series: flag, -inf, NaN gradients; its neighbours in the wavefront untouched
        a2 = a.copy(); a2[1, N // 2] = -5.0
        (a2d,) = dev(a2)
        ll4, grads4, flag4 = ops.loglik_grad(td, cd, a2d, Ud, Vd, yd)
        assert int(flag4[1]) == N // 2 and int(flag4[0]) == 0 and np.isneginf(float(ll4[1]))
        good = [b for b in range(B) if b != 1]
        close(ll4[good], llo[good])
        for g, e in zip(grads4, go):
            assert bool(np.isnan(g[1].cpu().numpy()).all())
            close(g[good], e[good])
    t0d, c0d = dev(t[0].copy(), c[0].copy())
    ts, cs = np.tile(t[0], (B, 1)), np.tile(c[0], (B, 1))
    llo5, go5, flo5 = oracle.loglik_grad_batched(ts, cs, a, U, V, y, nthreads=2)
    ll5, grads5, flag5 = ops.loglik_grad(t0d, c0d, ad, Ud, Vd, yd)
    ok = flo5 == 0
    assert np.array_equal(flag5.cpu().numpy() != 0, ~ok)
    close(ll5[ok], llo5[ok])
    for g, e in zip(grads5, go5):
        close(g[ok], e[ok])
    if N >= 4 and B >= 64:   # long gaps in the second half of the batch's first group of 64 series only
        tg = t.copy(); tg[40:64, N // 2:] += 300.0
        (tgd,) = dev(tg)
        llo6, go6, flo6 = oracle.loglik_grad_batched(tg, c, a, U, V, y, nthreads=2)
        import torch
        work = ops.loglik_grad_workspace(B, N, J, ad.device)
        ll6, grads6, flag6 = ops.loglik_grad(tgd, cd, ad, Ud, Vd, yd, work=work)
        torch.cuda.synchronize()
        # (one wavefront of 32 series raised the word of its group, if c * gap is beyond the guard at all)
        assert int(work[:2].view(torch.int64)[1]) == (1 if float(work[0]) > 2.0 else 0)
        ok = flo6 == 0
        close(ll6[ok], llo6[ok])
        for g, e in zip(grads6, go6):
            close(g[ok], e[ok])


@pytest.mark.parametrize("lanes,J", [(1, 8), (1, 6), (1, 4), (1, 2), (2, 8)])
def test_one_lane_gaps_are_reanchored(ops, oracle, monkeypatch, lanes, J):
    """Gaps in time (nights, seasons) under the one-lane mapping: the backward recursion of the reverse sweep cannot invert
    the decay across a gap, so the forward pass records an EXTRA checkpoint in front of it (wavefront-uniform, c2_loglik_t.hip)
    and the sweep stays on the one-lane kernels -- the guard word (first double of the workspace: the largest of the
    launch) stays below 2 -- for a few gappy series among many, for a shared grid with gaps, for gaps next to / on
    checkpoint rows and back to back; a wavefront whose series all have their own gaps runs out of extra slots and the
    replay kernels take ITS 64 series (the second word of the workspace counts the wavefronts that fell back), the rest
    of the batch stays on the one-lane kernels.  Results are the oracle's in every case."""
    import torch
    monkeypatch.setenv("C2_LANES", str(lanes))
    # (14 regular checkpoints and twice as many extra slots per wavefront; the two-lane mapping of c2_loglik_k2.hip -- 32
    # series per wavefront, whole groups of 64 -- re-anchors the same way)
    B, N = (150 if lanes == 1 else 192), 420
    wave = 64 // lanes
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    rng = np.random.default_rng(99)

    def run(tg, shared=False, floor=FLOOR):
        if shared:   # ONE problem (grid, matrices) for the whole batch, a right-hand side per series
            rep = lambda x: np.ascontiguousarray(np.tile(x[:1], (B,) + (1,) * (x.ndim - 1)))
            ts, cs, as_, Us, Vs = rep(tg), rep(c), rep(a), rep(U), rep(V)
        else:
            ts, cs, as_, Us, Vs = tg, c, a, U, V
        llo, go, flo = oracle.loglik_grad_batched(ts, cs, as_, Us, Vs, y, nthreads=2)
        assert int(np.abs(flo).sum()) == 0
        args = dev(tg[0].copy() if shared else tg, cs, as_, Us, Vs, y)
        work = ops.loglik_grad_workspace(B, N, J, args[2].device)
        ll, grads, flag = ops.loglik_grad(*args, work=work)
        torch.cuda.synchronize()
        guard, nfall = float(work[0]), int(work[:2].view(torch.int64)[1])
        assert int(flag.abs().sum()) == 0
        close(ll, llo)
        for g, e in zip(grads, go):
            close(g, e, floor=floor)
        close(ops.loglik(*args)[0], llo)
        assert (guard > 2.0) == (nfall > 0)
        run.nfall = nfall
        return guard

    # (1) 5 % of the series with one gap of 100 mean spacings at a row of their own
    tg = t.copy()
    for b in rng.choice(B, size=8, replace=False):
        tg[b, int(rng.integers(1, N)):] += 10.0
    assert run(tg) <= 2.0
    # (2) a shared grid with five gaps, one of them enormous (the decays underflow)
    tg = t.copy()
    for n0, g in ((17, 10.0), (64, 25.0), (65, 1e4), (130, 10.0), (N - 1, 40.0)):
        tg[:, n0:] += g
    assert run(tg, shared=True) <= 2.0
    # (3) gaps next to and on the regular checkpoint rows (multiples of 32), back to back, in front of the last row
    tg = t.copy()
    for b, rows in enumerate(((31,), (32,), (33,), (32, 33), (63, 64, 65), (1,), (2,), (N - 1,), (N - 2, N - 1), (96, 97))):
        for n0 in rows:
            tg[b, n0:] += 10.0
    assert run(tg) <= 2.0
    # (4) spans that accumulate: every series 12x sparser over 32 rows (c_max * span of a segment four times the guard)
    tg = t.copy()
    lo, hi = 40, 72
    tg[:, lo:] += 12.0 * (t[:, lo:] - t[:, lo:lo + 1]) * (np.arange(lo, N) < hi)[None, :] \
        + 12.0 * (t[:, hi - 1:hi] - t[:, lo:lo + 1]) * (np.arange(lo, N) >= hi)[None, :]
    assert np.diff(tg, axis=1).min() > 0
    assert run(tg) <= 2.0
    # (5) every series with five gaps of its own (c_max * gap = 30 at every width): 320 gaps per full wavefront, more extras
    # than there are slots (28) -> the replay kernels, for all three wavefronts
    tg = t.copy()
    for b in range(B):
        for n0 in rng.integers(1, N, size=5):
            tg[b, int(n0):] += 30.0 / c.max()
    # (750 gaps: bt next to a gap is the difference of two gap gradients -- one element in 63000 lands at 2.2e-12 of the
    # largest, on the replay kernels that passed the single-gap cases of the round-2 suite unchanged)
    assert run(tg, floor=4e-12) > 2.0 and run.nfall == -(-B // wave)
    # (6) the same for the series of ONE wavefront only (64 .. 127): that wavefront falls back, its neighbours do not -- a
    # wavefront that runs out of slots costs 64 series, not the batch -- and one series with two gaps in the last wavefront
    tg = t.copy()
    for b in range(64, 128):
        for n0 in rng.integers(1, N, size=5):
            tg[b, int(n0):] += 30.0 / c.max()
    tg[140, 100:] += 10.0; tg[140, 300:] += 10.0
    assert run(tg, floor=4e-12) > 2.0 and run.nfall == lanes
    # (7) three gaps of its own in every series of a batch whose wavefronts have the slots for them (N = 4096 in the bench's
    # `gappy_all` object; here 8 series per wavefront would need B < 64: 9 series x 3 gaps = 27 <= 28 extras)
    tg = t.copy()
    for b in range(9):
        for n0 in rng.integers(1, N, size=3):
            tg[b, int(n0):] += 30.0 / c.max()
    assert run(tg, floor=4e-12) <= 2.0 and run.nfall == 0


@pytest.mark.parametrize("J", [8, 7, 4, 3, 2, 1])
@pytest.mark.parametrize("N", [2, 9, 10, 15, 16, 17, 31, 32, 33, 41, 130, 258])   # (blocks of 16 rows from row 0: their edges)
def test_group_mapping_backward_recursion_and_its_fallback(ops, oracle, monkeypatch, J, N):
    """The group mappings (up to eight lanes per series, c2_loglik.hip) run their reverse sweep by the BACKWARD recursion
    from recorded W rows, re-anchored at every fourth checkpoint (32 rows) or -- where c * span over 32 rows is beyond the
    guard -- at every one (8 rows); a wavefront with a segment it cannot invert (gaps in time) is taken by the replay sweep
    launched behind it.  Same results as
    the oracle for a batch in which some wavefronts have gaps and others do not, for gaps in every series, and with the
    backward form switched off (C2_LOGLIK_BACK=0: the replay alone)."""
    B = 40   # five wavefronts at J = 8, fewer for narrower groups
    Je = J if J % 2 == 0 else J + 1
    t, c, a, U, V, y = dense.synthetic_batch(B, N, Je)
    U = np.ascontiguousarray(U[:, :, :J]); V = np.ascontiguousarray(V[:, :, :J]); c = np.ascontiguousarray(c[:, :J])
    a = a + 1.0
    rng = np.random.default_rng(4100 + 10 * J + N)
    for gaps in ("none", "some", "all", "sparser grid"):
        tg = t.copy()
        if gaps == "sparser grid":   # c * span beyond the guard over 32 rows, within it over 8: an anchor at every segment
            tg = t[:, :1] + 6.0 * (t - t[:, :1])
        elif gaps != "none" and N > 2:
            for b in (range(B) if gaps == "all" else rng.choice(B, size=7, replace=False)):
                tg[b, int(rng.integers(1, N)):] += 40.0 / c.max()
        llo, go, flo = oracle.loglik_grad_batched(tg, c, a, U, V, y, nthreads=2)
        assert int(np.abs(flo).sum()) == 0
        args = dev(tg, c, a, U, V, y)
        # (backward recursion in the scaled frame -- the default --, in the plain frame, and the replay alone)
        # ... and the scaled form with the rows of U, V, bU, bV as 128-byte lines (C2_LOGLIK_LINES=1: J = 8, even N; off by default)
        for back, scaled, lines in (("1", "1", "0"), ("1", "0", "0"), ("0", "1", "0"), ("1", "1", "1")):
            monkeypatch.setenv("C2_LOGLIK_BACK", back)
            monkeypatch.setenv("C2_LOGLIK_SCALED", scaled)
            monkeypatch.setenv("C2_LOGLIK_LINES", lines)
            ll, grads, flag = ops.loglik_grad(*args)
            assert int(flag.abs().sum()) == 0
            close(ll, llo)
            for g, e in zip(grads, go):
                close(g, e, floor=4e-12)
        monkeypatch.delenv("C2_LOGLIK_BACK")
        monkeypatch.delenv("C2_LOGLIK_SCALED")
        monkeypatch.delenv("C2_LOGLIK_LINES")


@pytest.mark.parametrize("lanes", ["8", "4", "2", "1"])
def test_scalar_gradient_lines_every_length(ops, oracle, monkeypatch, lanes):
    """Round 6: the reverse sweeps hand the per-series scalar gradients (ba, by, bt) over as whole 128-byte lines -- sixteen-row tiles
    (two lanes), the upper half of a line held in a register until the lower half is ready (four and eight lanes) -- instead of
    8-row runs as they come.  Every series length from 1 to 50 rows (every position of the last row inside a line and a half line,
    an odd and an even number of 8-row segments, a first segment that is the upper or the lower half), ragged wavefronts, each lane
    mapping forced, all six gradients against the oracle."""
    monkeypatch.setenv("C2_LANES", lanes)
    if lanes == "4":
        monkeypatch.setenv("C2_LOGLIK_Q4_LINES", "1")
    J, B = 8, 70
    for N in range(1, 51):
        t, c, a, U, V, y = dense.synthetic_batch(B, max(N, 2), J)
        t, a, U, V, y = (np.ascontiguousarray(v[:, :N]) for v in (t, a, U, V, y))
        llo, go, flo = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=2)
        assert int(np.abs(flo).sum()) == 0
        ll, grads, flag = ops.loglik_grad(*dev(t, c, a, U, V, y))
        assert int(flag.abs().sum()) == 0, N
        close(ll, llo)
        for g, e in zip(grads, go):
            close(g, e, floor=4e-12)


@pytest.mark.parametrize("lines", ["1", "0"])
@pytest.mark.parametrize("N", [1, 2, 9, 32, 33, 34, 64, 65, 66, 97, 130, 300])
def test_four_lane_pair_scaled_frame_and_its_fallback(ops, oracle, monkeypatch, N, lines):
    """Four lanes per series (c2_loglik_q4.hip, J = 8: both kernels in a scaled frame between anchors 32 rows apart).  Series
    lengths around the anchors (a last segment that is empty, partial, exactly full, itself an anchor); batches in which some
    groups of 64 series have gaps in time (those groups are closed by k_q4_gate and taken by the replay pair behind), all
    of them, none; a grid sparse enough that every group is closed; a ragged last wavefront."""
    monkeypatch.setenv("C2_LANES", "4")
    # rows of U, V, bU, bV as 128-byte lines through LDS tiles (automatic from 11264 series, forced here; series with an odd
    # number of rows take the row-by-row instances either way) and one 64-byte row of sixteen series at a time
    monkeypatch.setenv("C2_LOGLIK_Q4_LINES", lines)
    B, J = 150, 8   # ten wavefronts of 16 series (the last one ragged), three groups of 64
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    rng = np.random.default_rng(4400 + N)
    for gaps in ("none", "some", "all", "sparser grid"):
        tg = t.copy()
        if gaps == "sparser grid":
            tg = t[:, :1] + 6.0 * (t - t[:, :1])
        elif gaps != "none" and N > 2:
            for b in (range(B) if gaps == "all" else rng.choice(64, size=5, replace=False)):   # (some: the first group only)
                tg[b, int(rng.integers(1, N)):] += 40.0 / c.max()
        llo, go, flo = oracle.loglik_grad_batched(tg, c, a, U, V, y, nthreads=2)
        assert int(np.abs(flo).sum()) == 0
        ll, grads, flag = ops.loglik_grad(*dev(tg, c, a, U, V, y))
        assert int(flag.abs().sum()) == 0
        close(ll, llo)
        for g, e in zip(grads, go):
            close(g, e, floor=4e-12)
    # shared time grid and rates; a failed series among its neighbours
    t0d, c0d = dev(t[0].copy(), c[0].copy())
    ad, Ud, Vd, yd = dev(a, U, V, y)
    ll4, g4, f4 = ops.loglik_grad(t0d, c0d, ad, Ud, Vd, yd)
    for b in (0, 17, B - 1):
        e, ge, fe = oracle.loglik_grad(t[0], c[0], a[b], U[b], V[b], y[b])
        if fe == 0 and int(f4[b]) == 0:
            close(ll4[b:b + 1], np.array([e]))
            close(g4[3][b], ge[3], floor=4e-12)
    if N > 2:
        a2 = a.copy(); a2[18, N // 2] = -5.0
        ll3, g3, f3 = ops.loglik_grad(*dev(t, c, a2, U, V, y))
        assert int(f3[18]) == N // 2 and int(f3[17]) == 0 and np.isneginf(float(ll3[18]))
        assert bool(torch_isnan_all(g3, 18))
        llo, go, _ = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=2)
        close(ll3[17:18], llo[17:18])
        close(g3[4][17], go[4][17], floor=4e-12)


def torch_isnan_all(grads, b):
    import torch
    return all(bool(torch.isnan(g[b]).all()) for g in grads)


@pytest.mark.parametrize("J", [1, 3, 5, 6, 7, 12, 16, 24, 32])
@pytest.mark.parametrize("N", [1, 2, 8, 9, 10, 17, 100])
def test_loglik_grad_widths_and_segment_edges(ops, oracle, J, N):
    """Fused log-lik + gradient on padded widths (J < G), on the wide groups (G = 16, 32: LDS-parked replay) and at
    series lengths around the checkpoint interval (N-1 = 0, 1, 7, 8, 9, 16: empty, partial and exactly full
    segments; the replayed W rows start from the W stored with each checkpoint)."""
    B = 10
    Je = J if J % 2 == 0 else J + 1
    t, c, a, U, V, y = dense.synthetic_batch(B, max(N, 2), Je)
    t = np.ascontiguousarray(t[:, :N]); a = np.ascontiguousarray(a[:, :N]) + (Je - J) * 0.0
    U = np.ascontiguousarray(U[:, :N, :J]); V = np.ascontiguousarray(V[:, :N, :J])
    c = np.ascontiguousarray(c[:, :J]); y = np.ascontiguousarray(y[:, :N])
    a = a + 1.0   # dropping a column keeps K positive definite only with some head room on the diagonal
    td, cd, ad, Ud, Vd, yd = dev(t, c, a, U, V, y)
    llo, go, flago = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=2)
    assert int(np.abs(flago).sum()) == 0
    ll, grads, flag = ops.loglik_grad(td, cd, ad, Ud, Vd, yd)
    assert int(flag.abs().sum()) == 0
    close(ll, llo)
    for g, e in zip(grads, go):
        close(g, e)
    ll0, flag0 = ops.loglik(td, cd, ad, Ud, Vd, yd)
    close(ll0, llo)


@pytest.mark.parametrize("J", [8, 16, 24, 32])
def test_loglik_long_series_log_det_range(ops, oracle, J):
    """4096 pivots far from 1: the running product behind log det must be renormalised in every block of every
    group size (the widest groups use 4-row blocks) -- caught as ll = -inf for J > 16 once."""
    B, N = 3, 4096
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    s = 1e-3                        # K -> s K: every pivot ~1e-3, product of 4096 of them ~1e-12288
    a, U, y = a * s, U * s, y * np.sqrt(s)
    td, cd, ad, Ud, Vd, yd = dev(t, c, a, U, V, y)
    ll, flag = ops.loglik(td, cd, ad, Ud, Vd, yd)
    ll2, grads, flag2 = ops.loglik_grad(td, cd, ad, Ud, Vd, yd)
    assert int(flag.abs().sum()) == 0 and int(flag2.abs().sum()) == 0
    for b in range(B):
        e, f = oracle.loglik(t[b], c[b], a[b], U[b], V[b], y[b])
        assert f == 0 and np.isfinite(e)
        close(ll[b:b + 1], np.array([e])); close(ll2[b:b + 1], np.array([e]))


def test_loglik_grad_golden(ops, golden):
    x, c, a, U, V = (golden["py_" + k] for k in ("x", "c", "a", "U", "V"))
    y = np.ascontiguousarray(golden["py_Y"][:, 0])
    xd, cd, ad, Ud, Vd, yd = dev(x[None], c[None], a[None], U[None], V[None], y[None])
    ll, grads, flag = ops.loglik_grad(xd, cd, ad, Ud, Vd, yd)
    close(ll, np.array([golden["py_loglik"]]))
    for name, g in zip(("bt", "bc", "ba", "bU", "bV", "by"), grads):
        close(g[0], golden["py_grad_" + name])


def test_shared_t_and_c(ops, oracle):
    """t (N,) and c (J,) shared by the whole batch (batch stride 0)."""
    B, N, J = 4, 200, 4
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    t0, c0 = t[0].copy(), c[0].copy()
    td, cd, ad, Ud, Vd, yd = dev(t0, c0, a, U, V, y)
    ll, flag = ops.loglik(td, cd, ad, Ud, Vd, yd)
    for b in range(B):
        e, f = oracle.loglik(t0, c0, a[b], U[b], V[b], y[b])
        if f == 0:
            close(ll[b:b + 1], np.array([e]))
        else:
            assert int(flag[b]) == f


def test_not_positive_definite_flags(ops, oracle):
    B, N, J = 6, 128, 4
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    a[2, 57] = -3.0
    a[4, 1] = -1.0
    td, cd, ad, Ud, Vd, yd = dev(t, c, a, U, V, y)
    d, W, flag = ops.factor(td, cd, ad, Ud, Vd)
    assert flag.cpu().tolist() == [0, 0, 57, 0, 1, 0]
    ll, flag2 = ops.loglik(td, cd, ad, Ud, Vd, yd)
    assert flag2.cpu().tolist() == [0, 0, 57, 0, 1, 0]
    assert np.isneginf(ll.cpu().numpy()[[2, 4]]).all() and np.isfinite(ll.cpu().numpy()[[0, 1, 3, 5]]).all()


def test_edge_sizes(ops, oracle):
    """N = 1, N = 2, B not a multiple of the series-per-wave packing."""
    for N in (1, 2, 5):
        B, J = 11, 2
        t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
        td, cd, ad, Ud, Vd, yd = dev(t, c, a, U, V, y)
        llo, go, _ = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=1)
        ll, grads, flag = ops.loglik_grad(td, cd, ad, Ud, Vd, yd)
        close(ll, llo)
        for g, e in zip(grads, go):
            close(g, e)
    with pytest.raises(ValueError):
        import torch
        z = torch.zeros((1, 0, 2), dtype=torch.float64, device="cuda")
        ops.loglik(torch.zeros((1, 0), dtype=torch.float64, device="cuda"), torch.zeros((1, 2), dtype=torch.float64, device="cuda"),
                   torch.zeros((1, 0), dtype=torch.float64, device="cuda"), z, z, torch.zeros((1, 0), dtype=torch.float64, device="cuda"))


def test_get_celerite_matrices_batched(ops):
    import torch
    B, N, J = 3, 100, 6
    rng = np.random.default_rng(9)
    x = np.sort(rng.uniform(0, 10, (B, N)), axis=1); diag = rng.uniform(0.1, 0.3, (B, N))
    cos = [dense.sho_sum_coeffs(J, xi) for xi in (-0.5, 0.0, 0.7)]
    ac = np.stack([co.ac for co in cos]); bc = np.stack([co.bc for co in cos]); dc = np.stack([co.dc for co in cos])
    ar = np.zeros((B, 0))
    a, U, V = ops.get_celerite_matrices(*dev(ar, ac, bc, dc, x, diag))
    for b in range(B):
        c_, a_, U_, V_ = dense.celerite_matrices(cos[b], x[b], diag[b])
        close(a[b], a_, 1e-13); close(U[b], U_, 1e-12); close(V[b], V_, 1e-12)


def test_get_celerite_matrices_unsorted_large_phases(ops):
    """driver.cpp:456-474 has no sortedness precondition: an UNSORTED x whose interior rows leave the range of the
    branch-free sincos (|dc x| >= 1.6e6) while its ends do not must give the reference's values on every row (the
    row-parallel rare-path kernel rewrites them with the library sincos), not NaN -- next to a sorted series of raw Julian
    dates (whole terms on the rare path), a sorted small one, and a term whose rate keeps it in range throughout."""
    import torch
    B, N, J = 4, 300, 6
    rng = np.random.default_rng(19)
    x = np.sort(rng.uniform(0, 10, (B, N)), axis=1); diag = rng.uniform(0.1, 0.3, (B, N))
    x[1, 40:60] += 2.4e6                 # unsorted: interior rows out of range, both ends small
    x[1, 150] = -3.0e6
    x[2] += 2.45e6                       # sorted raw Julian dates
    cos = [dense.sho_sum_coeffs(J, xi) for xi in (-0.5, 0.0, 0.7, 0.2)]
    ac = np.stack([co.ac for co in cos]); bc = np.stack([co.bc for co in cos]); dc = np.stack([co.dc for co in cos])
    dc[:, 0] *= 1e-3                     # this term stays within range on every row of every series
    ar = np.zeros((B, 0))
    a, U, V = ops.get_celerite_matrices(*dev(ar, ac, bc, dc, x, diag))
    assert bool(torch.isfinite(U).all()) and bool(torch.isfinite(V).all())
    for b in range(B):
        co = dense.Coeffs(ar=ar[b], cr=np.zeros(0), ac=ac[b], bc=bc[b], cc=cos[b].cc, dc=dc[b])
        c_, a_, U_, V_ = dense.celerite_matrices(co, x[b], diag[b])
        close(a[b], a_, 1e-13); close(U[b], U_, 1e-12); close(V[b], V_, 1e-12)
    # shared coefficients, shared unsorted grid
    a, U, V = ops.get_celerite_matrices(*dev(ar[0], ac[1], bc[1], dc[1], x[1], diag))
    co = dense.Coeffs(ar=ar[1], cr=np.zeros(0), ac=ac[1], bc=bc[1], cc=cos[1].cc, dc=dc[1])
    for b in range(B):
        c_, a_, U_, V_ = dense.celerite_matrices(co, x[1], diag[b])
        close(a[b], a_, 1e-13); close(U[b], U_, 1e-12); close(V[b], V_, 1e-12)


# ---- BASELINE sizes: size-independent properties (the oracle is only run on a few series) --------------
@pytest.mark.parametrize("J,B", [(4, 1024), (8, 512)])
def test_full_size_properties(ops, oracle, J, B):
    import torch
    N = 4096
    nb = 4  # distinct series; the batch tiles them
    t, c, a, U, V, y = dense.synthetic_batch(nb, N, J)
    rep = B // nb
    td, cd, ad, Ud, Vd, yd = [x.repeat((rep,) + (1,) * (x.dim() - 1)).contiguous() for x in dev(t, c, a, U, V, y)]
    ll, flag = ops.loglik(td, cd, ad, Ud, Vd, yd)
    assert int(flag.abs().sum()) == 0
    # (1) identical series -> bit-identical results wherever they sit in the batch / wavefront
    assert torch.equal(ll[:nb].repeat(rep), ll)
    # (2) oracle parity on the distinct series
    llo, go, _ = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=4)
    close(ll[:nb], llo)
    # (3) gradient: parity + directional finite difference of the GPU forward
    ll2, grads, flag2 = ops.loglik_grad(td[:64], cd[:64], ad[:64], Ud[:64], Vd[:64], yd[:64])
    close(ll2[:nb], llo)
    for g, e in zip(grads, go):
        close(g[:nb], e)
    # (4) solve/matmul round trip: L^-1 (L z) == z with L = I + tril(U W^T)
    d, W, _ = ops.factor(td[:8], cd[:8], ad[:8], Ud[:8], Vd[:8])
    z = torch.randn((8, N, 2), dtype=torch.float64, device="cuda")
    Lz = ops.matmul_lower(td[:8], cd[:8], Ud[:8], W, z, Z=z.clone())
    back = ops.solve_lower(td[:8], cd[:8], Ud[:8], W, Lz)
    assert float((back - z).abs().max()) <= 1e-10 * max(1.0, float(z.abs().max()))
    # (5) linearity of matmul_lower in Y
    y1 = torch.randn((8, N, 1), dtype=torch.float64, device="cuda")
    y2 = torch.randn((8, N, 1), dtype=torch.float64, device="cuda")
    f = lambda v: ops.matmul_lower(td[:8], cd[:8], Ud[:8], Vd[:8], v.contiguous())
    lhs, rhs = f(2.0 * y1 - 3.0 * y2), 2.0 * f(y1) - 3.0 * f(y2)
    assert float((lhs - rhs).abs().max()) <= 1e-10 * float(rhs.abs().max())


def test_bench_scale_batch(ops, oracle):
    """A chip-filling batch at the bench shape (16384 x N=4096 x J=8: two wavefronts per SIMD, and the batch size
    from which c2_loglik switches to the two-columns-per-lane kernel by itself): results identical for identical
    series wherever they sit, equal to the oracle on the distinct ones, forward-only == forward of the gradient."""
    import torch
    B, N, J, nb = 16384, 4096, 8, 4
    t, c, a, U, V, y = dense.synthetic_batch(nb, N, J)
    rep = B // nb
    td, cd, ad, Ud, Vd, yd = [x.repeat((rep,) + (1,) * (x.dim() - 1)).contiguous() for x in dev(t, c, a, U, V, y)]
    llo, go, _ = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=4)
    ll, flag = ops.loglik(td, cd, ad, Ud, Vd, yd)
    assert int(flag.abs().sum()) == 0
    assert torch.equal(ll[:nb].repeat(rep), ll)
    close(ll[:nb], llo)
    ll2, grads, flag2 = ops.loglik_grad(td, cd, ad, Ud, Vd, yd)
    assert int(flag2.abs().sum()) == 0
    close(ll2[:nb], llo)
    assert float((ll2 - ll).abs().max()) <= 1e-10 * float(ll.abs().max())
    for g, e in zip(grads, go):
        close(g[:nb], e)
        assert torch.equal(g[:nb].repeat((rep,) + (1,) * (g.dim() - 1)), g)


@pytest.mark.parametrize("B", [65536, 32768, 16384, 8192])
def test_config2_full_batch_one_gpu(ops, oracle, B):
    """BASELINE configs[2] at the size the bench runs on ONE GPU: 65536 series x N=4096 x J=8, forward + gradient
    (41 GB of inputs, 41 GB of gradients, the replay records) -- and at its 2-, 4- and 8-GPU shards, which the dispatch gives
    to the two-lanes-per-series kernels (32768) and to the eight-lane pair (16384: two wavefronts per SIMD; 8192: one).  8 distinct series tile the batch: every replica must equal its original bit
    for bit wherever it sits (even / odd pair of a wavefront, any wavefront), the distinct ones must match the oracle."""
    import torch
    N, J, nb = 4096, 8, 8
    t, c, a, U, V, y = dense.synthetic_batch(nb, N, J)
    rep = B // nb
    td, cd, ad, Ud, Vd, yd = [x.repeat((rep,) + (1,) * (x.dim() - 1)).contiguous() for x in dev(t, c, a, U, V, y)]
    llo, go, _ = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=8)
    ll, grads, flag = ops.loglik_grad(td, cd, ad, Ud, Vd, yd)
    torch.cuda.synchronize()
    assert int(flag.abs().sum()) == 0
    close(ll[:nb], llo)
    assert bool((ll.view(rep, nb) == ll[:nb]).all())
    for g, e in zip(grads, go):
        close(g[:nb], e)
        assert bool((g.view((rep, nb) + tuple(g.shape[1:])) == g[:nb]).all())
    del grads
    ll1, flag1 = ops.loglik(td, cd, ad, Ud, Vd, yd)   # forward-only kernel at the same batch
    assert int(flag1.abs().sum()) == 0
    close(ll1[:nb], llo)
    assert bool((ll1.view(rep, nb) == ll1[:nb]).all())


@pytest.mark.parametrize("B", [65536, 32768, 16384, 8192])
def test_bench_generator_batch_sample_vs_oracle(ops, oracle, B):
    """The batch bench.py TIMES -- synth.device_batch_fast: B DISTINCT series from the device generator, not replicas of
    eight -- at configs[2]'s full size and at its shards: 32 series drawn at random (bench.take_sample, the `parity_sample`
    leg of the bench line) against the CPU oracle, log-likelihood and all six gradients."""
    import torch
    import bench
    from celerite2_amd import synth
    N, J = 4096, 8
    args = synth.device_batch_fast(0, B, N, J, torch.device("cuda:0"))
    ll, grads, flag = ops.loglik_grad(*args)
    torch.cuda.synchronize()
    assert int(flag.abs().sum()) == 0
    smp = bench.take_sample(args, ll, grads, 32, 1)
    del args, grads
    llo, go, flo = oracle.loglik_grad_batched(*smp["inputs"], nthreads=16)
    assert int(np.abs(flo).sum()) == 0
    close(smp["ll"], llo)
    for g, e in zip(smp["grads"], go):
        for b in range(len(llo)):   # (per series: the floor term scales with that series' largest entry)
            close(g[b], e[b])


@pytest.mark.parametrize("B", [65536, 32768, 16384, 8192])
def test_bench_generator_full_batch_vs_oracle(ops, oracle, B):
    """EVERY series of the batch bench.py times (configs[2] at full size, and its 2 / 4 / 8-GPU shards: a different lane
    mapping each) against the CPU oracle -- distinct data at every lane position of every wavefront, log-likelihood and all
    six gradients under the tests' element-wise criterion (bench.parity_all: inputs stream to the host 2048 series at a
    time, the oracle runs on all host threads, the comparison on the device)."""
    import torch
    import bench
    from celerite2_amd import synth
    N, J = 4096, 8
    args = synth.device_batch_fast(0, B, N, J, torch.device("cuda:0"))
    ll, grads, flag = ops.loglik_grad(*args)
    torch.cuda.synchronize()
    assert int(flag.abs().sum()) == 0
    r = bench.parity_all(args, ll, grads)
    print("B = %d: all series vs oracle: criterion (<= 1 passes) %s; rel. to largest %s; worst series %s"
          % (B, {k: "%.3f" % v for k, v in r["criterion"].items()}, {k: "%.1e" % v for k, v in r["rel_to_largest"].items()},
             r["worst_series"]))
    assert r["series"] == B and r["oracle_failed"] == 0
    assert r["passes"], r


@pytest.mark.parametrize("B,N,J", [(5, 257, 8), (3, 1000, 2), (4100, 64, 4), (2, 300, 20)])
def test_condition_number_per_series(ops, oracle, B, N, J):
    """c2_condition: kappa[b] = max_n a_n / d_n against the oracle's factor (two slices of the batch at 4100 series), +inf and
    the failing row where a series does not factor; shared t / c."""
    import torch
    t, c, a, U, V, y = dense.synthetic_batch(min(B, 8), N, J)
    if B > 8:
        rng = np.random.default_rng(1)
        rep = (B + 7) // 8
        t, c, U, V = (np.ascontiguousarray(np.tile(v, (rep,) + (1,) * (v.ndim - 1))[:B]) for v in (t, c, U, V))
        a = np.ascontiguousarray(np.tile(a, (rep, 1))[:B] * rng.uniform(1.0, 1.5, (B, 1)))   # (lifting a keeps K positive definite)
    bad = B // 2
    a[bad, N // 3] = -1.0
    kappa, flag = ops.condition(*dev(t, c, a, U, V))
    kap, fl = kappa.cpu().numpy(), flag.cpu().numpy()
    for b in list(range(min(B, 6))) + [bad, B - 1]:
        d = np.empty(N); W = np.empty((N, J))
        f = oracle.factor_flag(t[b], c[b], a[b], U[b], V[b], d, W)
        assert int(fl[b]) == f
        if f:
            assert np.isposinf(kap[b]) and b == bad
        else:
            assert abs(kap[b] - np.max(a[b] / d)) <= 1e-10 * kap[b]
    assert np.isfinite(np.delete(kap, bad)).all()
    k2, _ = ops.condition(*dev(t[0].copy(), c[0].copy(), a, U, V))   # shared grid and rates
    assert k2.shape == (B,) and bool(torch.isposinf(k2[bad]))


def test_config3_full_length_dot_tril(ops, oracle):
    """BASELINE configs[3] at full length: ONE series, N = 10^7, J = 16, nrhs = 32, dot_tril (numpy.py:100-102)
    against the sequential CPU oracle on every element."""
    import torch
    B, N, J, nrhs = 1, 10_000_000, 16, 32
    rng = np.random.default_rng(5)
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    td, cd, ad, Ud, Vd = dev(t, c, a, U, V)
    d, W, flag = ops.factor(td, cd, ad, Ud, Vd)
    assert int(flag[0]) == 0
    del ad, Vd
    Y = rng.standard_normal((N, nrhs))
    (Yd,) = dev(Y[None])
    Zd = ops.dot_tril(td, cd, Ud, W, d, Yd)
    torch.cuda.synchronize()
    z = Y * np.sqrt(d[0].cpu().numpy())[:, None]
    oracle.matmul_lower(t[0], c[0], U[0], np.ascontiguousarray(W[0].cpu().numpy()), z, z)   # in place, as numpy.py:102
    close(Zd[0], z)
    Zi = ops.dot_tril(td, cd, Ud, W, d, Yd, Z=Yd)   # in place on the device too
    close(Zi[0], z)


@pytest.mark.parametrize("J,lanes", [(4, None), (8, "8"), (8, "4"), (3, None)])
def test_failed_series_gradients_are_nan(ops, oracle, monkeypatch, J, lanes):
    """One non-positive-definite series in a batch that shares t and c: its log-likelihood is -inf, its flag the
    failing row, and its six gradients are NaN -- never stale memory (the reference raises, driver.hpp:13-19);
    every other series is untouched.  Both lane mappings of the fused pair."""
    import torch
    from celerite2_amd import autograd as ag
    if lanes:
        monkeypatch.setenv("C2_LANES", lanes)
    B, N = 10, 200
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J + (J % 2))
    U, V, c = np.ascontiguousarray(U[:, :, :J]), np.ascontiguousarray(V[:, :, :J]), np.ascontiguousarray(c[:, :J])
    a = a + 1.0
    for b in range(B):   # one shared grid and one shared set of rates: same kernel matrices, own diagonal and data
        t[b], c[b], U[b], V[b] = t[0], c[0], U[0], V[0]
        a[b] = a[0] + 0.01 * b
    bad = 6
    a[bad, 77] = -5.0
    td, cd, ad, Ud, Vd, yd = dev(t[0], c[0], a, U, V, y)
    out = tuple(torch.full(s, 12345.0, dtype=torch.float64, device="cuda")
                for s in ((B, N), (B, J), (B, N), (B, N, J), (B, N, J), (B, N)))
    ll, grads, flag = ops.loglik_grad(td, cd, ad, Ud, Vd, yd, out=out)
    assert flag.cpu().tolist() == [0] * bad + [77] + [0] * (B - bad - 1)
    assert np.isneginf(float(ll[bad]))
    good = [b for b in range(B) if b != bad]
    llo, go, _ = oracle.loglik_grad_batched(t[good], c[good], a[good], U[good], V[good], y[good], nthreads=2)
    close(ll[good], llo)
    for g, e in zip(grads, go):
        assert bool(torch.isnan(g[bad]).all())
        close(g[good], e)
    # through torch.autograd with shared t and c: with the failed series left in the objective the batch-summed
    # gradients are NaN (not finite garbage); masked out (zero cotangent) it contributes exactly zero
    leaves = [x.clone().requires_grad_(True) for x in (td, cd, ad, Ud, Vd, yd)]
    ag.log_likelihood(*leaves).sum().backward()
    assert bool(torch.isnan(leaves[0].grad).all()) and bool(torch.isnan(leaves[1].grad).all())
    assert bool(torch.isnan(leaves[2].grad[bad]).all()) and bool(torch.isfinite(leaves[2].grad[good]).all())
    leaves = [x.clone().requires_grad_(True) for x in (td, cd, ad, Ud, Vd, yd)]
    ag.log_likelihood(*leaves)[good].sum().backward()
    close(leaves[0].grad, go[0].sum(0)); close(leaves[1].grad, go[1].sum(0))
    assert float(leaves[2].grad[bad].abs().max()) == 0.0
    close(leaves[2].grad[good], go[2])


def test_shape_validation(ops):
    """Every op refuses wrongly shaped arguments with the reference's "Invalid shape: <name>" (driver.cpp:40-46)
    before any pointer reaches a kernel."""
    import torch
    from celerite2_amd import gp as gpm, terms
    B, N, J = 3, 40, 4
    t, c, a, U, V, y = dev(*dense.synthetic_batch(B, N, J))
    Y = y[..., None].contiguous()
    with pytest.raises(ValueError, match="Invalid shape: y"):
        ops.loglik(t, c, a, U, V, y[0])
    with pytest.raises(ValueError, match="Invalid shape: t"):
        ops.loglik(t[:1].contiguous(), c, a, U, V, y)
    with pytest.raises(ValueError, match="Invalid shape: a"):
        ops.factor(t, c, a[:, :-1].contiguous(), U, V)
    with pytest.raises(ValueError, match="Invalid shape: W"):
        ops.solve_lower(t, c, U, V[:, :, :2].contiguous(), Y)
    with pytest.raises(ValueError, match="Invalid shape: Y"):
        ops.solve_lower(t, c, U, V, Y[:2].contiguous())
    with pytest.raises(ValueError, match="Invalid shape: Z"):
        ops.matmul_lower(t, c, U, V, Y, Z=Y[:, :-1].contiguous())
    with pytest.raises(ValueError, match="Invalid shape: c"):
        ops.loglik_grad(t, c[:, :3].contiguous(), a, U, V, y)
    with pytest.raises(ValueError, match="Invalid shape: Y"):
        ops.general_matmul_lower(t, t, c, U, V, Y[:, :5].contiguous())
    k = terms.SHOTerm(S0=5.0, w0=0.1, Q=3.45)
    diag = torch.full((B, N), 0.2, dtype=torch.float64, device="cuda")
    g = gpm.GaussianProcess(k, t, diag=diag)
    with pytest.raises(ValueError, match="Invalid shape: y"):
        g.log_likelihood(y[0])
    with pytest.raises(ValueError, match="Invalid shape"):
        gpm.GaussianProcess(k, t[:, :-1].contiguous(), diag=diag)
    with pytest.raises(ValueError, match="Q must be a scalar"):
        terms.SHOTerm(sigma=1.0, rho=np.array([2.0, 3.0]), tau=3.0)


@pytest.mark.parametrize("J,N", [(8, 1000), (4, 333), (3, 64), (16, 130), (32, 40)])
def test_fused_grad_matches_composite_chain(ops, J, N):
    """The checkpoint/recompute kernels vs the literal op chain with S/F materialised in HBM (both on the GPU)."""
    B = 13
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J if J % 2 == 0 else J + 1)
    if J % 2:  # odd width: drop one column (still a valid celerite system: U V^T low-rank part)
        c, U, V = c[:, :J].copy(), np.ascontiguousarray(U[:, :, :J]), np.ascontiguousarray(V[:, :, :J])
    td, cd, ad, Ud, Vd, yd = dev(t, c, a, U, V, y)
    ll1, g1, f1 = ops.loglik_grad(td, cd, ad, Ud, Vd, yd)
    ll2, g2, f2 = ops._loglik_grad_composite(td, cd, ad, Ud, Vd, yd)
    assert int(f1.abs().sum()) == 0 and int(f2.abs().sum()) == 0
    close(ll1, ll2.cpu().numpy())
    for u, v in zip(g1, g2):
        close(u, v.cpu().numpy())


def test_gp_frontend_matches_dense(ops):
    """The thin GaussianProcess frontend (compute / log_likelihood / apply_inverse / dot_tril / predict)
    against dense linear algebra, as the reference's test_celerite2.py does against celerite v1."""
    import torch
    from celerite2_amd import gp as gpmod, terms

    rng = np.random.default_rng(40582)          # test_celerite2.py:11-19 recipe
    B, N, M = 3, 50, 100
    x = np.sort(rng.uniform(0, 10, (B, N)), axis=1)
    ts = np.sort(rng.uniform(-1, 12, (B, M)), axis=1)
    diag = rng.uniform(0.1, 0.3, (B, N))
    y = np.sin(x)
    kernel = terms.SHOTerm(S0=5.0, w0=0.1, Q=3.45) + terms.RealTerm(a=1.0, c=0.1) + terms.Matern32Term(sigma=0.5, rho=2.0)
    xd, tsd, dd, yd = dev(x, ts, diag, y)
    gp = gpmod.GaussianProcess(kernel, mean=0.3)
    gp.compute(xd, diag=dd)
    ll = gp.log_likelihood(yd).cpu().numpy()
    llf = gp.log_likelihood_fused(yd).cpu().numpy()
    mu_self = gp.predict(yd).cpu().numpy()
    mu_star = gp.predict(yd, tsd).cpu().numpy()
    alpha = gp.apply_inverse(yd).cpu().numpy()
    lz = gp.dot_tril(yd).cpu().numpy()
    for b in range(B):
        K = kernel.get_value(x[b][:, None] - x[b][None, :]) + np.diag(diag[b])
        Ks = kernel.get_value(ts[b][:, None] - x[b][None, :])
        r = y[b] - 0.3
        want = dense.dense_loglik(K, r)
        assert abs(ll[b] - want) <= 1e-10 * abs(want) and abs(llf[b] - want) <= 1e-10 * abs(want)
        np.testing.assert_allclose(alpha[b], np.linalg.solve(K, y[b]), rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(lz[b], np.linalg.cholesky(K) @ y[b], rtol=1e-9, atol=1e-10)
        a_r = np.linalg.solve(K, r)
        np.testing.assert_allclose(mu_star[b], Ks @ a_r + 0.3, rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(mu_self[b], y[b] - diag[b] * a_r, rtol=1e-8, atol=1e-9)
    s = gp.sample(size=4)
    assert s.shape == (B, 4, N) and bool(torch.isfinite(s).all())


@pytest.mark.parametrize("J,nrhs,N", [(16, 32, 40000), (4, 1, 70001), (6, 5, 20000)])
def test_long_series_chunked_matmul(ops, oracle, J, nrhs, N):
    """Long-series matmul_lower/upper take the time-chunked scan (c2_scan.hip): compare with the sequential
    oracle, incl. the F workspace, accumulation into Z, and in-place dot_tril (BASELINE config 4 shape)."""
    import torch
    B = 1
    rng = np.random.default_rng(17)
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J if J % 2 == 0 else J + 1)
    Y = rng.standard_normal((B, N, nrhs))
    td, cd, ad, Ud, Vd, Yd = dev(t, c, a, U, V, Y)
    for name in ("matmul_lower", "matmul_upper"):
        Z0 = rng.standard_normal((B, N, nrhs))
        (Zd,) = dev(Z0)
        Zd, Fd = getattr(ops, name)(td, cd, Ud, Vd, Yd, Z=Zd, workspace=True)
        Zo = Z0[0].copy(); Fo = np.empty((N, J, nrhs))
        getattr(oracle, name)(t[0], c[0], U[0], V[0], Y[0], Zo, Fo)
        close(Zd[0], Zo); close(Fd[0], Fo)
    d, W, flag = ops.factor(td, cd, ad, Ud, Vd)
    assert int(flag[0]) == 0
    Yc = Yd.clone()
    Zi = ops.dot_tril(td, cd, Ud, W, d, Yc, Z=Yc)   # in place
    z = np.ascontiguousarray(Y[0] * np.sqrt(d[0].cpu().numpy())[:, None])
    oracle.matmul_lower(t[0], c[0], U[0], W[0].cpu().numpy(), z, z)
    close(Zi[0], z)
    # size-independent property: linearity in Y
    y2 = torch.randn_like(Yd)
    f = lambda v: ops.matmul_lower(td, cd, Ud, Vd, v.contiguous())
    lhs, rhs = f(Yd - 2.0 * y2), f(Yd) - 2.0 * f(y2)
    assert float((lhs - rhs).abs().max()) <= 1e-10 * max(1.0, float(rhs.abs().max()))


@pytest.mark.parametrize("B,N,nrhs,gap", [(1, 40000, 32, False), (1, 40003, 32, False), (3, 20001, 16, False),
                                          (2, 17000, 64, False), (1, 40000, 32, True), (1, 16384, 32, False)])
def test_matrix_core_matmul_lower(ops, oracle, monkeypatch, B, N, nrhs, gap):
    """Long series with J = 16 and 16 / 32 / 64 right-hand sides take the matrix-core path (c2_mfma.hip: blocks of 16
    rows as fp64 MFMA contractions): matmul_lower with accumulation and dot_tril in place against the sequential oracle,
    ragged last block, several series, and gaps long enough that blocks fall back to the row-by-row walk; the VALU
    path (C2_MFMA=0) stays covered and must agree."""
    import torch
    J = 16
    rng = np.random.default_rng(3)
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    if gap:
        t[:, N // 3:] += 5000.0
        t[:, N // 2 + 5:] += 1e5
    Y = rng.standard_normal((B, N, nrhs))
    Z0 = rng.standard_normal((B, N, nrhs))
    td, cd, ad, Ud, Vd, Yd = dev(t, c, a, U, V, Y)
    d, W, flag = ops.factor(td, cd, ad, Ud, Vd)
    assert int(flag.abs().sum()) == 0
    dh, Wh = d.cpu().numpy(), W.cpu().numpy()
    for mode in ("1", "0"):
        monkeypatch.setenv("C2_MFMA", mode)
        (Zd,) = dev(Z0)
        Zd = ops.matmul_lower(td, cd, Ud, Vd, Yd, Z=Zd)
        Yc = Yd.clone()
        Zi = ops.dot_tril(td, cd, Ud, W, d, Yc, Z=Yc)      # in place
        for b in range(B):
            Zo = Z0[b].copy()
            oracle.matmul_lower(t[b], c[b], U[b], V[b], Y[b], Zo)
            close(Zd[b], Zo)
            z = np.ascontiguousarray(Y[b] * np.sqrt(dh[b])[:, None])
            oracle.matmul_lower(t[b], c[b], U[b], np.ascontiguousarray(Wh[b]), z, z)
            close(Zi[b], z)


@pytest.mark.parametrize("J", [1, 2, 3, 5, 8, 12, 16, 32])
@pytest.mark.parametrize("N", [1, 2, 9, 300])
def test_single_rhs_sweeps(ops, oracle, J, N):
    """nrhs = 1 without the F workspace takes the tuned single-rhs kernels (c2_sweep.hip, transposed scalar streams):
    all four sweeps against the oracle, out of place, in place (Z is Y), accumulating into Z and with zero_z, on
    every group size incl. padded widths, a ragged last wavefront (B = 11) and a shared time grid."""
    B = 11
    rng = np.random.default_rng(1000 + 10 * J + N)
    Je = J if J % 2 == 0 else J + 1
    t, c, a, U, V, y = dense.synthetic_batch(B, max(N, 2), Je)
    t = np.ascontiguousarray(t[:, :N]); U = np.ascontiguousarray(U[:, :N, :J]); V = np.ascontiguousarray(V[:, :N, :J])
    c = np.ascontiguousarray(c[:, :J])
    W = (0.3 / J) * rng.standard_normal((B, N, J))   # keeps L = I + tril(U W^T) well conditioned
    Y = rng.standard_normal((B, N, 1))
    td, cd, Ud, Vd, Wd, Yd = dev(t, c, U, V, W, Y)
    for name in ("solve_lower", "solve_upper", "matmul_lower", "matmul_upper"):
        solve = name.startswith("solve")
        sec, secd = (W, Wd) if solve else (V, Vd)
        Z0 = rng.standard_normal((B, N, 1))
        Zo = Z0.copy()
        for b in range(B):
            getattr(oracle, name)(t[b], c[b], U[b], sec[b], Y[b], Zo[b])   # solve overwrites, matmul accumulates
        (Zd,) = dev(Z0)
        close(getattr(ops, name)(td, cd, Ud, secd, Yd, Z=Zd), Zo)
        Yc = Yd.clone()
        Zi = getattr(ops, name)(td, cd, Ud, secd, Yc, Z=Yc)                  # in place
        assert Zi.data_ptr() == Yc.data_ptr()
        ref = Zo if solve else (Zo - Z0) + Y
        close(Zi, ref)
        if not solve:
            close(getattr(ops, name)(td, cd, Ud, secd, Yd, zero_z=True), Zo - Z0)
        # with the F workspace (backprop.*_fwd: Z zeroed first), every element of F
        Zf, Ff = getattr(ops, name)(td, cd, Ud, secd, Yd, workspace=True, zero_z=True)
        for b in (0, B - 1):
            zo = np.empty((N, 1)); fo = np.empty((N, J, 1))
            getattr(oracle, name + "_fwd")(t[b], c[b], U[b], sec[b], Y[b], zo, fo)
            close(Zf[b], zo); close(Ff[b], fo)
        # reverse pass (single-rhs kernel): bt, bc, bU, bV|bW, bY against the oracle
        bZ = rng.standard_normal((B, N, 1))
        (bZd,) = dev(bZ)
        res = getattr(ops, name + "_rev")(td, cd, Ud, secd, Yd, Zf, Ff, bZd)
        for b in (0, B // 2, B - 1):
            zo = np.empty((N, 1)); fo = np.empty((N, J, 1))
            getattr(oracle, name + "_fwd")(t[b], c[b], U[b], sec[b], Y[b], zo, fo)
            outs = [np.empty(N), np.empty(J), np.empty((N, J)), np.empty((N, J)), np.empty((N, 1))]
            getattr(oracle, name + "_rev")(t[b], c[b], U[b], sec[b], Y[b], zo, fo, bZ[b], *outs)
            for r_, e_ in zip(res, outs):
                close(r_[b], e_)
    # shared time grid and decay rates (batch stride 0)
    t0, c0 = t[0].copy(), c[0].copy()
    t0d, c0d = dev(t0, c0)
    Zs = ops.solve_lower(t0d, c0d, Ud, Wd, Yd)
    for b in (0, B - 1):
        zo = Y[b].copy()
        oracle.solve_lower(t0, c0, U[b], W[b], Y[b], zo)
        close(Zs[b], zo)


@pytest.mark.parametrize("J,nrhs", [(8, 5), (8, 8), (3, 8), (16, 8), (16, 16), (16, 20), (32, 32), (32, 33), (8, 64), (6, 70), (32, 7), (8, 3),
                                    (8, 2), (8, 4), (5, 3), (16, 4), (2, 2), (7, 5), (8, 6), (8, 7), (8, 12)])
def test_multi_rhs_sweeps(ops, oracle, J, nrhs):
    """Two to five right-hand sides (two or three with the workspace) run lanes over J with transposed scalar streams
    (c2_sweep_small.hip), more the kernel with lanes over the right-hand sides (c2_sweep.hip) when J fits the lanes of a
    series, the generic kernel otherwise ((16, 8), (32, 7)): all four sweeps with the F workspace, in place, accumulating,
    with ragged rhs tiles (70 = 64 + 6) and a ragged last wavefront -- and their reverse passes."""
    B, N = 5, 131
    rng = np.random.default_rng(7 * J + nrhs)
    Je = J if J % 2 == 0 else J + 1
    t, c, a, U, V, y = dense.synthetic_batch(B, N, Je)
    U = np.ascontiguousarray(U[:, :, :J]); V = np.ascontiguousarray(V[:, :, :J]); c = np.ascontiguousarray(c[:, :J])
    W = (0.3 / J) * rng.standard_normal((B, N, J))
    Y = rng.standard_normal((B, N, nrhs))
    td, cd, Ud, Vd, Wd, Yd = dev(t, c, U, V, W, Y)
    for name in ("solve_lower", "solve_upper", "matmul_lower", "matmul_upper"):
        solve = name.startswith("solve")
        sec, secd = (W, Wd) if solve else (V, Vd)
        Zo = np.empty_like(Y); Fo = np.empty((B, N, J, nrhs))
        for b in range(B):
            getattr(oracle, name + "_fwd")(t[b], c[b], U[b], sec[b], Y[b], Zo[b], Fo[b])
        Zd, Fd = getattr(ops, name)(td, cd, Ud, secd, Yd, workspace=True, zero_z=True)
        close(Zd, Zo); close(Fd, Fo)
        Yc = Yd.clone()
        Zi = getattr(ops, name)(td, cd, Ud, secd, Yc, Z=Yc)   # in place: solve overwrites, matmul adds to Y
        close(Zi, Zo if solve else Zo + Y)
        if not solve:
            Z0 = rng.standard_normal((B, N, nrhs))
            (Z0d,) = dev(Z0)
            close(getattr(ops, name)(td, cd, Ud, secd, Yd, Z=Z0d), Z0 + Zo)
        # the reverse pass of the same shape (two to seven right-hand sides: lanes over J with transposed scalar streams,
        # c2_sweep_small_rev.hip; more: lanes over the right-hand sides), every series against the oracle
        bZ = rng.standard_normal((B, N, nrhs))
        (bZd,) = dev(bZ)
        res = getattr(ops, name + "_rev")(td, cd, Ud, secd, Yd, Zd, Fd, bZd)
        for b in range(B):
            outs = [np.empty(N), np.empty(J), np.empty((N, J)), np.empty((N, J)), np.empty((N, nrhs))]
            getattr(oracle, name + "_rev")(t[b], c[b], U[b], sec[b], Y[b], Zo[b], Fo[b], bZ[b], *outs)
            for r_, e_ in zip(res, outs):
                close(r_[b], e_)


@pytest.mark.parametrize("J,nrhs,B,N", [(8, 128, 3, 150), (8, 256, 2, 97), (8, 500, 2, 64), (16, 128, 2, 70), (16, 256, 1, 130),
                                        (4, 500, 2, 51), (6, 130, 3, 66), (2, 256, 1, 40), (32, 128, 1, 40)])
def test_large_nrhs_solves(ops, oracle, J, nrhs, B, N):
    """SURVEY.md 8f-4 at its own size: solve_lower / solve_upper with HUNDREDS of right-hand sides (apply_inverse on an
    N x M matrix, core.py:62-66) with and without the F workspace, in place, and their reverse passes -- every element
    against the oracle; plus the matmul pair at the same counts (the covariance's general products feed on them)."""
    rng = np.random.default_rng(11 * J + nrhs)
    Je = J if J % 2 == 0 else J + 1
    t, c, a, U, V, y = dense.synthetic_batch(B, N, Je)
    d_, W_ = np.empty_like(a), np.empty_like(V)
    for b in range(B):
        oracle.factor(t[b], c[b], a[b], U[b], V[b], d_[b], W_[b], np.empty((N, Je, Je)))
    U = np.ascontiguousarray(U[:, :, :J]); V = np.ascontiguousarray(V[:, :, :J]); c = np.ascontiguousarray(c[:, :J])
    W = np.ascontiguousarray(W_[:, :, :J])
    Y = rng.standard_normal((B, N, nrhs))
    td, cd, Ud, Vd, Wd, Yd = dev(t, c, U, V, W, Y)
    for name in ("solve_lower", "solve_upper", "matmul_lower", "matmul_upper"):
        solve = name.startswith("solve")
        sec, secd = (W, Wd) if solve else (V, Vd)
        Zo = np.empty_like(Y); Fo = np.empty((B, N, J, nrhs))
        for b in range(B):
            getattr(oracle, name + "_fwd")(t[b], c[b], U[b], sec[b], Y[b], Zo[b], Fo[b])
        close(getattr(ops, name)(td, cd, Ud, secd, Yd, zero_z=True), Zo)      # without the workspace: the production path
        Zd, Fd = getattr(ops, name)(td, cd, Ud, secd, Yd, workspace=True, zero_z=True)
        close(Zd, Zo); close(Fd, Fo)
        Yc = Yd.clone()
        close(getattr(ops, name)(td, cd, Ud, secd, Yc, Z=Yc), Zo if solve else Zo + Y)
        bZ = rng.standard_normal((B, N, nrhs))
        (bZd,) = dev(bZ)
        res = getattr(ops, name + "_rev")(td, cd, Ud, secd, Yd, Zd, Fd, bZd)
        for b in range(B):
            outs = [np.empty(N), np.empty(J), np.empty((N, J)), np.empty((N, J)), np.empty((N, nrhs))]
            getattr(oracle, name + "_rev")(t[b], c[b], U[b], sec[b], Y[b], Zo[b], Fo[b], bZ[b], *outs)
            for r_, e_ in zip(res, outs):
                close(r_[b], e_)


@pytest.mark.parametrize("B,N,nrhs", [(8, 8, 9), (8, 9, 16), (16, 10, 10), (8, 11, 13), (13, 12, 12), (8, 131, 17), (24, 65, 24),
                                      (9, 200, 31), (8, 66, 32), (16, 403, 11), (8, 37, 14), (8, 64, 20), (8, 8, 32), (8, 13, 25)])
def test_forward_sweeps_nine_to_32_rhs_by_groups_of_rows(ops, oracle, monkeypatch, B, N, nrhs):
    """9 .. 32 right-hand sides at J = 8 without the workspace (c2_sweep_cols.hip: eight lanes per series, two to four
    columns per lane, Y / Z in groups of four rows): the four forward sweeps on whole wavefronts of eight series with the
    B % 8 series left over on k_sweepK, odd and even counts (8- and 16-byte pieces), N from two groups up with every
    remainder mod 4, in place (solves), products with zero_z and accumulating (the latter stays on the old kernel) --
    against the oracle and against the lanes-over-rhs kernel alone (option sweep_cols = 0)."""
    import torch
    rng = np.random.default_rng(100 * nrhs + N + B)
    J = 8
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    W = (0.3 / J) * rng.standard_normal((B, N, J))
    Y = rng.standard_normal((B, N, nrhs))
    td, cd, Ud, Vd, Wd, Yd = dev(t, c, U, V, W, Y)
    for name in ("solve_lower", "solve_upper", "matmul_lower", "matmul_upper"):
        solve = name.startswith("solve")
        sec, secd = (W, Wd) if solve else (V, Vd)
        Zo = np.empty_like(Y); Fo = np.empty((B, N, J, nrhs))
        for b in range(B):
            getattr(oracle, name + "_fwd")(t[b], c[b], U[b], sec[b], Y[b], Zo[b], Fo[b])
        Z1 = getattr(ops, name)(td, cd, Ud, secd, Yd, zero_z=True)
        close(Z1, Zo)
        Zn = torch.full_like(Yd, float("nan"))
        close(getattr(ops, name)(td, cd, Ud, secd, Yd, Z=Zn, zero_z=True), Zo)   # every element written, none read
        Yc = Yd.clone()
        close(getattr(ops, name)(td, cd, Ud, secd, Yc, Z=Yc), Zo if solve else Zo + Y)   # in place
        monkeypatch.setenv("C2_SWEEP_COLS", "0")
        Z0 = getattr(ops, name)(td, cd, Ud, secd, Yd, zero_z=True)
        monkeypatch.delenv("C2_SWEEP_COLS")
        close(Z0, Zo)
        assert float((Z1 - Z0).abs().max()) <= 1e-12 * max(1.0, float(Z0.abs().max()))
    # a grid and rates shared by the batch
    t0d, c0d = dev(np.ascontiguousarray(t[0]), np.ascontiguousarray(c[0]))
    U0 = np.ascontiguousarray(np.tile(U[:1], (B, 1, 1))); W0 = np.ascontiguousarray(np.tile(W[:1], (B, 1, 1)))
    Zs = ops.solve_upper(t0d, c0d, *dev(U0, W0), Yd)
    Zo = np.empty_like(Y)
    for b in range(B):
        oracle.solve_upper_fwd(t[0], c[0], U0[b], W0[b], Y[b], Zo[b], np.empty((N, J, nrhs)))
    close(Zs, Zo)


@pytest.mark.parametrize("B,N,nrhs", [(8, 8, 9), (8, 9, 16), (16, 10, 10), (8, 11, 13), (13, 12, 12), (8, 131, 15), (16, 65, 16),
                                      (9, 200, 11), (8, 66, 14), (8, 403, 9), (8, 13, 12), (24, 37, 16), (8, 21, 17), (8, 34, 24),
                                      (16, 11, 25), (8, 40, 32), (8, 9, 31)])
def test_reverse_sweeps_nine_to_16_rhs_two_columns_per_lane(ops, oracle, monkeypatch, B, N, nrhs):
    """The reverse sweeps with 9 .. 16 right-hand sides at J = 8 (k_sweepC_rev, c2_sweep_cols.hip: eight lanes per series,
    two columns per lane, groups of four rows, the row of the next group carried in slot 4): all four on whole wavefronts
    of eight series with the B % 8 left over on k_sweepK_rev, odd and even counts, every N mod 4 from two groups up --
    (bt, bc, bU, bV / bW, bY) of every series against the oracle and against the old kernel (option sweep_cols = 0)."""
    rng = np.random.default_rng(77 * nrhs + N + B)
    J = 8
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    W = (0.3 / J) * rng.standard_normal((B, N, J))
    Y = rng.standard_normal((B, N, nrhs)); bZ = rng.standard_normal((B, N, nrhs))
    td, cd, Ud, Vd, Wd, Yd, bZd = dev(t, c, U, V, W, Y, bZ)
    for name in ("solve_lower", "solve_upper", "matmul_lower", "matmul_upper"):
        solve = name.startswith("solve")
        sec, secd = (W, Wd) if solve else (V, Vd)
        Zo = np.empty_like(Y); Fo = np.empty((B, N, J, nrhs))
        for b in range(B):
            getattr(oracle, name + "_fwd")(t[b], c[b], U[b], sec[b], Y[b], Zo[b], Fo[b])
        Zd, Fd = dev(Zo, Fo)
        res = getattr(ops, name + "_rev")(td, cd, Ud, secd, Yd, Zd, Fd, bZd)
        monkeypatch.setenv("C2_SWEEP_COLS", "0")
        old = getattr(ops, name + "_rev")(td, cd, Ud, secd, Yd, Zd, Fd, bZd)
        monkeypatch.delenv("C2_SWEEP_COLS")
        for b in range(B):
            outs = [np.empty(N), np.empty(J), np.empty((N, J)), np.empty((N, J)), np.empty((N, nrhs))]
            getattr(oracle, name + "_rev")(t[b], c[b], U[b], sec[b], Y[b], Zo[b], Fo[b], bZ[b], *outs)
            for r_, o_, e_ in zip(res, old, outs):
                close(r_[b], e_)
                close(o_[b], e_)


def test_many_rhs_at_full_size(ops, oracle):
    """The round-4 kernels at the sizes their profiles quote: (1) 8192 series x 4096 rows x 16 right-hand sides through
    k_sweepC / k_sweepC_rev -- four distinct series tile the batch, every replica bit-identical to its original, the
    originals against the oracle (solve_lower, solve_upper, solve_lower_rev), the solve / product round trip on all of it;
    (2) ONE series of 4096 rows with 1024 right-hand sides through the chunk maps over the columns against the oracle."""
    import torch
    B, N, J, nrhs, nb = 8192, 4096, 8, 16, 4
    t, c, a, U, V, y = dense.synthetic_batch(nb, N, J)
    d_, W = np.empty_like(a), np.empty_like(V)
    for b in range(nb):
        oracle.factor(t[b], c[b], a[b], U[b], V[b], d_[b], W[b], np.empty((N, J, J)))
    rng = np.random.default_rng(4)
    Y = rng.standard_normal((nb, N, nrhs)); bZ = rng.standard_normal((nb, N, nrhs))
    rep = B // nb
    tile = lambda x: x.repeat((rep,) + (1,) * (x.dim() - 1)).contiguous()
    td, cd, Ud, Wd, Yd, bZd = [tile(x) for x in dev(t, c, U, W, Y, bZ)]
    Zo = np.empty_like(Y); Fo = np.empty((nb, N, J, nrhs)); Zu = np.empty_like(Y)
    for b in range(nb):
        oracle.solve_lower_fwd(t[b], c[b], U[b], W[b], Y[b], Zo[b], Fo[b])
        oracle.solve_upper_fwd(t[b], c[b], U[b], W[b], Y[b], Zu[b], np.empty((N, J, nrhs)))
    Zl = ops.solve_lower(td, cd, Ud, Wd, Yd)
    close(Zl[:nb], Zo)
    assert bool((Zl.view(rep, nb, N, nrhs) == Zl[:nb]).all())
    Zup = ops.solve_upper(td, cd, Ud, Wd, Yd)
    close(Zup[:nb], Zu)
    assert bool((Zup.view(rep, nb, N, nrhs) == Zup[:nb]).all())
    back = ops.matmul_lower(td, cd, Ud, Wd, Zl, Z=Zl.clone())           # L (L^-1 Y) = Y with L = I + tril(U W^T)
    assert float((back - Yd).abs().max()) <= 1e-10 * float(Yd.abs().max())
    del back, Zup
    Zd, Fd = ops.solve_lower(td, cd, Ud, Wd, Yd, workspace=True)
    res = ops.solve_lower_rev(td, cd, Ud, Wd, Yd, Zd, Fd, bZd)
    for b in range(nb):
        outs = [np.empty(N), np.empty(J), np.empty((N, J)), np.empty((N, J)), np.empty((N, nrhs))]
        oracle.solve_lower_rev(t[b], c[b], U[b], W[b], Y[b], Zo[b], Fo[b], bZ[b], *outs)
        for r_, e_ in zip(res, outs):
            close(r_[b], e_)
    for r_ in res:
        assert bool((r_.view((rep, nb) + tuple(r_.shape[1:])) == r_[:nb]).all())
    del res, Zd, Fd, td, cd, Ud, Wd, Yd, bZd, Zl
    torch.cuda.empty_cache()
    M = 1024
    Y1 = rng.standard_normal((1, N, M))
    Z1 = np.empty_like(Y1)
    oracle.solve_lower_fwd(t[0], c[0], U[0], W[0], Y1[0], Z1[0], np.empty((N, J, M)))
    t1, c1, U1, W1, Y1d = dev(t[:1], c[:1], U[:1], W[:1], Y1)
    close(ops.solve_lower(t1, c1, U1, W1, Y1d), Z1)      # (automatic dispatch: chunk maps over the columns)
    oracle.solve_upper_fwd(t[0], c[0], U[0], W[0], Y1[0], Z1[0], np.empty((N, J, M)))
    close(ops.solve_upper(t1, c1, U1, W1, Y1d), Z1)


@pytest.mark.parametrize("J,nrhs,B,N", [(8, 64, 1, 700), (8, 130, 3, 129), (16, 65, 2, 257), (12, 5, 2, 64), (3, 1, 4, 65),
                                        (5, 70, 1, 2100), (1, 64, 2, 63), (8, 16, 70, 200), (2, 200, 1, 2), (7, 33, 2, 1025)])
def test_many_rhs_solves_as_chunk_maps_over_columns(ops, oracle, monkeypatch, J, nrhs, B, N):
    """c2_solve_cols.hip forced (option solve_cols = 1): solve_lower / solve_upper without workspace as chunk maps with
    lanes over the right-hand sides -- every width up to 16 (padded to 8 / 16), ragged column tiles (65, 130, 200), series
    shorter than / equal to / just beyond a chunk of 64 rows, N = 2, more series than one launch dimension needs, shared
    grid -- out of place and in place, against the oracle and against the row-by-row kernels (option = 0)."""
    import torch
    rng = np.random.default_rng(13 * J + nrhs + N)
    Je = J if J % 2 == 0 else J + 1
    t, c, a, U, V, y = dense.synthetic_batch(B, N, Je)
    d_, W_ = np.empty_like(a), np.empty_like(V)
    for b in range(B):
        oracle.factor(t[b], c[b], a[b], U[b], V[b], d_[b], W_[b], np.empty((N, Je, Je)))
    U = np.ascontiguousarray(U[:, :, :J]); c = np.ascontiguousarray(c[:, :J]); W = np.ascontiguousarray(W_[:, :, :J])
    Y = rng.standard_normal((B, N, nrhs))
    td, cd, Ud, Wd, Yd = dev(t, c, U, W, Y)
    for name in ("solve_lower", "solve_upper"):
        Zo = np.empty_like(Y); Fo = np.empty((B, N, J, nrhs))
        for b in range(B):
            getattr(oracle, name + "_fwd")(t[b], c[b], U[b], W[b], Y[b], Zo[b], Fo[b])
        monkeypatch.setenv("C2_SOLVE_COLS", "1")
        Z1 = getattr(ops, name)(td, cd, Ud, Wd, Yd)
        close(Z1, Zo)
        Yc = Yd.clone()
        close(getattr(ops, name)(td, cd, Ud, Wd, Yc, Z=Yc), Zo)
        monkeypatch.setenv("C2_SOLVE_COLS", "0")
        Z0 = getattr(ops, name)(td, cd, Ud, Wd, Yd)
        close(Z0, Zo)
        assert float((Z1 - Z0).abs().max()) <= 1e-11 * max(1.0, float(Z0.abs().max()))
        monkeypatch.delenv("C2_SOLVE_COLS")
    # a grid shared by the batch
    monkeypatch.setenv("C2_SOLVE_COLS", "1")
    t0 = np.ascontiguousarray(t[0]); c0 = np.ascontiguousarray(c[0])
    (t0d, c0d) = dev(t0, c0)
    U0 = np.ascontiguousarray(np.tile(U[:1], (B, 1, 1))); W0 = np.ascontiguousarray(np.tile(W[:1], (B, 1, 1)))
    Zs = ops.solve_lower(t0d, c0d, *dev(U0, W0), Yd)
    Zo = np.empty_like(Y)
    for b in range(B):
        oracle.solve_lower_fwd(t0, c0, U0[b], W0[b], Y[b], Zo[b], np.empty((N, J, nrhs)))
    close(Zs, Zo)


@pytest.mark.parametrize("N,M", [(2000, 128), (1200, 256), (700, 500)])
def test_large_nrhs_apply_inverse_vs_dense(ops, N, M):
    """apply_inverse on an N x M matrix (M in the hundreds) against the dense K^-1 (numpy Cholesky) on N <= 2000:
    solve_lower, /d, solve_upper through the production dispatch -- the row-by-row kernels on the batch of 3, the
    chunk maps parallel along time on one series (B = 1)."""
    from celerite2_amd import gp as gpmod, terms
    kernel = terms.SHOTerm(S0=5.0, w0=0.1, Q=3.45) + terms.SHOTerm(S0=0.7, w0=2.1, Q=1.3) + terms.RealTerm(a=0.4, c=0.2)
    for B in (3, 1):
        rng = np.random.default_rng(5 + B)
        x = np.sort(rng.uniform(0, N / 10.0, (B, N)), axis=1)
        diag = rng.uniform(0.1, 0.3, (B, N))
        Y = rng.standard_normal((B, N, M))
        xd, dd, Yd = dev(x, diag, Y)
        gp = gpmod.GaussianProcess(kernel)
        gp.compute(xd, diag=dd)
        X = gp.apply_inverse(Yd).cpu().numpy()
        for b in range(B):
            K = kernel.get_value(x[b][:, None] - x[b][None, :]) + np.diag(diag[b])
            want = np.linalg.solve(K, Y[b])
            assert np.abs(X[b] - want).max() <= 1e-9 * np.abs(want).max()


def test_predictive_variance_and_covariance(ops):
    """gp.predict(..., return_var=True / return_cov=True) = core.py:134-150 against dense linear algebra: on the observed
    grid and on a new one, with M = 300 prediction points (apply_inverse with 300 right-hand sides, the general products
    with 300), per-series hyper-parameters, a separate `kernel=` component (core.py:74-86), and conditional samples."""
    import torch
    from celerite2_amd import gp as gpmod, terms

    rng = np.random.default_rng(40582)
    B, N, M = 3, 120, 300
    x = np.sort(rng.uniform(0, 10, (B, N)), axis=1)
    ts = np.sort(rng.uniform(-1, 12, (B, M)), axis=1)
    diag = rng.uniform(0.1, 0.3, (B, N))
    y = np.sin(x)
    S0 = np.array([5.0, 4.0, 6.0])
    comp = terms.SHOTerm(S0=S0, w0=0.1, Q=3.45)
    kernel = comp + terms.RealTerm(a=1.0, c=0.1)
    xd, tsd, dd, yd = dev(x, ts, diag, y)
    gp = gpmod.GaussianProcess(kernel, mean=0.3)
    gp.compute(xd, diag=dd)
    mu, var = gp.predict(yd, tsd, return_var=True)
    mu2, cov = gp.predict(yd, tsd, return_cov=True)
    mu0, var0 = gp.predict(yd, return_var=True)
    mu0c, cov0 = gp.predict(yd, return_cov=True)
    muk, vark = gp.predict(yd, tsd, return_var=True, kernel=comp)
    _, covk = gp.predict(yd, tsd, return_cov=True, kernel=comp, include_mean=False)
    assert torch.equal(mu, mu2) and torch.equal(mu0, mu0c)
    for b in range(B):
        kb = terms.SHOTerm(S0=float(S0[b]), w0=0.1, Q=3.45) + terms.RealTerm(a=1.0, c=0.1)
        cb = terms.SHOTerm(S0=float(S0[b]), w0=0.1, Q=3.45)
        K = kb.get_value(x[b][:, None] - x[b][None, :]) + np.diag(diag[b])
        Ks = kb.get_value(ts[b][:, None] - x[b][None, :])
        Kss = kb.get_value(ts[b][:, None] - ts[b][None, :])
        r = y[b] - 0.3
        want_mu = Ks @ np.linalg.solve(K, r) + 0.3
        want_cov = Kss - Ks @ np.linalg.solve(K, Ks.T)
        np.testing.assert_allclose(mu[b].cpu().numpy(), want_mu, rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(cov[b].cpu().numpy(), want_cov, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(var[b].cpu().numpy(), np.diag(want_cov), rtol=1e-7, atol=1e-9)
        K0 = K - np.diag(diag[b])
        want_cov0 = K0 - K0 @ np.linalg.solve(K, K0)
        np.testing.assert_allclose(cov0[b].cpu().numpy(), want_cov0, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(var0[b].cpu().numpy(), np.diag(want_cov0), rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(mu0[b].cpu().numpy(), y[b] - diag[b] * np.linalg.solve(K, r), rtol=1e-8, atol=1e-9)
        # the component kernel (core.py:74-86): cross-covariances of the component, inverse of the full matrix
        Ksc = cb.get_value(ts[b][:, None] - x[b][None, :])
        Kssc = cb.get_value(ts[b][:, None] - ts[b][None, :])
        np.testing.assert_allclose(muk[b].cpu().numpy(), Ksc @ np.linalg.solve(K, r) + 0.3, rtol=1e-8, atol=1e-9)
        want_covk = Kssc - Ksc @ np.linalg.solve(K, Ksc.T)
        np.testing.assert_allclose(covk[b].cpu().numpy(), want_covk, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(vark[b].cpu().numpy(), np.diag(want_covk), rtol=1e-7, atol=1e-9)
    s = gp.condition(yd, tsd).sample(size=3, regularize=1e-8)
    assert s.shape == (B, 3, M) and bool(torch.isfinite(s).all())
    # the kernel on two grids (c2_kernel_values, terms.py:58-79): per-series coefficients against numpy, shared ones with a
    # shared grid, and a phase beyond the range of the branch-free sincos (raw Julian dates against an origin at 0)
    Kd = kernel.get_value_grid(xd, tsd).cpu().numpy()
    for b in range(B):
        kb = terms.SHOTerm(S0=float(S0[b]), w0=0.1, Q=3.45) + terms.RealTerm(a=1.0, c=0.1)
        np.testing.assert_allclose(Kd[b], kb.get_value(x[b][:, None] - ts[b][None, :]), rtol=1e-12, atol=1e-13)
    fast = terms.SHOTerm(S0=1.0, w0=40.0, Q=30.0)
    big = np.array([2.45e6, 2.45e6 + 0.37, 5.0e5]); small = np.array([0.0, 1.5])
    Kb = fast.get_value_grid(*dev(big, small)).cpu().numpy()
    np.testing.assert_allclose(Kb[0], fast.get_value(big[:, None] - small[None, :]), rtol=1e-9, atol=1e-12)
    with pytest.raises(ValueError):
        gp.condition(yd, tsd[:2])


@pytest.mark.parametrize("tile", ["1", "0"])
def test_kernel_values_factored_exponential(ops, monkeypatch, tile):
    """c2_kernel_values on 32 x 64 tiles (round 6: the exponential factored into a row table and one exponential per column, with tmin / tmax
    the extreme times of a wavefront's eight rows): columns strictly inside a row group's range, exactly on its smallest / largest time,
    far outside on either side (every factor an exponential of a non-positive argument: gaps of 1e4 time units between the rows of one
    group), UNSORTED rows and columns, a growing real term (c < 0: the exponential per entry), ragged tile edges; per-series and shared
    coefficients -- against numpy, and the tile kernel against the thread-per-entry kernel."""
    import torch
    monkeypatch.setenv("C2_KERNEL_VALUES_TILE", tile)
    rng = np.random.default_rng(77)
    B, N, M, Jr, Jc = 3, 77, 150, 2, 3
    t1 = rng.uniform(0.0, 40.0, (B, N))
    t1[0] = np.sort(t1[0]); t1[0, 40:] += 1.0e4             # a gap inside a tile (rows 32 .. 63) and inside a group of eight rows
    t2 = rng.uniform(-5.0, 45.0, (B, M))
    t2[0, :8] = t1[0, 8:16]                                   # columns ON the rows of a group (its tmin, its tmax, in between)
    t2[0, 8:12] = [t1[0, 39] + 10.0, t1[0, 40] - 10.0, 0.5 * (t1[0, 39] + t1[0, 40]), 2.0e4]
    t2[1] = np.sort(t2[1])
    ar = rng.uniform(0.5, 2.0, (B, Jr)); cr = rng.uniform(0.05, 0.5, (B, Jr)); cr[2, 1] = -0.01   # one growing term
    ac = rng.uniform(0.5, 2.0, (B, Jc)); bc = rng.uniform(-0.5, 0.5, (B, Jc)); cc = rng.uniform(0.02, 0.3, (B, Jc)); dc = rng.uniform(0.1, 3.0, (B, Jc))
    dc[1, 0] = -dc[1, 0]
    def want(b):
        tau = np.abs(t1[b][:, None] - t2[b][None, :])
        k = np.zeros_like(tau)
        for i in range(Jr): k += ar[b, i] * np.exp(-cr[b, i] * tau)
        for i in range(Jc): k += np.exp(-cc[b, i] * tau) * (ac[b, i] * np.cos(dc[b, i] * tau) + bc[b, i] * np.sin(dc[b, i] * tau))
        return k
    # criterion: 2e-13 of k(0) = sum ar + sum ac per entry (the terms oscillate and cancel: an entry can be 1e-5 of its largest term;
    # measured against an extended-precision evaluation: 5.5e-14 on tiles -- phases of 3e4 rad through the angle addition -- 9e-16 per entry)
    K = ops.kernel_values(*dev(ar, cr, ac, bc, cc, dc, t1, t2)).cpu().numpy()
    for b in range(B):
        np.testing.assert_allclose(K[b], want(b), rtol=0.0, atol=2e-13 * (ar[b].sum() + ac[b].sum()))
    Ks = ops.kernel_values(*dev(ar[1], cr[1], ac[1], bc[1], cc[1], dc[1], t1[1], t2[1])).cpu().numpy()   # everything shared: B = 1
    np.testing.assert_allclose(Ks[0], want(1), rtol=0.0, atol=2e-13 * (ar[1].sum() + ac[1].sum()))


@pytest.mark.parametrize("J,K", [(8, 1), (8, 2), (8, 3), (8, 5), (8, 7), (8, 65), (8, 130), (4, 67), (16, 5), (16, 33)])
def test_many_rhs_solves_over_awkward_chunk_counts(ops, oracle, monkeypatch, J, K):
    """c2_solve_cols.hip: the chain over the chunks keeps the chunk maps of a series in LDS 64 chunks at a time and requests g four chunks
    ahead (round 6): series of K chunks of 64 rows with K below the ring depth, not a multiple of it, and beyond one LDS piece (65, 130),
    a last chunk that is short -- solve_lower / solve_upper with 70 right-hand sides, forced onto the chunk maps, against the oracle."""
    monkeypatch.setenv("C2_SOLVE_COLS", "1")
    B, nrhs = 2, 70
    N = 64 * K - (13 if K > 1 else 20)
    rng = np.random.default_rng(5 * J + K)
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    d_, W = np.empty_like(a), np.empty_like(V)
    for b in range(B):
        oracle.factor(t[b], c[b], a[b], U[b], V[b], d_[b], W[b], np.empty((N, J, J)))
    Y = rng.standard_normal((B, N, nrhs))
    td, cd, Ud, Wd, Yd = dev(t, c, U, W, Y)
    for name in ("solve_lower", "solve_upper"):
        Zo = np.empty_like(Y); Fo = np.empty((B, N, J, nrhs))
        for b in range(B):
            getattr(oracle, name + "_fwd")(t[b], c[b], U[b], W[b], Y[b], Zo[b], Fo[b])
        close(getattr(ops, name)(td, cd, Ud, Wd, Yd, zero_z=True), Zo)


@pytest.mark.parametrize("B,N", [(8, 8), (8, 10), (16, 12), (8, 14), (8, 34), (16, 64), (8, 132), (16, 200), (7, 64), (8, 33), (13, 9), (21, 65), (9, 130)])
def test_multi_rhs_forward_sweeps_by_lines(ops, oracle, B, N):
    """nrhs = J = 8: the four forward sweeps by aligned 128-byte lines (k_sweep8_lines) on the whole wavefronts of the batch,
    the row-by-row kernel on the B % 8 series left over (B = 13, 21, 9; B = 7: all of it), even and odd N -- with the
    F workspace, without, in place (Z is Y), accumulating into Z -- against the oracle and against the row-by-row kernel
    alone (option sweepk_lines = 0)."""
    from celerite2_amd import _lib
    J = nrhs = 8
    rng = np.random.default_rng(57 * B + N)
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    W = (0.3 / J) * rng.standard_normal((B, N, J))
    Y = rng.standard_normal((B, N, nrhs))
    Z0 = rng.standard_normal((B, N, nrhs))
    td, cd, Ud, Vd, Wd, Yd = dev(t, c, U, V, W, Y)
    for name in ("solve_lower", "solve_upper", "matmul_lower", "matmul_upper"):
        solve = name.startswith("solve")
        sec, secd = (W, Wd) if solve else (V, Vd)
        Zo = np.empty_like(Y); Fo = np.empty((B, N, J, nrhs))
        for b in range(B):
            getattr(oracle, name + "_fwd")(t[b], c[b], U[b], sec[b], Y[b], Zo[b], Fo[b])
        def run_all():
            out = {}
            out["ws"] = getattr(ops, name)(td, cd, Ud, secd, Yd, workspace=True, zero_z=True)
            out["plain"] = getattr(ops, name)(td, cd, Ud, secd, Yd) if solve else getattr(ops, name)(td, cd, Ud, secd, Yd, zero_z=True)
            Yc = Yd.clone()
            out["inplace"] = getattr(ops, name)(td, cd, Ud, secd, Yc, Z=Yc)
            if not solve:
                (Z0d,) = dev(Z0)
                out["acc"] = getattr(ops, name)(td, cd, Ud, secd, Yd, Z=Z0d)
            return out
        res = run_all()
        _lib.set_option("sweepk_lines", 0)
        try:
            ref = run_all()
        finally:
            _lib.set_option("sweepk_lines", None)
        close(res["ws"][0], Zo); close(res["ws"][1], Fo)
        close(res["plain"], Zo)
        close(res["inplace"], Zo if solve else Zo + Y)
        if not solve:
            close(res["acc"], Z0 + Zo)
        close(res["ws"][0], ref["ws"][0].cpu().numpy()); close(res["ws"][1], ref["ws"][1].cpu().numpy())
        close(res["inplace"], ref["inplace"].cpu().numpy())


@pytest.mark.parametrize("B,N", [(8, 8), (8, 10), (16, 12), (8, 14), (24, 33), (8, 34), (16, 64), (8, 131), (16, 200), (7, 64), (13, 9), (21, 65), (9, 130), (8, 11), (8, 13)])
def test_multi_rhs_reverse_sweeps_by_lines(ops, oracle, B, N):
    """nrhs = J = 8: the four reverse sweeps move every width-8 row as half of an aligned 128-byte line through LDS rings
    (k_sweep8_rev_lines) on the whole wavefronts of the batch, the row-by-row kernel on the B % 8 series left over -- every
    length class of the main loop (N - 1 mod 4, even and odd N, the peeled first step, the guarded last ones), against the
    oracle; the row-by-row kernel alone (option sweep_rev_lines = 0) must agree to rounding."""
    from celerite2_amd import _lib
    J = nrhs = 8
    rng = np.random.default_rng(31 * B + N)
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    W = (0.3 / J) * rng.standard_normal((B, N, J))
    Y = rng.standard_normal((B, N, nrhs))
    bZ = rng.standard_normal((B, N, nrhs))
    td, cd, Ud, Vd, Wd, Yd, bZd = dev(t, c, U, V, W, Y, bZ)
    for name in ("solve_lower", "solve_upper", "matmul_lower", "matmul_upper"):
        solve = name.startswith("solve")
        sec, secd = (W, Wd) if solve else (V, Vd)
        Zd, Fd = getattr(ops, name)(td, cd, Ud, secd, Yd, workspace=True, zero_z=True)
        res = getattr(ops, name + "_rev")(td, cd, Ud, secd, Yd, Zd, Fd, bZd)
        _lib.set_option("sweep_rev_lines", 0)
        try:
            ref = getattr(ops, name + "_rev")(td, cd, Ud, secd, Yd, Zd, Fd, bZd)
        finally:
            _lib.set_option("sweep_rev_lines", None)
        for r_, e_ in zip(res, ref):
            close(r_, e_.cpu().numpy())
        for b in sorted({0, B // 2, B - 1}):
            zo = np.empty((N, nrhs)); fo = np.empty((N, J, nrhs))
            getattr(oracle, name + "_fwd")(t[b], c[b], U[b], sec[b], Y[b], zo, fo)
            outs = [np.empty(N), np.empty(J), np.empty((N, J)), np.empty((N, J)), np.empty((N, nrhs))]
            getattr(oracle, name + "_rev")(t[b], c[b], U[b], sec[b], Y[b], zo, fo, bZ[b], *outs)
            for r_, e_ in zip(res, outs):
                close(r_[b], e_)


@pytest.mark.parametrize("nrhs", [1, 8])
@pytest.mark.parametrize("B,N", [(16, 66), (8, 40)])
def test_sweeps_by_lines_shared_grid_and_rates(ops, oracle, B, N, nrhs):
    """The line-pairing sweeps (J = 8; one and eight right-hand sides, forward and reverse) with the time grid and the
    rates shared by the batch (batch stride 0): against the oracle, every series."""
    J = 8
    rng = np.random.default_rng(911 * B + N + nrhs)
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    t0, c0 = t[0].copy(), c[0].copy()
    W = (0.3 / J) * rng.standard_normal((B, N, J))
    Y = rng.standard_normal((B, N, nrhs)); bZ = rng.standard_normal((B, N, nrhs))
    t0d, c0d, Ud, Vd, Wd, Yd, bZd = dev(t0, c0, U, V, W, Y, bZ)
    for name in ("solve_lower", "solve_upper", "matmul_lower", "matmul_upper"):
        solve = name.startswith("solve")
        sec, secd = (W, Wd) if solve else (V, Vd)
        Zd, Fd = getattr(ops, name)(t0d, c0d, Ud, secd, Yd, workspace=True, zero_z=True)
        Zp = getattr(ops, name)(t0d, c0d, Ud, secd, Yd) if solve else getattr(ops, name)(t0d, c0d, Ud, secd, Yd, zero_z=True)
        res = getattr(ops, name + "_rev")(t0d, c0d, Ud, secd, Yd, Zd, Fd, bZd)
        for b in range(B):
            zo = np.empty((N, nrhs)); fo = np.empty((N, J, nrhs))
            getattr(oracle, name + "_fwd")(t0, c0, U[b], sec[b], Y[b], zo, fo)
            close(Zd[b], zo); close(Fd[b], fo); close(Zp[b], zo)
            outs = [np.empty(N), np.empty(J), np.empty((N, J)), np.empty((N, J)), np.empty((N, nrhs))]
            getattr(oracle, name + "_rev")(t0, c0, U[b], sec[b], Y[b], zo, fo, bZ[b], *outs)
            for r_, e_ in zip(res, outs):
                close(r_[b], e_)


@pytest.mark.parametrize("tile", ["1", "0"])
@pytest.mark.parametrize("J,nrhs,N,M", [(8, 1, 97, 64), (3, 3, 40, 131), (6, 5, 200, 33), (16, 2, 50, 50), (2, 7, 1, 1)])
def test_general_matmul_batched(ops, oracle, monkeypatch, J, nrhs, N, M, tile):
    """general_matmul_lower/upper on the device, batched (tile = "1": a wavefront per series with lanes over rows,
    c2_general_tile.hip, where the shape fits it; "0": two-phase state sweep + one lane group per output row, or lanes
    over the right-hand sides): against the sequential-merge oracle with and without the F workspace, accumulation
    into Z, ties between the two grids, output rows entirely before / after the input grid, and rows of F the merge
    never visits (left untouched, forward.hpp:313/375)."""
    monkeypatch.setenv("C2_GENERAL_TILE", tile)
    B = 5
    rng = np.random.default_rng(100 * J + nrhs)
    Je = J if J % 2 == 0 else J + 1
    t2, c, a, Ue, Ve, y = dense.synthetic_batch(B, max(M, 2), Je)
    t2 = np.ascontiguousarray(t2[:, :M]); V = np.ascontiguousarray(Ve[:, :M, :J]); c = np.ascontiguousarray(c[:, :J])
    lo, hi = t2[:, :1], t2[:, -1:]
    t1 = np.sort(lo - 0.3 * (hi - lo + 1.0) + (1.6 * (hi - lo + 1.0)) * rng.random((B, N)), axis=1)
    if N > 4 and M > 4:
        t1[0, 3] = t2[0, 2]; t1[0, 4] = t2[0, 2]   # ties: t1 == t2, repeated
        t1[1, :] = np.sort(t2[1, 0] - 1.0 - rng.random(N))   # every output row before the input grid
        t1[2, :] = np.sort(t2[2, -1] + 0.5 + rng.random(N))  # every output row after it
        t1 = np.sort(t1, axis=1)
    U = rng.standard_normal((B, N, J))
    Y = rng.standard_normal((B, M, nrhs))
    t1d, t2d, cd, Ud, Vd, Yd = dev(t1, t2, c, U, V, Y)
    for name in ("general_matmul_lower", "general_matmul_upper"):
        Z0 = rng.standard_normal((B, N, nrhs))
        Zo = Z0.copy(); Fo = np.full((B, M, J, nrhs), -7.0)
        for b in range(B):
            getattr(oracle, name)(t1[b], t2[b], c[b], U[b], V[b], Y[b], Zo[b], Fo[b])
        (Zd,) = dev(Z0)
        (Fd,) = dev(np.full((B, M, J, nrhs), -7.0))
        Zd, Fd = getattr(ops, name)(t1d, t2d, cd, Ud, Vd, Yd, Z=Zd, F=Fd)
        close(Zd, Zo); close(Fd, Fo)
        (Zd2,) = dev(Z0)
        Zd2 = getattr(ops, name)(t1d, t2d, cd, Ud, Vd, Yd, Z=Zd2)   # internal temporary for the state rows
        close(Zd2, Zo)
        Zz = getattr(ops, name)(t1d, t2d, cd, Ud, Vd, Yd, zero_z=True)
        close(Zz, Zo - Z0)


@pytest.mark.parametrize("J,nrhs,N,M,Lc", [(8, 33, 50, 300, 64), (8, 64, 257, 1030, 128), (8, 100, 700, 701, 96), (4, 256, 64, 2100, 256),
                                           (12, 70, 300, 520, 8), (16, 64, 129, 1500, 200), (24, 40, 90, 400, 56), (8, 65, 1, 900, 128),
                                           (8, 64, 400, 2, 8), (8, 256, 256, 4096, None)])
def test_general_matmul_many_rhs_chunked_along_time(ops, oracle, monkeypatch, J, nrhs, N, M, Lc):
    """Round 6: general_matmul_* with many right-hand sides on a small batch cut into chunks of the t2 grid (c2_general.hip, CH: chunk
    sums, a diagonal chain, the event loop per chunk from its true start state over the outputs whose last absorbed row lies in the
    chunk).  Forced chunk lengths that do not divide the grid, chunks without any output, every output before / after the grid, ties
    between the grids ON chunk boundaries and runs of identical times across them, a grid of two rows, one output; the last case is
    the shape of the predictive covariance under the automatic plan -- against the sequential-merge oracle and against the unchunked
    kernel (C2_GENERAL_RHS_CHUNKS=0)."""
    B = 3
    rng = np.random.default_rng(31 * J + nrhs + N + M)
    Je = J if J % 2 == 0 else J + 1
    t2, c, a, Ue, Ve, y = dense.synthetic_batch(B, max(M, 2), Je)
    t2 = np.ascontiguousarray(t2[:, :M]); V = np.ascontiguousarray(Ve[:, :M, :J]); c = np.ascontiguousarray(c[:, :J])
    lo, hi = t2[:, :1], t2[:, -1:]
    t1 = np.sort(lo - 0.2 * (hi - lo + 1.0) + (1.4 * (hi - lo + 1.0)) * rng.random((B, N)), axis=1)
    L = Lc or 256
    if N > 8 and M > 2 * L + 2:
        # outputs exactly ON the rows around the first two chunk boundaries (positions 1 + k L), repeated; a run of equal t2 across one
        t2[0, L + 1] = t2[0, L]; t2[0, L + 2] = t2[0, L]
        t1[0, :6] = [t2[0, L - 1], t2[0, L], t2[0, L], t2[0, L + 1], t2[0, 2 * L], t2[0, 2 * L + 1]]
        t1[1, :] = np.sort(t2[1, 0] - 1.0 - rng.random(N))    # every output before the grid
        t1[2, :N // 2] = np.sort(t2[2, -1] + 0.5 + rng.random(N // 2))   # half of them after it, none inside the middle chunks
        t1[2, N // 2:] = t2[2, 0] + 1e-3 * rng.random(N - N // 2)
        t1 = np.sort(t1, axis=1)
    U = rng.standard_normal((B, N, J))
    Y = rng.standard_normal((B, M, nrhs))
    t1d, t2d, cd, Ud, Vd, Yd = dev(t1, t2, c, U, V, Y)
    for name in ("general_matmul_lower", "general_matmul_upper"):
        Z0 = rng.standard_normal((B, N, nrhs))
        Zo = Z0.copy(); Fo = np.zeros((B, M, J, nrhs))
        for b in range(B):
            getattr(oracle, name)(t1[b], t2[b], c[b], U[b], V[b], Y[b], Zo[b], Fo[b])
        if Lc is None: monkeypatch.delenv("C2_GENERAL_RHS_CHUNKS", raising=False)
        else: monkeypatch.setenv("C2_GENERAL_RHS_CHUNKS", str(Lc))
        (Zd,) = dev(Z0)
        Zd = getattr(ops, name)(t1d, t2d, cd, Ud, Vd, Yd, Z=Zd)
        close(Zd, Zo)
        monkeypatch.setenv("C2_GENERAL_RHS_CHUNKS", "0")
        (Zu,) = dev(Z0)
        Zu = getattr(ops, name)(t1d, t2d, cd, Ud, Vd, Yd, Z=Zu)
        close(Zu, Zo)
        close(Zd, Zu.cpu().numpy(), tol=1e-12, floor=1e-13)


@pytest.mark.parametrize("J,nrhs,N,M", [(8, 1, 64, 64), (8, 1, 128, 129), (8, 1, 1000, 300), (8, 1, 70, 2000), (4, 2, 513, 511),
                                        (8, 4, 300, 257), (5, 3, 190, 640), (16, 1, 200, 200), (11, 2, 130, 65),
                                        (1, 1, 65, 1), (2, 4, 1, 200), (8, 1, 4096, 4096), (8, 1, 4500, 5000),
                                        (4, 2, 3000, 2600), (8, 4, 2100, 2049), (16, 1, 900, 2300), (6, 1, 9000, 70000),
                                        (8, 8, 300, 257), (7, 5, 2500, 2300), (16, 5, 400, 2100), (3, 11, 130, 70)])
@pytest.mark.parametrize("chunks", ["1", "0"])
def test_general_matmul_row_tiles(ops, oracle, monkeypatch, J, nrhs, N, M, chunks):
    """c2_general_tile.hip (a wavefront per series, 64 rows of either grid per pass) on shapes around its tile sizes:
    grids of very different density (many output tiles per state tile and the reverse), ties that fall on tile
    boundaries, outputs before / after / between the input rows only, runs of identical times, a series whose outputs
    stop early (F rows beyond stay untouched) -- Z and the F workspace against the sequential-merge oracle.  Series of
    2048 rows and more in a small batch are cut into chunks along time (chunks = "0": one wavefront per series); more
    right-hand sides than one tile holds (4; 2 at widths above 8) run as several tiles on small batches."""
    if chunks == "0" and M < 2048:
        pytest.skip("not a chunked shape")
    monkeypatch.setenv("C2_GENERAL_CHUNKS", chunks)
    B = 6
    rng = np.random.default_rng(7 * J + 13 * nrhs + N + M)
    Je = J if J % 2 == 0 else J + 1
    t2, c, a, Ue, Ve, y = dense.synthetic_batch(B, max(M, 2), Je)
    t2 = np.ascontiguousarray(t2[:, :M]); V = np.ascontiguousarray(Ve[:, :M, :J]); c = np.ascontiguousarray(c[:, :J])
    lo, hi = t2[:, :1], t2[:, -1:]
    t1 = np.sort(lo - 0.1 * (hi - lo + 1.0) + (1.2 * (hi - lo + 1.0)) * rng.random((B, N)), axis=1)
    if M > 130 and N > 8:
        t1[0, :8] = t2[0, [63, 63, 64, 64, 127, 128, 128, 129]]      # ties on the tile boundaries
        t1[0] = np.sort(t1[0])
    if N > 3:
        t1[1] = np.sort(t2[1, min(M - 1, 70)] + 1e-3 * rng.random(N))       # all outputs behind one input row
        t1[2] = np.sort(lo[2] - 1.0 + (0.5 * (hi[2] - lo[2]) + 1.0) * rng.random(N))   # outputs stop half-way
        t1[3, N // 2:] = t1[3, N // 2]                                       # a run of identical output times
    if M > 3:
        t2[4, M // 3: M // 3 + 3] = t2[4, M // 3]                            # identical input times
    U = rng.standard_normal((B, N, J))
    Y = rng.standard_normal((B, M, nrhs))
    t1d, t2d, cd, Ud, Vd, Yd = dev(t1, t2, c, U, V, Y)
    for name in ("general_matmul_lower", "general_matmul_upper"):
        Z0 = rng.standard_normal((B, N, nrhs))
        Zo = Z0.copy(); Fo = np.full((B, M, J, nrhs), -7.0)
        for b in range(B):
            getattr(oracle, name)(t1[b], t2[b], c[b], U[b], V[b], Y[b], Zo[b], Fo[b])
        (Zd,) = dev(Z0)
        (Fd,) = dev(np.full((B, M, J, nrhs), -7.0))
        Zd, Fd = getattr(ops, name)(t1d, t2d, cd, Ud, Vd, Yd, Z=Zd, F=Fd)
        close(Zd, Zo); close(Fd, Fo)
        (Zd2,) = dev(Z0)
        close(getattr(ops, name)(t1d, t2d, cd, Ud, Vd, Yd, Z=Zd2), Zo)


def test_torch_autograd_adapter(ops, oracle):
    """autograd.log_likelihood: values and gradients (incl. a shared time grid and shared c) vs the oracle."""
    import torch
    from celerite2_amd import autograd as ag

    B, N, J = 6, 150, 4
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    t0, c0 = t[0].copy(), c[0].copy()
    # per-series t and c
    td, cd, ad, Ud, Vd, yd = [x.requires_grad_(True) for x in dev(t, c, a, U, V, y)]
    ll = ag.log_likelihood(td, cd, ad, Ud, Vd, yd)
    wts = torch.linspace(0.5, 1.5, B, dtype=torch.float64, device="cuda")
    (ll * wts).sum().backward()
    llo, go, _ = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=2)
    close(ll.detach(), llo)
    w = wts.cpu().numpy()
    for x, e in zip((td, cd, ad, Ud, Vd, yd), go):
        close(x.grad, e * w.reshape((B,) + (1,) * (e.ndim - 1)))
    # shared t and c (one light-curve grid, one kernel, different noise / data): gradients are summed over the batch
    rng = np.random.default_rng(2)
    co = dense.sho_sum_coeffs(J, 0.0)
    for b in range(B):
        c0, a[b], U[b], V[b] = dense.celerite_matrices(co, t0, rng.uniform(0.1, 0.3, N))
        y[b] = np.sin(t0) + 0.1 * rng.standard_normal(N)
    td, cd = [x.requires_grad_(True) for x in dev(t0, c0)]
    ad, Ud, Vd, yd = dev(a, U, V, y)
    ll = ag.log_likelihood(td, cd, ad, Ud, Vd, yd)
    ll.sum().backward()
    gt = np.zeros(N); gc = np.zeros(J)
    for b in range(B):
        l1, g1, f1 = oracle.loglik_grad(t0, c0, a[b], U[b], V[b], y[b])
        assert f1 == 0
        gt += g1[0]; gc += g1[1]
    close(td.grad, gt); close(cd.grad, gc)
    # no-grad call takes the forward-only kernel
    with torch.no_grad():
        ll2 = ag.log_likelihood(td, cd, ad, Ud, Vd, yd)
    close(ll2, ll.detach().cpu().numpy())


def test_torch_autograd_ops(ops, oracle):
    """The five differentiable ops as torch.autograd Functions: a log-likelihood composed from them
    (factor -> solve_lower -> reductions, numpy.py:84-87) has the value and gradients of the fused kernel, and
    matmul / solve_upper pass a directional finite-difference check."""
    import torch
    from celerite2_amd import autograd as ag

    B, N, J = 4, 60, 4
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    leaves = [x.requires_grad_(True) for x in dev(t, c, a, U, V, y)]
    td, cd, ad, Ud, Vd, yd = leaves
    d, W = ag.factor(td, cd, ad, Ud, Vd)
    z = ag.solve_lower(td, cd, Ud, W, yd[:, :, None])[:, :, 0]
    ll = -0.5 * (torch.log(d).sum(1) + N * np.log(2 * np.pi)) - 0.5 * (z * z / d).sum(1)
    wts = torch.linspace(0.5, 1.5, B, dtype=torch.float64, device="cuda")
    (ll * wts).sum().backward()
    llo, go, _ = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=2)
    close(ll.detach(), llo)
    w = wts.cpu().numpy()
    for x, e in zip(leaves, go):
        close(x.grad, e * w.reshape((B,) + (1,) * (e.ndim - 1)))
    # directional finite differences through matmul_lower / matmul_upper / solve_upper, 3 right-hand sides
    rng = np.random.default_rng(4)
    Y = rng.standard_normal((B, N, 3))
    for name in ("matmul_lower", "matmul_upper", "solve_upper"):
        base = [x.detach().clone() for x in dev(t, c, U, 0.05 * V, Y)]
        dirs = [torch.from_numpy(rng.standard_normal(x.shape)).cuda() for x in base]
        f = lambda xs: (getattr(ag, name)(*xs) ** 2).sum()
        xs = [x.clone().requires_grad_(True) for x in base]
        f(xs).backward()
        lin = sum(float((x.grad * dv).sum()) for x, dv in zip(xs, dirs))
        eps = 1e-6
        fp = f([x + eps * dv for x, dv in zip(base, dirs)]); fm = f([x - eps * dv for x, dv in zip(base, dirs)])
        fd = float(fp - fm) / (2 * eps)
        assert abs(fd - lin) <= 1e-6 * max(1.0, abs(lin)), (name, fd, lin)
    # a non-positive-definite series raises like the reference
    a2 = a.copy(); a2[1, 7] = -4.0
    with pytest.raises(ag.LinAlgError):
        ag.factor(*dev(t, c, a2, U, V))


def test_million_row_series(ops, oracle):
    """N = 1.2e6 rows per series (64-bit row offsets everywhere, 150 000 checkpoint segments): fused log-lik + gradient,
    single- and multi-rhs solves with their reverse, factor_rev from sparse checkpoints and the two-phase prediction
    product against the oracle on one of the series."""
    B, N, J = 3, 1_200_000, 4
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    td, cd, ad, Ud, Vd, yd = dev(t, c, a, U, V, y)
    b = 1
    llo, go, flago = oracle.loglik_grad(t[b], c[b], a[b], U[b], V[b], y[b])
    assert flago == 0
    ll, grads, flag = ops.loglik_grad(td, cd, ad, Ud, Vd, yd)
    assert int(flag.abs().sum()) == 0
    close(ll[b:b + 1], np.array([llo]))
    for g, e in zip(grads, go):
        close(g[b], e)
    d, W, S, _ = ops.factor(td, cd, ad, Ud, Vd, workspace=True)
    do = np.empty(N); Wo = np.empty((N, J)); So = np.empty((N, J, J))
    assert oracle.factor_flag(t[b], c[b], a[b], U[b], V[b], do, Wo, So) == 0
    close(d[b], do); close(W[b], Wo); close(S[b], So)
    rng = np.random.default_rng(8)
    bd = rng.standard_normal((B, N)); bW = rng.standard_normal((B, N, J))
    res = ops.factor_rev(td, cd, ad, Ud, Vd, d, W, S, *dev(bd, bW))
    outs = [np.empty(N), np.empty(J), np.empty(N), np.empty((N, J)), np.empty((N, J))]
    oracle.factor_rev(t[b], c[b], a[b], U[b], V[b], do, Wo, So, bd[b], bW[b], *outs)
    for r_, e_ in zip(res, outs):
        close(r_[b], e_)
    for nrhs in (1, 6):
        Y = rng.standard_normal((B, N, nrhs)); (Yd,) = dev(Y)
        Zd, Fd = ops.solve_lower(td, cd, Ud, W, Yd, workspace=True)
        Zo = np.empty((N, nrhs)); Fo = np.empty((N, J, nrhs))
        oracle.solve_lower_fwd(t[b], c[b], U[b], Wo, Y[b], Zo, Fo)
        close(Zd[b], Zo)
        close(ops.solve_upper(td, cd, Ud, W, Yd)[b], oracle.solve_upper(t[b], c[b], U[b], Wo, Y[b], np.empty((N, nrhs))))
        bZ = rng.standard_normal((B, N, nrhs)); (bZd,) = dev(bZ)
        res = ops.solve_lower_rev(td, cd, Ud, W, Yd, Zd, Fd, bZd)
        outs = [np.empty(N), np.empty(J), np.empty((N, J)), np.empty((N, J)), np.empty((N, nrhs))]
        oracle.solve_lower_rev(t[b], c[b], U[b], Wo, Y[b], Zo, Fo, bZ[b], *outs)
        for r_, e_ in zip(res, outs):
            close(r_[b], e_)
    M = 900_001
    t2 = np.ascontiguousarray(t[:, :M]); V2 = np.ascontiguousarray(V[:, :M]); Y2 = rng.standard_normal((B, M, 1))
    Zg = ops.general_matmul_lower(td, *dev(t2), cd, Ud, *dev(V2, Y2))
    zo = np.zeros((N, 1))
    oracle.general_matmul_lower(t[b], t2[b], c[b], U[b], V2[b], Y2[b], zo)
    close(Zg[b], zo)


def test_hot_path_is_graph_capturable(ops, oracle, monkeypatch):
    """The fused gradient is stream-ordered end to end -- no host round trip, no allocation with caller-provided
    workspace / outputs, the choice between the backward-recursion sweep and the replay kernels made on the device --
    so it can be captured once in a HIP graph and replayed on new data (the group, one-lane and two-lane mappings, and a
    batch that trips their stability gates on replay)."""
    import torch
    J = 8
    for lanes, B, N in (("8", 9, 200), ("1", 70, 200), ("2", 70, 200)):
        monkeypatch.setenv("C2_LANES", lanes)
        t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
        td, cd, ad, Ud, Vd, yd = dev(t, c, a, U, V, y)
        work = ops.loglik_grad_workspace(B, N, J, td.device)
        ll, out, flag = ops.loglik_grad(td, cd, ad, Ud, Vd, yd, work=work)      # warm-up outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ll_g, out_g, flag_g = ops.loglik_grad(td, cd, ad, Ud, Vd, yd, work=work, out=out)
        for variant in range(3):
            y2 = y + 0.01 * variant
            t2 = t.copy()
            if variant == 2:
                t2[:, N // 2:] += 500.0     # beyond the guard of the one-lane sweep: the gated replay kernels answer
            yd.copy_(torch.from_numpy(y2)); td.copy_(torch.from_numpy(t2))
            g.replay()
            torch.cuda.synchronize()
            llo, go, flo = oracle.loglik_grad_batched(t2, c, a, U, V, y2, nthreads=2)
            assert flag_g.cpu().tolist() == list(flo)
            close(ll_g, llo)
            for x, e in zip(out_g, go):
                close(x, e)


def test_loglik_grad_buffers_placement_search(ops, oracle):
    """ops.loglik_grad_buffers: workspace + gradient arrays chosen among a few placements by timing one step each; the
    buffers it returns are ordinary `work=` / `out=` arguments."""
    B, N, J = 70, 300, 8
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    args = dev(t, c, a, U, V, y)
    work, out, report = ops.loglik_grad_buffers(*args, candidates=3)
    assert len(report["candidates"]) == 3 and report["chosen_ms"] == min(x["ms"] for x in report["candidates"])
    ll, grads, flag = ops.loglik_grad(*args, work=work, out=out)
    assert all(g is o for g, o in zip(grads, out)) and int(flag.abs().sum()) == 0
    llo, go, _ = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=2)
    close(ll, llo)
    for g, e in zip(grads, go):
        close(g, e)
