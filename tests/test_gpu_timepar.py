# -*- coding: utf-8 -*-
"""Time-parallel forms of c2_timepar.hip against the CPU oracle: the forward log-likelihood (widths 8, 4, 2) and `factor`
(widths 4, 2) on chunk ELEMENTS combined in a tree / a scan, the single-rhs solves as chunked affine maps, with the row-by-row
kernels as the stream-ordered fallback behind a device-side gate (failed factorisations).  Forced with C2_TIMEPAR=1 (the
dispatch takes them by a measured cost model)."""
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


def close(a, b, tol=1e-10):
    a = a.cpu().numpy() if hasattr(a, "cpu") else a
    np.testing.assert_allclose(a, b, rtol=tol, atol=0.0)


def wide_batch(B, N, J):
    """A positive definite batch of any width: an odd width drops the last column of the next even one and lifts the
    diagonal (the dropped term's variance becomes white noise)."""
    Je = J + (J % 2)
    t, c, a, U, V, y = dense.synthetic_batch(B, max(N, 2), Je)
    t, a, y = (np.ascontiguousarray(v[:, :N]) for v in (t, a, y))
    U, V = np.ascontiguousarray(U[:, :N, :J]), np.ascontiguousarray(V[:, :N, :J])
    c = np.ascontiguousarray(c[:, :J])
    if Je != J:
        a = a + 1.0
    return t, c, a, U, V, y


def oracle_ll(oracle, t, c, a, U, V, y):
    ll, _, fl = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=2)
    return ll, np.asarray(fl)


@pytest.mark.parametrize("J", [8, 4, 2])
@pytest.mark.parametrize("B,N", [(1, 1), (2, 2), (3, 63), (5, 64), (4, 65), (3, 128), (7, 700), (2, 4096), (70, 1000), (1, 20000),
                                 (2, 4097), (1, 300000)])
def test_timepar_matches_oracle(ops, oracle, monkeypatch, B, N, J):
    """Chunk elements in scattering form combined in a tree: one wavefront finishes a series of up to 4096 rows, k_tp_join
    beyond (4097 rows = two wavefronts, 300000 = 74: a second round of the join)."""
    t, c, a, U, V, y = dense.synthetic_batch(B, max(N, 2), J)
    t, a, U, V, y = (np.ascontiguousarray(v[:, :N]) for v in (t, a, U, V, y))
    llo, flo = oracle_ll(oracle, t, c, a, U, V, y)
    assert not flo.any()
    monkeypatch.setenv("C2_TIMEPAR", "1")
    ll, flag = ops.loglik(*dev(t, c, a, U, V, y))
    assert int(flag.abs().sum()) == 0
    close(ll, llo)
    monkeypatch.setenv("C2_TIMEPAR", "0")
    ll0, _ = ops.loglik(*dev(t, c, a, U, V, y))
    close(ll, ll0.cpu().numpy(), tol=1e-12)   # (nothing ill-conditioned in the tree)
    # shared time grid and rates
    ts, cs = np.tile(t[0], (B, 1)), np.tile(c[0], (B, 1))
    lls, fls = oracle_ll(oracle, ts, cs, a, U, V, y)
    monkeypatch.setenv("C2_TIMEPAR", "1")
    ll2, flag2 = ops.loglik(*dev(t[0].copy(), c[0].copy(), a, U, V, y))
    ok = fls == 0
    assert np.array_equal(flag2.cpu().numpy() != 0, ~ok)
    close(ll2[ok], lls[ok])


@pytest.mark.parametrize("B,N", [(3, 140), (3, 270), (1, 390), (3, 390), (600, 390), (2, 520), (300, 912), (3, 1600), (2, 8200),
                                 (2, 147000)])
def test_timepar_width8_tree_spans_that_are_not_powers_of_two(ops, oracle, monkeypatch, B, N):
    """k_e8_tree with K = 9, 17, 25, 33, 57, 100 chunk elements per workgroup (and Kin = 9 / 17 in a later launch: 8200 rows =
    513 chunks -> 9 results of 64, 147000 rows): the levels of such a span need MORE than `span` records (25 -> 13 + 7 + 4 + 2).
    Round 5 sized the level scratch by `span` and put the final record at span - 1: a write into the next workgroup's region
    (a race once the batch exceeds the resident workgroups: 600 series) or past the buffer.  Every series
    against the oracle and against the row-by-row kernel, twice (a stale neighbour would differ between runs)."""
    t, c, a, U, V, y = dense.synthetic_batch(min(B, 6), N, 8)
    if B > 6:   # distinct data in every series without 600 calls of the generator: shifted copies of y, scaled a
        reps = (B + 5) // 6
        rng = np.random.default_rng(B)
        t, c, U, V = (np.ascontiguousarray(np.tile(v, (reps,) + (1,) * (v.ndim - 1))[:B]) for v in (t, c, U, V))
        a = np.ascontiguousarray(np.tile(a, (reps, 1))[:B] * rng.uniform(1.0, 1.3, (B, 1)))
        y = np.ascontiguousarray(np.tile(y, (reps, 1))[:B] + 0.05 * rng.standard_normal((B, N)))
    llo, flo = oracle_ll(oracle, t, c, a, U, V, y)
    assert not flo.any()
    d = dev(t, c, a, U, V, y)
    monkeypatch.setenv("C2_TIMEPAR", "1")
    for _ in range(2):
        ll, flag = ops.loglik(*d)
        assert int(flag.abs().sum()) == 0
        close(ll, llo)
    monkeypatch.setenv("C2_TIMEPAR", "0")
    ll0, _ = ops.loglik(*d)
    close(ll, ll0.cpu().numpy(), tol=1e-12)


@pytest.mark.parametrize("J", [8, 4, 2])
def test_timepar_falls_back_when_it_cannot_be_trusted(ops, oracle, monkeypatch, J):
    """Failed factorisations, zero white noise (kappa = 0: the maps are singular), gaps long enough to underflow a decay:
    the verification word sends the batch to the row-by-row kernel, which reports what the reference reports."""
    B, N = 6, 900
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    monkeypatch.setenv("C2_TIMEPAR", "1")
    # (1) a series that is not positive definite
    a1 = a.copy(); a1[2, 500] = -3.0
    llo, flo = oracle_ll(oracle, t, c, a1, U, V, y)
    ll, flag = ops.loglik(*dev(t, c, a1, U, V, y))
    assert flag.cpu().tolist() == list(flo) and flo[2] == 500
    ok = flo == 0
    close(ll[ok], llo[ok]); assert np.isneginf(float(ll[2]))
    # (2) no white noise on a stretch of rows: a = sum(U V) exactly
    # (numerically singular for that series: its value is noise in any implementation -- what is checked is that the
    # batch comes back from the row-by-row kernel, bit for bit, and that the healthy series match the oracle)
    a2 = a.copy(); a2[1, 100:140] = (U[1, 100:140] * V[1, 100:140]).sum(-1)
    llo, flo = oracle_ll(oracle, t, c, a2, U, V, y)
    ll, flag = ops.loglik(*dev(t, c, a2, U, V, y))
    monkeypatch.setenv("C2_TIMEPAR", "0")
    ll_rows, flag_rows = ops.loglik(*dev(t, c, a2, U, V, y))
    monkeypatch.setenv("C2_TIMEPAR", "1")
    # nothing is singular for the scattering form (d from the zero state = a_n > 0): no fallback unless a pivot fails
    keep = np.arange(B) != 1
    assert np.array_equal(flag.cpu().numpy()[keep], flag_rows.cpu().numpy()[keep])
    ok = (flo == 0) & (np.arange(B) != 1)
    close(ll[ok], llo[ok])
    # (3) a gap of 1e6 time units inside a chunk: exp(-c dt) underflows, its reciprocal overflows
    t3 = t.copy(); t3[:, 450:] += 1e6
    llo, flo = oracle_ll(oracle, t3, c, a, U, V, y)
    ll, flag = ops.loglik(*dev(t3, c, a, U, V, y))
    assert flag.cpu().tolist() == list(flo)
    close(ll, llo)


def test_timepar_is_what_small_batches_of_long_series_run(ops, oracle, monkeypatch):
    """BASELINE configs[1] (1024 series, N = 4096, J = 4, forward): default dispatch, oracle on a few series."""
    import torch
    from celerite2_amd import synth
    monkeypatch.delenv("C2_TIMEPAR", raising=False)
    B, N, J = 1024, 4096, 4
    t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, torch.device("cuda:0"))
    ll, flag = ops.loglik(t, c, a, U, V, y)
    assert int(flag.abs().sum()) == 0
    idx = [0, 1, 511, 1023]
    h = [x[idx].cpu().numpy() for x in (t, c, a, U, V, y)]
    llo, flo = oracle_ll(oracle, *h)
    close(ll[idx], llo)
    monkeypatch.setenv("C2_TIMEPAR", "0")
    ll0, _ = ops.loglik(t, c, a, U, V, y)
    close(ll, ll0.cpu().numpy())


@pytest.mark.parametrize("J", [4, 2])
@pytest.mark.parametrize("B,N", [(1, 1), (2, 2), (3, 63), (5, 64), (4, 65), (3, 129), (2, 4096), (9, 1000), (1, 30000)])
def test_timepar_factor_matches_oracle(ops, oracle, monkeypatch, B, N, J):
    """`factor` (d, W) by the same composition: every row against the oracle; a failed series falls back and reports the
    reference's flag; in-place calls (d is a, W is V) stay on the row-by-row kernel and still agree."""
    t, c, a, U, V, y = dense.synthetic_batch(B, max(N, 2), J)
    t, a, U, V = (np.ascontiguousarray(v[:, :N]) for v in (t, a, U, V))
    monkeypatch.setenv("C2_TIMEPAR", "1")
    d, W, flag = ops.factor(*dev(t, c, a, U, V))
    assert int(flag.abs().sum()) == 0
    for b in range(B):
        do = np.empty(N); Wo = np.empty((N, J)); So = np.empty((N, J, J))
        assert oracle.factor_flag(t[b], c[b], a[b], U[b], V[b], do, Wo, So) == 0
        close(d[b], do); np.testing.assert_allclose(W[b].cpu().numpy(), Wo, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(Wo).max()))
    if N > 40:
        a1 = a.copy(); a1[0, N // 2] = -2.0
        d1, W1, flag1 = ops.factor(*dev(t, c, a1, U, V))
        assert int(flag1[0]) == N // 2 and int(flag1[1:].abs().sum()) == 0
        monkeypatch.setenv("C2_TIMEPAR", "0")
        d0, W0, flag0 = ops.factor(*dev(t, c, a1, U, V))
        assert np.array_equal(d1.cpu().numpy()[1:], d0.cpu().numpy()[1:]) if B > 1 else True
        monkeypatch.setenv("C2_TIMEPAR", "1")
    ad, Vd = dev(a, V)
    td, cd, Ud = dev(t, c, U)
    d2, W2, _ = ops.factor(td, cd, ad, Ud, Vd, d=ad, W=Vd)   # in place
    close(d2, d.cpu().numpy()); np.testing.assert_allclose(W2.cpu().numpy(), W.cpu().numpy(), rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("J", [8, 7, 6, 4, 3, 2])
@pytest.mark.parametrize("B,N", [(1, 1), (2, 2), (3, 63), (5, 64), (4, 65), (3, 131), (2, 4096), (9, 1000), (1, 30001), (2, 4100),
                                 (3, 16500)])
def test_timepar_solves_match_oracle(ops, oracle, monkeypatch, B, N, J):
    """solve_lower / solve_upper with one right-hand side as chunked affine maps (k_tps_* at widths 8, 4, 2; from 16384
    rows the chunk maps of c2_timepar_grad.hip with their two-level chain, at every width up to 8): every row against the
    oracle, out of place and in place, per-series and shared time grids."""
    t, c, a, U, V, y = wide_batch(B, N, J)
    W = np.empty_like(V); d = np.empty_like(a)
    for b in range(B):
        assert oracle.factor_flag(t[b], c[b], a[b], U[b], V[b], d[b], W[b]) == 0
    Y = y[:, :, None].copy()
    monkeypatch.setenv("C2_TIMEPAR", "1")
    td, cd, Ud, Wd, Yd = dev(t, c, U, W, Y)
    for name in ("solve_lower", "solve_upper"):
        want = np.empty_like(Y)
        for b in range(B):
            zb = Y[b].copy()
            getattr(oracle, name)(t[b], c[b], U[b], W[b], Y[b], zb)
            want[b] = zb
        got = getattr(ops, name)(td, cd, Ud, Wd, Yd)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(want).max()))
        Zin = Yd.clone()
        got2 = getattr(ops, name)(td, cd, Ud, Wd, Zin, Z=Zin)      # in place
        np.testing.assert_allclose(got2.cpu().numpy(), want, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(want).max()))
        monkeypatch.setenv("C2_TIMEPAR", "0")
        rows = getattr(ops, name)(td, cd, Ud, Wd, Yd)
        monkeypatch.setenv("C2_TIMEPAR", "1")
        np.testing.assert_allclose(got.cpu().numpy(), rows.cpu().numpy(), rtol=1e-10, atol=1e-12 * max(1.0, np.abs(want).max()))
    # shared time grid and rates (the factorisation of series 0 applied to every right-hand side)
    t0d, c0d = dev(t[0].copy(), c[0].copy())
    U0 = np.tile(U[0], (B, 1, 1)); W0 = np.tile(W[0], (B, 1, 1))
    U0d, W0d = dev(U0, W0)
    want = np.empty_like(Y)
    for b in range(B):
        zb = Y[b].copy(); oracle.solve_lower(t[0], c[0], U[0], W[0], Y[b], zb); want[b] = zb
    got = ops.solve_lower(t0d, c0d, U0d, W0d, Yd)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(want).max()))


@pytest.mark.parametrize("rows", [None, "32", "64"])
@pytest.mark.parametrize("B,N,J", [(1, 2048, 8), (2, 5000, 7), (3, 2100, 6), (1, 9000, 5), (2, 4097, 4), (1, 2500, 3),
                                   (2, 3000, 2), (1, 2049, 1), (1, 20000, 8), (1, 9000, 4), (1, 512, 8), (40, 1000, 6),
                                   (3, 700, 3)])
def test_factor_with_s_workspace_parallel_along_time(ops, oracle, monkeypatch, B, N, J, rows):
    """factor WITH the S workspace of the drop-in on a small batch of long series: d, W by the Newton iterations, the S
    rows (half-decayed states, forward.hpp:115-123) by chunks (k_s_rows) -- every element against the oracle and against
    the row-by-row kernels (C2_FACTOR_ITER=0); a failed series keeps its flag."""
    t, c, a, U, V, y = wide_batch(B, N, J)
    if B > 1:
        a[B - 1, N // 2] = -1.0
    if rows: monkeypatch.setenv("C2_TPG_ROWS", rows)
    args = dev(t, c, a, U, V)
    monkeypatch.delenv("C2_FACTOR_ITER", raising=False)
    d, W, S, flag = ops.factor(*args, workspace=True)
    monkeypatch.setenv("C2_FACTOR_ITER", "0")
    d0, W0, S0, flag0 = ops.factor(*args, workspace=True)
    fl = flag.cpu().numpy()
    assert np.array_equal(fl != 0, flag0.cpu().numpy() != 0)
    for b in range(B):
        do = np.empty(N); Wo = np.empty((N, J)); So = np.empty((N, J * J))
        bad = oracle.factor_flag(t[b], c[b], a[b], U[b], V[b], do, Wo, So) != 0
        assert bad == (fl[b] != 0)
        if bad:
            continue
        close(d[b], do)
        np.testing.assert_allclose(W[b].cpu().numpy(), Wo, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(Wo).max()))
        tol = dict(rtol=1e-10, atol=1e-12 * max(1.0, np.abs(So).max()))
        np.testing.assert_allclose(S[b].cpu().numpy().reshape(N, J * J), So, **tol)
        np.testing.assert_allclose(S[b].cpu().numpy(), S0[b].cpu().numpy(), **tol)


@pytest.mark.parametrize("rows", [None, "32", "64"])
@pytest.mark.parametrize("B,N,J", [(1, 2048, 8), (2, 5000, 7), (3, 2100, 6), (1, 9000, 5), (2, 4097, 4), (1, 2500, 3),
                                   (2, 3000, 2), (1, 2049, 1), (1, 20000, 8), (1, 9000, 4), (1, 512, 8), (40, 1000, 6),
                                   (3, 700, 3)])
def test_factor_rev_parallel_along_time(ops, oracle, monkeypatch, B, N, J, rows):
    """factor_rev (reverse.hpp:26-85) on a small batch of long series: the reverse pass of the time-parallel gradient with
    the adjoints of d and W handed in -- all five outputs against the oracle and against the row-by-row kernel
    (C2_TIMEPAR_GRAD=0), each relative to its largest entry; shared time grid and rates as well."""
    t, c, a, U, V, y = wide_batch(B, N, J)
    rng = np.random.default_rng(3 * N + J)
    d = np.empty_like(a); W = np.empty_like(V); S = np.empty((B, N, J * J))
    for b in range(B):
        assert oracle.factor_flag(t[b], c[b], a[b], U[b], V[b], d[b], W[b], S[b]) == 0
    bd = rng.standard_normal((B, N)); bW = rng.standard_normal((B, N, J))
    want = [np.empty((B, N)), np.empty((B, J)), np.empty((B, N)), np.empty((B, N, J)), np.empty((B, N, J))]
    for b in range(B):
        outs = [np.zeros(N), np.zeros(J), np.zeros(N), np.zeros((N, J)), np.zeros((N, J))]
        oracle.factor_rev(t[b], c[b], a[b], U[b], V[b], d[b], W[b], S[b], bd[b], bW[b], *outs)
        for w, o in zip(want, outs):
            w[b] = o
    if rows: monkeypatch.setenv("C2_TPG_ROWS", rows)
    args = dev(t, c, a, U, V, d, W, S.reshape(B, N, J, J), bd, bW)
    monkeypatch.delenv("C2_TIMEPAR_GRAD", raising=False)
    got = ops.factor_rev(*args)
    monkeypatch.setenv("C2_TIMEPAR_GRAD", "0")
    rowsgot = ops.factor_rev(*args)
    monkeypatch.delenv("C2_TIMEPAR_GRAD", raising=False)
    for g, r, w in zip(got, rowsgot, want):
        gclose(g, w)
        gclose(g, r.cpu().numpy())
    if B > 1:   # one time grid and one set of rates for the whole batch (the problem of series 0, other adjoints)
        rep = lambda x: np.ascontiguousarray(np.tile(x[0], (B,) + (1,) * (x.ndim - 1)))
        a0, U0, V0, d0, W0, S0 = rep(a), rep(U), rep(V), rep(d), rep(W), rep(S)
        for b in range(B):
            outs = [np.zeros(N), np.zeros(J), np.zeros(N), np.zeros((N, J)), np.zeros((N, J))]
            oracle.factor_rev(t[0], c[0], a0[b], U0[b], V0[b], d0[b], W0[b], S0[b], bd[b], bW[b], *outs)
            for w, o in zip(want, outs):
                w[b] = o
        args = dev(t[0].copy(), c[0].copy(), a0, U0, V0, d0, W0, S0.reshape(B, N, J, J), bd, bW)
        got = ops.factor_rev(*args)
        for g, w in zip(got, want):      # (bt, bc stay per series)
            gclose(g, w)


@pytest.mark.parametrize("B,N,J,nrhs", [(1, 1024, 8, 1), (3, 4100, 6, 5), (100, 2048, 4, 2), (2, 9000, 16, 8), (1, 20000, 3, 3)])
def test_chunked_products_on_mid_length_series(ops, oracle, B, N, J, nrhs):
    """matmul_lower / matmul_upper on small batches of series from 1024 rows (c2_scan.hip with chunks of ~sqrt(N) / 2
    rows): Z (accumulated into, and zeroed first) and the F workspace against the oracle."""
    t, c, a, U, V, y = wide_batch(B, N, J) if J <= 8 else dense.synthetic_batch(B, N, J)
    rng = np.random.default_rng(N + J)
    Y = rng.standard_normal((B, N, nrhs)); Z0 = rng.standard_normal((B, N, nrhs))
    td, cd, Ud, Vd, Yd = dev(t, c, U, V, Y)
    for name in ("matmul_lower", "matmul_upper"):
        want = Z0.copy(); wantF = np.empty((B, N, J * nrhs))
        for b in range(B):
            getattr(oracle, name)(t[b], c[b], U[b], V[b], Y[b], want[b], wantF[b])
        (Zd,) = dev(Z0)
        got, F = getattr(ops, name)(td, cd, Ud, Vd, Yd, Z=Zd, workspace=True)
        tol = dict(rtol=1e-10, atol=1e-12 * max(1.0, np.abs(want).max()))
        np.testing.assert_allclose(got.cpu().numpy(), want, **tol)
        np.testing.assert_allclose(F.cpu().numpy().reshape(B, N, J * nrhs), wantF, rtol=1e-10,
                                   atol=1e-12 * max(1.0, np.abs(wantF).max()))
        np.testing.assert_allclose(getattr(ops, name)(td, cd, Ud, Vd, Yd, zero_z=True).cpu().numpy(), want - Z0, **tol)


@pytest.mark.parametrize("name", ["solve_lower_rev", "solve_upper_rev", "matmul_lower_rev", "matmul_upper_rev"])
@pytest.mark.parametrize("B,N,J,nrhs", [(1, 16500, 8, 1), (2, 17000, 5, 3), (1, 16384, 2, 8), (3, 16411, 6, 2), (1, 20000, 16, 2),
                                        (1, 600, 8, 1), (3, 1100, 6, 2), (40, 1024, 4, 1), (2, 4096, 8, 8), (2, 1500, 16, 5)])
def test_reverse_sweeps_on_long_series(ops, oracle, monkeypatch, name, B, N, J, nrhs):
    """The four reverse sweeps (internal.hpp:191-303) on a small batch of long series (the solves: from 512 rows): the opposite sweep applied to bZ,
    with its workspace, plus a pass local to the rows (c2_internal_sweep_rev_long) -- all five outputs against the oracle
    and against the row-by-row kernels (C2_REV_LONG=0), each relative to its largest entry."""
    solve = name.startswith("solve")
    if solve and J > 8:
        pytest.skip("the chunk-map solves cover widths up to 8")
    t, c, a, U, V, y = wide_batch(B, N, J) if J <= 8 else dense.synthetic_batch(B, N, J)
    rng = np.random.default_rng(N + 7 * J + nrhs)
    if solve:
        W = np.empty_like(V); d = np.empty_like(a)
        for b in range(B):
            assert oracle.factor_flag(t[b], c[b], a[b], U[b], V[b], d[b], W[b]) == 0
    else:
        W = V
    Y = rng.standard_normal((B, N, nrhs)); bZ = rng.standard_normal((B, N, nrhs))
    Z = np.empty_like(Y); F = np.empty((B, N, J * nrhs))
    want = [np.zeros((B, N)), np.zeros((B, J)), np.zeros((B, N, J)), np.zeros((B, N, J)), np.zeros((B, N, nrhs))]
    fwd = getattr(oracle, name[:-4])
    for b in range(B):
        zb = Y[b].copy() if solve else np.zeros((N, nrhs))
        fwd(t[b], c[b], U[b], W[b], Y[b], zb, F[b])
        Z[b] = zb
        outs = [np.zeros(N), np.zeros(J), np.zeros((N, J)), np.zeros((N, J)), np.zeros((N, nrhs))]
        getattr(oracle, name)(t[b], c[b], U[b], W[b], Y[b], Z[b], F[b], bZ[b], *outs)
        for w, o in zip(want, outs):
            w[b] = o
    args = dev(t, c, U, W, Y, Z, F.reshape(B, N, J, nrhs), bZ)
    monkeypatch.delenv("C2_REV_LONG", raising=False)
    got = getattr(ops, name)(*args)
    monkeypatch.setenv("C2_REV_LONG", "0")
    rows = getattr(ops, name)(*args)
    for g, r, w in zip(got, rows, want):
        gclose(g, w)
        gclose(g, r.cpu().numpy())
    if B > 1:   # one time grid and one set of rates for the batch (the problem of series 0, other right-hand sides)
        rep = lambda x: np.ascontiguousarray(np.tile(x[0], (B,) + (1,) * (x.ndim - 1)))
        U0, W0 = rep(U), rep(W)
        for b in range(B):
            zb = Y[b].copy() if solve else np.zeros((N, nrhs))
            fwd(t[0], c[0], U0[b], W0[b], Y[b], zb, F[b])
            Z[b] = zb
            outs = [np.zeros(N), np.zeros(J), np.zeros((N, J)), np.zeros((N, J)), np.zeros((N, nrhs))]
            getattr(oracle, name)(t[0], c[0], U0[b], W0[b], Y[b], Z[b], F[b], bZ[b], *outs)
            for w, o in zip(want, outs):
                w[b] = o
        monkeypatch.delenv("C2_REV_LONG", raising=False)
        got = getattr(ops, name)(*dev(t[0].copy(), c[0].copy(), U0, W0, Y, Z, F.reshape(B, N, J, nrhs), bZ))
        for g, w in zip(got, want):
            gclose(g, w)


@pytest.mark.parametrize("rows", [None, "16", "64"])
@pytest.mark.parametrize("B,N,J,nrhs", [(1, 16500, 8, 3), (2, 17000, 5, 8), (1, 16384, 2, 2), (3, 16411, 6, 1),
                                        (1, 600, 8, 1), (2, 1500, 7, 3), (64, 1024, 4, 1), (5, 4096, 8, 8)])
def test_chunk_map_solves_with_several_right_hand_sides_and_workspace(ops, oracle, monkeypatch, B, N, J, nrhs, rows):
    """solve_lower / solve_upper on a small batch of series of 512 rows and more by chunk maps, right-hand side by right-hand side
    (c2_internal_solve_chunks*), with and without the workspace F of the drop-in (internal.hpp:140-141, 179-180): every
    row of Z and F against the oracle and against the row-by-row kernels, out of place and in place."""
    t, c, a, U, V, y = wide_batch(B, N, J)
    W = np.empty_like(V); d = np.empty_like(a)
    for b in range(B):
        assert oracle.factor_flag(t[b], c[b], a[b], U[b], V[b], d[b], W[b]) == 0
    rng = np.random.default_rng(N + J)
    Y = np.ascontiguousarray(y[:, :, None] * rng.uniform(0.5, 1.5, (1, 1, nrhs)) + 0.1 * rng.standard_normal((B, N, nrhs)))
    if rows: monkeypatch.setenv("C2_TPG_ROWS", rows)
    td, cd, Ud, Wd, Yd = dev(t, c, U, W, Y)
    for name in ("solve_lower", "solve_upper"):
        want = np.empty_like(Y); wantF = np.empty((B, N, J * nrhs))
        for b in range(B):
            zb = Y[b].copy()
            getattr(oracle, name)(t[b], c[b], U[b], W[b], Y[b], zb, wantF[b])
            want[b] = zb
        tolz = dict(rtol=1e-10, atol=1e-12 * max(1.0, np.abs(want).max()))
        tolf = dict(rtol=1e-10, atol=1e-12 * max(1.0, np.abs(wantF).max()))
        monkeypatch.setenv("C2_TIMEPAR", "1")
        got = getattr(ops, name)(td, cd, Ud, Wd, Yd)
        np.testing.assert_allclose(got.cpu().numpy(), want, **tolz)
        got2, F2 = getattr(ops, name)(td, cd, Ud, Wd, Yd, workspace=True)
        np.testing.assert_allclose(got2.cpu().numpy(), want, **tolz)
        np.testing.assert_allclose(F2.cpu().numpy().reshape(B, N, J * nrhs), wantF, **tolf)
        Zin = Yd.clone()
        got3 = getattr(ops, name)(td, cd, Ud, Wd, Zin, Z=Zin)      # in place
        np.testing.assert_allclose(got3.cpu().numpy(), want, **tolz)
        monkeypatch.setenv("C2_TIMEPAR", "0")
        rowsZ, rowsF = getattr(ops, name)(td, cd, Ud, Wd, Yd, workspace=True)
        np.testing.assert_allclose(got2.cpu().numpy(), rowsZ.cpu().numpy(), **tolz)
        np.testing.assert_allclose(F2.cpu().numpy(), rowsF.cpu().numpy(), **tolf)


@pytest.mark.parametrize("J", [8, 7, 6, 5, 4, 3, 2, 1])
@pytest.mark.parametrize("B,N", [(1, 1), (2, 2), (3, 63), (5, 64), (4, 65), (3, 128), (7, 700), (2, 4096), (70, 300), (1, 20000)])
def test_timepar_gradient_matches_oracle(ops, oracle, monkeypatch, B, N, J):
    """The gradient parallel along time (c2_timepar_grad.hip: d, W, z from factor + solve, linear recurrences for the
    states, the adjoint recursion as affine chunk maps), forced on shapes around the chunk length (64 rows): log-likelihood
    and all six gradients against the oracle; a failed series gets -inf / NaN and leaves its neighbours alone."""
    t, c, a, U, V, y = wide_batch(B, N, J)
    llo, go, flo = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=2)
    assert not np.asarray(flo).any()
    monkeypatch.setenv("C2_TIMEPAR_GRAD", "1")
    ll, grads, flag = ops.loglik_grad(*dev(t, c, a, U, V, y))
    assert int(flag.abs().sum()) == 0
    close(ll, llo)
    for g, e in zip(grads, go):
        gclose(g, e)
    # the row-by-row kernels give the same
    monkeypatch.setenv("C2_TIMEPAR_GRAD", "0")
    ll0, grads0, _ = ops.loglik_grad(*dev(t, c, a, U, V, y))
    close(ll, ll0.cpu().numpy())
    for g, e in zip(grads, grads0):
        gclose(g, e.cpu().numpy())
    # shared time grid and rates (strides 0); bt, bc stay per series
    monkeypatch.setenv("C2_TIMEPAR_GRAD", "1")
    ts, cs = np.tile(t[0], (B, 1)), np.tile(c[0], (B, 1))
    lls, gs, fls = oracle.loglik_grad_batched(ts, cs, a, U, V, y, nthreads=2)
    ok = np.asarray(fls) == 0
    ll3, grads3, flag3 = ops.loglik_grad(*dev(t[0].copy(), c[0].copy(), a, U, V, y))
    assert np.array_equal(flag3.cpu().numpy() != 0, ~ok)
    if ok.any():
        close(ll3[ok], lls[ok])
        for g, e in zip(grads3, gs):
            for b in np.nonzero(ok)[0]:
                gclose(g[b], e[b])
    if B > 2 and N > 2:
        monkeypatch.setenv("C2_TIMEPAR_GRAD", "1")
        a2 = a.copy(); a2[1, N // 2] = -5.0
        ll2, grads2, flag2 = ops.loglik_grad(*dev(t, c, a2, U, V, y))
        fl = flag2.cpu().numpy()
        assert fl[1] != 0 and not fl[[0, 2]].any()
        assert np.isneginf(ll2.cpu().numpy()[1])
        for g, e in zip(grads2, go):
            gn = g.cpu().numpy()
            assert np.isnan(gn[1]).all()
            gclose(gn[0], e[0]); gclose(gn[2], e[2])


def gclose(a, b, tol=1e-10):
    """gradients: relative to the largest entry of the array (the small entries of a gradient are sums of large terms)"""
    a = a.cpu().numpy() if hasattr(a, "cpu") else a
    np.testing.assert_allclose(a, b, rtol=0.0, atol=tol * max(np.abs(b).max(), 1e-300))


@pytest.mark.parametrize("J", [8, 7, 6, 5, 3, 1])
@pytest.mark.parametrize("B,N", [(1, 1), (2, 2), (3, 63), (5, 64), (4, 65), (3, 129), (2, 4096), (9, 1000), (1, 30000)])
def test_newton_factor_matches_oracle(ops, oracle, monkeypatch, B, N, J):
    """`factor` (d, W) at widths 6 and 8 by Newton iterations on the chunk start states (c2_timepar_grad.hip), forced:
    every row against the oracle; a failed series hands the batch to the row-by-row kernel, which reports the
    reference's flag; in-place calls stay on the row-by-row kernel and agree."""
    t, c, a, U, V, y = wide_batch(B, N, J)
    monkeypatch.setenv("C2_FACTOR_ITER", "1")
    d, W, flag = ops.factor(*dev(t, c, a, U, V))
    assert int(flag.abs().sum()) == 0
    for b in range(B):
        do = np.empty(N); Wo = np.empty((N, J)); So = np.empty((N, J, J))
        assert oracle.factor_flag(t[b], c[b], a[b], U[b], V[b], do, Wo, So) == 0
        close(d[b], do); np.testing.assert_allclose(W[b].cpu().numpy(), Wo, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(Wo).max()))
    if N > 40:
        a1 = a.copy(); a1[0, N // 2] = -2.0
        d1, W1, flag1 = ops.factor(*dev(t, c, a1, U, V))
        assert int(flag1[0]) == N // 2 and int(flag1[1:].abs().sum()) == 0
        monkeypatch.setenv("C2_FACTOR_ITER", "0")
        d0, W0, flag0 = ops.factor(*dev(t, c, a1, U, V))
        assert np.array_equal(d1.cpu().numpy()[1:], d0.cpu().numpy()[1:]) if B > 1 else True
        monkeypatch.setenv("C2_FACTOR_ITER", "1")
    ad, Vd = dev(a, V)
    td, cd, Ud = dev(t, c, U)
    d2, W2, _ = ops.factor(td, cd, ad, Ud, Vd, d=ad, W=Vd)   # in place
    close(d2, d.cpu().numpy()); np.testing.assert_allclose(W2.cpu().numpy(), W.cpu().numpy(), rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("scan8,group", [("1", "1"), ("1", "0"), ("0", "1")])
@pytest.mark.parametrize("B,N", [(1, 4096), (3, 390), (70, 1000), (2, 8200), (1, 100001), (5, 64), (2, 33)])
def test_width8_scanned_states_both_sides_of_the_switches(ops, oracle, monkeypatch, B, N, scan8, group):
    """Width 8, round 6: `factor`, the forward log-likelihood and the log-likelihood + gradient on the time-parallel forms with
    the chunk-start states from the scanned chunk elements (C2_FACTOR_SCAN8: up-sweep + k_e8_down) or from Newton iterations
    (= 0), and the chunk pass with the element spread over a group's lanes (C2_E8_GROUP_CHUNKS: k_e8_chunks) or in one lane
    (= 0, k_tp_onepass<8, 2>) -- every row of d, W, the log-likelihood and all six gradients against the oracle; a failed
    series hands the batch to the row-by-row kernel, which reports the reference's flag."""
    t, c, a, U, V, y = dense.synthetic_batch(min(B, 4), N, 8)
    if B > 4:
        rng = np.random.default_rng(B)
        rep = (B + 3) // 4
        t, c, U, V = (np.ascontiguousarray(np.tile(v, (rep,) + (1,) * (v.ndim - 1))[:B]) for v in (t, c, U, V))
        a = np.ascontiguousarray(np.tile(a, (rep, 1))[:B] * rng.uniform(1.0, 1.3, (B, 1)))
        y = np.ascontiguousarray(np.tile(y, (rep, 1))[:B] + 0.05 * rng.standard_normal((B, N)))
    monkeypatch.setenv("C2_FACTOR_SCAN8", scan8)
    monkeypatch.setenv("C2_E8_GROUP_CHUNKS", group)
    monkeypatch.setenv("C2_FACTOR_ITER", "1")
    args = dev(t, c, a, U, V)
    d, W, flag = ops.factor(*args)
    assert int(flag.abs().sum()) == 0
    for b in range(min(B, 5)):
        do = np.empty(N); Wo = np.empty((N, 8))
        assert oracle.factor_flag(t[b], c[b], a[b], U[b], V[b], do, Wo) == 0
        close(d[b], do); np.testing.assert_allclose(W[b].cpu().numpy(), Wo, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(Wo).max()))
    llo, go, flo = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=4)
    assert not np.asarray(flo).any()
    monkeypatch.setenv("C2_TIMEPAR", "1")
    ll, fl = ops.loglik(*args, dev(y)[0])
    assert int(fl.abs().sum()) == 0
    close(ll, llo)
    monkeypatch.setenv("C2_TIMEPAR_GRAD", "1")
    ll2, grads, fl2 = ops.loglik_grad(*args, dev(y)[0])
    assert int(fl2.abs().sum()) == 0
    close(ll2, llo)
    for g, e in zip(grads, go):
        for b in range(B):
            np.testing.assert_allclose(g[b].cpu().numpy(), e[b], rtol=0.0, atol=1e-10 * max(np.abs(e[b]).max(), 1e-300))
    if N > 40 and B > 1:
        a1 = a.copy(); a1[1, N // 2] = -2.0
        d1, W1, flag1 = ops.factor(*dev(t, c, a1, U, V))
        f = flag1.cpu().numpy()
        assert int(f[1]) == N // 2 and int(np.abs(np.delete(f, 1)).sum()) == 0
        close(d1[0], d[0].cpu().numpy(), 1e-12)


def test_newton_factor_on_hard_series(ops, oracle, monkeypatch):
    """Series the Newton iteration has to work for: a long gap (the states decouple), repeated times, a nearly singular
    stretch (tiny white noise), very slow and very fast rates side by side -- every row against the oracle, whether the
    iteration converged or the row-by-row kernel took over."""
    J, N, B = 8, 3000, 4
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    t[0, N // 2:] += 500.0
    t[1, 100:110] = t[1, 100]
    a[2, 1000:1400] -= 0.999 * (a[2, 1000:1400] - np.sum(U[2, 1000:1400] * V[2, 1000:1400], axis=1))
    c[3] = c[3] * np.array([1e-3, 1e-3, 1.0, 1.0, 30.0, 30.0, 300.0, 300.0])
    monkeypatch.setenv("C2_FACTOR_ITER", "1")
    d, W, flag = ops.factor(*dev(t, c, a, U, V))
    for b in range(B):
        do = np.empty(N); Wo = np.empty((N, J)); So = np.empty((N, J, J))
        fo = oracle.factor_flag(t[b], c[b], a[b], U[b], V[b], do, Wo, So)
        assert int(flag[b]) == fo
        if fo == 0:
            close(d[b], do, tol=1e-9 if b == 2 else 1e-10)
            np.testing.assert_allclose(W[b].cpu().numpy(), Wo, rtol=1e-9 if b == 2 else 1e-10, atol=1e-12 * max(1.0, np.abs(Wo).max()))


@pytest.mark.parametrize("J", [8, 7, 6, 5, 3, 1])
@pytest.mark.parametrize("B,N", [(1, 1), (3, 63), (4, 65), (2, 4096), (9, 1000), (1, 30000)])
def test_wide_loglik_matches_oracle(ops, oracle, monkeypatch, B, N, J):
    """Forward-only log-likelihood at widths 6 / 8 composed from the Newton factor, the time-parallel solve and a
    reduction (forced): against the oracle and the row-by-row kernel; a failed series gets its flag and -inf."""
    t, c, a, U, V, y = wide_batch(B, N, J)
    llo, flo = oracle_ll(oracle, t, c, a, U, V, y)
    monkeypatch.setenv("C2_FACTOR_ITER", "1")
    ll, flag = ops.loglik(*dev(t, c, a, U, V, y))
    assert int(flag.abs().sum()) == 0
    close(ll, llo)
    monkeypatch.setenv("C2_FACTOR_ITER", "0")
    ll0, _ = ops.loglik(*dev(t, c, a, U, V, y))
    close(ll, ll0.cpu().numpy())
    if B > 2 and N > 40:
        monkeypatch.setenv("C2_FACTOR_ITER", "1")
        a1 = a.copy(); a1[1, N // 3] = -3.0
        ll1, flag1 = ops.loglik(*dev(t, c, a1, U, V, y))
        fl = flag1.cpu().numpy()
        assert fl[1] != 0 and not fl[[0, 2]].any()
        assert np.isneginf(ll1.cpu().numpy()[1])
        close(ll1[[0, 2]], llo[[0, 2]])


def test_time_parallel_gradient_is_the_default_for_long_series(ops, oracle, monkeypatch):
    """Without any switch a small batch of long series takes the time-parallel gradient (bitwise the forced result) at
    every width it covers, and agrees with the oracle; the workspace query covers it."""
    import torch
    for J, B, N in ((8, 2, 5000), (6, 3, 3000), (4, 1, 9000), (2, 5, 2048)):
        t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
        args = dev(t, c, a, U, V, y)
        monkeypatch.delenv("C2_TIMEPAR_GRAD", raising=False)
        work = ops.loglik_grad_workspace(B, N, J, torch.device("cuda:0"))
        ll, grads, flag = ops.loglik_grad(*args, work=work)
        monkeypatch.setenv("C2_TIMEPAR_GRAD", "1")
        ll1, grads1, _ = ops.loglik_grad(*args)
        assert torch.equal(ll, ll1) and all(torch.equal(g, g1) for g, g1 in zip(grads, grads1))
        llo, go, _ = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=2)
        close(ll, llo)
        for g, e in zip(grads, go):
            gclose(g, e)


@pytest.mark.parametrize("J,N", [(4, 40000), (2, 140000)])
def test_newton_factor_takes_over_on_long_series_at_widths_4_and_2(ops, oracle, monkeypatch, J, N):
    """Widths 4 / 2 have the composed linear-fractional maps (sequential chain over the chunks); on long series the
    Newton iterations with their two-level chains run instead (default dispatch): d, W against the oracle and against
    the composed maps (C2_FACTOR_ITER=0), and the gradient built on top against the oracle."""
    t, c, a, U, V, y = dense.synthetic_batch(1, N, J)
    monkeypatch.delenv("C2_FACTOR_ITER", raising=False)
    d, W, flag = ops.factor(*dev(t, c, a, U, V))
    assert int(flag.abs().sum()) == 0
    do = np.empty(N); Wo = np.empty((N, J)); So = np.empty((N, J, J))
    assert oracle.factor_flag(t[0], c[0], a[0], U[0], V[0], do, Wo, So) == 0
    close(d[0], do); np.testing.assert_allclose(W[0].cpu().numpy(), Wo, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(Wo).max()))
    monkeypatch.setenv("C2_FACTOR_ITER", "0")
    d0, W0, _ = ops.factor(*dev(t, c, a, U, V))
    close(d0, d.cpu().numpy())
    monkeypatch.delenv("C2_FACTOR_ITER", raising=False)
    ll, grads, flag = ops.loglik_grad(*dev(t, c, a, U, V, y))
    llo, go, _ = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=2)
    close(ll, llo)
    for g, e in zip(grads, go):
        gclose(g, e)


@pytest.mark.parametrize("J,N,shared", [(2, 17000, True), (6, 16500, False), (8, 33000, True), (4, 40001, False)])
def test_time_parallel_gradient_two_level_chains(ops, oracle, monkeypatch, J, N, shared):
    """From 128 chunks per series every chain over the chunks runs in two levels (blocks of 32 chunks composed in
    parallel): a small batch with a ragged last chunk and block, shared or per-series grids, one failed series -- the
    log-likelihoods, flags and all six gradients against the oracle."""
    B = 3
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    if shared:
        t, c = np.tile(t[0], (B, 1)), np.tile(c[0], (B, 1))
    a = a + 0.3
    a[1, N // 3] = -2.0
    llo, go, flo = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=2)
    ok = np.asarray(flo) == 0      # (on a shared grid the other series' U, V need not give a positive definite matrix)
    assert ok[0] and not ok[1]
    monkeypatch.setenv("C2_TIMEPAR_GRAD", "1")
    args = dev(t[0].copy() if shared else t, c[0].copy() if shared else c, a, U, V, y)
    ll, grads, flag = ops.loglik_grad(*args)
    assert np.array_equal(flag.cpu().numpy() != 0, ~ok)
    close(ll[ok], llo[ok])
    for g, e in zip(grads, go):
        gn = g.cpu().numpy()
        assert np.isnan(gn[~ok]).all()
        for b in np.nonzero(ok)[0]:
            gclose(gn[b], e[b])
    ll0, flag0 = ops.loglik(*args)
    close(ll0[ok], llo[ok])


@pytest.mark.parametrize("J,N", [(8, 1500), (6, 20000), (3, 900)])
def test_time_parallel_gradient_inside_a_graph_capture(ops, oracle, J, N):
    """The time-parallel gradient makes no allocation (the factor's Newton iterations / composed maps work in a piece of
    the caller's workspace, z comes from the chunk maps), so the whole call -- gated iterations and fallbacks included --
    can be captured in a HIP graph; replayed on new data it gives the oracle's numbers."""
    import torch
    B = 2
    t, c, a, U, V, y = wide_batch(B, N, J)
    td, cd, ad, Ud, Vd, yd = dev(t, c, a, U, V, y)
    work = ops.loglik_grad_workspace(B, N, J, td.device)
    ll, out, flag = ops.loglik_grad(td, cd, ad, Ud, Vd, yd, work=work)      # warm-up outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ll_g, out_g, flag_g = ops.loglik_grad(td, cd, ad, Ud, Vd, yd, work=work, out=out)
    for variant in range(2):
        y2 = y + 0.01 * (variant + 1)
        yd.copy_(torch.from_numpy(y2))
        g.replay()
        torch.cuda.synchronize()
        llo, go, flo = oracle.loglik_grad_batched(t, c, a, U, V, y2, nthreads=2)
        assert flag_g.cpu().tolist() == list(flo)
        close(ll_g, llo)
        for x, e in zip(out_g, go):
            gclose(x, e)
