# -*- coding: utf-8 -*-
"""Log-likelihood and its gradient w.r.t. the celerite COEFFICIENTS on the device (c2_terms.hip; SURVEY.md section
8f-1) against (1) the CPU oracle's gradients w.r.t. (t, c, a, U, V, y) pushed through a numpy restatement of the reverse
of get_celerite_matrices (driver.cpp:456-474), and (2) central finite differences of the DENSE log-likelihood."""
import numpy as np
import pytest

from oracle import dense

pytestmark = pytest.mark.gpu
NAMES = ("bar", "bcr", "bac", "bbc", "bcc", "bdc", "bx", "bdiag", "by")


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


def force(monkeypatch, which):
    """composed chain (matrices in memory) / one lane per series (c2_loglik_t.hip) / two lanes (c2_loglik_k2.hip) / four (c2_loglik_q4.hip:
    k_q4_*<..., TT>, gradient only) / eight (c2_loglik.hip: k_loglik_*<..., TT>; its gradient of a one-row series stays composed)"""
    monkeypatch.setenv("C2_TERMS_FUSED", "1" if which == "one" else "0")
    monkeypatch.setenv("C2_TERMS_TWO_LANES", "1" if which == "two" else "0")
    monkeypatch.setenv("C2_TERMS_EIGHT_LANES", "1" if which == "eight" else "0")   # (the gradient at N >= 2; forward any N)
    monkeypatch.setenv("C2_TERMS_FOUR_LANES", "1" if which == "four" else "0")


def coeffs(B, Jr, Jc, rng):
    ar = rng.uniform(0.5, 1.5, (B, Jr)); cr = rng.uniform(0.05, 0.5, (B, Jr))
    ac = rng.uniform(0.5, 2.0, (B, Jc))
    cc = rng.uniform(0.02, 0.3, (B, Jc)); dc = rng.uniform(0.2, 3.0, (B, Jc))
    bc = ac * cc / dc * rng.uniform(0.0, 0.9, (B, Jc))   # a valid (positive semi-definite) term needs ac cc >= bc dc
    return ar, cr, ac, bc, cc, dc


def oracle_chain(oracle, ar, cr, ac, bc, cc, dc, x, diag, y):
    """ll and the nine gradients for ONE series: CPU oracle + the reverse of the matrix recipe in numpy (oracle/dense.py)."""
    ll, grads, flag = dense.coefficient_chain(oracle, ar, cr, ac, bc, cc, dc, x, diag, y)
    assert flag == 0
    return ll, grads


@pytest.mark.parametrize("B,N,Jr,Jc", [(5, 200, 1, 2), (3, 1000, 0, 4), (70, 64, 2, 0), (2, 33, 3, 1), (1, 1, 1, 1), (2, 20000, 0, 2),
                                      (1, 9000, 1, 3), (3, 8192, 2, 0)])
def test_loglik_terms_vs_oracle_chain(ops, oracle, B, N, Jr, Jc):
    rng = np.random.default_rng(11)
    ar, cr, ac, bc, cc, dc = coeffs(B, Jr, Jc, rng)
    x = np.sort(rng.uniform(0, N / 10.0, (B, N)), axis=1)
    diag = rng.uniform(0.1, 0.3, (B, N))
    y = np.sin(x) + 0.1 * rng.standard_normal((B, N))
    want = [oracle_chain(oracle, ar[b], cr[b], ac[b], bc[b], cc[b], dc[b], x[b], diag[b], y[b]) for b in range(B)]
    args = dev(ar, cr, ac, bc, cc, dc, x, diag, y)
    ll, flag = ops.loglik_terms(*args)
    assert int(flag.abs().sum()) == 0
    close(ll, np.array([w[0] for w in want]))
    ll2, grads, flag2 = ops.loglik_terms_grad(*args)
    close(ll2, np.array([w[0] for w in want]))
    for k, (nm, g) in enumerate(zip(NAMES, grads)):
        e = np.stack([w[1][k] for w in want])
        if e.size:
            close(g, e)
    # shared coefficients and a shared grid: same series replicated with its own data
    args_s = dev(ar[0], cr[0], ac[0], bc[0], cc[0], dc[0], x[0], diag, y)
    want_s = [oracle_chain(oracle, ar[0], cr[0], ac[0], bc[0], cc[0], dc[0], x[0], diag[b], y[b]) for b in range(B)]
    ll3, grads3, _ = ops.loglik_terms_grad(*args_s)
    close(ll3, np.array([w[0] for w in want_s]))
    for k, g in enumerate(grads3):
        e = np.stack([w[1][k] for w in want_s])
        if e.size:
            close(g, e)


def test_loglik_terms_vs_dense_finite_differences(ops):
    """Independent of the oracle: central differences of the dense Cholesky log-likelihood in every coefficient."""
    rng = np.random.default_rng(5)
    B, N, Jr, Jc = 1, 40, 1, 2
    ar, cr, ac, bc, cc, dc = coeffs(B, Jr, Jc, rng)
    x = np.sort(rng.uniform(0, 8.0, (B, N)), axis=1)
    diag = rng.uniform(0.1, 0.3, (B, N))
    y = np.sin(x) + 0.1 * rng.standard_normal((B, N))

    def ll_dense(vals):
        co = dense.Coeffs(**vals)
        return dense.dense_loglik(dense.dense_matrix(co, x[0], diag[0]), y[0])

    base = dict(ar=ar[0], cr=cr[0], ac=ac[0], bc=bc[0], cc=cc[0], dc=dc[0])
    ll, grads, flag = ops.loglik_terms_grad(*dev(ar, cr, ac, bc, cc, dc, x, diag, y))
    assert abs(float(ll[0]) - ll_dense(base)) <= 1e-10 * abs(ll_dense(base))
    got = dict(zip(("ar", "cr", "ac", "bc", "cc", "dc"), [g[0].cpu().numpy() for g in grads[:6]]))
    h = 1e-6
    for name in base:
        for k in range(len(base[name])):
            vp = {n: v.copy() for n, v in base.items()}; vm = {n: v.copy() for n, v in base.items()}
            vp[name][k] += h; vm[name][k] -= h
            fd = (ll_dense(vp) - ll_dense(vm)) / (2 * h)
            assert abs(fd - got[name][k]) <= 2e-6 * max(1.0, abs(fd)), (name, k, fd, got[name][k])


def test_log_likelihood_terms_autograd(ops, oracle):
    """torch.autograd over the coefficients: per-series and shared hyper-parameters."""
    import torch
    from celerite2_amd import autograd as ag
    rng = np.random.default_rng(2)
    B, N, Jr, Jc = 4, 120, 1, 2
    ar, cr, ac, bc, cc, dc = coeffs(1, Jr, Jc, rng)
    x = np.sort(rng.uniform(0, 12.0, N))
    diag = rng.uniform(0.1, 0.3, (B, N))
    y = np.sin(x)[None, :] + 0.1 * rng.standard_normal((B, N))
    want = [oracle_chain(oracle, ar[0], cr[0], ac[0], bc[0], cc[0], dc[0], x, diag[b], y[b]) for b in range(B)]
    leaves = [v.requires_grad_(True) for v in dev(ar[0], cr[0], ac[0], bc[0], cc[0], dc[0], x, diag, y)]
    w = torch.tensor([0.5, -1.0, 2.0, 0.25], dtype=torch.float64, device="cuda")
    (ag.log_likelihood_terms(*leaves) * w).sum().backward()
    wn = w.cpu().numpy()
    for k, leaf in enumerate(leaves):
        per = np.stack([wn[b] * want[b][1][k] for b in range(B)])
        close(leaf.grad, per.sum(0) if k < 7 else per)
    with pytest.raises(ValueError, match="Invalid shape: y"):
        ops.loglik_terms(*[v.detach() for v in leaves[:8]], leaves[8].detach()[:, :5].contiguous())


@pytest.mark.parametrize("lanes", ["one", "two", "four", "eight"])
@pytest.mark.parametrize("Jr,Jc", [(0, 4), (2, 3), (4, 2), (6, 1), (8, 0)])
@pytest.mark.parametrize("B,N", [(70, 200), (3, 1), (2, 2), (5, 9), (130, 67)])
def test_fused_terms_kernels(ops, oracle, monkeypatch, B, N, Jr, Jc, lanes):
    """The kernels that form U_n, V_n in registers -- one lane per series (c2_loglik_t.hip) and two lanes per series
    (c2_loglik_k2.hip: a slot of two columns per lane is a complex term or a pair of real terms), width 8 -- forced for any
    batch size, against the same oracle chain; and against the composed path (matrices in memory)."""
    rng = np.random.default_rng(100 * Jr + N)
    ar, cr, ac, bc, cc, dc = coeffs(B, Jr, Jc, rng)
    x = np.sort(rng.uniform(0, N / 10.0, (B, N)), axis=1)
    diag = rng.uniform(0.1, 0.3, (B, N))
    y = np.sin(x) + 0.1 * rng.standard_normal((B, N))
    nb = min(B, 6)
    want = [oracle_chain(oracle, ar[b], cr[b], ac[b], bc[b], cc[b], dc[b], x[b], diag[b], y[b]) for b in range(nb)]
    args = dev(ar, cr, ac, bc, cc, dc, x, diag, y)
    force(monkeypatch, "composed")
    ll_c, flag_c = ops.loglik_terms(*args)
    ll_cg, grads_c, _ = ops.loglik_terms_grad(*args)
    force(monkeypatch, lanes)
    ll, flag = ops.loglik_terms(*args)
    assert int(flag.abs().sum()) == 0
    close(ll[:nb], np.array([w[0] for w in want]))
    close(ll, ll_c.cpu().numpy())
    ll2, grads, flag2 = ops.loglik_terms_grad(*args)
    assert int(flag2.abs().sum()) == 0
    close(ll2, ll_c.cpu().numpy())
    for k, (nm, g) in enumerate(zip(NAMES, grads)):
        e = np.stack([w[1][k] for w in want])
        if e.size:
            close(g[:nb], e)
            close(g, grads_c[k].cpu().numpy(), tol=1e-9, floor=1e-11)
    # shared coefficients and grid
    args_s = dev(ar[0], cr[0], ac[0], bc[0], cc[0], dc[0], x[0], diag, y)
    ll3, grads3, _ = ops.loglik_terms_grad(*args_s)
    force(monkeypatch, "composed")
    ll4, grads4, _ = ops.loglik_terms_grad(*args_s)
    close(ll3, ll4.cpu().numpy())
    for g3, g4 in zip(grads3, grads4):
        if g4.numel():
            close(g3, g4.cpu().numpy(), tol=1e-9, floor=1e-11)


@pytest.mark.parametrize("lanes", ["one", "eight"])
@pytest.mark.parametrize("Jr,Jc", [(0, 2), (2, 1), (4, 0), (0, 1), (2, 0)])
@pytest.mark.parametrize("B,N", [(70, 200), (3, 1), (5, 9), (130, 67)])
def test_fused_terms_kernels_widths_four_and_two(ops, oracle, monkeypatch, B, N, Jr, Jc, lanes):
    """Widths 4 and 2 with the rows formed in the lanes: one lane per series (c2_loglik_t.hip) and a group of J lanes
    ("eight": k_loglik_*<..., TT> at G = 4, 2) against the oracle chain and the composed path."""
    rng = np.random.default_rng(7 + 10 * Jr + Jc + N)
    ar, cr, ac, bc, cc, dc = coeffs(B, Jr, Jc, rng)
    x = np.sort(rng.uniform(0, N / 10.0, (B, N)), axis=1)
    diag = rng.uniform(0.1, 0.3, (B, N))
    y = np.sin(x) + 0.1 * rng.standard_normal((B, N))
    nb = min(B, 4)
    want = [oracle_chain(oracle, ar[b], cr[b], ac[b], bc[b], cc[b], dc[b], x[b], diag[b], y[b]) for b in range(nb)]
    args = dev(ar, cr, ac, bc, cc, dc, x, diag, y)
    force(monkeypatch, "composed")
    ll_c, _ = ops.loglik_terms(*args)
    _, grads_c, _ = ops.loglik_terms_grad(*args)
    force(monkeypatch, lanes)
    ll, flag = ops.loglik_terms(*args)
    ll2, grads, flag2 = ops.loglik_terms_grad(*args)
    assert int(flag.abs().sum()) == 0 and int(flag2.abs().sum()) == 0
    close(ll[:nb], np.array([w[0] for w in want]))
    close(ll, ll_c.cpu().numpy()); close(ll2, ll_c.cpu().numpy())
    for k, g in enumerate(grads):
        e = np.stack([w[1][k] for w in want])
        if e.size:
            close(g[:nb], e)
            close(g, grads_c[k].cpu().numpy(), tol=1e-9, floor=1e-11)


@pytest.mark.parametrize("lanes", ["one", "two", "four", "eight"])
def test_fused_terms_fallback_when_backward_recursion_is_unsafe(ops, oracle, monkeypatch, lanes):
    """Rates x segment span beyond kBackwardGuard: the fused reverse sweep declines on the device and the gated composed
    chain delivers the gradients (same outputs); a batch inside the guard next to it takes the fused sweep."""
    rng = np.random.default_rng(8)
    B, N, Jr, Jc = 66, 150, 2, 3
    ar, cr, ac, bc, cc, dc = coeffs(B, Jr, Jc, rng)
    diag = rng.uniform(0.1, 0.3, (B, N))
    for scale in (4.0, 0.02):     # c_max * span of 32 rows ~ 0.5 * 32 * scale
        x = np.sort(rng.uniform(0, N * scale, (B, N)), axis=1)
        y = np.sin(x) + 0.1 * rng.standard_normal((B, N))
        want = [oracle_chain(oracle, ar[b], cr[b], ac[b], bc[b], cc[b], dc[b], x[b], diag[b], y[b]) for b in range(4)]
        force(monkeypatch, lanes)
        ll, grads, flag = ops.loglik_terms_grad(*dev(ar, cr, ac, bc, cc, dc, x, diag, y))
        assert int(flag.abs().sum()) == 0
        close(ll[:4], np.array([w[0] for w in want]))
        for k, g in enumerate(grads):
            close(g[:4], np.stack([w[1][k] for w in want]))


@pytest.mark.parametrize("lanes", ["one", "two", "four", "eight"])
def test_fused_terms_failed_series_gradients_are_nan(ops, monkeypatch, lanes):
    import torch
    rng = np.random.default_rng(3)
    B, N, Jr, Jc = 70, 40, 2, 3
    ar, cr, ac, bc, cc, dc = coeffs(B, Jr, Jc, rng)
    x = np.sort(rng.uniform(0, 4.0, (B, N)), axis=1)
    diag = rng.uniform(0.1, 0.3, (B, N))
    diag[9, 17] = -50.0      # not positive definite
    y = rng.standard_normal((B, N))
    force(monkeypatch, lanes)
    ll, grads, flag = ops.loglik_terms_grad(*dev(ar, cr, ac, bc, cc, dc, x, diag, y))
    assert int(flag[9]) != 0 and int(flag.abs().sum()) == int(flag[9].abs())
    for g in grads:
        assert bool(torch.isnan(g[9]).all()), g[9]
        ok = torch.cat([g[:9], g[10:]])
        assert bool(torch.isfinite(ok).all())


@pytest.mark.parametrize("lanes", ["one", "two", "four", "eight"])
def test_fused_terms_large_phases_take_the_library_reduction(ops, oracle, monkeypatch, lanes):
    """Raw Julian dates: dc * x beyond the range of the branch-free sincos -> the wavefront runs the instantiation with
    the library's large-argument reduction; a neighbouring wavefront with small phases keeps the fast one."""
    rng = np.random.default_rng(21)
    B, N, Jr, Jc = 128, 90, 2, 3
    ar, cr, ac, bc, cc, dc = coeffs(B, Jr, Jc, rng)
    x = np.sort(rng.uniform(0, 9.0, (B, N)), axis=1)
    x[:64] += 2.45e6
    diag = rng.uniform(0.1, 0.3, (B, N))
    y = np.sin(x) + 0.1 * rng.standard_normal((B, N))
    # the matrix recipe on its own: the rare-path kernel (library sincos) owns the columns of the large-phase series
    a_d, U_d, V_d = ops.get_celerite_matrices(*dev(ar, ac, bc, dc, x, diag))
    for b in (0, 63, 64, 127):
        _, a_o, U_o, V_o = dense.celerite_matrices(dense.Coeffs(ar=ar[b], cr=cr[b], ac=ac[b], bc=bc[b], cc=cc[b], dc=dc[b]),
                                                   x[b], diag[b])
        close(a_d[b], a_o); close(U_d[b], U_o); close(V_d[b], V_o)
    force(monkeypatch, lanes)
    ll, grads, flag = ops.loglik_terms_grad(*dev(ar, cr, ac, bc, cc, dc, x, diag, y))
    ll_f, _ = ops.loglik_terms(*dev(ar, cr, ac, bc, cc, dc, x, diag, y))
    assert int(flag.abs().sum()) == 0
    for b in (0, 5, 63, 64, 100):
        want = oracle_chain(oracle, ar[b], cr[b], ac[b], bc[b], cc[b], dc[b], x[b], diag[b], y[b])
        close(ll[b:b + 1], np.array([want[0]]))
        close(ll_f[b:b + 1], np.array([want[0]]))
        for k, g in enumerate(grads):
            # bdc = sum_n g_n x_n with x ~ 2.5e6 cancels to O(1): its rounding error scales with eps * N * max|x|
            # whatever the summation order (the numpy chain and the device differ there), so that is its floor
            floor = 1e-15 * N * float(np.abs(x[b]).max()) if NAMES[k] == "bdc" else 1e-10
            close(g[b], want[1][k], tol=1e-10, floor=max(floor, 1e-10))


@pytest.mark.parametrize("lanes", ["two", "four", "eight"])
def test_fused_terms_row_beyond_the_fast_sincos_on_an_unsorted_grid_is_loud(ops, monkeypatch, lanes):
    """The fused kernels' range test for the branch-free sincos looks at the ENDS of a series (x sorted is a precondition of the
    recursions).  An unsorted grid with a huge time in the middle must not come back as a finite number: that series is NaN /
    flagged, its neighbours are untouched."""
    import torch
    rng = np.random.default_rng(77)
    B, N, Jr, Jc = 70, 64, 2, 3
    ar, cr, ac, bc, cc, dc = coeffs(B, Jr, Jc, rng)
    x = np.sort(rng.uniform(0, 6.0, (B, N)), axis=1)
    diag = rng.uniform(0.1, 0.3, (B, N))
    y = np.sin(x) + 0.1 * rng.standard_normal((B, N))
    force(monkeypatch, "composed")
    ll_c, g_c, _ = ops.loglik_terms_grad(*dev(ar, cr, ac, bc, cc, dc, x, diag, y))
    x[3, 5] = 1e9       # dc x = O(1e9) >> 1.6e6 between ends of O(1)
    args = dev(ar, cr, ac, bc, cc, dc, x, diag, y)
    force(monkeypatch, lanes)
    ll, g, flag = ops.loglik_terms_grad(*args)
    ll_f, flag_f = ops.loglik_terms(*args)
    for v, f in ((ll, flag), (ll_f, flag_f)):
        assert (not bool(torch.isfinite(v[3]))) or int(f[3]) != 0
    ok = [b for b in range(B) if b != 3]
    close(ll[ok], ll_c.cpu().numpy()[ok]); close(ll_f[ok], ll_c.cpu().numpy()[ok])
    for a, b in zip(g, g_c):
        if b.numel():
            close(a[ok], b.cpu().numpy()[ok], tol=1e-9, floor=1e-11)


@pytest.mark.parametrize("lanes", ["one", "two", "four", "eight"])
def test_fused_terms_mixed_groups_and_long_series(ops, oracle, monkeypatch, lanes):
    """Groups of 64 series decide on their own: in one batch the first group stays inside the backward guard (fused kernels),
    the second has gaps in time beyond it (composed chain behind the same gate words), the third -- a partial group -- is
    inside again; and a forced fused path on a few LONG series (thousands of anchors)."""
    rng = np.random.default_rng(41)
    B, N, Jr, Jc = 150, 120, 2, 3
    ar, cr, ac, bc, cc, dc = coeffs(B, Jr, Jc, rng)
    scale = np.where((np.arange(B) >= 64) & (np.arange(B) < 128), 4.0, 0.02)
    x = np.sort(rng.uniform(0, 1.0, (B, N)), axis=1) * (N * scale)[:, None]
    diag = rng.uniform(0.1, 0.3, (B, N))
    y = np.sin(x) + 0.1 * rng.standard_normal((B, N))
    args = dev(ar, cr, ac, bc, cc, dc, x, diag, y)
    force(monkeypatch, "composed")
    ll_c, g_c, _ = ops.loglik_terms_grad(*args)
    force(monkeypatch, lanes)
    ll, g, flag = ops.loglik_terms_grad(*args)
    assert int(flag.abs().sum()) == 0
    close(ll, ll_c.cpu().numpy())
    for a, b in zip(g, g_c):
        close(a, b.cpu().numpy(), tol=1e-9, floor=1e-11)
    for b in (0, 63, 64, 100, 127, 128, 149):
        want = oracle_chain(oracle, ar[b], cr[b], ac[b], bc[b], cc[b], dc[b], x[b], diag[b], y[b])
        close(ll[b:b + 1], np.array([want[0]]))
        for k, gg in enumerate(g):
            close(gg[b], want[1][k])
    # long series
    B, N = 3, 5003
    ar, cr, ac, bc, cc, dc = coeffs(B, Jr, Jc, rng)
    x = np.sort(rng.uniform(0, N / 10.0, (B, N)), axis=1)
    diag = rng.uniform(0.1, 0.3, (B, N))
    y = np.sin(x) + 0.1 * rng.standard_normal((B, N))
    args = dev(ar, cr, ac, bc, cc, dc, x, diag, y)
    force(monkeypatch, "composed")
    ll_c, g_c, _ = ops.loglik_terms_grad(*args)
    force(monkeypatch, lanes)
    ll, g, flag = ops.loglik_terms_grad(*args)
    ll_f, _ = ops.loglik_terms(*args)
    assert int(flag.abs().sum()) == 0
    close(ll, ll_c.cpu().numpy()); close(ll_f, ll_c.cpu().numpy())
    for a, b in zip(g, g_c):
        close(a, b.cpu().numpy(), tol=1e-9, floor=1e-11)


@pytest.mark.parametrize("lanes", ["composed", "one", "two", "four", "eight"])
def test_loglik_terms_grad_is_graph_capturable(ops, oracle, monkeypatch, lanes):
    """The coefficient-level gradient is stream-ordered end to end on every mapping -- caller-provided workspace and outputs, the
    choice between the fused kernels and the composed chain made per group of 64 series on the device -- so one HIP graph captured
    once replays on new data: new observations, new coefficients, and times whose gaps close the gates of the fused kernels."""
    import torch
    rng = np.random.default_rng(91)
    B, N, Jr, Jc = 70, 150, 2, 3
    ar, cr, ac, bc, cc, dc = coeffs(B, Jr, Jc, rng)
    x = np.sort(rng.uniform(0, N * 0.02, (B, N)), axis=1)
    diag = rng.uniform(0.1, 0.3, (B, N))
    y = np.sin(x) + 0.1 * rng.standard_normal((B, N))
    force(monkeypatch, lanes)
    args = dev(ar, cr, ac, bc, cc, dc, x, diag, y)
    work = ops.loglik_terms_workspace(B, N, Jr, Jc, args[0].device)
    ll, out, flag = ops.loglik_terms_grad(*args, work=work)          # warm-up outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ll_g, out_g, flag_g = ops.loglik_terms_grad(*args, work=work, out=out)
    for variant in range(3):
        y2 = y + 0.01 * variant
        ac2 = ac * (1.0 + 0.05 * variant)
        x2 = x.copy()
        if variant == 2:
            x2[:64, N // 2:] += 500.0      # beyond the backward guard for the first group: the composed chain answers there
        args[8].copy_(torch.from_numpy(y2)); args[2].copy_(torch.from_numpy(ac2)); args[6].copy_(torch.from_numpy(x2))
        g.replay()
        torch.cuda.synchronize()
        assert int(flag_g.abs().sum()) == 0
        for b in (0, 5, 63, 64, 69):
            want = oracle_chain(oracle, ar[b], cr[b], ac2[b], bc[b], cc[b], dc[b], x2[b], diag[b], y2[b])
            close(ll_g[b:b + 1], np.array([want[0]]))
            for k, gg in enumerate(out_g):
                floor = 1e-15 * N * float(np.abs(x2[b]).max()) if NAMES[k] == "bdc" else 1e-10
                close(gg[b], want[1][k], tol=1e-10, floor=max(floor, 1e-10))
