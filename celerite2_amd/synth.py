# -*- coding: utf-8 -*-
"""Synthetic batches of independent GPs (SURVEY.md section 8d recipe), used by bench.py and smoke().

Per series b: rng = default_rng(seed0 + b) (the reference's seed convention, testing.py:18);
t = sort(U(0, N/10)); diag ~ U(0.1, 0.3); xi ~ U(-1, 1); y = sin t + 0.1 N(0,1);
kernel = sum of J/2 underdamped SHOTerms, S0 = 5*0.7^k, w0 = 0.1*3^k*(1 + 0.05 xi), Q = 3.45 + k
(k = 0, xi = 0 is the reference test kernel SHOTerm(S0=5, w0=0.1, Q=3.45), testing.py:30).
Coefficients follow python/celerite2/terms.py:658-691; the matrices (a, U, V) are then built ON DEVICE
by the c2_get_celerite_matrices kernel (driver.cpp:422-477)."""
import numpy as np


def sho_underdamped(S0, w0, Q, eps=1e-5):
    f = np.sqrt(np.maximum(4.0 * Q**2 - 1.0, eps))
    a = S0 * w0 * Q
    c = 0.5 * w0 / Q
    return a, a / f, c, c * f  # ac, bc, cc, dc


def host_inputs(first, count, N, J, seed0=721, gap_fraction=0.0, gap=10.0):
    """t, diag, y (count, N) and complex-term coefficients ac, bc, cc, dc (count, J/2) on the host.
    gap_fraction > 0: that fraction of the series gets one gap of `gap` time units (100 mean spacings by default: a
    night, a season) at a random row -- drawn after everything else, so the other numbers do not change."""
    assert J % 2 == 0, "the synthetic kernel is a sum of J/2 complex (underdamped SHO) terms"
    Jc = J // 2
    t = np.empty((count, N)); diag = np.empty((count, N)); y = np.empty((count, N)); xi = np.empty(count)
    for i in range(count):
        rng = np.random.default_rng(seed0 + first + i)
        t[i] = np.sort(rng.uniform(0, N / 10.0, N))
        diag[i] = rng.uniform(0.1, 0.3, N)
        xi[i] = rng.uniform(-1, 1)
        y[i] = np.sin(t[i]) + 0.1 * rng.standard_normal(N)
        if gap_fraction > 0.0 and N > 2 and rng.uniform() < gap_fraction:
            n0 = int(rng.integers(1, N))
            t[i, n0:] += gap
            y[i, n0:] = np.sin(t[i, n0:]) + (y[i, n0:] - np.sin(t[i, n0:] - gap))
    k = np.arange(Jc, dtype=np.float64)
    S0 = 5.0 * 0.7**k
    w0 = 0.1 * 3.0**k[None, :] * (1.0 + 0.05 * xi[:, None])
    Q = 3.45 + k
    ac, bc, cc, dc = sho_underdamped(S0[None, :], w0, Q[None, :])
    return t, diag, y, ac, bc, cc, dc


def device_batch(first, count, N, J, device, seed0=721, gap_fraction=0.0, gap=10.0):
    """Device-resident (t, c, a, U, V, y) for series [first, first+count)."""
    import torch

    from . import ops

    t, diag, y, ac, bc, cc, dc = host_inputs(first, count, N, J, seed0, gap_fraction, gap)
    to = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)
    td, diagd, yd, acd, bcd, dcd = map(to, (t, diag, y, ac, bc, dc))
    ar = torch.zeros((count, 0), dtype=torch.float64, device=device)
    c = np.repeat(cc, 2, axis=1)  # c = [cc0, cc0, cc1, cc1, ...] (terms.py:171-173)
    a, U, V = ops.get_celerite_matrices(ar, acd, bcd, dcd, td, diagd)
    return td, to(c), a, U, V, yd


def device_coeffs_fast(first, count, N, J, device, seed0=721, gap_fraction=0.0, gap=10.0, gaps_per_series=1):
    """(t, diag, y, ac, bc, cc, dc) of device_batch_fast: the coefficient-level view of the same series.
    gap_fraction / gap: as in host_inputs (drawn last); gaps_per_series: that many gaps, each at a row of its own, in
    every series that is hit."""
    import torch

    assert J % 2 == 0
    Jc = J // 2
    gen = torch.Generator(device=device)
    gen.manual_seed(seed0 + int(first))
    f64 = dict(dtype=torch.float64, device=device)
    t = torch.sort(torch.rand((count, N), generator=gen, **f64) * (N / 10.0), dim=1).values.contiguous()
    diag = 0.1 + 0.2 * torch.rand((count, N), generator=gen, **f64)
    xi = 2.0 * torch.rand((count, 1), generator=gen, **f64) - 1.0
    noise = 0.1 * torch.randn((count, N), generator=gen, **f64)
    if gap_fraction > 0.0 and N > 2:
        hit = torch.rand((count, 1), generator=gen, **f64) < gap_fraction
        rows = torch.arange(N, device=device)[None, :]
        for _ in range(gaps_per_series):
            n0 = 1 + (torch.rand((count, 1), generator=gen, **f64) * (N - 1)).long().clamp_(max=N - 2)
            t = t + gap * (hit & (rows >= n0)).to(torch.float64)
        t = t.contiguous()
    y = torch.sin(t) + noise
    k = torch.arange(Jc, **f64)[None, :]
    S0 = 5.0 * 0.7**k
    w0 = 0.1 * 3.0**k * (1.0 + 0.05 * xi)
    Q = 3.45 + k
    f = torch.sqrt(torch.clamp(4.0 * Q**2 - 1.0, min=1e-5))
    ac = (S0 * w0 * Q).contiguous()
    bc = (ac / f).contiguous()
    cc = (0.5 * w0 / Q).contiguous()
    dc = (cc * f).contiguous()
    return t, diag, y, ac, bc, cc, dc


def device_batch_fast(first, count, N, J, device, seed0=721, gap_fraction=0.0, gap=10.0, gaps_per_series=1):
    """Same synthetic distribution as device_batch, but drawn with torch's device generator (seeded with
    seed0 + first) so that very large shards (65536 x 4096) are ready in a fraction of a second.  Used by
    bench.py; parity tests use the numpy recipe (host_inputs) so that the CPU oracle sees identical numbers."""
    import torch

    from . import ops

    t, diag, y, ac, bc, cc, dc = device_coeffs_fast(first, count, N, J, device, seed0, gap_fraction, gap, gaps_per_series)
    ar = torch.zeros((count, 0), dtype=torch.float64, device=device)
    c = torch.repeat_interleave(cc, 2, dim=1).contiguous()
    a, U, V = ops.get_celerite_matrices(ar, ac, bc, dc, t, diag)
    return t, c, a, U, V, y


def timed_steady(fn, reps=9, warm_ms=40.0, what="median"):
    """HIP-event time (ms) of `fn` at the device's STEADY clock: the clock ramps for ~25 ms after any pause of the host
    (profiles/r05_clock_ramp.md: the first kernels after an idle millisecond run up to 35 % slow), so `fn` is first repeated
    until `warm_ms` of device time have gone by, then `reps` calls are timed back to back with events created beforehand.
    Measurement helper of tools/ and tests/ (like the rest of this module: not part of the product path)."""
    import torch

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    one = max(e0.elapsed_time(e1), 1e-3)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for _ in range(min(2000, max(2, int(warm_ms / one) + 1))):
        fn()
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2] if what == "median" else (sum(ts) / len(ts) if what == "mean" else ts[0])
