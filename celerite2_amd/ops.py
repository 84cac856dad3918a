# -*- coding: utf-8 -*-
"""Batched, device-resident operator API over the C-ABI (include/celerite2_amd.h).

Same op names and argument meaning as the reference's backend-op layer
(python/celerite2/definitions.json; jax/ops.py:40-72; pymc/ops.py:38-159) with a
leading batch dimension B of independent series.  All tensors are float64,
contiguous, on one HIP device; t may be (B,N) or shared (N,), c (B,J) or (J,).
Launches go on torch's current stream.  torch is plumbing here (device memory,
streams); the arithmetic is in the gfx950 kernels.
"""
import ctypes

import torch

from . import _lib

__all__ = [
    "factor", "solve_lower", "solve_upper", "matmul_lower", "matmul_upper", "general_matmul_lower",
    "general_matmul_upper", "factor_rev", "solve_lower_rev", "solve_upper_rev", "matmul_lower_rev",
    "matmul_upper_rev", "get_celerite_matrices", "kernel_values", "colsumsq_over_d", "loglik", "loglik_grad", "loglik_grad_workspace", "condition", "dot_tril",
    "kron_loglik", "kron_loglik_grad", "loglik_terms", "loglik_terms_grad",
]

_i64 = ctypes.c_int64


def _p(x):
    return ctypes.c_void_p(0 if x is None else x.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(*xs):
    dev = None
    for x in xs:
        if x is None:
            continue
        if not (x.is_cuda and x.dtype == torch.float64 and x.is_contiguous()):
            raise ValueError("celerite2_amd ops need contiguous float64 tensors on the GPU")
        if dev is None:
            dev = x.device
        elif x.device != dev:
            raise ValueError("celerite2_amd ops need every tensor on the same device")


def _shape(name, x, *allowed):
    """The reference raises "Invalid shape: <name>" for every argument (driver.cpp:40-46,94-99); so does this layer,
    BEFORE any pointer reaches a kernel (the kernels compute their strides from (B, N, J, nrhs) alone)."""
    if x is None:
        return
    if tuple(x.shape) not in allowed:
        raise ValueError("Invalid shape: %s (got %s, expected %s)"
                         % (name, tuple(x.shape), " or ".join(str(a) for a in allowed)))


def _bs(x, per):
    """(tensor, batch stride in elements): 0 when shared by the batch."""
    return 0 if x.dim() == 1 else per


def _dims(U):
    if U.dim() != 3:
        raise ValueError("U must be (B, N, J)")
    return U.shape


def factor(t, c, a, U, V, d=None, W=None, S=None, *, workspace=False):
    """Batched core::factor.  Returns (d, W, flag) or (d, W, S, flag); d may alias a, W may alias V."""
    B, N, J = _dims(U)
    d = torch.empty_like(a) if d is None else d
    W = torch.empty_like(V) if W is None else W
    if workspace and S is None:
        S = torch.empty((B, N, J, J), dtype=torch.float64, device=U.device)
    flag = torch.empty(B, dtype=torch.int32, device=U.device)
    _chk(t, c, a, U, V, d, W, S)
    _shape("t", t, (N,), (B, N)); _shape("c", c, (J,), (B, J)); _shape("a", a, (B, N)); _shape("V", V, (B, N, J))
    _shape("d", d, (B, N)); _shape("W", W, (B, N, J)); _shape("S", S, (B, N, J, J))
    rc = _lib.load().c2_factor(_i64(B), _i64(N), _i64(J), _p(t), _i64(_bs(t, N)), _p(c), _i64(_bs(c, J)), _p(a),
                               _p(U), _p(V), _p(d), _p(W), _p(S), _p(flag), _stream())
    _lib.check(rc, "factor")
    return (d, W, S, flag) if S is not None else (d, W, flag)


def condition(t, c, a, U, V):
    """kappa[b] = max_n a_n / d_n per series (+inf where the factorisation fails), and the factor flag: whether north_star's
    1e-10 agreement with the reference is attainable in float64 for these inputs (include/celerite2_amd.h, c2_condition:
    any evaluation order carries ~0.4 eps kappa^2 of the largest gradient entry).  Returns (kappa, flag)."""
    B, N, J = _dims(U)
    kappa = torch.empty(B, dtype=torch.float64, device=U.device)
    flag = torch.empty(B, dtype=torch.int32, device=U.device)
    _chk(t, c, a, U, V)
    _shape("t", t, (N,), (B, N)); _shape("c", c, (J,), (B, J)); _shape("a", a, (B, N)); _shape("V", V, (B, N, J))
    rc = _lib.load().c2_condition(_i64(B), _i64(N), _i64(J), _p(t), _i64(_bs(t, N)), _p(c), _i64(_bs(c, J)), _p(a),
                                  _p(U), _p(V), _p(kappa), _p(flag), _stream())
    _lib.check(rc, "condition")
    return kappa, flag


def _sweep(name, matmul):
    def op(t, c, U, W, Y, Z=None, F=None, *, workspace=False, zero_z=False):
        B, N, J = _dims(U)
        if Y.dim() != 3:
            raise ValueError("Invalid shape: Y (must be (B, N, nrhs))")
        nrhs = Y.shape[-1]
        if Z is None:
            Z = torch.zeros_like(Y) if matmul else torch.empty_like(Y)
        if workspace and F is None:
            F = torch.empty((B, N, J, nrhs), dtype=torch.float64, device=U.device)
        _chk(t, c, U, W, Y, Z, F)
        _shape("t", t, (N,), (B, N)); _shape("c", c, (J,), (B, J)); _shape("W", W, (B, N, J))
        _shape("Y", Y, (B, N, nrhs)); _shape("Z", Z, (B, N, nrhs)); _shape("F", F, (B, N, J, nrhs))
        args = [_i64(B), _i64(N), _i64(J), _i64(nrhs), _p(t), _i64(_bs(t, N)), _p(c), _i64(_bs(c, J)), _p(U), _p(W),
                _p(Y), _p(Z), _p(F)]
        if matmul:
            args.append(ctypes.c_int(1 if zero_z else 0))
        rc = getattr(_lib.load(), "c2_" + name)(*args, _stream())
        _lib.check(rc, name)
        return (Z, F) if F is not None else Z
    op.__name__ = name
    return op


solve_lower = _sweep("solve_lower", False)
solve_upper = _sweep("solve_upper", False)
matmul_lower = _sweep("matmul_lower", True)
matmul_upper = _sweep("matmul_upper", True)


def _general(name):
    def op(t1, t2, c, U, V, Y, Z=None, F=None, *, workspace=False, zero_z=False):
        B, N, J = _dims(U)
        if V.dim() != 3 or Y.dim() != 3:
            raise ValueError("Invalid shape: V must be (B, M, J) and Y (B, M, nrhs)")
        M, nrhs = V.shape[1], Y.shape[-1]
        if Z is None:
            Z = torch.zeros((B, N, nrhs), dtype=torch.float64, device=U.device)
        if workspace and F is None:
            F = torch.zeros((B, M, J, nrhs), dtype=torch.float64, device=U.device)
        _chk(t1, t2, c, U, V, Y, Z, F)
        _shape("t1", t1, (N,), (B, N)); _shape("t2", t2, (M,), (B, M)); _shape("c", c, (J,), (B, J))
        _shape("V", V, (B, M, J)); _shape("Y", Y, (B, M, nrhs)); _shape("Z", Z, (B, N, nrhs))
        _shape("F", F, (B, M, J, nrhs))
        rc = getattr(_lib.load(), "c2_" + name)(
            _i64(B), _i64(N), _i64(M), _i64(J), _i64(nrhs), _p(t1), _i64(_bs(t1, N)), _p(t2), _i64(_bs(t2, M)), _p(c),
            _i64(_bs(c, J)), _p(U), _p(V), _p(Y), _p(Z), _p(F), ctypes.c_int(1 if zero_z else 0), _stream())
        _lib.check(rc, name)
        return (Z, F) if F is not None else Z
    op.__name__ = name
    return op


general_matmul_lower = _general("general_matmul_lower")
general_matmul_upper = _general("general_matmul_upper")


def factor_rev(t, c, a, U, V, d, W, S, bd, bW):
    """Batched core::factor_rev (reverse.hpp:10-85).  `S` is the workspace `factor(..., workspace=True)` returned.  The
    row-by-row kernel reads every C-th row of it and replays the rows in between from d, W; on small batches of series of
    512 rows and more the reverse pass runs parallel along time and replays ALL its states from d, W --
    `S` is then only read if the device-side verification of that pass fails and the row-by-row kernel recomputes the
    batch behind it.  Either way the result is that of the reference on the same (t, c, U, d, W, bd, bW)."""
    B, N, J = _dims(U)
    dev = U.device
    bt = torch.empty((B, N), dtype=torch.float64, device=dev)
    bc = torch.empty((B, J), dtype=torch.float64, device=dev)
    ba = torch.empty((B, N), dtype=torch.float64, device=dev)
    bU = torch.empty_like(U)
    bV = torch.empty_like(U)
    _chk(t, c, a, U, V, d, W, S, bd, bW)
    _shape("t", t, (N,), (B, N)); _shape("c", c, (J,), (B, J)); _shape("a", a, (B, N)); _shape("V", V, (B, N, J))
    _shape("d", d, (B, N)); _shape("W", W, (B, N, J)); _shape("S", S, (B, N, J, J)); _shape("bd", bd, (B, N))
    _shape("bW", bW, (B, N, J))
    rc = _lib.load().c2_factor_rev(_i64(B), _i64(N), _i64(J), _p(t), _i64(_bs(t, N)), _p(c), _i64(_bs(c, J)), _p(a),
                                   _p(U), _p(V), _p(d), _p(W), _p(S), _p(bd), _p(bW), _p(bt), _p(bc), _p(ba), _p(bU),
                                   _p(bV), _stream())
    _lib.check(rc, "factor_rev")
    return bt, bc, ba, bU, bV


def _sweep_rev(name):
    def op(t, c, U, W, Y, Z, F, bZ):
        B, N, J = _dims(U)
        if Y.dim() != 3:
            raise ValueError("Invalid shape: Y (must be (B, N, nrhs))")
        nrhs = Y.shape[-1]
        dev = U.device
        bt = torch.empty((B, N), dtype=torch.float64, device=dev)
        bc = torch.empty((B, J), dtype=torch.float64, device=dev)
        bU = torch.empty_like(U)
        bW = torch.empty_like(U)
        bY = torch.empty_like(Y)
        _chk(t, c, U, W, Y, Z, F, bZ)
        _shape("t", t, (N,), (B, N)); _shape("c", c, (J,), (B, J)); _shape("W", W, (B, N, J))
        _shape("Y", Y, (B, N, nrhs)); _shape("Z", Z, (B, N, nrhs)); _shape("F", F, (B, N, J, nrhs))
        _shape("bZ", bZ, (B, N, nrhs))
        rc = getattr(_lib.load(), "c2_" + name)(
            _i64(B), _i64(N), _i64(J), _i64(nrhs), _p(t), _i64(_bs(t, N)), _p(c), _i64(_bs(c, J)), _p(U), _p(W), _p(Y),
            _p(Z), _p(F), _p(bZ), _p(bt), _p(bc), _p(bU), _p(bW), _p(bY), _stream())
        _lib.check(rc, name)
        return bt, bc, bU, bW, bY
    op.__name__ = name
    return op


solve_lower_rev = _sweep_rev("solve_lower_rev")
solve_upper_rev = _sweep_rev("solve_upper_rev")
matmul_lower_rev = _sweep_rev("matmul_lower_rev")
matmul_upper_rev = _sweep_rev("matmul_upper_rev")


def get_celerite_matrices(ar, ac, bc, dc, x, diag):
    """Batched driver.get_celerite_matrices.  ar (Jr,)|(B,Jr); ac,bc,dc (Jc,)|(B,Jc); x (N,)|(B,N); diag (B,N)."""
    if diag.dim() != 2:
        raise ValueError("Invalid shape: diag (must be (B, N))")
    B, N = diag.shape
    Jr, Jc = ar.shape[-1], ac.shape[-1]
    J = Jr + 2 * Jc
    batched = ar.dim() == 2 or ac.dim() == 2
    if batched and (ar.dim() != 2 or ac.dim() != 2 or bc.dim() != 2 or dc.dim() != 2):
        raise ValueError("coefficients must be all shared or all per-series")
    dev = diag.device
    a = torch.empty((B, N), dtype=torch.float64, device=dev)
    U = torch.empty((B, N, J), dtype=torch.float64, device=dev)
    V = torch.empty((B, N, J), dtype=torch.float64, device=dev)
    _chk(ar, ac, bc, dc, x, diag)
    _shape("x", x, (N,), (B, N))
    _shape("ar", ar, (Jr,), (B, Jr))
    for nm, v in (("ac", ac), ("bc", bc), ("dc", dc)):
        _shape(nm, v, (Jc,), (B, Jc))
    rc = _lib.load().c2_get_celerite_matrices(
        _i64(B), _i64(N), _i64(Jr), _i64(Jc), _p(ar if Jr else None), _p(ac if Jc else None), _p(bc if Jc else None),
        _p(dc if Jc else None), ctypes.c_int(1 if batched else 0), _p(x), _i64(_bs(x, N)), _p(diag), _p(a), _p(U),
        _p(V), _stream())
    _lib.check(rc, "get_celerite_matrices")
    return a, U, V


def kernel_values(ar, cr, ac, bc, cc, dc, t1, t2, B=None):
    """K[b, n, m] = k(t1[b, n] - t2[b, m]) (terms.py:58-79 on two grids): coefficients (Jr,)|(B,Jr) / (Jc,)|(B,Jc), t1 (N,)|(B,N),
    t2 (M,)|(B,M); B is taken from whichever argument carries it (or the keyword when everything is shared)."""
    Jr, Jc = ar.shape[-1], ac.shape[-1]
    batched = ar.dim() == 2 or ac.dim() == 2
    if batched and any(v.dim() != 2 for v in (ar, cr, ac, bc, cc, dc)):
        raise ValueError("coefficients must be all shared or all per-series")
    for v in (t1, t2, ar, ac):
        if v.dim() == 2:
            B = v.shape[0] if B is None else B
    if B is None:
        B = 1
    N, M = t1.shape[-1], t2.shape[-1]
    _chk(ar, cr, ac, bc, cc, dc, t1, t2)
    _shape("t1", t1, (N,), (B, N)); _shape("t2", t2, (M,), (B, M))
    for nm, v, w in (("ar", ar, Jr), ("cr", cr, Jr), ("ac", ac, Jc), ("bc", bc, Jc), ("cc", cc, Jc), ("dc", dc, Jc)):
        _shape(nm, v, (w,), (B, w))
    K = torch.empty((B, N, M), dtype=torch.float64, device=t1.device)
    rc = _lib.load().c2_kernel_values(
        _i64(B), _i64(N), _i64(M), _i64(Jr), _i64(Jc), _p(ar if Jr else None), _p(cr if Jr else None), _p(ac if Jc else None),
        _p(bc if Jc else None), _p(cc if Jc else None), _p(dc if Jc else None), ctypes.c_int(1 if batched else 0), _p(t1),
        _i64(_bs(t1, N)), _p(t2), _i64(_bs(t2, M)), _p(K), _stream())
    _lib.check(rc, "kernel_values")
    return K


def colsumsq_over_d(Z, d):
    """out[b, m] = sum_n Z[b, n, m]^2 / d[b, n]  (Z (B, N, M), d (B, N)): the quadratic form of the predictive variance from the
    lower solve alone (c2_colsumsq_over_d; core.py:134-140)."""
    if Z.dim() != 3:
        raise ValueError("Invalid shape: Z (must be (B, N, M))")
    B, N, M = Z.shape
    _chk(Z, d)
    _shape("d", d, (B, N))
    out = torch.empty((B, M), dtype=torch.float64, device=Z.device)
    rc = _lib.load().c2_colsumsq_over_d(_i64(B), _i64(N), _i64(M), _p(Z), _p(d), _p(out), _stream())
    _lib.check(rc, "colsumsq_over_d")
    return out


def _loglik_shapes(B, N, J, t, c, a, V, y):
    _shape("t", t, (N,), (B, N)); _shape("c", c, (J,), (B, J)); _shape("a", a, (B, N)); _shape("V", V, (B, N, J))
    _shape("y", y, (B, N))


def loglik(t, c, a, U, V, y):
    """Fused batched log-likelihood.  Returns (ll (B,), flag (B,) int32)."""
    B, N, J = _dims(U)
    ll = torch.empty(B, dtype=torch.float64, device=U.device)
    flag = torch.empty(B, dtype=torch.int32, device=U.device)
    _chk(t, c, a, U, V, y)
    _loglik_shapes(B, N, J, t, c, a, V, y)
    rc = _lib.load().c2_loglik(_i64(B), _i64(N), _i64(J), _p(t), _i64(_bs(t, N)), _p(c), _i64(_bs(c, J)), _p(a),
                               _p(U), _p(V), _p(y), _p(ll), _p(flag), _stream())
    _lib.check(rc, "loglik")
    return ll, flag


def loglik_grad_workspace(B, N, J, device):
    nbytes = _lib.load().c2_loglik_grad_workspace_bytes(B, N, J)
    return torch.empty(nbytes // 8, dtype=torch.float64, device=device)


def loglik_grad_buffers(t, c, a, U, V, y, *, candidates=3):
    """Workspace and gradient arrays for repeated `loglik_grad(..., work=, out=)` calls on one shape, PLACED by
    measurement.  A chip-filling step streams a dozen 2 - 16 GiB arrays at once, and WHERE in the 288 GB of HBM the
    workspace and the gradient arrays lie relative to the inputs moves its time by 6 - 10 % -- deterministically: the same
    process re-allocating the same arrays behind a spacer of a few tens of GiB switches between 28.2 and 31.5 ms for
    65536 x 4096 x 8, shifts of MiB change nothing, a plain copy between 16-GiB buffers does not care
    (profiles/r04_headline_spread.md).  So: time one
    step on the allocator's own placement, then on fresh allocations behind spacers of 24 and 48 GiB (more with
    `candidates`), and keep the fastest.  Returns (work, out, report) -- `report` lists every candidate's time."""
    B, N, J = _dims(U)
    dev = U.device

    def fresh():
        return (loglik_grad_workspace(B, N, J, dev),
                (torch.empty((B, N), dtype=torch.float64, device=dev), torch.empty((B, J), dtype=torch.float64, device=dev),
                 torch.empty((B, N), dtype=torch.float64, device=dev), torch.empty_like(U), torch.empty_like(U),
                 torch.empty((B, N), dtype=torch.float64, device=dev)))

    def one_step_ms(w_, o_):
        for _ in range(2):
            loglik_grad(t, c, a, U, V, y, work=w_, out=o_)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        loglik_grad(t, c, a, U, V, y, work=w_, out=o_)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    work, out = fresh()
    cand = [{"spacer_GiB": 0, "ms": one_step_ms(work, out)}]
    best_ms = cand[0]["ms"]
    # spacers scaled to the memory that is actually free (24 / 48 GiB on a 288-GB part holding the bench shape)
    free_gib = torch.cuda.mem_get_info(dev)[0] / 2**30
    need_gib = (work.numel() * 8 + sum(o.numel() for o in out) * 8) / 2**30
    room = max(0.0, free_gib - need_gib - 2.0)
    for gib in [g for g in (24, 48, 12, 72, 96) if g <= room][:max(0, candidates - 1)]:
        sp = w_ = o_ = None
        try:
            sp = torch.empty(gib * 2**30, dtype=torch.uint8, device=dev)
            w_, o_ = fresh()
        except RuntimeError:   # (not enough memory left for a second set behind this spacer)
            del sp, w_, o_
            torch.cuda.empty_cache()
            break
        del sp                 # (only the position of the arrays matters: the spacer itself goes back at once)
        ms_ = one_step_ms(w_, o_)
        cand.append({"spacer_GiB": gib, "ms": ms_})
        if ms_ < best_ms:
            best_ms, work, out = ms_, w_, o_
        del w_, o_
        torch.cuda.empty_cache()   # the next candidate must not simply get the loser's blocks back
    return work, out, {"candidates": cand, "chosen_ms": best_ms,
                       "note": "setup, untimed: one step per candidate placement of the workspace and the gradient arrays "
                               "(behind spacers of tens of GiB; profiles/r04_headline_spread.md)"}


def loglik_grad(t, c, a, U, V, y, *, work=None, out=None):
    """Fused batched log-likelihood + gradient.  Returns (ll, (bt, bc, ba, bU, bV, by), flag)."""
    B, N, J = _dims(U)
    dev = U.device
    if work is None:
        work = loglik_grad_workspace(B, N, J, dev)
    if out is None:
        out = (torch.empty((B, N), dtype=torch.float64, device=dev), torch.empty((B, J), dtype=torch.float64, device=dev),
               torch.empty((B, N), dtype=torch.float64, device=dev), torch.empty_like(U), torch.empty_like(U),
               torch.empty((B, N), dtype=torch.float64, device=dev))
    bt, bc, ba, bU, bV, by = out
    ll = torch.empty(B, dtype=torch.float64, device=dev)
    flag = torch.empty(B, dtype=torch.int32, device=dev)
    _chk(t, c, a, U, V, y, bt, bc, ba, bU, bV, by)
    _loglik_shapes(B, N, J, t, c, a, V, y)
    _shape("bt", bt, (B, N)); _shape("bc", bc, (B, J)); _shape("ba", ba, (B, N)); _shape("bU", bU, (B, N, J))
    _shape("bV", bV, (B, N, J)); _shape("by", by, (B, N))
    rc = _lib.load().c2_loglik_grad(_i64(B), _i64(N), _i64(J), _p(t), _i64(_bs(t, N)), _p(c), _i64(_bs(c, J)), _p(a),
                                    _p(U), _p(V), _p(y), _p(ll), _p(bt), _p(bc), _p(ba), _p(bU), _p(bV), _p(by),
                                    _p(flag), _p(work), ctypes.c_size_t(work.numel() * 8), _stream())
    _lib.check(rc, "loglik_grad")
    return ll, out, flag


def dot_tril(t, c, U, W, d, Y, Z=None):
    """Z = L sqrt(D) Y (numpy.py:100-102).  Y may be passed as Z for in-place use."""
    B, N, J = _dims(U)
    if Y.dim() != 3:
        raise ValueError("Invalid shape: Y (must be (B, N, nrhs))")
    nrhs = Y.shape[-1]
    Z = torch.empty_like(Y) if Z is None else Z
    _chk(t, c, U, W, d, Y, Z)
    _shape("t", t, (N,), (B, N)); _shape("c", c, (J,), (B, J)); _shape("W", W, (B, N, J)); _shape("d", d, (B, N))
    _shape("Y", Y, (B, N, nrhs)); _shape("Z", Z, (B, N, nrhs))
    rc = _lib.load().c2_dot_tril(_i64(B), _i64(N), _i64(J), _i64(nrhs), _p(t), _i64(_bs(t, N)), _p(c),
                                 _i64(_bs(c, J)), _p(U), _p(W), _p(d), _p(Y), _p(Z), _stream())
    _lib.check(rc, "dot_tril")
    return Z


_KRON_METHODS = {"collapsed": 0, "interleaved": 1}


def _kron_args(t, c, a, U, V, alpha, diag, y, method):
    B, N, J = _dims(U)
    if diag.dim() != 3:
        raise ValueError("Invalid shape: diag (must be (B, N, M))")
    M = diag.shape[-1]
    if method not in _KRON_METHODS:
        raise ValueError("method must be 'collapsed' or 'interleaved'")
    _chk(t, c, a, U, V, alpha, diag, y)
    _shape("t", t, (N,), (B, N)); _shape("c", c, (J,), (B, J)); _shape("a", a, (B, N)); _shape("V", V, (B, N, J))
    _shape("alpha", alpha, (M,), (B, M)); _shape("diag", diag, (B, N, M)); _shape("y", y, (B, N, M))
    return B, N, M, J, _KRON_METHODS[method]


def kron_loglik(t, c, a, U, V, alpha, diag, y, *, method="collapsed", work=None):
    """2-D (multi-band) log-likelihood, rank-1 band covariance K = T (x) alpha alpha^T + diag (extension; the
    reference has no 2-D code).  (t, c, a, U, V): celerite matrices of the EPOCH grid built with zero white noise
    (a = k(0)); alpha (M,)|(B,M); diag, y (B,N,M).  Returns (ll (B,), flag (B,) int32)."""
    B, N, M, J, meth = _kron_args(t, c, a, U, V, alpha, diag, y, method)
    lib = _lib.load()
    nbytes = lib.c2_kron_loglik_workspace_bytes(B, N, M, J, meth, 0)
    if work is None or work.numel() * 8 < nbytes:
        work = torch.empty(nbytes // 8, dtype=torch.float64, device=U.device)
    ll = torch.empty(B, dtype=torch.float64, device=U.device)
    flag = torch.empty(B, dtype=torch.int32, device=U.device)
    rc = lib.c2_kron_loglik(_i64(B), _i64(N), _i64(M), _i64(J), _p(t), _i64(_bs(t, N)), _p(c), _i64(_bs(c, J)), _p(a),
                            _p(U), _p(V), _p(alpha), _i64(_bs(alpha, M)), _p(diag), _p(y), _p(ll), _p(flag),
                            ctypes.c_int(meth), _p(work), ctypes.c_size_t(work.numel() * 8), _stream())
    _lib.check(rc, "kron_loglik")
    return ll, flag


def kron_loglik_grad(t, c, a, U, V, alpha, diag, y, *, method="collapsed", work=None):
    """kron_loglik + reverse-mode gradient.  Returns (ll, (bt, bc, ba, bU, bV, balpha, bdiag, by), flag); balpha is
    per series (B, M) also for a shared alpha.  (ba, bU, bV) are the partials of the method's own parametrisation
    ("collapsed": T_nn = a_n, the literal Kronecker definition; "interleaved": same-epoch cross-band terms through
    U_n.V_n); the total derivatives bU + ba V, bV + ba U along a = U.V agree."""
    B, N, M, J, meth = _kron_args(t, c, a, U, V, alpha, diag, y, method)
    lib = _lib.load()
    dev = U.device
    nbytes = lib.c2_kron_loglik_workspace_bytes(B, N, M, J, meth, 1)
    if work is None or work.numel() * 8 < nbytes:
        work = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
    f64 = dict(dtype=torch.float64, device=dev)
    bt, bc, ba = torch.empty((B, N), **f64), torch.empty((B, J), **f64), torch.empty((B, N), **f64)
    bU, bV = torch.empty_like(U), torch.empty_like(U)
    balpha, bdiag, by = torch.empty((B, M), **f64), torch.empty_like(diag), torch.empty_like(y)
    ll = torch.empty(B, **f64)
    flag = torch.empty(B, dtype=torch.int32, device=dev)
    rc = lib.c2_kron_loglik_grad(_i64(B), _i64(N), _i64(M), _i64(J), _p(t), _i64(_bs(t, N)), _p(c), _i64(_bs(c, J)),
                                 _p(a), _p(U), _p(V), _p(alpha), _i64(_bs(alpha, M)), _p(diag), _p(y), _p(ll), _p(bt),
                                 _p(bc), _p(ba), _p(bU), _p(bV), _p(balpha), _p(bdiag), _p(by), _p(flag),
                                 ctypes.c_int(meth), _p(work), ctypes.c_size_t(work.numel() * 8), _stream())
    _lib.check(rc, "kron_loglik_grad")
    return ll, (bt, bc, ba, bU, bV, balpha, bdiag, by), flag


def _terms_args(ar, cr, ac, bc, cc, dc, x, diag, y):
    if diag.dim() != 2:
        raise ValueError("Invalid shape: diag (must be (B, N))")
    B, N = diag.shape
    Jr, Jc = ar.shape[-1], ac.shape[-1]
    batched = any(v.dim() == 2 for v in (ar, cr, ac, bc, cc, dc))
    if batched and not all(v.dim() == 2 for v in (ar, cr, ac, bc, cc, dc)):
        raise ValueError("coefficients must be all shared or all per-series")
    _chk(ar, cr, ac, bc, cc, dc, x, diag, y)
    _shape("x", x, (N,), (B, N)); _shape("y", y, (B, N))
    for nm, v, w in (("ar", ar, Jr), ("cr", cr, Jr), ("ac", ac, Jc), ("bc", bc, Jc), ("cc", cc, Jc), ("dc", dc, Jc)):
        _shape(nm, v, (w,), (B, w))
    return B, N, Jr, Jc, batched


def _coef_ptrs(ar, cr, ac, bc, cc, dc, Jr, Jc):
    return [_p(ar if Jr else None), _p(cr if Jr else None), _p(ac if Jc else None), _p(bc if Jc else None),
            _p(cc if Jc else None), _p(dc if Jc else None)]


def loglik_terms(ar, cr, ac, bc, cc, dc, x, diag, y, *, work=None):
    """Batched log-likelihood straight from the celerite coefficients (terms.py:117-177 + numpy.py:84-109).
    ar, cr (Jr,)|(B,Jr); ac, bc, cc, dc (Jc,)|(B,Jc); x (N,)|(B,N); diag, y (B,N).  Returns (ll, flag)."""
    B, N, Jr, Jc, batched = _terms_args(ar, cr, ac, bc, cc, dc, x, diag, y)
    lib = _lib.load()
    nbytes = lib.c2_loglik_terms_workspace_bytes(B, N, Jr, Jc, 0)
    if work is None or work.numel() * 8 < nbytes:
        work = torch.empty(nbytes // 8, dtype=torch.float64, device=diag.device)
    ll = torch.empty(B, dtype=torch.float64, device=diag.device)
    flag = torch.empty(B, dtype=torch.int32, device=diag.device)
    rc = lib.c2_loglik_terms(_i64(B), _i64(N), _i64(Jr), _i64(Jc), *_coef_ptrs(ar, cr, ac, bc, cc, dc, Jr, Jc),
                             ctypes.c_int(1 if batched else 0), _p(x), _i64(_bs(x, N)), _p(diag), _p(y), _p(ll),
                             _p(flag), _p(work), ctypes.c_size_t(work.numel() * 8), _stream())
    _lib.check(rc, "loglik_terms")
    return ll, flag


def loglik_terms_workspace(B, N, Jr, Jc, device, grad=True):
    """Scratch for loglik_terms[_grad] (reusable across calls of the same shape)."""
    lib = _lib.load()
    return torch.empty(lib.c2_loglik_terms_workspace_bytes(B, N, Jr, Jc, 1 if grad else 0) // 8, dtype=torch.float64,
                       device=device)


def loglik_terms_grad(ar, cr, ac, bc, cc, dc, x, diag, y, *, work=None, out=None):
    """loglik_terms + reverse-mode gradient w.r.t. every input: returns
    (ll, (bar, bcr, bac, bbc, bcc, bdc, bx, bdiag, by), flag); coefficient gradients are per series (B, Jr|Jc) also when
    the coefficients are shared by the batch.  `out`: a previous call's nine gradient tensors to write into."""
    B, N, Jr, Jc, batched = _terms_args(ar, cr, ac, bc, cc, dc, x, diag, y)
    lib = _lib.load()
    dev = diag.device
    nbytes = lib.c2_loglik_terms_workspace_bytes(B, N, Jr, Jc, 1)
    if work is None or work.numel() * 8 < nbytes:
        work = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
    f64 = dict(dtype=torch.float64, device=dev)
    if out is not None:
        outs = list(out)
        shapes = [(B, Jr)] * 2 + [(B, Jc)] * 4 + [(B, N)] * 3
        if len(outs) != 9 or any(tuple(o.shape) != sh or o.dtype != torch.float64 or o.device != dev or not o.is_contiguous()
                                 for o, sh in zip(outs, shapes)):
            raise ValueError("Invalid shape: out (nine contiguous float64 tensors as returned by loglik_terms_grad)")
    else:
        outs = [torch.empty((B, Jr), **f64), torch.empty((B, Jr), **f64)] + [torch.empty((B, Jc), **f64) for _ in range(4)] + \
               [torch.empty((B, N), **f64) for _ in range(3)]
    ll = torch.empty(B, **f64)
    flag = torch.empty(B, dtype=torch.int32, device=dev)
    optr = [_p(o if o.numel() else None) for o in outs[:6]] + [_p(o) for o in outs[6:]]
    rc = lib.c2_loglik_terms_grad(_i64(B), _i64(N), _i64(Jr), _i64(Jc), *_coef_ptrs(ar, cr, ac, bc, cc, dc, Jr, Jc),
                                  ctypes.c_int(1 if batched else 0), _p(x), _i64(_bs(x, N)), _p(diag), _p(y), _p(ll),
                                  *optr, _p(flag), _p(work), ctypes.c_size_t(work.numel() * 8), _stream())
    _lib.check(rc, "loglik_terms_grad")
    return ll, tuple(outs), flag


def _loglik_grad_composite(t, c, a, U, V, y):
    """Internal cross-check: the literal op chain (factor_fwd -> solve_lower_fwd -> seeds -> solve_lower_rev ->
    factor_rev) with S/F workspaces materialised in HBM, as the reference's autodiff frontends run it."""
    B, N, J = _dims(U)
    dev = U.device
    lib = _lib.load()
    lib.c2_loglik_grad_composite_workspace_bytes.restype = ctypes.c_size_t
    lib.c2_loglik_grad_composite_workspace_bytes.argtypes = [ctypes.c_int64] * 3
    work = torch.empty(lib.c2_loglik_grad_composite_workspace_bytes(B, N, J) // 8, dtype=torch.float64, device=dev)
    out = (torch.empty((B, N), dtype=torch.float64, device=dev), torch.empty((B, J), dtype=torch.float64, device=dev),
           torch.empty((B, N), dtype=torch.float64, device=dev), torch.empty_like(U), torch.empty_like(U),
           torch.empty((B, N), dtype=torch.float64, device=dev))
    bt, bc, ba, bU, bV, by = out
    ll = torch.empty(B, dtype=torch.float64, device=dev)
    flag = torch.empty(B, dtype=torch.int32, device=dev)
    rc = lib.c2_loglik_grad_composite(_i64(B), _i64(N), _i64(J), _p(t), _i64(_bs(t, N)), _p(c), _i64(_bs(c, J)), _p(a),
                                      _p(U), _p(V), _p(y), _p(ll), _p(bt), _p(bc), _p(ba), _p(bU), _p(bV), _p(by),
                                      _p(flag), _p(work), ctypes.c_size_t(work.numel() * 8), _stream())
    _lib.check(rc, "loglik_grad_composite")
    return ll, out, flag
