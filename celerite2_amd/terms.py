# -*- coding: utf-8 -*-
"""Kernel terms -> celerite matrices (c, a, U, V), on the device.

A minimal mirror of the reference's term algebra (python/celerite2/terms.py): each term exposes
`get_coefficients() -> (ar, cr, ac, bc, cc, dc)`; sums concatenate coefficient lists (terms.py:233-235);
`get_celerite_matrices(x, diag)` fills c with the reference's interleaved layout (terms.py:171-173) and runs
the c2_get_celerite_matrices kernel (driver.cpp:422-477) for (a, U, V).  Parameters may be python floats
(one kernel shared by the batch) or 1-D arrays of length B (one hyper-parameter draw per series).
"""
import numpy as np

__all__ = ["Term", "TermSum", "RealTerm", "ComplexTerm", "SHOTerm", "Matern32Term", "RotationTerm"]


def _col(x):
    return np.atleast_1d(np.asarray(x, dtype=np.float64))


class Term:
    """Base class: subclasses implement get_coefficients() returning six arrays of shape (Jr,)|(B,Jr) / (Jc,)|(B,Jc)."""

    def get_coefficients(self):
        raise NotImplementedError

    def __add__(self, other):
        return TermSum(self, other)

    @property
    def width(self):
        ar, _, ac, _, _, _ = self.get_coefficients()
        return ar.shape[-1] + 2 * ac.shape[-1]

    def get_value(self, tau):
        """k(tau) (terms.py:58-79), numpy, for dense cross-checks; shared coefficients only."""
        ar, cr, ac, bc, cc, dc = self.get_coefficients()
        tau = np.abs(np.asarray(tau, dtype=np.float64))[..., None]
        k = np.sum(ar * np.exp(-cr * tau), axis=-1)
        return k + np.sum(np.exp(-cc * tau) * (ac * np.cos(dc * tau) + bc * np.sin(dc * tau)), axis=-1)

    def _dev_coefs(self, device, nb=None):
        """The six coefficient arrays (+ the interleaved c of terms.py:171-173 as a seventh) as device tensors, ONE upload,
        cached on the term while the coefficient VALUES stay what they were (parameters are plain attributes a caller may
        change): a pageable host-to-device copy is a synchronisation, and predict() used to make ten of them per call.
        Per-series coefficients keep their (B, .) shape; with any of them per-series the shared ones are broadcast to
        (nb or their own B, .).  Returns (tensors, batched)."""
        import torch

        coefs = [np.asarray(v, dtype=np.float64) for v in self.get_coefficients()]
        batched = any(v.ndim == 2 for v in coefs)
        if batched:
            B = max([v.shape[0] for v in coefs if v.ndim == 2] + [nb or 0])
            coefs = [np.broadcast_to(v, (B, v.shape[-1])) if v.ndim == 1 else v for v in coefs]
        ar, cr, ac, bc, cc, dc = coefs
        Jr, Jc = ar.shape[-1], ac.shape[-1]
        c = np.empty(cr.shape[:-1] + (Jr + 2 * Jc,))
        c[..., :Jr] = cr       # c = [cr, cc0, cc0, cc1, cc1, ...]  (terms.py:171-173)
        c[..., Jr::2] = cc
        c[..., Jr + 1::2] = cc
        host = [np.ascontiguousarray(v) for v in coefs + [c]]
        key = (str(device), batched)
        cache = self.__dict__.get("_dev_cache")
        if cache is not None and cache[0] == key and all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(cache[1], host)):
            return cache[2], batched
        flat = torch.from_numpy(np.concatenate([h.ravel() for h in host] + [np.zeros(1)])).to(device)
        out, at = [], 0
        for h in host:
            out.append(flat[at:at + h.size].view(h.shape))
            at += h.size
        self.__dict__["_dev_cache"] = (key, host, out)
        return out, batched

    def get_value_device(self, tau):
        """k(tau) (terms.py:58-79) on the device: `tau` a float64 tensor whose LEADING axis is the batch (B, ...);
        coefficients shared by the batch or one row per series.  What the conditional distribution needs for its
        cross-covariances `KxsT`, `k(0)` and `k(xs - xs')` (core.py:57-66, 134-150)."""
        import torch

        host = self.get_coefficients()
        batched = any(np.ndim(v) == 2 for v in host)
        (ar, cr, ac, bc, cc, dc) = self._dev_coefs(tau.device, tau.shape[0])[0][:6] if batched else host
        tau = tau.abs()
        extra = (1,) * (tau.dim() - 1)

        def co(v, j):   # coefficient j broadcastable against tau: a python float (shared) or a (B, 1, ..) device view
            if not batched:
                return float(v[j])
            return v[:, j].reshape((-1,) + extra)

        k = torch.zeros_like(tau)
        for j in range(ar.shape[-1]):
            k = k + co(ar, j) * torch.exp(-co(cr, j) * tau)
        for j in range(ac.shape[-1]):
            arg = co(dc, j) * tau
            k = k + torch.exp(-co(cc, j) * tau) * (co(ac, j) * torch.cos(arg) + co(bc, j) * torch.sin(arg))
        return k

    def get_value_grid(self, t1, t2, B=None):
        """k(t1[n] - t2[m]) on two grids, (B, N, M), by the c2_kernel_values kernel (ops.kernel_values): what the
        conditional distribution needs for KxsT and for the prior covariance of the prediction grid (core.py:46-54, 142-148)."""
        import torch

        from . import ops

        dev, _ = self._dev_coefs(t1.device, B)
        return ops.kernel_values(*dev[:6], t1.contiguous(), t2.contiguous(), B=B)

    def get_celerite_matrices(self, x, diag):
        """x (N,)|(B,N), diag (B,N) torch float64 device tensors -> (c, a, U, V) device tensors."""
        import torch

        from . import ops

        (ar, cr, ac, bc, cc, dc, c), _ = self._dev_coefs(diag.device, diag.shape[0])
        a, U, V = ops.get_celerite_matrices(ar, ac, bc, dc, x, diag)
        return c, a, U, V


class TermSum(Term):
    def __init__(self, *terms):
        self.terms = []
        for t in terms:
            self.terms += t.terms if isinstance(t, TermSum) else [t]

    def get_coefficients(self):
        parts = [t.get_coefficients() for t in self.terms]
        nb = max(max(v.shape[0] if v.ndim == 2 else 0 for v in p) for p in parts)

        def cat(i):
            arrs = [p[i] for p in parts]
            if nb:
                arrs = [np.broadcast_to(v, (nb, v.shape[-1])) if v.ndim == 1 else v for v in arrs]
            return np.concatenate(arrs, axis=-1)

        return tuple(cat(i) for i in range(6))


def _stack(*cols):
    cols = [_col(v) for v in cols]
    n = max(v.shape[0] for v in cols)
    if n == 1:
        return np.array([float(v[0]) for v in cols])
    return np.stack([np.broadcast_to(v, (n,)) for v in cols], axis=1)


_E = np.empty(0)


def _empty_like(v):
    return np.empty(v.shape[:-1] + (0,))


class RealTerm(Term):
    """k(tau) = a exp(-c tau)  (terms.py:515-521)."""

    def __init__(self, *, a, c):
        self.a, self.c = a, c

    def get_coefficients(self):
        ar, cr = _stack(self.a), _stack(self.c)
        e = _empty_like(ar)
        return ar, cr, e, e, e, e


class ComplexTerm(Term):
    """k(tau) = exp(-c tau) (a cos(d tau) + b sin(d tau))  (terms.py:554-569)."""

    def __init__(self, *, a, b, c, d):
        self.a, self.b, self.c, self.d = a, b, c, d

    def get_coefficients(self):
        ac, bc, cc, dc = _stack(self.a), _stack(self.b), _stack(self.c), _stack(self.d)
        e = _empty_like(ac)
        return e, e, ac, bc, cc, dc


class SHOTerm(Term):
    """Stochastically driven damped harmonic oscillator (terms.py:641-691).  Q is a python float
    (the over/under-damped branch is chosen once); S0 and w0 may be per-series arrays."""

    def __init__(self, *, S0=None, w0=None, Q=None, sigma=None, rho=None, tau=None, eps=1e-5):
        if w0 is None:
            w0 = 2 * np.pi / _col(rho)
        if Q is None:
            Q = 0.5 * np.asarray(w0, dtype=np.float64) * tau
        # One over/under-damped branch serves the whole batch, so Q must be ONE number: a per-series Q (given, or
        # derived from a per-series rho / w0 and tau) would be truncated silently -- refuse it instead.
        Qv = np.unique(np.ravel(np.asarray(Q, dtype=np.float64)))
        if Qv.size != 1:
            raise ValueError("SHOTerm: Q must be a scalar shared by the batch (got %d distinct values); "
                             "pass a scalar Q, or scalar rho/w0 together with tau" % Qv.size)
        Q = float(Qv[0])
        if S0 is None:
            S0 = _col(sigma) ** 2 / (_col(w0) * Q)  # the same Q that is stored and used below
        self.S0, self.w0, self.Q, self.eps = S0, w0, Q, float(eps)

    def get_coefficients(self):
        S0, w0, Q = _col(self.S0), _col(self.w0), self.Q
        if Q < 0.5:
            f = np.sqrt(max(1.0 - 4.0 * Q**2, self.eps))
            ar = _stack(0.5 * S0 * w0 * Q * (1.0 + 1.0 / f), 0.5 * S0 * w0 * Q * (1.0 - 1.0 / f))
            cr = _stack(0.5 * w0 / Q * (1.0 - f), 0.5 * w0 / Q * (1.0 + f))
            e = _empty_like(ar)
            return ar, cr, e, e, e, e
        f = np.sqrt(max(4.0 * Q**2 - 1.0, self.eps))
        a = S0 * w0 * Q
        c = 0.5 * w0 / Q
        ac, bc, cc, dc = _stack(a), _stack(a / f), _stack(c), _stack(c * f)
        e = _empty_like(ac)
        return e, e, ac, bc, cc, dc


class Matern32Term(Term):
    """Approximate Matern-3/2 (terms.py:729-745)."""

    def __init__(self, *, sigma, rho, eps=0.01):
        self.sigma, self.rho, self.eps = sigma, rho, float(eps)

    def get_coefficients(self):
        w0 = np.sqrt(3.0) / _col(self.rho)
        S0 = _col(self.sigma) ** 2 / w0
        ac, bc, cc, dc = _stack(w0 * S0), _stack(w0 * w0 * S0 / self.eps), _stack(w0), _stack(np.full_like(w0, self.eps))
        e = _empty_like(ac)
        return e, e, ac, bc, cc, dc


class RotationTerm(TermSum):
    """Mixture of two SHO terms at period and period/2 (terms.py:791-812)."""

    def __init__(self, *, sigma, period, Q0, dQ, f):
        amp = float(sigma) ** 2 / (1 + float(f))
        Q1 = 0.5 + Q0 + dQ
        w1 = 4 * np.pi * Q1 / (period * np.sqrt(4 * Q1**2 - 1))
        S1 = amp / (w1 * Q1)
        Q2 = 0.5 + Q0
        w2 = 8 * np.pi * Q2 / (period * np.sqrt(4 * Q2**2 - 1))
        S2 = f * amp / (w2 * Q2)
        super().__init__(SHOTerm(S0=S1, w0=w1, Q=Q1), SHOTerm(S0=S2, w0=w2, Q=Q2))
