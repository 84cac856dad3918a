# -*- coding: utf-8 -*-
"""Thin batched GaussianProcess frontend over the device ops.

Reproduces, for a batch of B independent series, what the reference's numpy backend does around the ops
(python/celerite2/numpy.py:66-121 and core.py:262-501): compute -> factor; log_likelihood -> solve_lower +
reductions; apply_inverse -> solve_lower, /d, solve_upper; dot_tril; sample; the conditional distribution
(core.py:9-150, numpy.py:14-32): mean at new coordinates via general_matmul_lower/upper, variance and covariance
via apply_inverse on the N x M cross-covariance (solves with M right-hand sides).  Everything stays on the GPU.
`log_likelihood_and_grad` exposes the fused kernels' gradients w.r.t. (t, c, a, U, V, y).
"""
import math

import torch

from . import ops

__all__ = ["GaussianProcess", "ConditionalDistribution", "LinAlgError"]


class LinAlgError(Exception):
    pass


class GaussianProcess:
    def __init__(self, kernel, t=None, *, mean=0.0, **kwargs):
        self.kernel = kernel
        self.mean = float(mean)
        self._t = None
        if t is not None:
            self.compute(t, **kwargs)

    # -- core.py:262-310 + numpy.py:66-92 -------------------------------------------------------------
    def compute(self, t, *, yerr=None, diag=None, check_sorted=True, quiet=False):
        if t.dim() not in (1, 2):
            raise ValueError("The input coordinates must be (N,) or (B, N)")
        if check_sorted and bool((t[..., 1:] < t[..., :-1]).any()):
            raise ValueError("The input coordinates must be sorted")
        if yerr is not None and diag is not None:
            raise ValueError("only one of 'diag' and 'yerr' can be provided")
        if yerr is not None:
            diag = yerr**2
        if diag is None:
            raise ValueError("'diag' or 'yerr' (B, N) is required: it defines the batch")
        if diag.dim() != 2:
            raise ValueError("diag / yerr must be (B, N)")
        if t.shape[-1] != diag.shape[-1] or (t.dim() == 2 and t.shape[0] != diag.shape[0]):
            raise ValueError("Invalid shape: t %s does not match diag %s" % (tuple(t.shape), tuple(diag.shape)))
        self._t, self._diag = t.contiguous(), diag.contiguous()
        self._size = t.shape[-1]
        self._c, self._a, self._U, self._V = self.kernel.get_celerite_matrices(self._t, self._diag)
        self._d, self._W, self._flag = ops.factor(self._t, self._c, self._a, self._U, self._V)
        failed = self._flag != 0
        if bool(failed.any()) and not quiet:
            raise LinAlgError("failed to factorize or solve matrix")
        log_det = torch.log(self._d).sum(dim=1)
        self._log_det = torch.where(failed, torch.full_like(log_det, -math.inf), log_det)
        self._norm = torch.where(failed, torch.full_like(log_det, math.inf),
                                 -0.5 * (log_det + self._size * math.log(2 * math.pi)))
        return self

    def _need(self):
        if self._t is None:
            raise RuntimeError("you must call 'compute' first")

    def _as_matrix(self, y):
        if y.dim() not in (2, 3) or tuple(y.shape[:2]) != tuple(self._diag.shape):
            raise ValueError("Invalid shape: y %s, expected (B, N) or (B, N, nrhs) with (B, N) = %s"
                             % (tuple(y.shape), tuple(self._diag.shape)))
        return (y[..., None], True) if y.dim() == 2 else (y, False)

    def _check_vector(self, y):
        if tuple(y.shape) != tuple(self._diag.shape):  # core.py:312-330 (_process_input)
            raise ValueError("Invalid shape: y %s, expected (B, N) = %s" % (tuple(y.shape), tuple(self._diag.shape)))

    # -- core.py:407-428 + numpy.py:104-109 -------------------------------------------------------------
    def log_likelihood(self, y):
        self._need()
        self._check_vector(y)
        r = (y - self.mean)[..., None].contiguous()
        z = ops.solve_lower(self._t, self._c, self._U, self._W, r)[..., 0]
        return self._norm - 0.5 * (z * z / self._d).sum(dim=1)

    def log_likelihood_fused(self, y):
        """Same value straight from the one-pass fused kernel (no d / W / z materialised)."""
        self._need()
        self._check_vector(y)
        ll, flag = ops.loglik(self._t, self._c, self._a, self._U, self._V, (y - self.mean).contiguous())
        return ll

    def log_likelihood_and_grad(self, y, work=None):
        """(ll, (bt, bc, ba, bU, bV, by), flag): gradients w.r.t. the celerite matrices and the data."""
        self._need()
        self._check_vector(y)
        return ops.loglik_grad(self._t, self._c, self._a, self._U, self._V, (y - self.mean).contiguous(), work=work)

    # -- core.py:342-376 + numpy.py:94-98 ----------------------------------------------------------------
    def apply_inverse(self, y):
        self._need()
        Y, vec = self._as_matrix(y)
        z = ops.solve_lower(self._t, self._c, self._U, self._W, Y.contiguous())   # (a fresh array: the caller keeps y)
        z.div_(self._d[..., None])                                                 # ... scaled and solved again in place
        z = ops.solve_upper(self._t, self._c, self._U, self._W, z, Z=z)
        return z[..., 0] if vec else z

    # -- core.py:378-405 + numpy.py:100-102 --------------------------------------------------------------
    def dot_tril(self, y):
        self._need()
        Y, vec = self._as_matrix(y)
        z = ops.dot_tril(self._t, self._c, self._U, self._W, self._d, Y.contiguous())
        return z[..., 0] if vec else z

    # -- numpy.py:111-121 ------------------------------------------------------------------------------
    def sample(self, *, size=None, include_mean=True, generator=None):
        self._need()
        B, N = self._diag.shape
        k = 1 if size is None else size
        n = torch.randn((B, N, k), dtype=torch.float64, device=self._diag.device, generator=generator)
        out = self.dot_tril(n).transpose(1, 2)
        if include_mean:
            out = out + self.mean
        return out[:, 0] if size is None else out

    # -- conditional distribution, core.py:430-478 ------------------------------------------------------
    def condition(self, y, t=None, *, include_mean=True, kernel=None):
        self._need()
        self._check_vector(y)
        return ConditionalDistribution(self, y, t=t, include_mean=include_mean, kernel=kernel)

    def predict(self, y, t=None, *, return_cov=False, return_var=False, include_mean=True, kernel=None):
        """core.py:430-472: the conditional mean (B, M), and with `return_var` its variance (B, M), with `return_cov`
        its covariance (B, M, M) -- `apply_inverse` on the N x M cross-covariance, i.e. solves with M right-hand sides
        (SURVEY.md 8f-4)."""
        cond = self.condition(y, t=t, include_mean=include_mean, kernel=kernel)
        if return_var:
            return cond.mean, cond.variance
        if return_cov:
            return cond.mean, cond.covariance
        return cond.mean


class ConditionalDistribution:
    """Batched mirror of core.py:9-150 (BaseConditionalDistribution) + numpy.py:14-32: the distribution of the process at
    coordinates `t` (B, M) | (M,) -- default: the observed grid -- given observations `y` (B, N).  Properties are evaluated
    on demand and cached like the reference's (`KxsT`, `Kinv_KxsT`)."""

    def __init__(self, gp, y, t=None, *, include_mean=True, kernel=None):
        self.gp, self.y, self.t, self.include_mean, self.kernel = gp, y, t, include_mean, kernel
        self._KxsT = self._Kinv_KxsT = self._Linv_KxsT = self._mean = None
        self._mats2 = self._mats1 = None
        if t is None:
            self._xs = gp._t
        else:
            if t.dim() not in (1, 2) or (t.dim() == 2 and t.shape[0] != gp._diag.shape[0]):
                raise ValueError("'t' must be (M,) or (B, M)")   # core.py:39-40
            self._xs = t.contiguous()

    def _kernel(self):
        return self.gp.kernel if self.kernel is None else self.kernel

    def _batched(self, x):
        B = self.gp._diag.shape[0]
        return x if x.dim() == 2 else x[None].expand(B, x.shape[0])

    @property
    def KxsT(self):      # core.py:46-54: k(t_n - xs_m), (B, N, M)
        if self._KxsT is None:
            self._KxsT = self._kernel().get_value_grid(self.gp._t, self._xs, B=self.gp._diag.shape[0])
        return self._KxsT

    @property
    def Linv_KxsT(self):  # L^-1 KxsT (K = L D L^T): the lower solve with M right-hand sides, shared by variance and Kinv_KxsT
        if self._Linv_KxsT is None:
            gp = self.gp
            self._Linv_KxsT = ops.solve_lower(gp._t, gp._c, gp._U, gp._W, self.KxsT.contiguous())
        return self._Linv_KxsT

    @property
    def Kinv_KxsT(self):  # core.py:56-60: apply_inverse on the N x M matrix -- solves with M right-hand sides
        if self._Kinv_KxsT is None:
            gp = self.gp
            z = self.Linv_KxsT / gp._d[..., None]
            self._Kinv_KxsT = ops.solve_upper(gp._t, gp._c, gp._U, gp._W, z, Z=z)
        return self._Kinv_KxsT

    def _do_dot(self, inp, target):
        """core.py:68-113 + numpy.py:15-22: target += K(xs, t) inp through general_matmul_lower/upper."""
        gp = self.gp
        if self.kernel is None:
            U1, V1 = gp._U, gp._V
        else:
            if self._mats1 is None:
                self._mats1 = self.kernel.get_celerite_matrices(gp._t, torch.zeros_like(gp._diag))
            U1, V1 = self._mats1[2], self._mats1[3]
        if self._mats2 is None:
            B = gp._diag.shape[0]
            zero = torch.zeros((B, self._xs.shape[-1]), dtype=torch.float64, device=gp._diag.device)
            self._mats2 = self._kernel().get_celerite_matrices(self._xs, zero)
        c, _, U2, V2 = self._mats2
        vec = inp.dim() == 2
        if vec:
            inp, target = inp[..., None], target[..., None]
        inp, target = inp.contiguous(), target.contiguous()
        target = ops.general_matmul_lower(self._xs, gp._t, c, U2, V1, inp, Z=target)
        target = ops.general_matmul_upper(self._xs, gp._t, c, V2, U1, inp, Z=target)
        return target[..., 0] if vec else target

    @property
    def mean(self):      # core.py:115-132 (computed once per distribution: predict(return_var / return_cov) and sample() reuse it)
        if self._mean is None:
            gp = self.gp
            alpha = gp.apply_inverse(self.y - gp.mean)
            if self.t is None and self.kernel is None:
                mu = self.y - gp._diag * alpha
                self._mean = mu if self.include_mean else mu - gp.mean
            else:
                B = gp._diag.shape[0]
                mu = torch.zeros((B, self._xs.shape[-1]), dtype=torch.float64, device=gp._diag.device)
                mu = self._do_dot(alpha, mu)
                self._mean = mu + gp.mean if self.include_mean else mu
        return self._mean

    @property
    def variance(self):  # core.py:134-140 + numpy.py:24-25: k(0) - diag(KxsT' K^-1 KxsT), (B, M)
        # diag(Kxs K^-1 KxsT)_m = sum_n (L^-1 KxsT)_nm^2 / d_n: the lower solve and ONE pass over its result
        # (ops.colsumsq_over_d) -- the reference's apply_inverse + diagdot (numpy.py:24-25) is both solves and a pass over
        # two N x M arrays for the same number
        return self._k0() - ops.colsumsq_over_d(self.Linv_KxsT, self.gp._d)

    def _k0(self):
        """k(0) = sum ar + sum ac (terms.py:58-79 at tau = 0): a python float, or (B, 1) on the device for per-series coefficients."""
        co = self._kernel().get_coefficients()
        if all(v.ndim == 1 for v in co):
            return float(co[0].sum() + co[2].sum())
        dev, _ = self._kernel()._dev_coefs(self.gp._diag.device, self.gp._diag.shape[0])
        return (dev[0].sum(dim=-1) + dev[2].sum(dim=-1))[:, None]

    @property
    def covariance(self):  # core.py:142-150: k(xs - xs') - K(xs, t) K^-1 K(t, xs), (B, M, M)
        neg_cov = -self._kernel().get_value_grid(self._xs, self._xs, B=self.gp._diag.shape[0])
        neg_cov = self._do_dot(self.Kinv_KxsT, neg_cov)
        return -neg_cov

    def sample(self, *, size=None, regularize=None, generator=None):
        """numpy.py:27-32: draws from N(mean, covariance), O(M^3) per series (a dense Cholesky of the M x M covariance
        by torch -- outside the hot path, as in the reference)."""
        mu, cov = self.mean, self.covariance
        if regularize is not None:
            cov = cov + regularize * torch.eye(cov.shape[-1], dtype=cov.dtype, device=cov.device)
        L = torch.linalg.cholesky(0.5 * (cov + cov.transpose(1, 2)))
        k = 1 if size is None else size
        n = torch.randn((cov.shape[0], cov.shape[-1], k), dtype=torch.float64, device=cov.device, generator=generator)
        out = (L @ n).transpose(1, 2) + mu[:, None, :]
        return out[:, 0] if size is None else out
