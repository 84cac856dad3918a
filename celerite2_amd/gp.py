# -*- coding: utf-8 -*-
"""Thin batched GaussianProcess frontend over the device ops.

Reproduces, for a batch of B independent series, what the reference's numpy backend does around the ops
(python/celerite2/numpy.py:66-121 and core.py:262-501): compute -> factor; log_likelihood -> solve_lower +
reductions; apply_inverse -> solve_lower, /d, solve_upper; dot_tril; sample; conditional mean at new
coordinates via general_matmul_lower/upper (core.py:74-132, numpy.py:15-22).  Everything stays on the GPU.
`log_likelihood_and_grad` exposes the fused kernels' gradients w.r.t. (t, c, a, U, V, y).
"""
import math

import torch

from . import ops

__all__ = ["GaussianProcess", "LinAlgError"]


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
        z = ops.solve_lower(self._t, self._c, self._U, self._W, Y.contiguous())
        z = z / self._d[..., None]
        z = ops.solve_upper(self._t, self._c, self._U, self._W, z.contiguous())
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

    # -- conditional mean, core.py:115-132 + numpy.py:15-22 ----------------------------------------------
    def predict(self, y, t=None, *, include_mean=True):
        self._need()
        self._check_vector(y)
        alpha = self.apply_inverse(y - self.mean)
        if t is None:
            mu = y - self._diag * alpha
            return mu if include_mean else mu - self.mean
        ts = t.contiguous()
        zero = torch.zeros((self._diag.shape[0], ts.shape[-1]), dtype=torch.float64, device=ts.device)
        _, _, U2, V2 = self.kernel.get_celerite_matrices(ts, zero)
        inp = alpha[..., None].contiguous()
        mu = ops.general_matmul_lower(ts, self._t, self._c, U2, self._V, inp)
        mu = ops.general_matmul_upper(ts, self._t, self._c, V2, self._U, inp, Z=mu)
        mu = mu[..., 0]
        return mu + self.mean if include_mean else mu
