# -*- coding: utf-8 -*-
"""torch.autograd adapter over the fused kernels (framework adapter on the device path).

`log_likelihood(t, c, a, U, V, y)` is differentiable w.r.t. every argument: the forward call runs
c2_loglik_grad once (value and the six cotangents come out of the same checkpoint/replay pass, exactly what the
reference's PyMC/JAX ops do in two steps -- pymc/ops.py:104-141), backward just scales the saved gradients by
the incoming cotangent.  Shared `t` (N,) / `c` (J,) receive the batch-summed gradient.

`factor`, `solve_lower`, `solve_upper`, `matmul_lower`, `matmul_upper` are the reference's five differentiable ops
(python/celerite2/pymc/ops.py:61-141, jax/ops.py:33-172: forward = `backprop.<op>_fwd` with its workspace, gradient =
`backprop.<op>_rev`), batched, on the device kernels -- for models that compose the ops themselves."""
import torch

from . import ops

__all__ = ["log_likelihood", "log_likelihood_terms", "factor", "solve_lower", "solve_upper", "matmul_lower", "matmul_upper", "LinAlgError"]


class LinAlgError(RuntimeError):
    """failed to factorize or solve matrix (driver.hpp:13-19); `.flag` holds the per-series first bad row."""


class _LogLik(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, c, a, U, V, y):
        args = [x.detach().contiguous() for x in (t, c, a, U, V, y)]
        needs_grad = any(x.requires_grad for x in (t, c, a, U, V, y))
        if not needs_grad:
            ll, flag = ops.loglik(*args)
            ctx.grads = None
            return ll
        ll, grads, flag = ops.loglik_grad(*args)
        ctx.shared_t = t.dim() == 1
        ctx.shared_c = c.dim() == 1
        ctx.save_for_backward(*grads)
        ctx.mark_non_differentiable()
        return ll

    @staticmethod
    def backward(ctx, g):
        # A series whose factorisation failed has ll = -inf and NaN gradients (c2_loglik_grad, celerite2_amd.h).
        # A series that receives a ZERO cotangent contributes exactly zero -- so masking failed series out of the
        # objective keeps the batch-summed gradients of a shared t / c finite; left in, they turn NaN, never garbage.
        bt, bc, ba, bU, bV, by = ctx.saved_tensors
        g1, g2 = g[:, None], g[:, None, None]
        z1, z2 = g1 == 0, g2 == 0
        sc = lambda x, gg, zz: torch.where(zz, torch.zeros((), dtype=x.dtype, device=x.device), x * gg)
        gt = sc(bt, g1, z1).sum(0) if ctx.shared_t else sc(bt, g1, z1)
        gc = sc(bc, g1, z1).sum(0) if ctx.shared_c else sc(bc, g1, z1)
        return gt, gc, sc(ba, g1, z1), sc(bU, g2, z2), sc(bV, g2, z2), sc(by, g1, z1)


def log_likelihood(t, c, a, U, V, y):
    """Batched GP log-likelihood (B,), differentiable through torch.autograd."""
    return _LogLik.apply(t, c, a, U, V, y)


class _LogLikTerms(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ar, cr, ac, bc, cc, dc, x, diag, y):
        args = [v.detach().contiguous() for v in (ar, cr, ac, bc, cc, dc, x, diag, y)]
        if not any(v.requires_grad for v in (ar, cr, ac, bc, cc, dc, x, diag, y)):
            ll, flag = ops.loglik_terms(*args)
            return ll
        ll, grads, flag = ops.loglik_terms_grad(*args)
        ctx.shared = [v.dim() == 1 for v in (ar, cr, ac, bc, cc, dc, x)]
        ctx.save_for_backward(*grads)
        return ll

    @staticmethod
    def backward(ctx, g):
        grads = ctx.saved_tensors
        g1 = g[:, None]
        zero = torch.zeros((), dtype=g.dtype, device=g.device)
        out = []
        for k, gr in enumerate(grads):
            v = torch.where(g1 == 0, zero, gr * g1)   # a masked-out (failed) series contributes exactly zero
            out.append(v.sum(0) if (k < 7 and ctx.shared[k]) else v)
        return tuple(out)


def log_likelihood_terms(ar, cr, ac, bc, cc, dc, x, diag, y):
    """Batched GP log-likelihood (B,) as a differentiable function of the celerite coefficients, the times, the
    white-noise diagonal and the data -- the gradient a sampler needs, computed by the device chain of c2_terms.hip.
    Shared coefficients / times (one fewer dimension) receive the batch-summed gradient."""
    return _LogLikTerms.apply(ar, cr, ac, bc, cc, dc, x, diag, y)


def _reduce(g, like):
    """Gradient of an argument that was shared by the batch (one fewer dimension): sum over the batch."""
    return g.sum(0) if like.dim() == g.dim() - 1 else g


class _Factor(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, c, a, U, V):
        args = [x.detach().contiguous() for x in (t, c, a, U, V)]
        d, W, S, flag = ops.factor(*args, workspace=True)
        if bool((flag != 0).any()):
            err = LinAlgError("failed to factorize or solve matrix")
            err.flag = flag
            raise err
        ctx.save_for_backward(*args, d, W, S)
        return d, W

    @staticmethod
    def backward(ctx, bd, bW):
        t, c, a, U, V, d, W, S = ctx.saved_tensors
        bt, bc, ba, bU, bV = ops.factor_rev(t, c, a, U, V, d, W, S, bd.contiguous(), bW.contiguous())
        return _reduce(bt, t), _reduce(bc, c), ba, bU, bV


def factor(t, c, a, U, V):
    """(d, W) = LDL^T factors of the batch (forward.hpp:69-135), differentiable (reverse.hpp:10-85)."""
    return _Factor.apply(t, c, a, U, V)


def _sweep(name):
    fwd, rev = getattr(ops, name), getattr(ops, name + "_rev")

    class _Op(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t, c, U, W, Y):
            args = [x.detach().contiguous() for x in (t, c, U, W, Y)]
            Z, F = fwd(*args, workspace=True, zero_z=True) if name.startswith("matmul") else fwd(*args, workspace=True)
            ctx.save_for_backward(*args, Z, F)
            return Z

        @staticmethod
        def backward(ctx, bZ):
            t, c, U, W, Y, Z, F = ctx.saved_tensors
            bt, bc, bU, bW, bY = rev(t, c, U, W, Y, Z, F, bZ.contiguous())
            return _reduce(bt, t), _reduce(bc, c), bU, bW, bY

    _Op.__name__ = "_" + name

    def op(t, c, U, W, Y):
        return _Op.apply(t, c, U, W, Y)

    op.__name__ = name
    op.__doc__ = "Batched %s (B,N,nrhs), differentiable w.r.t. (t, c, U, W|V, Y); internal.hpp:105-303." % name
    return op


solve_lower = _sweep("solve_lower")
solve_upper = _sweep("solve_upper")
matmul_lower = _sweep("matmul_lower")
matmul_upper = _sweep("matmul_upper")
