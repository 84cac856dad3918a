# -*- coding: utf-8 -*-
"""torch.autograd adapter over the fused kernels (framework adapter on the device path).

`log_likelihood(t, c, a, U, V, y)` is differentiable w.r.t. every argument: the forward call runs
c2_loglik_grad once (value and the six cotangents come out of the same checkpoint/replay pass, exactly what the
reference's PyMC/JAX ops do in two steps -- pymc/ops.py:104-141), backward just scales the saved gradients by
the incoming cotangent.  Shared `t` (N,) / `c` (J,) receive the batch-summed gradient."""
import torch

from . import ops

__all__ = ["log_likelihood"]


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
        bt, bc, ba, bU, bV, by = ctx.saved_tensors
        g1, g2 = g[:, None], g[:, None, None]
        gt = (bt * g1).sum(0) if ctx.shared_t else bt * g1
        gc = (bc * g1).sum(0) if ctx.shared_c else bc * g1
        return gt, gc, ba * g1, bU * g2, bV * g2, by * g1


def log_likelihood(t, c, a, U, V, y):
    """Batched GP log-likelihood (B,), differentiable through torch.autograd."""
    return _LogLik.apply(t, c, a, U, V, y)
