"""configs[4] shape (collapsed method): time under the dispatch options that could matter (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import _lib, ops, synth
dev = "cuda"
B, N, M, J = 32, 50000, 16, 6
t, c, a, U, V, _ = synth.device_batch_fast(0, B, N, J, dev)
gen = torch.Generator(device=dev); gen.manual_seed(5)
f64 = dict(dtype=torch.float64, device=dev)
alpha = 0.5 + torch.rand((B, M), generator=gen, **f64)
diag = 0.1 + 0.2 * torch.rand((B, N, M), generator=gen, **f64)
y = alpha[:, None, :] * torch.sin(t)[:, :, None] + diag.sqrt() * torch.randn((B, N, M), generator=gen, **f64)
a0 = (U * V).sum(-1).contiguous()
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
for opts in ({}, {"verify_fallback": 0}, {"timepar_grad": 0}, {"factor_iter": 0}, {"kron_banded": 0}):
    for k, v in opts.items(): _lib.set_option(k, v)
    ms = timed(lambda: ops.kron_loglik_grad(t, c, a0, U, V, alpha, diag, y, method="collapsed"))
    print(opts, "%.3f ms" % ms, flush=True)
    for k in opts: _lib.set_option(k, None)
# the 1-D problem inside it
y1 = torch.sin(t)
work = ops.loglik_grad_workspace(B, N, J, torch.device("cuda:0"))
ms = timed(lambda: ops.loglik_grad(t, c, a0 + 0.2, U, V, y1, work=work))
print("1-D loglik_grad 32 x 50000, J = 6: %.3f ms; verify words:" % ms, work[:8].tolist())
