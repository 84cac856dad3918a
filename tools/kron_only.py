import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = "cuda"
B, N, M, J = 32, 50000, 16, 6
t, c, a, U, V, _ = synth.device_batch_fast(0, B, N, J, dev)
gen = torch.Generator(device=dev); gen.manual_seed(5)
f64 = dict(dtype=torch.float64, device=dev)
alpha = 0.5 + torch.rand((B, M), generator=gen, **f64)
diag = 0.1 + 0.2 * torch.rand((B, N, M), generator=gen, **f64)
y = alpha[:, None, :] * torch.sin(t)[:, :, None] + diag.sqrt() * torch.randn((B, N, M), generator=gen, **f64)
a0 = (U * V).sum(-1).contiguous()
for _ in range(5):
    ops.kron_loglik_grad(t, c, a0, U, V, alpha, diag, y, method="collapsed")
torch.cuda.synchronize()
