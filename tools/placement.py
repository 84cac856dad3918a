# -*- coding: utf-8 -*-
"""Does the relative placement of the big arrays matter?  The bench shape's arrays are 2^34 bytes each (65536 x 4096 x 8
doubles): carved back to back they sit a power of two apart.  Carve them from one buffer with a stagger of k x PAD bytes
between consecutive arrays and time the gradient pair."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
B, N, J = 65536, 4096, 8
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
wsz = ops.loglik_grad_workspace(B, N, J, dev).numel()
def timed(fn, reps=6, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
del U, V
torch.cuda.empty_cache()
shapes = [("U", (B, N, J)), ("V", (B, N, J)), ("bU", (B, N, J)), ("bV", (B, N, J)), ("bt", (B, N)), ("ba", (B, N)), ("by", (B, N)), ("bc", (B, J)), ("work", (wsz,))]
total = sum(int(torch.tensor(s).prod()) for _, s in shapes)
for PAD in (0, 4096 + 256, 65536 + 4096, (1 << 20) + 4096 + 256, 3 * (1 << 20) + 8192 + 512):
    buf = torch.empty(total + len(shapes) * (PAD // 8 + 64), dtype=torch.float64, device=dev)
    off = 0; ten = {}
    for k, (name, shp) in enumerate(shapes):
        n = int(torch.tensor(shp).prod())
        ten[name] = buf[off:off + n].view(shp)
        off += n + PAD // 8
        off = (off + 31) // 32 * 32   # keep 256-byte alignment
    ar = torch.zeros((B, 0), dtype=torch.float64, device=dev)
    tt, diag, yy, ac, bc_, cc, dc = synth.device_coeffs_fast(0, B, N, J, dev)
    a2, U2, V2 = ops.get_celerite_matrices(ar, ac, bc_, dc, tt, diag)
    ten["U"].copy_(U2); ten["V"].copy_(V2); del U2, V2
    out = (ten["bt"], ten["bc"], ten["ba"], ten["bU"], ten["bV"], ten["by"])
    ms = timed(lambda: ops.loglik_grad(t, c, a, ten["U"], ten["V"], y, work=ten["work"], out=out))
    print("stagger %8d B between arrays: %.2f ms per step" % (PAD, ms), flush=True)
    del buf, ten, out
    torch.cuda.empty_cache()
