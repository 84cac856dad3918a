# -*- coding: utf-8 -*-
"""One-lane-per-series gradient with UNPAIRED rates (c[2k] != c[2k+1]: real terms) against the paired case and the
8-lane replay kernels, bench shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
B, N = 65536, 4096
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, 8, dev)
def timed(fn, reps=4, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, cc in (("paired", c), ("unpaired", (c * torch.tensor([1, 1.01, 1, 1.02, 1, 0.99, 1, 1.03], device=dev, dtype=c.dtype)).contiguous())):
    for lanes in ("1", "8"):
        os.environ["C2_LANES"] = lanes
        work = ops.loglik_grad_workspace(B, N, 8, dev)
        out = None
        ll, out, fl = ops.loglik_grad(t, cc, a, U, V, y, work=work, out=out)
        ms = timed(lambda: ops.loglik_grad(t, cc, a, U, V, y, work=work, out=out))
        print("%-9s C2_LANES=%s  %.2f ms  %.3f M GP/s  failed %d" % (name, lanes, ms, B / ms / 1e3, int((fl != 0).sum())), flush=True)
        del work, out
        torch.cuda.empty_cache()
