#!/usr/bin/env python
"""Pure-store, pure-load and copy rates of this GPU with PyTorch's own elementwise kernels (16 GiB operands):
the practical HBM ceilings the roofline fractions in DESIGN.md should be read against."""
import torch

n = 2 * 1024**3  # doubles = 16 GiB
x = torch.empty(n, dtype=torch.float64, device="cuda")
y = torch.empty(n, dtype=torch.float64, device="cuda")


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


gb = n * 8 / 1e9
ms = timed(lambda: x.fill_(1.5)); print("fill  (store only): %.2f ms  %.0f GB/s" % (ms, gb / ms * 1e3))
ms = timed(lambda: x.sum()); print("sum   (load only) : %.2f ms  %.0f GB/s" % (ms, gb / ms * 1e3))
ms = timed(lambda: y.copy_(x)); print("copy  (load+store): %.2f ms  %.0f GB/s moved" % (ms, 2 * gb / ms * 1e3))
ms = timed(lambda: x.mul_(1.0000001)); print("scale (load+store, in place): %.2f ms  %.0f GB/s moved" % (ms, 2 * gb / ms * 1e3))
