import torch
x = torch.empty(4_560_000_000 // 8, dtype=torch.float64, device="cuda")
for f, name in ((lambda: x.fill_(1.5), "fill"), (lambda: x.zero_(), "zero"), (lambda: torch.mul(x, 2.0, out=x), "rmw x*=2")):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(name, "%.3f ms  %.2f TB/s written" % (ms, x.numel() * 8 / ms / 1e9))
