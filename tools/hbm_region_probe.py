"""Is the bandwidth of a plain device copy a function of WHERE in HBM its buffers lie?  One process; for every offset S a
spacer of S GiB is held while two 16-GiB buffers are allocated behind it, a copy between them is timed (median of 5), and
everything is freed again.  (profiles/r04_headline_spread.md: the step of the bench changes mode with such an offset.)"""
import sys
import torch
dev = torch.device("cuda:0")
G = 2**30
n = 16 * G
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
for S in [int(v) for v in sys.argv[1:]] or list(range(0, 132, 8)):
    sp = torch.empty(S * G, dtype=torch.uint8, device=dev) if S else None
    a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty(n, dtype=torch.uint8, device=dev)
    a.fill_(1)
    ms_copy = timed(lambda: b.copy_(a))
    ms_fill = timed(lambda: b.fill_(2))
    ms_sum = timed(lambda: a.view(torch.int64).sum())
    print("offset %3d GiB: copy %.2f ms (%.2f TB/s)   fill %.2f ms (%.2f TB/s)   sum %.2f ms (%.2f TB/s)   a at %#x" % (
        S, ms_copy, 2 * n / ms_copy / 1e9, ms_fill, n / ms_fill / 1e9, ms_sum, n / ms_sum / 1e9, a.data_ptr()), flush=True)
    del a, b, sp
    torch.cuda.empty_cache()
