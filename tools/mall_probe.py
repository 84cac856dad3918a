"""Does a second read of a buffer come from the 256 MB Infinity Cache?  Reads (torch.sum) of buffers of growing size, repeated
back to back: GB/s of the repeats against the buffer size.  Also: read A, then write B of the same size, then read A again."""
import torch
dev = torch.device("cuda:0")
def rate(fn, nbytes, reps=6):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    return nbytes / ts[len(ts) // 2] / 1e6
big = torch.empty(1 << 30, dtype=torch.float64, device=dev).fill_(1.0)   # 8 GiB: flushes every cache
for mb in (32, 64, 96, 128, 160, 192, 256, 384, 512, 1024, 4096):
    n = mb * (1 << 20) // 8
    a = torch.ones(n, dtype=torch.float64, device=dev)
    b = torch.empty_like(a)
    r_rep = rate(lambda: a.sum(), n * 8)
    def cold():
        big[: (1 << 27)].sum()   # 1 GiB of other data in between
        return a.sum()
    t_flush = rate(lambda: big[: (1 << 27)].sum(), 8 << 27)
    def rw():
        b.fill_(2.0); return a.sum()
    print("%5d MB: repeated read %7.0f GB/s | fill %7.0f GB/s | read after a fill of the same size: pair at %7.0f GB/s (of the read bytes + written bytes)"
          % (mb, r_rep, rate(lambda: b.fill_(2.0), n * 8), rate(rw, 2 * n * 8)), flush=True)
    del a, b
