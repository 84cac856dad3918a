"""general_matmul_lower over the number of right-hand sides at B = 8192, N = M = 4096, J = 8: ms and fraction of 8 TB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
B, N, J = 8192, 4096, 8
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
ts = (t + 0.03).contiguous()
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts_ = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts_.append(e0.elapsed_time(e1))
    return sorted(ts_)[len(ts_) // 2]
for nrhs in [int(v) for v in sys.argv[1:]] or [1, 2, 3, 4, 5, 6, 8, 12, 16, 32]:
    Y = torch.randn((B, N, nrhs), dtype=torch.float64, device=dev); Zg = torch.zeros_like(Y)
    Fg = torch.empty((B, N, J, nrhs), dtype=torch.float64, device=dev)
    m1 = timed(lambda: ops.general_matmul_lower(ts, t, c, U, V, Y, Z=Zg))
    m2 = timed(lambda: ops.general_matmul_lower(ts, t, c, U, V, Y, Z=Zg, F=Fg))
    by = lambda b: B * N * b / 8e12 * 1e3
    print("nrhs %2d: general_matmul_lower %.2f ms (%.2f)   with F %.2f ms (%.2f)" % (nrhs, m1, by(16 * (1 + J + nrhs)) / m1, m2, by(16 * (1 + J + nrhs) + 8 * J * nrhs) / m2), flush=True)
    del Y, Zg, Fg
