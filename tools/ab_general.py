"""general_matmul_lower at B = 8192, N = M = 4096, J = 8: row tiles (c2_general_tile.hip) against lanes over the right-hand sides
(c2_general.hip) for small numbers of right-hand sides, in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import _lib, ops, synth
dev = torch.device("cuda:0")
B, N, J = 8192, 4096, 8
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
ts = (t + 0.03).contiguous()
def once(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)
for nrhs in (1, 2, 3, 4):
    Y = torch.randn((B, N, nrhs), dtype=torch.float64, device=dev); Zg = torch.zeros_like(Y)
    Fg = torch.empty((B, N, J, nrhs), dtype=torch.float64, device=dev)
    for label, fn in (("", lambda: ops.general_matmul_lower(ts, t, c, U, V, Y, Z=Zg)), (" with F", lambda: ops.general_matmul_lower(ts, t, c, U, V, Y, Z=Zg, F=Fg))):
        res = {}
        for v in (1, 0):
            _lib.set_option("general_tile", v); fn(); fn()
            res[v] = sorted(once(fn) for _ in range(5))[2]
        _lib.set_option("general_tile", None)
        print("nrhs %d%s: row tiles %.2f ms   without them %.2f ms" % (nrhs, label, res[1], res[0]), flush=True)
