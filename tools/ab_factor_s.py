"""factor with the S workspace at B = 8192, N = 4096, J = 8: option s_replay_lines on / off in one process, and parity between them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import _lib, ops, synth
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
t, c, a, U, V, y = synth.device_batch_fast(0, B, 4096, 8, dev)
d2, W2 = torch.empty_like(a), torch.empty_like(V)
S = torch.empty((B, 4096, 8, 8), dtype=torch.float64, device=dev)
fn = lambda: ops.factor(t, c, a, U, V, d=d2, W=W2, S=S)
def once():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
res = {0: [], 1: []}; keep = {}
for rep in range(6):
    for v in (1, 0):
        _lib.set_option("s_replay_lines", v)
        once()
        if rep: res[v].append(once())
        if rep == 1: keep[v] = S.clone()
_lib.set_option("s_replay_lines", None)
med = lambda x: sorted(x)[len(x) // 2]
print("factor + S, B=%d: lines %.3f ms   rows %.3f ms   max |difference| of S %.1e" % (B, med(res[1]), med(res[0]), float((keep[1] - keep[0]).abs().max())))
