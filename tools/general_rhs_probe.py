"""general_matmul_lower / upper at B = 8192, N = M = 4096, J = 8 over the number of right-hand sides, by 64-row tiles
(C2_GENERAL_TILE_MAX_RHS = nrhs) against lanes over the right-hand sides (= 0 ... the default table): ms and fraction of 8 TB/s
on the algorithmic bytes 16 (1 + J + nrhs) per row."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
B, N, J = 8192, 4096, 8
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
ts = (t + 0.03).contiguous()
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); out.append(e0.elapsed_time(e1))
    return sorted(out)[len(out) // 2]
for nrhs in [int(v) for v in sys.argv[1:]] or [2, 3, 4, 5, 6, 8, 12, 16]:
    Y = torch.randn((B, N, nrhs), dtype=torch.float64, device=dev)
    Z = torch.zeros_like(Y)
    by = B * N * 16 * (1 + J + nrhs) / 8e12 * 1e3
    res = {}
    for name, val in (("lanes over rhs", "1"), ("tiles", str(nrhs))):
        os.environ["C2_GENERAL_TILE_MAX_RHS"] = val
        lo = timed(lambda: ops.general_matmul_lower(ts, t, c, U, V, Y, Z=Z))
        Zl = Z.clone()
        up = timed(lambda: ops.general_matmul_upper(ts, t, c, U, V, Y, Z=Z))
        res[name] = (lo, up, Zl)
    os.environ.pop("C2_GENERAL_TILE_MAX_RHS")
    d = float((res["tiles"][2] - res["lanes over rhs"][2]).abs().max() / res["tiles"][2].abs().max())
    print("nrhs %2d: lanes over rhs %.2f / %.2f ms (%.2f / %.2f)   tiles %.2f / %.2f ms (%.2f / %.2f)   diff %.1e" % (
        nrhs, res["lanes over rhs"][0], res["lanes over rhs"][1], by / res["lanes over rhs"][0], by / res["lanes over rhs"][1],
        res["tiles"][0], res["tiles"][1], by / res["tiles"][0], by / res["tiles"][1], d), flush=True)
