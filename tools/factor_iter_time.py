"""factor (d, W) at widths 6 / 8 on small batches of long series: fixed-point passes over chunks (C2_FACTOR_ITER=1,
c2_timepar_grad.hip) against row by row (=0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
def timed(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]
shapes = [(1, 4096, 8), (1, 100000, 8), (1, 100000, 6), (32, 50000, 6), (64, 4096, 8), (1024, 4096, 8), (1, 1000000, 8)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for B, N, J in shapes:
    t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
    d = torch.empty_like(a); W = torch.empty_like(V)
    res = {}
    for mode in ("0", "1"):
        os.environ["C2_FACTOR_ITER"] = mode
        ms = timed(lambda: ops.factor(t, c, a, U, V, d=d, W=W))
        res[mode] = (ms, d.clone(), W.clone())
    print("B %5d N %7d J %d: factor row by row %8.3f ms, chunk passes %8.3f ms (%5.1fx)  rel diff d %.1e W %.1e" % (
        B, N, J, res["0"][0], res["1"][0], res["0"][0] / res["1"][0],
        float(((res["1"][1] - res["0"][1]) / res["0"][1]).abs().max()),
        float((res["1"][2] - res["0"][2]).abs().max() / res["0"][2].abs().max())), flush=True)
