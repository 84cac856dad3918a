"""Fused log-likelihood + gradient on small batches of long series: parallel along time (C2_TIMEPAR_GRAD=1,
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
shapes = [(1, 1000, 2), (1, 4096, 8), (1, 100000, 2), (1, 100000, 4), (1, 100000, 6), (1, 100000, 8), (32, 50000, 6),
          (64, 4096, 8), (256, 4096, 8), (1024, 4096, 4), (1024, 4096, 8), (2048, 4096, 4), (4096, 4096, 2)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for B, N, J in shapes:
    args = synth.device_batch_fast(0, B, N, J, dev)
    res = {}
    for mode in ("0", "1"):
        os.environ["C2_TIMEPAR_GRAD"] = mode
        work = ops.loglik_grad_workspace(B, N, J, dev)
        ll, out, fl = ops.loglik_grad(*args, work=work)
        ms = timed(lambda: ops.loglik_grad(*args, work=work, out=out))
        res[mode] = (ms, ll.clone(), [o.clone() for o in out])
        del work
    err = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(res["1"][2], res["0"][2]))
    print("B %5d N %6d J %d: row by row %8.3f ms, time-parallel %8.3f ms (%5.1fx)  ll diff %.1e  grad diff %.1e" % (
        B, N, J, res["0"][0], res["1"][0], res["0"][0] / res["1"][0],
        float(((res["1"][1] - res["0"][1]) / res["0"][1]).abs().max()), err), flush=True)
