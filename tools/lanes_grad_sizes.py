"""Fused log-likelihood + gradient by lane mapping (8, 4, 1 lanes per series) over shard sizes; N = 4096, J = 8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import _lib, ops, synth
dev = torch.device("cuda:0")
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
for B in [int(v) for v in sys.argv[1:]] or [8192, 16384, 24576, 32768]:
    args = synth.device_batch_fast(0, B, 4096, 8, dev)
    row = []
    for lanes in (8, 4, 1):
        _lib.set_option("lanes", lanes)
        work = ops.loglik_grad_workspace(B, 4096, 8, dev)
        out = ops.loglik_grad(*args, work=work)[1]
        row.append("%d lanes %.2f ms" % (lanes, timed(lambda: ops.loglik_grad(*args, work=work, out=out))))
        del work, out
    _lib.set_option("lanes", None)
    print("B %6d: %s" % (B, "   ".join(row)), flush=True)
    del args; torch.cuda.empty_cache()
