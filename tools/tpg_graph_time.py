"""The time-parallel gradient captured in a HIP graph (it makes no allocation) against the eager call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for B, N, J in ((1, 1000, 2), (1, 4096, 2), (1, 4096, 8), (1, 100000, 8), (32, 50000, 6)):
    args = synth.device_batch_fast(0, B, N, J, dev)
    work = ops.loglik_grad_workspace(B, N, J, dev)
    ll, out, fl = ops.loglik_grad(*args, work=work)
    eager = timed(lambda: ops.loglik_grad(*args, work=work, out=out))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.loglik_grad(*args, work=work, out=out)
    graph = timed(g.replay)
    print("B %3d N %6d J %d: eager %.3f ms, graph replay %.3f ms" % (B, N, J, eager, graph), flush=True)
