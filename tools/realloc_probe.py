"""After the box has idled, is the slow mode of the headline step tied to the FIRST allocation of the process?  One process:
allocate everything, time; free everything (empty_cache), allocate again, time; and once more."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
B, N, J = 65536, 4096, 8
def timed(fn, reps=8, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
res = []
for rnd in range(3):
    t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
    work = ops.loglik_grad_workspace(B, N, J, dev)
    out = ops.loglik_grad(t, c, a, U, V, y, work=work)[1]
    res.append(timed(lambda: ops.loglik_grad(t, c, a, U, V, y, work=work, out=out)))
    del t, c, a, U, V, y, work, out
    torch.cuda.synchronize(); torch.cuda.empty_cache()
print("ms per step, allocation rounds 1 / 2 / 3 of one process: " + " / ".join("%.2f" % x for x in res), flush=True)
