import os, sys
sys.path.insert(0, os.getcwd())
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
def timed(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for J, B, N in ((4, 32, 50000), (4, 1, 20000), (4, 2, 4097), (2, 32, 50000), (4, 1, 100000), (4, 1, 1000000)):
    t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
    out = []
    for tp, op in (("0", "1"), ("1", "1"), ("1", "0")):
        os.environ["C2_TIMEPAR"] = tp; os.environ["C2_TIMEPAR_ONEPASS"] = op
        ll, f = ops.loglik(t, c, a, U, V, y)
        out.append((tp, op, round(timed(lambda: ops.loglik(t, c, a, U, V, y)), 3), float(ll[0])))
    print(J, B, N, out, flush=True)
