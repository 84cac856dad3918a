"""Per-step HIP-event times of the bench workload over many steps in ONE process (is a slow first process of a box slow throughout, or
ramping?): python tools/step_trace.py [steps] [series]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda:0")
t, c, a, U, V, y = synth.device_batch_fast(0, B, 4096, 8, dev)
work = ops.loglik_grad_workspace(B, 4096, 8, dev)
out = None
def step():
    global out
    out = ops.loglik_grad(t, c, a, U, V, y, workspace=work, out=out) if False else ops.loglik_grad(t, c, a, U, V, y)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
torch.cuda.synchronize()
t0 = time.time()
ev[0].record()
for i in range(steps):
    step(); ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
print("wall %.1f s; steps 1-5: %s" % (time.time() - t0, " ".join("%.2f" % x for x in ms[:5])))
for lo in range(0, steps, max(steps // 12, 1)):
    seg = ms[lo:lo + max(steps // 12, 1)]
    print("steps %4d .. %4d: mean %.2f ms (min %.2f max %.2f)" % (lo, lo + len(seg) - 1, sum(seg) / len(seg), min(seg), max(seg)))
