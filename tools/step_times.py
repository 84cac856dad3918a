import os, sys, torch
sys.path.insert(0, os.getcwd())
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
B, N, J = 65536, 4096, 8
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
work = ops.loglik_grad_workspace(B, N, J, dev); out = None
ev = []
for i in range(40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ll, out, fl = ops.loglik_grad(t, c, a, U, V, y, work=work, out=out); e1.record(); ev.append((e0, e1))
torch.cuda.synchronize()
print(" ".join("%.1f" % a.elapsed_time(b) for a, b in ev))
