# -*- coding: utf-8 -*-
"""Which array's placement decides between the fast and the slow mode of the one-lane gradient pair (27.6 vs 30.6 ms per
65536 series)?  Two copies of every class of arrays (inputs U, V / outputs bU, bV, bt, ba, by / workspace) in one process,
every combination timed; the virtual addresses printed."""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
B, N, J = 65536, 4096, 8
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
def timed(fn, reps=4, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ins = [(U, V), (U.clone(), V.clone())]
works = [ops.loglik_grad_workspace(B, N, J, dev) for _ in range(2)]
outs = []
for _ in range(2):
    ll, g, fl = ops.loglik_grad(t, c, a, U, V, y, work=works[0])
    outs.append(g)
for name, objs in (("in", ins), ("work", [(w,) for w in works]), ("out", outs)):
    for i, o in enumerate(objs):
        print(name, i, " ".join("%x" % x.data_ptr() for x in o))
for i, w, o in itertools.product(range(2), range(2), range(2)):
    ms = timed(lambda: ops.loglik_grad(t, c, a, ins[i][0], ins[i][1], y, work=works[w], out=outs[o]))
    print("inputs %d  workspace %d  outputs %d : %.2f ms" % (i, w, o, ms), flush=True)
