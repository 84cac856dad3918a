"""Fine shifts: the six gradient arrays carved from ONE pool (fixed relative layout) whose start moves in steps of 2 MiB / 64 KiB;
inputs and workspace fixed.  Is the step time a periodic function of the offset between the output rows and the input rows?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
B, N, J = 65536, 4096, 8
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
torch.cuda.empty_cache()
work = ops.loglik_grad_workspace(B, N, J, dev)
nb = [B * N, B * J, B * N, B * N * J, B * N * J, B * N]          # doubles of bt, bc, ba, bU, bV, by
shapes = [(B, N), (B, J), (B, N), (B, N, J), (B, N, J), (B, N)]
pad = 64 * 2**20 // 8
pool = torch.empty(sum(nb) + pad + 6 * 1100 * 2**17, dtype=torch.float64, device=dev)
def carve(off, stagger=0):
    out, o = [], off
    for i, (n_, sh) in enumerate(zip(nb, shapes)):
        out.append(pool[o:o + n_].view(sh)); o += n_ + stagger
    return tuple(out)
def timed(out, reps=3):
    for _ in range(2): ops.loglik_grad(t, c, a, U, V, y, work=work, out=out)
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.loglik_grad(t, c, a, U, V, y, work=work, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
print("U %#x  V %#x  work %#x  pool %#x" % (U.data_ptr(), V.data_ptr(), work.data_ptr(), pool.data_ptr()))
print("pool offset 0: %.2f ms" % timed(carve(0)), flush=True)
for mb in (0, 2, 6, 10, 18, 34, 66, 130, 258, 514, 1026, 3, 7, 100, 1000):   # stagger BETWEEN the arrays, MiB
    print("stagger %5d MiB between the arrays: %.2f ms" % (mb, timed(carve(0, mb * 2**17))), flush=True)
