"""A/B of one dispatch option inside one process (alternating, median of 7 each):
    python tools/ab_option.py <option> <op> [nrhs] [B]      op: solve_lower | solve_upper | matmul_lower | matmul_upper | <op>_rev | <op>_ws (with the F workspace)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import _lib, ops, synth
opt, name = sys.argv[1], sys.argv[2]
nrhs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
N, J = 4096, 8
dev = torch.device("cuda:0")
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
d, W, flag = ops.factor(t, c, a, U, V)
torch.manual_seed(0)
Y = torch.randn((B, N, nrhs), dtype=torch.float64, device=dev)
ws = name.endswith("_ws")      # forward sweep writing the F workspace
if ws: name = name[:-3]
base = name[:-4] if name.endswith("_rev") else name
sec = W if base.startswith("solve") else V
if name.endswith("_rev"):
    kw = dict(workspace=True) if base.startswith("solve") else dict(workspace=True, zero_z=True)
    Z, F = getattr(ops, base)(t, c, U, sec, Y, **kw)
    bZ = torch.randn_like(Y)
    fn = lambda: getattr(ops, name)(t, c, U, sec, Y, Z, F, bZ)
elif ws:
    Zo = torch.empty_like(Y); Fo = torch.empty((B, N, J, nrhs), dtype=torch.float64, device=dev)
    fn = (lambda: getattr(ops, name)(t, c, U, sec, Y, Z=Zo, F=Fo)) if base.startswith("solve") else (lambda: getattr(ops, name)(t, c, U, sec, Y, Z=Zo, F=Fo, zero_z=True))
else:
    Zo = torch.empty_like(Y)
    fn = (lambda: getattr(ops, name)(t, c, U, sec, Y, Z=Zo)) if base.startswith("solve") else (lambda: getattr(ops, name)(t, c, U, sec, Y, Z=Zo, zero_z=True))
def once():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
res = {0: [], 1: []}
for rep in range(8):
    for v in (1, 0):
        _lib.set_option(opt, v)
        once()
        if rep: res[v].append(once())
_lib.set_option(opt, None)
med = lambda x: sorted(x)[len(x) // 2]
print("%s nrhs=%d B=%d: %s=1 %.3f ms   %s=0 %.3f ms" % (name, nrhs, B, opt, med(res[1]), opt, med(res[0])))
