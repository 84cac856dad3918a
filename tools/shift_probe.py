"""Which arrays of the bench step decide its mode?  One process, inputs allocated once; then the workspace and / or the six
gradient arrays are re-allocated behind spacers of S GiB and three steps are timed for each placement."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
B, N, J = 65536, 4096, 8
G = 2**30
pre = int(os.environ.get("C2_PRE_SPACER_GB", "0"))
presp = torch.empty(pre * G, dtype=torch.uint8, device=dev) if pre else None
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
torch.cuda.empty_cache()
def outs():
    return (torch.empty((B, N), dtype=torch.float64, device=dev), torch.empty((B, J), dtype=torch.float64, device=dev),
            torch.empty((B, N), dtype=torch.float64, device=dev), torch.empty_like(U), torch.empty_like(U),
            torch.empty((B, N), dtype=torch.float64, device=dev))
def timed(work, out, reps=3):
    for _ in range(2): ops.loglik_grad(t, c, a, U, V, y, work=work, out=out)
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.loglik_grad(t, c, a, U, V, y, work=work, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
print("inputs: U at %#x, V at %#x (pre-spacer %d GiB)" % (U.data_ptr(), V.data_ptr(), pre))
for what in ("both", "work", "out"):
    base_w = ops.loglik_grad_workspace(B, N, J, dev) if what == "out" else None
    base_o = outs() if what == "work" else None
    for S in (0, 4, 8, 16, 24, 32, 48):
        sp = torch.empty(S * G, dtype=torch.uint8, device=dev) if S else None
        w = base_w if base_w is not None else ops.loglik_grad_workspace(B, N, J, dev)
        o = base_o if base_o is not None else outs()
        ms = timed(w, o)
        print("shift %-4s by %2d GiB: %.2f ms   (work at %#x, bU at %#x)" % (what, S, ms, w.data_ptr(), o[3].data_ptr()), flush=True)
        if base_w is None: del w
        if base_o is None: del o
        del sp
        torch.cuda.empty_cache()
    del base_w, base_o
    torch.cuda.empty_cache()
