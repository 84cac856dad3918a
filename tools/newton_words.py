"""Newton-factor iteration words (updates, kappas) of the time-parallel gradient on a small-white-noise problem
(the 1-D problem inside the collapsed 2-D method: a_eff = k(0) + 1 / A with 1 / A ~ 0.0125)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import _lib, ops, synth
dev = torch.device("cuda:0")
B, N, J = 32, 50000, 6
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
a0 = (U * V).sum(-1).contiguous()
lib = _lib.load()
sink = torch.zeros(64, dtype=torch.float64, device=dev)
lib.c2_internal_set_debug_sink.argtypes = [ctypes.c_void_p]; lib.c2_internal_set_debug_sink.restype = None
lib.c2_internal_set_debug_sink(ctypes.c_void_p(sink.data_ptr()))
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
work = ops.loglik_grad_workspace(B, N, J, dev)
for noise in (0.2, 0.05, 0.0125, 0.003):
    an = (a0 + noise).contiguous()
    ms = timed(lambda: ops.loglik_grad(t, c, an, U, V, y, work=work))
    w = sink.cpu().numpy()
    print("white noise %.4f: %.3f ms; verify %s" % (noise, ms, ["%.2e" % v for v in w[:6]]))
    print("   updates / half tolerance %s" % ["%.1e" % v for v in w[8:18]]); print("   kappas  %s" % ["%.1e" % v for v in w[18:28]])
    print("   updates %s" % ["%.1e" % v for v in w[28:38]])
