"""One op of the device ABI a few times, for profilers: python tools/op_run.py <op> <B> <N> <J> [reps]
op: loglik | loglik_grad | terms | terms_grad (coefficient-level, bench coefficients) | factor | factor_s | factor_rev | solve_rhs<nrhs> | rev_rhs<nrhs> (solve_lower_rev) | predict[_var|_cov] | chain (factor_s + solve_lower F + solve_lower_rev + factor_rev)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
op, B, N, J = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda:0")
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
Y = y[:, :, None].contiguous()
gp = ts = None
if op.startswith("predict"):
    from celerite2_amd import gp as gpmod, terms
    import numpy as np
    M = 256
    xi = np.zeros(B)
    k = np.arange(J // 2, dtype=np.float64)
    kernel = None
    for kk in range(J // 2):
        term = terms.SHOTerm(S0=5.0 * 0.7**kk, w0=0.1 * 3.0**kk, Q=3.45 + kk)
        kernel = term if kernel is None else kernel + term
    gp = gpmod.GaussianProcess(kernel, mean=0.0)
    diag = torch.rand((B, N), dtype=torch.float64, device=dev) * 0.2 + 0.1
    gp.compute(t, diag=diag)
    ts = torch.sort(torch.rand((B, M), dtype=torch.float64, device=dev) * (N / 10.0), dim=1).values
Wf = Yr = None
targs = None
if op.startswith("terms"):
    import numpy as np
    th, dgh, yh, ach, bch, cch, dch = synth.host_inputs(0, 8, N, J)
    f = lambda x: torch.from_numpy(np.ascontiguousarray(np.tile(x, ((B + 7) // 8,) + (1,) * (x.ndim - 1))[:B])).to(dev)
    e_ = torch.zeros((B, 0), dtype=torch.float64, device=dev)
    targs = (e_, e_, f(ach), f(bch), f(cch), f(dch), f(th), f(dgh), f(yh))
revargs = None
if op.startswith("rev_rhs"):   # rev_rhs<nrhs>: solve_lower_rev with that many right-hand sides
    d_, Wf, fl_ = ops.factor(t, c, a, U, V)
    Yq = torch.randn((B, N, int(op[7:])), dtype=torch.float64, device=dev)
    Zq, Fq = ops.solve_lower(t, c, U, Wf, Yq, workspace=True)
    revargs = (t, c, U, Wf, Yq, Zq, Fq, torch.ones_like(Zq))
if op.startswith("solve_rhs"):   # solve_rhs<nrhs>: solve_lower with that many right-hand sides, in place
    d_, Wf, fl_ = ops.factor(t, c, a, U, V)
    Yr = torch.randn((B, N, int(op[9:])), dtype=torch.float64, device=dev)
def run():
    if Yr is not None: return ops.solve_lower(t, c, U, Wf, Yr, Z=Yr)
    if revargs is not None: return ops.solve_lower_rev(*revargs)
    if op == "terms": return ops.loglik_terms(*targs)
    if op == "terms_grad": return ops.loglik_terms_grad(*targs)
    if op == "predict_var": return gp.predict(y, ts, return_var=True)
    if op == "predict_cov": return gp.predict(y, ts, return_cov=True)
    if op == "predict": return gp.predict(y, ts)
    if op == "loglik": return ops.loglik(t, c, a, U, V, y)
    if op == "loglik_grad": return ops.loglik_grad(t, c, a, U, V, y)
    if op == "factor": return ops.factor(t, c, a, U, V)
    if op == "factor_s": return ops.factor(t, c, a, U, V, workspace=True)
    d, W, S, fl = ops.factor(t, c, a, U, V, workspace=True)
    if op == "factor_rev": return ops.factor_rev(t, c, a, U, V, d, W, S, torch.ones_like(d), torch.ones_like(W))
    Z, F = ops.solve_lower(t, c, U, W, Y, workspace=True)
    r1 = ops.solve_lower_rev(t, c, U, W, Y, Z, F, torch.ones_like(Z))
    return ops.factor_rev(t, c, a, U, V, d, W, S, torch.ones_like(d), torch.ones_like(W))
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
print("%s B=%d N=%d J=%d: %.3f ms per call (back to back)" % (op, B, N, J, e0.elapsed_time(e1) / reps))
