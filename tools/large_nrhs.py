"""SURVEY.md 8f-4 at its own size: solve_lower + solve_upper (= apply_inverse on an N x M matrix, core.py:62-66) with
nrhs in {64, 256, 1024} at B in {1, 64, 2048}, N = 4096, J in {8, 16}; in place (Z is Y).  ms, GB/s of algorithmic bytes
(8 (1 + 2 J + 2 nrhs) per row: t, the two width-J rows, Y in, Z out) and fraction of 8 TB/s; and gp.predict(return_var)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
N = 4096
def timed(fn, reps=5):
    return synth.timed_steady(fn, reps=reps)   # (steady clock: profiles/r05_clock_ramp.md)
print("| J | B | nrhs | solve_lower ms | GB/s | frac | solve_upper ms | GB/s | frac |")
print("|---|---|---|---|---|---|---|---|---|")
Bs = [int(x) for x in os.environ.get("LARGE_NRHS_B", "1,64,2048").split(",")]
Js = [int(x) for x in os.environ.get("LARGE_NRHS_J", "8,16").split(",")]
for J in Js:
    for B in Bs:
        t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
        d, W, flag = ops.factor(t, c, a, U, V)
        for nrhs in (64, 256, 1024):
            Y = torch.randn((B, N, nrhs), dtype=torch.float64, device=dev)
            nbytes = B * N * 8.0 * (1 + 2 * J + 2 * nrhs)
            lo = timed(lambda: ops.solve_lower(t, c, U, W, Y, Z=Y))
            up = timed(lambda: ops.solve_upper(t, c, U, W, Y, Z=Y))
            print("| %d | %d | %d | %.3f | %.0f | %.3f | %.3f | %.0f | %.3f |" % (
                J, B, nrhs, lo, nbytes / lo / 1e6, nbytes / lo / 1e6 / 8000, up, nbytes / up / 1e6, nbytes / up / 1e6 / 8000), flush=True)
            del Y
        del t, c, a, U, V, y, d, W
        torch.cuda.empty_cache()
if os.environ.get("LARGE_NRHS_B"): sys.exit(0)
# the frontend: predictive variance of 64 light curves of 4096 points at 256 new times (apply_inverse with 256 right-hand sides)
from celerite2_amd import gp as gpmod, terms
B, M = 64, 256
gen = torch.Generator(device=dev); gen.manual_seed(3)
x = torch.sort(torch.rand((B, N), generator=gen, dtype=torch.float64, device=dev) * (N / 10.0), dim=1).values.contiguous()
xs = torch.sort(torch.rand((B, M), generator=gen, dtype=torch.float64, device=dev) * (N / 10.0), dim=1).values.contiguous()
diag = 0.1 + 0.2 * torch.rand((B, N), generator=gen, dtype=torch.float64, device=dev)
yv = torch.sin(x)
kernel = terms.SHOTerm(S0=5.0, w0=0.1, Q=3.45) + terms.SHOTerm(S0=3.5, w0=0.3, Q=4.45) + terms.SHOTerm(S0=2.4, w0=0.9, Q=5.45) + terms.SHOTerm(S0=1.7, w0=2.7, Q=6.45)
g = gpmod.GaussianProcess(kernel); g.compute(x, diag=diag)
ms = timed(lambda: g.predict(yv, xs, return_var=True))
cond = g.condition(yv, xs); K = cond.KxsT; torch.cuda.synchronize()
ms_inv = timed(lambda: g.apply_inverse(K))
print("gp.predict(return_var=True): B = %d light curves x N = %d, M = %d new times, J = 8: %.2f ms (apply_inverse on the N x M matrix: %.2f ms)" % (B, N, M, ms, ms_inv))
