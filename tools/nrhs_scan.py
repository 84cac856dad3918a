"""Forward and reverse sweeps over the number of right-hand sides at B = 8192, N = 4096, J = 8: ms and fraction of 8 TB/s
(algorithmic bytes as in tools/bench_ops.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
B, N, J = 8192, 4096, 8
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
d, W, flag = ops.factor(t, c, a, U, V)
def timed(fn, reps=7):
    return synth.timed_steady(fn, reps=reps)   # (steady clock: profiles/r05_clock_ramp.md)
FWD_ONLY = os.environ.get("C2_SCAN_FWD_ONLY", "0") == "1"
for nrhs in [int(v) for v in sys.argv[1:]] or [1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16]:
    Y = torch.randn((B, N, nrhs), dtype=torch.float64, device=dev); Zo = torch.empty_like(Y)
    F = torch.empty((B, N, J, nrhs), dtype=torch.float64, device=dev); bZ = torch.randn_like(Y)
    f1 = timed(lambda: ops.solve_lower(t, c, U, W, Y, Z=Zo))
    if FWD_ONLY:
        byf = B * N * 8 * (1 + 2 * J + 2 * nrhs) / 8e12 * 1e3
        fu = timed(lambda: ops.solve_upper(t, c, U, W, Y, Z=Zo))
        fm = timed(lambda: ops.matmul_lower(t, c, U, V, Y, Z=Zo, zero_z=True))
        print("nrhs %2d: solve_lower %.2f ms (%.2f)   solve_upper %.2f ms (%.2f)   matmul_lower (zero_z) %.2f ms (%.2f)" % (
            nrhs, f1, byf / f1, fu, byf / fu, fm, byf / fm), flush=True)
        del Y, Zo
        continue
    f2 = timed(lambda: ops.solve_lower(t, c, U, W, Y, Z=Zo, F=F))
    Zs, Fs = ops.solve_lower(t, c, U, W, Y, workspace=True)
    r = timed(lambda: ops.solve_lower_rev(t, c, U, W, Y, Zs, Fs, bZ))
    by = lambda b: B * N * b / 8e12 * 1e3
    print("nrhs %2d: solve_lower %.2f ms (%.2f)   with F %.2f ms (%.2f)   solve_lower_rev %.2f ms (%.2f)" % (
        nrhs, f1, by(8 * (1 + 2 * J + 2 * nrhs)) / f1, f2, by(8 * (1 + 2 * J + 2 * nrhs + J * nrhs)) / f2,
        r, by(8 * (2 + 4 * J + 4 * nrhs + J * nrhs)) / r), flush=True)
    del Y, Zo, F, bZ, Zs, Fs
