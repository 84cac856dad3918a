"""general_matmul_lower / _upper with several right-hand sides (lanes over the right-hand sides: c2_general.hip), ms and fraction of the
8 TB/s roofline on algorithmic bytes 8 (1 + J + nrhs) per row of either grid.  Usage: general_rhs_time.py [B [J [nrhs ...]]]
(A/B of builds: C2_LIB_PATH=<other library> python tools/general_rhs_time.py ...)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
J = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rhs = [int(x) for x in sys.argv[3:]] or [5, 6, 7, 8]
N = 4096
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
t1 = (t + 0.013).contiguous()
print("library:", os.environ.get("C2_LIB_PATH", "default"), " B", B, "N = M", N, "J", J, flush=True)
for nrhs in rhs:
    Y = torch.randn((B, N, nrhs), dtype=torch.float64, device=dev)
    Z = torch.empty((B, N, nrhs), dtype=torch.float64, device=dev)
    F = torch.empty((B, N, J, nrhs), dtype=torch.float64, device=dev) if B * N * J * nrhs * 8 < 40e9 else None
    for name in ("general_matmul_lower", "general_matmul_upper"):
        f = getattr(ops, name)
        ms = synth.timed_steady(lambda: f(t1, t, c, U, V, Y, Z=Z, zero_z=True), reps=7)
        alg = B * 8.0 * N * (1 + J + nrhs) * 2
        line = "%s nrhs=%3d: %.3f ms (frac %.3f)" % (name, nrhs, ms, alg / ms / 8e9)
        if F is not None:
            msf = synth.timed_steady(lambda: f(t1, t, c, U, V, Y, Z=Z, F=F, zero_z=True), reps=7)
            line += "   with F %.3f ms (frac %.3f)" % (msf, (alg + B * 8.0 * N * J * nrhs) / msf / 8e9)
        print(line, flush=True)
