"""Cycle breakdown of the multi-rhs reverse step (diagnostic build):
    tools/build_variant.sh prof c2_sweep_rev.hip -DC2R_PROF
    C2_LIB_PATH=celerite2_amd/libcelerite2_amd_prof.so python tools/sweepk_rev_sections.py [nrhs]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth, _lib
dev = torch.device("cuda:0")
B, N, J = 8192, 4096, 8
nrhs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
d, W, flag = ops.factor(t, c, a, U, V)
Y = torch.randn((B, N, nrhs), dtype=torch.float64, device=dev)
bZ = torch.randn((B, N, nrhs), dtype=torch.float64, device=dev)
Z, F = ops.solve_lower(t, c, U, W, Y, workspace=True)
_lib.set_option("sweep_rev_lines", 0)   # the row-by-row kernel k_sweepK_rev is the instrumented one
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()
names = ["ring -> registers (waits for the rows)", "exp, row vectors into LDS, fence", "LDS reads + per-column products", "cotangent of row m (store)",
         "three reduce-scatters, stores, phi", "requests for step s - R (issued right after the ring reads)"]
for rep in range(2):
    ops.solve_lower_rev(t, c, U, W, Y, Z, F, bZ); lib.c2_internal_sweep_rev_prof_read(out)
spw = 64 // (8 if nrhs <= 8 else 16)
nw = len(range(0, B // spw, 97))
tot = sum(out[k] for k in range(6))
print("solve_lower_rev nrhs=%d: cycles per step %.0f" % (nrhs, tot / nw / (N - 1)))
for k in range(6):
    print("   %-50s %8.0f  (%4.1f %%)" % (names[k], out[k] / nw / (N - 1), 100.0 * out[k] / tot))
