#!/usr/bin/env python
"""Per-phase cycle counts of k_loglik_rev from a library built with -DC2_REV_TIMING (s_memtime deltas accumulated per
wavefront).  Build: hipcc <HIPFLAGS of celerite2_amd/build.py> -DC2_REV_TIMING <sources> -o celerite2_amd/_exp_t.so, then
C2_LIB_PATH=$PWD/celerite2_amd/_exp_t.so python tools/rev_phase_timing.py [B ...]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from celerite2_amd import _lib, ops, synth

lib = _lib.load()
N, J = 4096, 8
for B in [int(x) for x in sys.argv[1:]] or [2048, 8192, 65536]:
    t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, "cuda")
    work = ops.loglik_grad_workspace(B, N, J, "cuda")
    out = (ctypes.c_ulonglong * 8)()
    for rep in range(2):
        ops.loglik_grad(t, c, a, U, V, y, work=work)
        torch.cuda.synchronize()
        lib.c2_internal_read_dbg(out)
    waves = out[0]
    nseg = (N - 1 + 7) // 8
    names = ["A: records -> LDS, exps", "B: replay", "C: reverse steps", "flush"]
    tot = sum(out[1:5])
    print("B=%d waves=%d  cycles per segment per wave (8 rows):" % (B, waves))
    for i, nm in enumerate(names):
        print("   %-26s %9.0f  (%.1f %%)" % (nm, out[1 + i] / waves / nseg, 100.0 * out[1 + i] / tot))
    print("   total %9.0f per segment = %.0f per row" % (tot / waves / nseg, tot / waves / nseg / 8))
    del t, c, a, U, V, y, work
