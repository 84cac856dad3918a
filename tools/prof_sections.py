# -*- coding: utf-8 -*-
"""Cycle breakdown of the one-lane reverse step (diagnostic build):
    tools/build_variant.sh prof c2_loglik_t.hip -DC2T_PROF
    C2_LIB_PATH=celerite2_amd/libcelerite2_amd_prof.so python tools/prof_sections.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celerite2_amd import ops, synth, _lib  # noqa: E402

N, B, J = 1000, 65536, 8
t, diag, y, ac, bc, cc, dc = synth.host_inputs(0, 8, N, J)
f = lambda x: torch.from_numpy(np.ascontiguousarray(np.tile(x, (B // 8,) + (1,) * (x.ndim - 1)))).cuda()
td, dg, yd, acd, bcd, ccd, dcd = map(f, (t, diag, y, ac, bc, cc, dc))
e = torch.zeros((B, 0), dtype=torch.float64, device="cuda")
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()
names = ["top: requests, rows (sincos), decay", "packed 36-element pass", "bU / sums / bc / seeds / tile writes (+ flush)",
         "waits for the requests, register rotation", "loop back-edge", "tile turn (every 8th step), checkpoint (every 32nd)"]
for label, fused in (("coefficient-level (fused)", "1"), ("matrix-level", None)):
    if fused:
        os.environ["C2_TERMS_FUSED"] = fused
        run = lambda: ops.loglik_terms_grad(e, e, acd, bcd, ccd, dcd, td, dg, yd)
    else:
        a, U, V = ops.get_celerite_matrices(e, acd, bcd, dcd, td, dg)
        os.environ["C2_LANES"] = "1"
        cfull = torch.cat([ccd[:, k // 2:k // 2 + 1] for k in range(8)], dim=1).contiguous()
        run = lambda: ops.loglik_grad(td, cfull, a, U, V, yd)
    run(); lib.c2_internal_prof_read(out)
    run(); lib.c2_internal_prof_read(out)
    nw = len(range(0, B // 64, 97))
    tot = sum(out[k] for k in range(6))
    print(label, "cycles per step: %.0f" % (tot / nw / (N - 1)))
    for k in range(6):
        print("   %-60s %8.0f  (%4.1f %%)" % (names[k], out[k] / nw / (N - 1), 100.0 * out[k] / tot))
