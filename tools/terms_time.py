# -*- coding: utf-8 -*-
"""Coefficient-level log-likelihood (+ gradient): fused one-lane kernels (C2_TERMS_FUSED=1) against the composed chain
(matrices in memory, =0) over batch sizes at the bench shape.  python tools/terms_time.py [N] [B ...]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celerite2_amd import ops, synth  # noqa: E402


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    Bs = [int(v) for v in sys.argv[2:]] or [4096, 8192, 16384, 32768, 65536]
    J = 8
    t, diag, y, ac, bc, cc, dc = synth.host_inputs(0, 8, N, J)
    for B in Bs:
        rep = B // 8
        f = lambda x: torch.from_numpy(np.ascontiguousarray(np.tile(x, (rep,) + (1,) * (x.ndim - 1)))).cuda()
        td, dg, yd, acd, bcd, ccd, dcd = map(f, (t, diag, y, ac, bc, cc, dc))
        e = torch.zeros((B, 0), dtype=torch.float64, device="cuda")
        row = {"B": B, "N": N, "J": J}
        ref = None
        for fused in ("0", "1"):
            os.environ["C2_TERMS_FUSED"] = fused
            ll, flag = ops.loglik_terms(e, e, acd, bcd, ccd, dcd, td, dg, yd)
            ll2, g, flag2 = ops.loglik_terms_grad(e, e, acd, bcd, ccd, dcd, td, dg, yd)
            assert int(flag.abs().sum()) == 0 and int(flag2.abs().sum()) == 0
            if ref is None:
                ref = (ll, g)
            else:
                row["max_rel_dll"] = float(((ll - ref[0]).abs() / ref[0].abs()).max())
                row["max_rel_dgrad"] = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(g, ref[1]) if b.numel())
            row["fwd_ms_fused" + fused] = timed(lambda: ops.loglik_terms(e, e, acd, bcd, ccd, dcd, td, dg, yd))
            row["grad_ms_fused" + fused] = timed(lambda: ops.loglik_terms_grad(e, e, acd, bcd, ccd, dcd, td, dg, yd))
        row["grad_GP_per_s_fused"] = B / row["grad_ms_fused1"] * 1e3
        row["grad_GP_per_s_composed"] = B / row["grad_ms_fused0"] * 1e3
        print(json.dumps(row), flush=True)
        del td, dg, yd, ref, g, ll, ll2
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
