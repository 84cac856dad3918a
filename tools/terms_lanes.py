# -*- coding: utf-8 -*-
"""Coefficient-level log-likelihood (+ gradient) at J = 8: composed chain / one lane per series / two lanes per series over batch
sizes at the bench shape.  python tools/terms_lanes.py [N] [B ...]   (TERMS_J=4 / 2: the narrower widths, composed / one / group)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celerite2_amd import ops, synth  # noqa: E402
from tools.terms_time import timed  # noqa: E402

MODES = {"composed": ("0", "0", "0", "0"), "one": ("1", "0", "0", "0"), "two": ("0", "1", "0", "0"), "eight": ("0", "0", "1", "0"),
         "four": ("0", "0", "0", "1")}


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    Bs = [int(v) for v in sys.argv[2:]] or [8192, 12288, 16384, 24576, 32768, 49152, 65536]
    J = int(os.environ.get("TERMS_J", "8"))
    t, diag, y, ac, bc, cc, dc = synth.host_inputs(0, 8, N, J)
    for B in Bs:
        rep = (B + 7) // 8
        f = lambda x: torch.from_numpy(np.ascontiguousarray(np.tile(x, (rep,) + (1,) * (x.ndim - 1))[:B])).cuda()
        td, dg, yd, acd, bcd, ccd, dcd = map(f, (t, diag, y, ac, bc, cc, dc))
        e = torch.zeros((B, 0), dtype=torch.float64, device="cuda")
        row = {"B": B, "N": N, "J": J}
        ref = None
        for name, (fu, two, eight, four) in MODES.items():
            if (name == "eight" and B * J > 65536) or (name == "four" and B > 24576) or (J != 8 and name in ("two", "four")):
                continue
            os.environ["C2_TERMS_FOUR_LANES"] = four
            os.environ["C2_TERMS_FUSED"] = fu
            os.environ["C2_TERMS_TWO_LANES"] = two
            os.environ["C2_TERMS_EIGHT_LANES"] = eight
            ll, flag = ops.loglik_terms(e, e, acd, bcd, ccd, dcd, td, dg, yd)
            ll2, g, flag2 = ops.loglik_terms_grad(e, e, acd, bcd, ccd, dcd, td, dg, yd)
            assert int(flag.abs().sum()) == 0 and int(flag2.abs().sum()) == 0
            if ref is None:
                ref = (ll, g)
            else:
                row["dll_" + name] = float(((ll - ref[0]).abs() / ref[0].abs()).max())
                row["dgrad_" + name] = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(g, ref[1]) if b.numel())
            row["fwd_ms_" + name] = round(timed(lambda: ops.loglik_terms(e, e, acd, bcd, ccd, dcd, td, dg, yd)), 3)
            row["grad_ms_" + name] = round(timed(lambda: ops.loglik_terms_grad(e, e, acd, bcd, ccd, dcd, td, dg, yd)), 3)
        print(json.dumps(row), flush=True)
        del td, dg, yd, ref, g, ll, ll2
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
