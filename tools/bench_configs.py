#!/usr/bin/env python
"""Secondary measurements for the BASELINE.json configs that are parity-test cases rather than the bench line:
config 2 (B=1024, N=4096, J=4, forward log-lik), config 4 (one series, N=1e7, J=16, nrhs=32 dot_tril; matrix cores vs
VALU), config 5 (2-D, rank-1 band covariance: 32 series per GPU x 50000 epochs x 16 bands, J=6, forward + gradient, both
methods) and the coefficient-level gradient (c2_loglik_terms_grad) at the bench shape.
Prints one JSON object per measurement with the HIP-event time and the algorithmic-byte rate."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from celerite2_amd import ops, synth


def timed(fn, reps=9, warm=2):
    from celerite2_amd import synth as _synth
    return _synth.timed_steady(fn, reps=reps)   # (steady clock: profiles/r05_clock_ramp.md)


def config2():
    B, N, J = 1024, 4096, 4
    t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, "cuda")
    ms = timed(lambda: ops.loglik(t, c, a, U, V, y))
    nbytes = B * (N * 8 * (3 + 2 * J) + 8 * J + 8)
    return {"config": "2: B=1024 N=4096 J=4 forward log-lik, 1 GPU", "ms": ms, "GP_per_s": B / ms * 1e3,
            "algorithmic_GB": nbytes / 1e9, "GB_per_s": nbytes / ms / 1e6, "frac_hbm_8TBs": nbytes / ms / 1e6 / 8000}


def config4(N=10_000_000, J=16, nrhs=32):
    t, c, a, U, V, _ = synth.device_batch_fast(0, 1, N, J, "cuda")
    d, W, flag = ops.factor(t, c, a, U, V)
    assert int(flag[0]) == 0
    Y = torch.randn((1, N, nrhs), dtype=torch.float64, device="cuda")
    Z = torch.empty_like(Y)
    ms = timed(lambda: ops.dot_tril(t, c, U, W, d, Y, Z=Z), reps=3, warm=1)
    nbytes = N * 8 * (1 + 2 * J + 1 + 2 * nrhs)
    return {"config": "4: 1 series N=%d J=%d nrhs=%d dot_tril, 1 GPU (time-chunked scan, fp64 MFMA blocks)" % (N, J, nrhs), "ms": ms,
            "algorithmic_GB": nbytes / 1e9, "GB_per_s": nbytes / ms / 1e6, "frac_hbm_8TBs": nbytes / ms / 1e6 / 8000}


def config4_valu():
    os.environ["C2_MFMA"] = "0"
    try:
        r = config4()
    finally:
        os.environ.pop("C2_MFMA", None)
    r["config"] = r["config"].replace("time-chunked scan, fp64 MFMA blocks", "time-chunked scan, VALU path, C2_MFMA=0")
    return r


def config5(B=32, N=50000, M=16, J=6):
    """naive 1-D view 192 MB per GP (16 (3 + 2 J) bytes per interleaved row, forward + gradient)."""
    dev = "cuda"
    t, c, a, U, V, _ = synth.device_batch_fast(0, B, N, J, dev)
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    f64 = dict(dtype=torch.float64, device=dev)
    alpha = 0.5 + torch.rand((B, M), generator=gen, **f64)
    diag = 0.1 + 0.2 * torch.rand((B, N, M), generator=gen, **f64)
    y = alpha[:, None, :] * torch.sin(t)[:, :, None] + diag.sqrt() * torch.randn((B, N, M), generator=gen, **f64)
    # the epoch grid's matrices with ZERO white noise: a = k(0)
    a0 = (U * V).sum(-1).contiguous()
    out = []
    nbytes = B * N * M * 16 * (3 + 2 * J)
    for method in ("collapsed", "interleaved"):
        ll, g, flag = ops.kron_loglik_grad(t, c, a0, U, V, alpha, diag, y, method=method)
        assert int(flag.abs().sum()) == 0
        ms = timed(lambda: ops.kron_loglik_grad(t, c, a0, U, V, alpha, diag, y, method=method), reps=3, warm=1)
        out.append({"config": "5: 2-D rank-1 bands, %d series x %d epochs x %d bands, J=%d, fwd+grad, 1 GPU, method=%s "
                              "(extension; parity unpinned by the reference)" % (B, N, M, J, method),
                    "ms": ms, "GP_per_s": B / ms * 1e3, "algorithmic_GB_naive_1D_view": nbytes / 1e9,
                    "GB_per_s": nbytes / ms / 1e6, "frac_hbm_8TBs": nbytes / ms / 1e6 / 8000})
    return out


def terms_grad(B=8192, N=4096, J=8):
    """Gradient w.r.t. the celerite coefficients, composed on the device."""
    dev = "cuda"
    t, diag, y, ac, bc, cc, dc = synth.host_inputs(0, 8, N, J)
    import numpy as np
    rep = B // 8
    f = lambda x: torch.from_numpy(np.ascontiguousarray(np.tile(x, (rep,) + (1,) * (x.ndim - 1)))).to(dev)
    td, dg, yd, acd, bcd, ccd, dcd = map(f, (t, diag, y, ac, bc, cc, dc))
    e = torch.zeros((B, 0), dtype=torch.float64, device=dev)
    ms = timed(lambda: ops.loglik_terms_grad(e, e, acd, bcd, ccd, dcd, td, dg, yd), reps=3, warm=1)
    return {"config": "terms: B=%d N=%d J=%d log-lik + gradient w.r.t. (ac, bc, cc, dc, x, diag, y), composed chain" % (B, N, J),
            "ms": ms, "GP_per_s": B / ms * 1e3}


if __name__ == "__main__":
    which = sys.argv[1:] or ["2", "4", "4v", "5", "terms"]
    if "2" in which:
        print(json.dumps(config2()), flush=True)
    if "4" in which:
        print(json.dumps(config4()), flush=True)
    if "4v" in which:
        print(json.dumps(config4_valu()), flush=True)
    if "5" in which:
        for r in config5():
            print(json.dumps(r), flush=True)
    if "terms" in which:
        print(json.dumps(terms_grad()), flush=True)
