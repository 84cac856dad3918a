#!/usr/bin/env python
"""Secondary measurements for the BASELINE.json configs that are parity-test cases rather than the bench line:
config 2 (B=1024, N=4096, J=4, forward log-lik) and config 4 (one series, N=1e7, J=16, nrhs=32 dot_tril).
Prints one JSON object per config with the HIP-event time and the algorithmic-byte rate."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from celerite2_amd import ops, synth


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    return sorted(e0.elapsed_time(e1) for e0, e1 in ev)[len(ev) // 2]


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
    return {"config": "4: 1 series N=%d J=%d nrhs=%d dot_tril, 1 GPU (time-chunked scan)" % (N, J, nrhs), "ms": ms,
            "algorithmic_GB": nbytes / 1e9, "GB_per_s": nbytes / ms / 1e6, "frac_hbm_8TBs": nbytes / ms / 1e6 / 8000}


if __name__ == "__main__":
    which = sys.argv[1:] or ["2", "4"]
    if "2" in which:
        print(json.dumps(config2()), flush=True)
    if "4" in which:
        print(json.dumps(config4()), flush=True)
