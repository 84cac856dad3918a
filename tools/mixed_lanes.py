# -*- coding: utf-8 -*-
"""Mid-size batches: part of the batch on the one-lane kernels, the rest on the 8-lane kernels, on two streams at once
(the one-lane kernels leave SIMDs idle below 65536 series; the 8-lane kernels can take them).
    python tools/mixed_lanes.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celerite2_amd import _lib, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
N, J = 4096, 8


def run(B, x, reps=4):
    """x series on one lane each, B - x on eight lanes; x = 0 / B: one call"""
    args = synth.device_batch_fast(0, B, N, J, dev)
    parts = []
    for lo, hi, lanes in ((0, x, 1), (x, B, 8)):
        if hi > lo:
            a = [v[lo:hi].contiguous() if v.shape[0] == B and v.dim() >= 1 else v for v in args]
            _lib.set_option("lanes", lanes)
            work = ops.loglik_grad_workspace(hi - lo, N, J, dev)
            parts.append([a, work, None, lanes, torch.cuda.Stream()])
    def once():
        for p in parts:
            _lib.set_option("lanes", p[3])
            with torch.cuda.stream(p[4]):
                ll, p[2], fl = ops.loglik_grad(*p[0], work=p[1], out=p[2])
    once(); once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record(cur)
    for _ in range(reps):
        for p in parts:
            p[4].wait_stream(cur)
        once()
        for p in parts:
            cur.wait_stream(p[4])
    e1.record(cur)
    torch.cuda.synchronize()
    _lib.set_option("lanes", None)
    return e0.elapsed_time(e1) / reps


for B in [int(v) for v in sys.argv[1:]] or [32768, 24576, 49152, 16384]:
    row = []
    for x in sorted({0, B} | {x for x in (4096, 8192, 12288, 16384, 20480, 24576, 28672, 32768, 40960) if x < B}):
        row.append("%d: %.2f" % (x, run(B, x)))
        torch.cuda.empty_cache()
    print("B %6d  ms by number of series on the one-lane kernels:  %s" % (B, "   ".join(row)), flush=True)
