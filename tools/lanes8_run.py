# -*- coding: utf-8 -*-
"""The matrix-level 8-lanes-per-series pair (k_loglik_fwd / k_loglik_rev) on the bench series, for profilers:
    python tools/lanes8_run.py [N] [B] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celerite2_amd import _lib, ops, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
_lib.set_option("lanes", 8)
args = synth.device_batch_fast(0, B, N, 8, dev)
work = ops.loglik_grad_workspace(B, N, 8, dev)
out = None
for _ in range(2):
    ll, out, fl = ops.loglik_grad(*args, work=work, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.loglik_grad(*args, work=work, out=out)
e1.record()
torch.cuda.synchronize()
print("B %d N %d: %.3f ms per step" % (B, N, e0.elapsed_time(e1) / reps))
