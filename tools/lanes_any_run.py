"""The fused gradient pair with a forced lane mapping, for profilers: python tools/lanes_any_run.py <lanes> [N] [B] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import _lib, ops, synth
lanes = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16384; reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
_lib.set_option("lanes", lanes)
args = synth.device_batch_fast(0, B, N, 8, dev)
work = ops.loglik_grad_workspace(B, N, 8, dev)
out = None
for _ in range(2 + reps): ll, out, fl = ops.loglik_grad(*args, work=work, out=out)
torch.cuda.synchronize()
