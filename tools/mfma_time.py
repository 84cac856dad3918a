import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
N, J, nrhs = 10_000_000, 16, 32
t, c, a, U, V, y = synth.device_batch_fast(0, 1, N, J, dev)
d, W, flag = ops.factor(t, c, a, U, V)
Yd = torch.randn((1, N, nrhs), dtype=torch.float64, device=dev)
for _ in range(6): ops.dot_tril(t, c, U, W, d, Yd)
torch.cuda.synchronize()
