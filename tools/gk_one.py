import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
B, N, J = 8192, 4096, 8
t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
t1 = (t + 0.013).contiguous()
Y = torch.randn((B, N, 8), dtype=torch.float64, device=dev)
for _ in range(3): Z = ops.general_matmul_lower(t1, t, c, U, V, Y)
torch.cuda.synchronize()
