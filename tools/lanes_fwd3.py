import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
args = synth.device_batch_fast(0, 65536, 4096, 8, dev)
for lanes in ("1", "4"):
    os.environ["C2_LANES"] = lanes
    for _ in range(3): ops.loglik(*args)
    torch.cuda.synchronize()
