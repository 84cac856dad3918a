"""Thread-per-series forward kernel: time vs batch size (occupancy) for the library selected by C2_LIB_PATH."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
N = int(os.environ.get("NN", "4096"))
os.environ["C2_LANES"] = "1"
for Bb in (16384, 32768, 49152, 65536):
    args = synth.device_batch_fast(0, Bb, N, 8, dev)
    for _ in range(2): ops.loglik(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ll, flag = ops.loglik(*args)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(os.environ.get("C2_LIB_PATH", "default")[-12:], "N", N, "B", Bb, "%.2f ms" % (dt * 1e3), "%.2f M GP/s" % (Bb / dt / 1e6), "alg TB/s %.2f" % (Bb * N * 152 / dt / 1e12), flush=True)
    del args
