"""Forward-only log-likelihood: lane mappings 8 / 4 / 1 at the bench shape; parity of mapping 1 against the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celerite2_amd import ops, synth
from oracle import cpu, dense

dev = torch.device("cuda:0")
# parity, small odd batch
B, N, J = 70, 1031, 8
t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
d = [torch.from_numpy(x).to(dev) for x in (t, c, a, U, V, y)]
llo, _ = cpu.loglik_batched(t, c, a, U, V, y, nthreads=4)
for lanes in ("8", "1"):
    os.environ["C2_LANES"] = lanes
    ll, flag = ops.loglik(*d)
    print("lanes", lanes, "max rel err", float(np.abs(ll.cpu().numpy() - llo).max() / np.abs(llo).max()), "flags", int(flag.abs().sum()))
# unpaired c
c2 = c.copy(); c2[:, 1] *= 1.01
d2 = [torch.from_numpy(x).to(dev) for x in (t, c2, a, U, V, y)]
llo2, _ = cpu.loglik_batched(t, c2, a, U, V, y, nthreads=4)
os.environ["C2_LANES"] = "1"
ll, flag = ops.loglik(*d2)
print("lanes 1 unpaired max rel err", float(np.abs(ll.cpu().numpy() - llo2).max() / np.abs(llo2).max()))
for Bb in (65536, 32768):
    args = synth.device_batch_fast(0, Bb, 4096, 8, dev)
    for lanes in ("8", "4", "1"):
        os.environ["C2_LANES"] = lanes
        for _ in range(2): ops.loglik(*args)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): ll, flag = ops.loglik(*args)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print("B", Bb, "lanes", lanes, "%.2f ms" % (dt * 1e3), "%.2f M GP/s" % (Bb / dt / 1e6), "alg TB/s %.2f" % (Bb * 4096 * 152 / dt / 1e12))
    del args
