"""Fused log-lik + gradient: lane mappings 8 vs 1 (one lane per series) -- parity vs the oracle on small batches (odd sizes,
segment edges), then timing at the bench shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celerite2_amd import ops, synth
from oracle import cpu, dense
dev = torch.device("cuda:0")
NAMES = ("bt", "bc", "ba", "bU", "bV", "by")
def check(B, N, tweak=None):
    t, c, a, U, V, y = dense.synthetic_batch(B, N, 8)
    if tweak: tweak(t, c, a, U, V, y)
    d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (t, c, a, U, V, y)]
    llo, go, flo = cpu.loglik_grad_batched(t, c, a, U, V, y, nthreads=4)
    os.environ["C2_LANES"] = "1"
    ll, g, flag = ops.loglik_grad(*d)
    torch.cuda.synchronize()
    ok = flo == 0
    errs = {"ll": float(np.abs(ll.cpu().numpy()[ok] - llo[ok]).max() / np.abs(llo[ok]).max())}
    for nm, x, e in zip(NAMES, g, go):
        x = x.cpu().numpy()
        errs[nm] = float(np.abs(x[ok] - e[ok]).max() / np.abs(e[ok]).max())
        if (~ok).any(): assert np.isnan(x[~ok]).all(), nm
    print("B %d N %d" % (B, N), "flags", flag.cpu().tolist() if B < 12 else int((flag != 0).sum()), {k: "%.1e" % v for k, v in errs.items()}, flush=True)
for (B, N) in [(1, 1), (3, 2), (5, 3), (70, 17), (64, 16), (65, 33), (130, 100), (7, 1031)]:
    check(B, N)
def unpaired(t, c, a, U, V, y): c[:, 1] *= 1.01
check(9, 257, unpaired)
def bad(t, c, a, U, V, y): a[2, 40] = -3.0
check(9, 100, bad)
def big_gap(t, c, a, U, V, y): t[:, 50:] += 400.0   # c * span >> guard: the gated replay kernels must take over
check(9, 100, big_gap)
if "--time" in sys.argv:
    for Bb in (65536,):
        args = synth.device_batch_fast(0, Bb, 4096, 8, dev)
        for lanes in ("8", "1"):
            os.environ["C2_LANES"] = lanes
            work = ops.loglik_grad_workspace(Bb, 4096, 8, dev)
            out = None
            for _ in range(2): ll, out, fl = ops.loglik_grad(*args, work=work, out=out)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(4): ops.loglik_grad(*args, work=work, out=out)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
            print("B", Bb, "lanes", lanes, "%.2f ms" % (dt * 1e3), "%.3f M GP/s" % (Bb / dt / 1e6), "frac %.3f" % (Bb * 1245320 / dt / 8e12), flush=True)
            del work, out
