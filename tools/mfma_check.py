"""Matrix-core matmul_lower / dot_tril (J = 16): parity vs the CPU oracle, then A/B timing against the VALU path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from celerite2_amd import ops
from oracle import cpu, dense
dev = torch.device("cuda:0")
def D(*xs): return [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in xs]
def err(a, b): return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
rng = np.random.default_rng(3)
for (B, N, nrhs, gap) in [(1, 40000, 32, False), (1, 40003, 32, False), (3, 20001, 16, False), (2, 17000, 64, False), (1, 40000, 32, True), (1, 16384, 32, False)]:
    J = 16
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    if gap:
        t[:, N // 3:] += 5000.0; t[:, N // 2 + 5:] += 1e5
    Y = rng.standard_normal((B, N, nrhs))
    td, cd, ad, Ud, Vd, Yd = D(t, c, a, U, V, Y)
    os.environ["C2_MFMA"] = "1"
    Z0 = rng.standard_normal((B, N, nrhs)); (Zd,) = D(Z0)
    Zd = ops.matmul_lower(td, cd, Ud, Vd, Yd, Z=Zd)
    e1 = 0.0
    for b in range(B):
        Zo = Z0[b].copy(); cpu.matmul_lower(t[b], c[b], U[b], V[b], Y[b], Zo)
        e1 = max(e1, err(Zd[b].cpu().numpy(), Zo))
    d, W, flag = ops.factor(td, cd, ad, Ud, Vd)
    Yc = Yd.clone()
    Zi = ops.dot_tril(td, cd, Ud, W, d, Yc, Z=Yc)
    e2 = 0.0
    for b in range(B):
        z = np.ascontiguousarray(Y[b] * np.sqrt(d[b].cpu().numpy())[:, None])
        cpu.matmul_lower(t[b], c[b], U[b], W[b].cpu().numpy(), z, z)
        e2 = max(e2, err(Zi[b].cpu().numpy(), z))
    os.environ["C2_MFMA"] = "0"
    Zv = ops.dot_tril(td, cd, Ud, W, d, Yd)
    e3 = err(Zi.cpu().numpy(), Zv.cpu().numpy())
    print("B %d N %d nrhs %d gap %s: matmul_lower err %.1e  dot_tril(in place) err %.1e  mfma vs valu %.1e" % (B, N, nrhs, gap, e1, e2, e3), flush=True)
if "--time" in sys.argv:
    N, J, nrhs = 10_000_000, 16, 32
    t, c, a, U, V, y = dense.synthetic_batch(1, N, J)
    td, cd, ad, Ud, Vd = D(t, c, a, U, V)
    d, W, flag = ops.factor(td, cd, ad, Ud, Vd)
    Yd = torch.randn((1, N, nrhs), dtype=torch.float64, device=dev)
    for m in ("0", "1"):
        os.environ["C2_MFMA"] = m
        for _ in range(2): ops.dot_tril(td, cd, Ud, W, d, Yd)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): ops.dot_tril(td, cd, Ud, W, d, Yd)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print("dot_tril N=1e7 J=16 nrhs=32  C2_MFMA=%s  %.2f ms  (%.2f TB/s of 7.84 GB algorithmic, frac %.3f)" % (m, dt * 1e3, 7.84e9 / dt / 1e12, 7.84e9 / dt / 8e12), flush=True)
