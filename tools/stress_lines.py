"""Stress run of the line-pairing sweeps: random shapes (B a multiple of 8, N even or odd, one or eight right-hand sides; grid /
rates shared or not; in place, accumulating, with the F workspace), every result against the row-by-row kernels (options off).
    python tools/stress_lines.py [seeds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from celerite2_amd import _lib, ops
from oracle import dense
dev = torch.device("cuda:0")
OPTS = ("sweep1_lines", "sweepk_lines", "sweep_rev_lines")
def setall(v):
    for o in OPTS: _lib.set_option(o, v)
def close(a, b, what):
    a = a.double().cpu().numpy(); b = b.double().cpu().numpy()
    m = max(np.abs(b).max(), 1e-300)
    err = np.abs(a - b).max() / m
    assert np.isfinite(a).all() and err < 1e-11, (what, err)
    return err
worst = 0.0
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for seed in range(nseeds):
    rng = np.random.default_rng(424242 + seed)
    B = int(rng.choice([8, 9, 15, 16, 17, 24, 29, 40, 64, 67])); N = int(rng.choice([8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 30, 31, 32, 33, 34, 63, 64, 65, 66, 100, 257, 258, 600]))
    nrhs = int(rng.choice([1, 8])); J = 8
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    if rng.random() < 0.3: t[:, N // 2:] += rng.choice([1.0, 30.0])
    W = (0.3 / J) * rng.standard_normal((B, N, J))
    Y = rng.standard_normal((B, N, nrhs)); bZ = rng.standard_normal((B, N, nrhs)); Z0 = rng.standard_normal((B, N, nrhs))
    sh_t, sh_c = rng.random() < 0.3, rng.random() < 0.3
    f = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    td, cd = f(t[0] if sh_t else t), f(c[0] if sh_c else c)
    Ud, Vd, Wd, Yd, bZd = map(f, (U, V, W, Y, bZ))
    for name in ("solve_lower", "solve_upper", "matmul_lower", "matmul_upper"):
        solve = name.startswith("solve"); secd = Wd if solve else Vd
        def run():
            out = []
            Zd, Fd = getattr(ops, name)(td, cd, Ud, secd, Yd, workspace=True, zero_z=True)
            out += [Zd, Fd]
            out.append(getattr(ops, name)(td, cd, Ud, secd, Yd) if solve else getattr(ops, name)(td, cd, Ud, secd, Yd, zero_z=True))
            Yc = Yd.clone(); out.append(getattr(ops, name)(td, cd, Ud, secd, Yc, Z=Yc))
            if not solve: out.append(getattr(ops, name)(td, cd, Ud, secd, Yd, Z=f(Z0)))
            out += list(getattr(ops, name + "_rev")(td, cd, Ud, secd, Yd, Zd, Fd, bZd))
            return out
        setall(None); res = run()
        setall(0); ref = run()
        setall(None)
        for i, (r, e) in enumerate(zip(res, ref)):
            worst = max(worst, close(r, e, (seed, B, N, nrhs, name, i, sh_t, sh_c)))
    if seed % 20 == 0: print("seed", seed, "B", B, "N", N, "nrhs", nrhs, "worst so far %.2e" % worst, flush=True)
print("OK: %d seeds, largest relative difference between the two kernel families %.2e" % (nseeds, worst))
