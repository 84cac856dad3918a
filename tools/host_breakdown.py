import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from celerite2_amd import ops, driver, backprop
from oracle import dense
for (N, J) in [(1000, 2), (4096, 8)]:
    co = dense.sho_sum_coeffs(J); rng = np.random.default_rng(1)
    t = np.sort(rng.uniform(0, N / 10.0, N)); diag = rng.uniform(0.1, 0.3, N)
    c, a, U, V = dense.celerite_matrices(co, t, diag)
    dv = lambda x: torch.from_numpy(x[None].copy()).cuda()
    td, cd, ad, Ud, Vd = map(dv, (t, c, a, U, V))
    Y = rng.standard_normal((N, 1)); Yd = dv(Y)
    d, W, fl = ops.factor(td, cd, ad, Ud, Vd)
    def timeit(fn, n=200):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    def timeit_sync(fn, n=200):
        for _ in range(5): fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    dd = torch.empty_like(ad); WW = torch.empty_like(Vd); SS = torch.empty((1, N, J, J), dtype=torch.float64, device="cuda")
    print("N=%d J=%d device-pointer ops, us per call (back-to-back / with a sync after each):" % (N, J))
    print("  factor            %.1f / %.1f" % (timeit(lambda: ops.factor(td, cd, ad, Ud, Vd, d=dd, W=WW)), timeit_sync(lambda: ops.factor(td, cd, ad, Ud, Vd, d=dd, W=WW))))
    print("  factor + S        %.1f / %.1f" % (timeit(lambda: ops.factor(td, cd, ad, Ud, Vd, d=dd, W=WW, S=SS)), timeit_sync(lambda: ops.factor(td, cd, ad, Ud, Vd, d=dd, W=WW, S=SS))))
    Z = torch.empty_like(Yd); F = torch.empty((1, N, J, 1), dtype=torch.float64, device="cuda")
    print("  solve_lower       %.1f / %.1f" % (timeit(lambda: ops.solve_lower(td, cd, Ud, W, Yd, Z=Z)), timeit_sync(lambda: ops.solve_lower(td, cd, Ud, W, Yd, Z=Z))))
    print("  solve_lower + F   %.1f / %.1f" % (timeit(lambda: ops.solve_lower(td, cd, Ud, W, Yd, Z=Z, F=F)), timeit_sync(lambda: ops.solve_lower(td, cd, Ud, W, Yd, Z=Z, F=F))))
    bZ = torch.randn_like(Z)
    print("  solve_lower_rev   %.1f / %.1f" % (timeit(lambda: ops.solve_lower_rev(td, cd, Ud, W, Yd, Z, F, bZ)), timeit_sync(lambda: ops.solve_lower_rev(td, cd, Ud, W, Yd, Z, F, bZ))))
    bd, bW = torch.randn_like(d), torch.randn_like(W)
    print("  factor_rev        %.1f / %.1f" % (timeit(lambda: ops.factor_rev(td, cd, ad, Ud, Vd, d, W, SS, bd, bW)), timeit_sync(lambda: ops.factor_rev(td, cd, ad, Ud, Vd, d, W, SS, bd, bW))))
    # raw copies of the same sizes through a pinned buffer
    nb = 8 * N * (3 + 3 * J)
    hp = torch.empty(nb, dtype=torch.uint8).pin_memory(); dp = torch.empty(nb, dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    def rt():
        with torch.cuda.stream(s):
            dp.copy_(hp, non_blocking=True); hp.copy_(dp, non_blocking=True)
        s.synchronize()
    for _ in range(5): rt()
    t0 = time.perf_counter()
    for _ in range(200): rt()
    print("  pinned H2D + D2H of %d KB + stream sync: %.1f us" % (nb // 1000, (time.perf_counter() - t0) / 200 * 1e6))
