import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
for B, N, J in ((1, 100000, 8), (1, 4096, 8), (4, 50000, 4)):
    x, diag, y, ac, bc, cc, dc = synth.device_coeffs_fast(0, B, N, J, dev)
    e = torch.zeros((B, 0), dtype=torch.float64, device=dev)
    for mode in ("0", "1"):
        os.environ["C2_TIMEPAR_GRAD"] = mode; os.environ["C2_FACTOR_ITER"] = mode
        work = ops.loglik_terms_workspace(B, N, 0, J // 2, dev, grad=True)
        outs = None
        for _ in range(2): ll, outs, flag = ops.loglik_terms_grad(e, e, ac, bc, cc, dc, x, diag, y, work=work, out=outs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): ll, outs, flag = ops.loglik_terms_grad(e, e, ac, bc, cc, dc, x, diag, y, work=work, out=outs)
        torch.cuda.synchronize(); print(B, N, J, "timepar=" + mode, "terms grad %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3), float(ll[0]))
