# -*- coding: utf-8 -*-
"""Calibration of the device-side verification of the time-parallel gradient (c2_timepar_grad.hip, k_verify_combine).

For every seed of tests/test_gpu_fuzz.py::test_fuzz_time_parallel_gradient (same draws) it runs the time-parallel form with
the gated fallback switched OFF (C2_VERIFY_FALLBACK=0), reads the verification words through the diagnostics sink and
prints them next to the actual distance to the CPU oracle:

    seed B N J rows | gate es eb kappa ef ez | newton updates ... | err_max (largest-entry norm) err_elem (close() units)

    python tools/verify_words.py 30 6030 [extra seeds ...]      # base count
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from celerite2_amd import _lib, ops  # noqa: E402
from oracle import cpu as orc  # noqa: E402
from oracle import dense  # noqa: E402


def problem(rng, B, N, J):
    Je = J if J % 2 == 0 else J + 1
    t, c, a, U, V, y = dense.synthetic_batch(B, max(N, 2), Je)
    t = np.ascontiguousarray(t[:, :N]); a = np.ascontiguousarray(a[:, :N]) + 1.0
    U = np.ascontiguousarray(U[:, :N, :J]); V = np.ascontiguousarray(V[:, :N, :J])
    c = np.ascontiguousarray(c[:, :J]); y = np.ascontiguousarray(y[:, :N])
    return t, c, a, U, V, y


def dev(*xs):
    return [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in xs]


def draw(seed):
    rng = np.random.default_rng(77000 + seed)
    B = int(rng.choice([1, 2, 3, 5, 9, 70]))
    N = int(rng.choice([1, 2, 3, 63, 64, 65, 127, 128, 129, 200, 449, 640, 1000, 2100]))
    J = int(rng.choice([8, 7, 6, 5, 4, 3, 2, 1]))
    rows = [None, "16", "32", "64"][seed % 4]
    if seed >= 30 and B < 70 and rng.random() < 0.3:
        N = int(rng.choice([2500, 4096, 4100, 7000]))
    t, c, a, U, V, y = problem(rng, B, N, J)
    if rng.random() < 0.4:
        c = c * rng.uniform(0.8, 1.25, c.shape)
    if N > 70 and rng.random() < 0.4:
        t[:, N // 2:] += rng.choice([2.0, 50.0, 3000.0])
    shared_t = rng.random() < 0.3
    shared_c = rng.random() < 0.3
    if shared_t: t = np.tile(t[0], (B, 1))
    if shared_c: c = np.tile(c[0], (B, 1))
    if B > 1 and N > 10 and rng.random() < 0.3:
        a[B // 2, N // 3] = -1.0
    return B, N, J, rows, t, c, a, U, V, y, shared_t, shared_c


def main():
    base, count = int(sys.argv[1]), int(sys.argv[2])
    seeds = list(range(base, base + count)) + [int(x) for x in sys.argv[3:]]
    orc.build()
    lib = _lib.load()
    sink = torch.zeros(64, dtype=torch.float64, device="cuda")
    lib.c2_internal_set_debug_sink.argtypes = [ctypes.c_void_p]
    lib.c2_internal_set_debug_sink.restype = None
    lib.c2_internal_set_debug_sink(ctypes.c_void_p(sink.data_ptr()))
    os.environ["C2_TIMEPAR_GRAD"] = "1"
    os.environ["C2_FACTOR_ITER"] = "1"
    worst = []
    for seed in seeds:
        B, N, J, rows, t, c, a, U, V, y, shared_t, shared_c = draw(seed)
        if N < 2:
            continue
        if rows: os.environ["C2_TPG_ROWS"] = rows
        else: os.environ.pop("C2_TPG_ROWS", None)
        llo, go, flago = orc.loglik_grad_batched(t, c, a, U, V, y, nthreads=4)
        ok = np.asarray(flago) == 0
        args = dev(t[0].copy() if shared_t else t, c[0].copy() if shared_c else c, a, U, V, y)
        out = {}
        for fb in ("0", "1"):
            os.environ["C2_VERIFY_FALLBACK"] = fb
            sink.zero_()
            ll, grads, flag = ops.loglik_grad(*args)
            torch.cuda.synchronize()
            w = sink.cpu().numpy().copy()
            emax = eel = 0.0
            for g, e in zip(grads, go):
                gn = g.cpu().numpy()
                for b in np.nonzero(ok)[0]:
                    m = max(np.abs(e[b]).max(), 1e-300)
                    diff = np.abs(gn[b] - e[b])
                    emax = max(emax, float(diff.max() / m))
                    eel = max(eel, float((diff / (1e-10 * np.abs(e[b]) + 1e-12 * max(1.0, m))).max()))
            if ok.any():
                emax = max(emax, float(np.max(np.abs(ll.cpu().numpy()[ok] - llo[ok]) / np.abs(llo[ok]))))
            out[fb] = (emax, eel, w)
        # the row-by-row kernels on the same draw
        os.environ["C2_TIMEPAR_GRAD"] = "0"; os.environ["C2_FACTOR_ITER"] = "0"
        ll, grads, flag = ops.loglik_grad(*args)
        os.environ["C2_TIMEPAR_GRAD"] = "1"; os.environ["C2_FACTOR_ITER"] = "1"
        erow = 0.0
        for g, e in zip(grads, go):
            gn = g.cpu().numpy()
            for b in np.nonzero(ok)[0]:
                erow = max(erow, float(np.abs(gn[b] - e[b]).max() / max(np.abs(e[b]).max(), 1e-300)))
        emax, eel, w = out["0"]
        nw = w[8:8 + 10]; kp = w[8 + 10:8 + 20]
        print("seed %d B %d N %d J %d rows %s ok %d | gate %.2e es %.1e eb %.1e kap %.1e ef %.1e ez %.1e | newton %s | kap %s | err_max %.2e err_elem %.2e | with fallback: err_max %.2e err_elem %.2e | row-by-row: err_max %.2e"
              % (seed, B, N, J, rows, ok.sum(), w[0], w[1], w[2], w[3], w[4], w[5],
                 " ".join("%.1e" % x for x in nw[1:9]), " ".join("%.0e" % x for x in kp[1:9]), emax, eel, out["1"][0], out["1"][1], erow), flush=True)
        worst.append((emax, seed))
    worst.sort(reverse=True)
    print("WORST", worst[:20])


if __name__ == "__main__":
    main()
