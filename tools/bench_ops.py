#!/usr/bin/env python
"""Per-op measurements on one MI355X: HIP-event time of each batched device op at
B series x N=4096 x J=8 and the algorithmic-byte rate (each input read once, each output written once).
Usage: [C2_BENCH_N=rows] [C2_BENCH_J=width] python tools/bench_ops.py [B [substring-of-op-name]]   -> markdown table on stdout."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from celerite2_amd import ops, synth


def timed(fn, reps=9):
    return synth.timed_steady(fn, reps=reps)   # (steady clock: profiles/r05_clock_ramp.md)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    N, J = int(os.environ.get("C2_BENCH_N", 4096)), int(os.environ.get("C2_BENCH_J", 8))
    dev = "cuda"
    t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
    f64 = dict(dtype=torch.float64, device=dev)
    d, W, S, flag = ops.factor(t, c, a, U, V, workspace=True)
    rows = []

    only = sys.argv[2] if len(sys.argv) > 2 else ""

    def add(name, row, fn, bytes_per_step):
        if only not in name:
            return
        ms = timed(fn)
        gb = B * N * bytes_per_step / 1e9
        rows.append((name, row, ms, B / ms * 1e3, gb / ms * 1e3, gb / ms * 1e3 / 8000))

    d2, W2 = torch.empty_like(a), torch.empty_like(V)
    add("factor (no workspace)", "A", lambda: ops.factor(t, c, a, U, V, d=d2, W=W2), 8 * (3 + 3 * J))
    add("factor + S workspace", "A", lambda: ops.factor(t, c, a, U, V, d=d2, W=W2, S=S), 8 * (3 + 3 * J + J * J))
    for nrhs in (1, 8):
        Y = torch.randn((B, N, nrhs), **f64)
        Z = torch.empty_like(Y)
        F = torch.empty((B, N, J, nrhs), **f64)
        bpf = 8 * (1 + 2 * J + 2 * nrhs)
        add("solve_lower nrhs=%d" % nrhs, "D", lambda: ops.solve_lower(t, c, U, W, Y, Z=Z), bpf)
        add("solve_upper nrhs=%d" % nrhs, "E", lambda: ops.solve_upper(t, c, U, W, Y, Z=Z), bpf)
        add("matmul_lower nrhs=%d" % nrhs, "F", lambda: ops.matmul_lower(t, c, U, V, Y, Z=Z, zero_z=True), bpf)
        add("matmul_upper nrhs=%d" % nrhs, "G", lambda: ops.matmul_upper(t, c, U, V, Y, Z=Z, zero_z=True), bpf)
        add("solve_lower + F workspace nrhs=%d" % nrhs, "D", lambda: ops.solve_lower(t, c, U, W, Y, Z=Z, F=F), bpf + 8 * J * nrhs)
        bZ = torch.randn_like(Y)
        Zs, Fs = ops.solve_lower(t, c, U, W, Y, workspace=True)
        add("solve_lower_rev nrhs=%d" % nrhs, "I", lambda: ops.solve_lower_rev(t, c, U, W, Y, Zs, Fs, bZ),
            8 * (2 + 4 * J + 4 * nrhs + J * nrhs))
        Zm, Fm = ops.matmul_upper(t, c, U, V, Y, workspace=True, zero_z=True)
        add("matmul_upper_rev nrhs=%d" % nrhs, "J", lambda: ops.matmul_upper_rev(t, c, U, V, Y, Zm, Fm, bZ),
            8 * (2 + 4 * J + 4 * nrhs + J * nrhs))
    bd, bW = torch.randn_like(a), torch.randn_like(V)
    add("factor_rev", "H", lambda: ops.factor_rev(t, c, a, U, V, d, W, S, bd, bW), 8 * (5 + 5 * J + J * J))
    M = N
    ts = (t + 0.03).contiguous()
    for nrhs in (1, 8):
        Y = torch.randn((B, N, nrhs), **f64)
        Zg = torch.zeros((B, N, nrhs), **f64)
        add("general_matmul_lower (M=N) nrhs=%d" % nrhs, "K", lambda: ops.general_matmul_lower(ts, t, c, U, V, Y, Z=Zg),
            16 * (1 + J + nrhs))
        add("general_matmul_upper (M=N) nrhs=%d" % nrhs, "K", lambda: ops.general_matmul_upper(ts, t, c, U, V, Y, Z=Zg),
            16 * (1 + J + nrhs))
        Fg = torch.empty((B, N, J, nrhs), **f64)
        add("general_matmul_lower + F workspace nrhs=%d" % nrhs, "K",
            lambda: ops.general_matmul_lower(ts, t, c, U, V, Y, Z=Zg, F=Fg), 16 * (1 + J + nrhs) + 8 * J * nrhs)
        del Fg
    ll = lambda: ops.loglik(t, c, a, U, V, y)
    add("fused log-lik", "L", ll, 8 * (3 + 2 * J))
    work = ops.loglik_grad_workspace(B, N, J, dev)
    add("fused log-lik + grad", "L", lambda: ops.loglik_grad(t, c, a, U, V, y, work=work), 16 * (3 + 2 * J))
    diag = torch.rand((B, N), **f64)
    Jc = J // 2
    ac = torch.rand((B, Jc), **f64); ar = torch.zeros((B, 0), **f64)
    add("get_celerite_matrices", "M", lambda: ops.get_celerite_matrices(ar, ac, ac, ac, t, diag), 8 * (3 + 2 * J))
    print("| op | row | ms | series/s | GB/s (algorithmic) | frac of 8 TB/s |")
    print("|---|---|---|---|---|---|")
    for name, row, ms, sps, gbs, frac in rows:
        print("| %s | %s | %.2f | %.3g | %.0f | %.3f |" % (name, row, ms, sps, gbs, frac))


if __name__ == "__main__":
    main()
