# -*- coding: utf-8 -*-
"""numpy prototype of VERDICT r05 item 2: drop the W and (d, z) records of the fused forward / reverse pair and RE-DERIVE them
in the backward recursion from the inputs (forward.hpp:126-131, internal.hpp:140-144).

    python tools/proto_rederive.py [J] [series] [diag scale]

With S the state row n - 1 saw and X = S + d w^T w = P^-1 S_n P^-1 (what the backward sweep forms from the state row n saw):
    r = V - U X,  e = a - U X U^T,  q = (U r^T) / e,  s = q / (1 - q) (= u . w),  d = e / (1 - s^2),  w = r / (d (1 - s)),
    z = (y - U G) / (1 - s)  with G = P^-1 F_n,            then  S = X - d w^T w,  F = G - w^T z.
Exact in exact arithmetic; 1 - s = (a - u.v) / d = (white noise) / d for celerite-built inputs.

What the prototype measures, on the bench generator (synth.host_inputs: diag ~ U(0.1, 0.3)) and on small-diag draws:
  * the error of the re-derived w, d (relative to the largest entry of W / d over the interval) as a function of the distance
    from the anchor the backward sweep started at (anchors every 8 / 16 / 32 rows: the pair's interval is 32), against the
    forward recursion's own values; the same recursion in long double shows what part is rounding;
  * the per-row amplification of a perturbation of X (the map X -> S is the INVERSE of the forward Riccati step, which
    contracts by (1 - s) = diag / d per row in the direction of w: the inverse expands by d / diag);
  * the recorded-W backward recursion (today's kernels: S = X - d w^T w with d, w READ) for comparison: no feedback.
Kill criterion of the item: error > 1e-11 of the largest entry on the bench generator."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celerite2_amd import synth  # noqa: E402
from oracle import cpu  # noqa: E402

J = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
DSCALE = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
N = 4096


def inputs(i, dtype=np.float64):
    t, diag, y, ac, bc, cc, dc = synth.host_inputs(0, NS, N, J)[:7]
    z = np.zeros(0)
    a = np.empty(N); U = np.empty((N, J)); V = np.empty((N, J))
    cpu.get_celerite_matrices(z, ac[i], bc[i], dc[i], t[i], diag[i] * DSCALE, a, U, V)
    return [np.asarray(x, dtype=dtype) for x in (t[i], np.repeat(cc[i], 2), a, U, V, y[i])], diag[i] * DSCALE


def forward(t, c, a, U, V, y):
    """forward.hpp:105-134 + internal.hpp:135-145: returns d, W, z and the states (S_n, F_n) every row SEES (post-decay)."""
    dt = t.dtype
    S = np.zeros((J, J), dt); F = np.zeros(J, dt)
    d = np.empty(N, dt); W = np.empty((N, J), dt); z = np.empty(N, dt)
    Ss = np.empty((N, J, J), dt); Fs = np.empty((N, J), dt)
    for n in range(N):
        if n:
            p = np.exp(-c * (t[n] - t[n - 1]))
            S = (S + d[n - 1] * np.outer(W[n - 1], W[n - 1])) * np.outer(p, p)
            F = p * (F + W[n - 1] * z[n - 1])
        Ss[n] = S; Fs[n] = F
        tau = U[n] @ S
        d[n] = a[n] - tau @ U[n]
        W[n] = (V[n] - tau) / d[n]
        z[n] = y[n] - U[n] @ F
    return d, W, z, Ss, Fs


def backward_rederive(t, c, a, U, V, y, Ss, Fs, n_hi, n_lo, perturb=None):
    """From the state row n_hi sees, walk down to row n_lo re-deriving (d, w, z) of rows n_hi - 1 ... n_lo."""
    S = Ss[n_hi].copy(); F = Fs[n_hi].copy()
    if perturb is not None:
        S = S + perturb
    out = {}
    for n in range(n_hi, n_lo, -1):
        pinv = np.exp(c * (t[n] - t[n - 1]))
        X = S * np.outer(pinv, pinv); G = pinv * F
        u, v = U[n - 1], V[n - 1]
        uX = u @ X
        r = v - uX
        e = a[n - 1] - uX @ u
        q = (u @ r) / e
        s = q / (1 - q)
        d = e / (1 - s * s)
        w = r / (d * (1 - s))
        zz = (y[n - 1] - u @ G) / (1 - s)
        S = X - d * np.outer(w, w); F = G - w * zz
        out[n - 1] = (d, w, zz, 1 - s)
    return out, S


def backward_recorded(t, c, d, W, z, Ss, Fs, n_hi, n_lo):
    S = Ss[n_hi].copy(); F = Fs[n_hi].copy()
    for n in range(n_hi, n_lo, -1):
        pinv = np.exp(c * (t[n] - t[n - 1]))
        S = S * np.outer(pinv, pinv) - d[n - 1] * np.outer(W[n - 1], W[n - 1])
        F = pinv * F - W[n - 1] * z[n - 1]
    return S, F


def main():
    print("J = %d, N = %d, %d series of the bench generator, diag x %g" % (J, N, NS, DSCALE))
    for interval in (8, 16, 32):
        worst_w = np.zeros(interval + 1); worst_d = np.zeros(interval + 1); worst_z = np.zeros(interval + 1)
        worst_rec = 0.0; min1s = 1.0; amp = []
        worst_w_ld = 0.0
        for i in range(NS):
            (t, c, a, U, V, y), diag = inputs(i)
            d, W, z, Ss, Fs = forward(t, c, a, U, V, y)
            ld = [np.asarray(x, dtype=np.longdouble) for x in (t, c, a, U, V, y)]
            dl, Wl, zl, Ssl, Fsl = forward(*ld) if interval == 32 and i < 2 else (None,) * 5
            for hi in range(interval, N, interval * 8):   # every 8th interval: enough for the statistics, 8 x faster
                lo = hi - interval
                got, S_lo = backward_rederive(t, c, a, U, V, y, Ss, Fs, hi, lo)
                wmax = np.abs(W[lo:hi]).max(); dmax = np.abs(d[lo:hi]).max(); zmax = max(np.abs(z[lo:hi]).max(), 1e-300)
                for n, (dd, w, zz, oms) in got.items():
                    k = hi - n
                    worst_w[k] = max(worst_w[k], np.abs(w - W[n]).max() / wmax)
                    worst_d[k] = max(worst_d[k], abs(dd - d[n]) / dmax)
                    worst_z[k] = max(worst_z[k], abs(zz - z[n]) / zmax)
                    min1s = min(min1s, oms)
                Sr, Fr = backward_recorded(t, c, d, W, z, Ss, Fs, hi, lo)
                worst_rec = max(worst_rec, np.abs(Sr - Ss[lo]).max() / np.abs(Ss[hi]).max())
                if hi % (interval * 64) == interval:   # amplification of a perturbation of the anchor state, per row
                    rng = np.random.default_rng(hi)
                    E = rng.standard_normal((J, J)); E = 1e-9 * (E + E.T) * np.abs(Ss[hi]).max()
                    _, S_p = backward_rederive(t, c, a, U, V, y, Ss, Fs, hi, lo, perturb=E)
                    amp.append((np.abs(S_p - S_lo).max() / np.abs(E).max()) ** (1.0 / interval))
                if dl is not None and hi < 600:
                    gl, _ = backward_rederive(*ld, Ssl, Fsl, hi, lo)
                    worst_w_ld = max(worst_w_ld, max(float(np.abs(w - Wl[n]).max() / wmax) for n, (_, w, _, _) in gl.items()))
        ks = [1, 2, 4, 8, 16, 32]
        print("anchors every %2d rows: re-derived w, error / max|W| at k rows below the anchor: %s"
              % (interval, "  ".join("k=%d: %.1e" % (k, worst_w[k]) for k in ks if k <= interval)))
        print("                       d: %s   z: %s" % ("  ".join("%.1e" % worst_d[k] for k in ks if k <= interval),
                                                        "  ".join("%.1e" % worst_z[k] for k in ks if k <= interval)))
        print("                       min(1 - s) = %.3f (= diag / d: the forward step contracts by it, the inverse expands by 1 / it);"
              " measured growth of a 1e-9 perturbation of the anchor state: x %.2f per row (median), x %.2f (max)"
              % (min1s, float(np.median(amp)), float(np.max(amp))))
        print("                       recorded-W backward recursion (today's kernels), state at the interval's start: %.1e" % worst_rec)
        if worst_w_ld:
            print("                       the same re-derivation in long double (first 600 rows of 2 series): %.1e" % worst_w_ld)
        verdict = "KILL (> 1e-11)" if worst_w[min(interval, 32)] > 1e-11 else "within 1e-11"
        print("                       -> %s" % verdict)


if __name__ == "__main__":
    main()
