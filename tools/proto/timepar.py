# -*- coding: utf-8 -*-
"""numpy prototype of the TIME-PARALLEL forward log-likelihood (c2_timepar.hip): the Riccati recursion of `factor`
(forward.hpp:105-134) is a linear-fractional map of the state, T' = (A T + B)(C T + D)^-1 with
[[A, B], [C, D]] = diag(P, P^-1) (kappa I + x y^T), x = [v; u], y = [-u; v], kappa = a - u.v (the white-noise diagonal),
so chunks of rows compose independently; the solve recursion (internal.hpp:135-145) is affine in F given (d, W)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import dense


def sequential(t, c, a, U, V, y):
    N, J = U.shape
    S = np.zeros((J, J)); F = np.zeros(J)
    d = a[0]; w = V[0] / d; z = y[0]
    logdet = np.log(d); quad = z * z / d
    for n in range(1, N):
        p = np.exp(-c * (t[n] - t[n - 1]))
        S = (S + d * np.outer(w, w)) * np.outer(p, p)
        F = p * (F + w * z)
        tmp = U[n] @ S
        d = a[n] - tmp @ U[n]; w = (V[n] - tmp) / d; z = y[n] - U[n] @ F
        logdet += np.log(d); quad += z * z / d
    return -0.5 * (logdet + quad + N * np.log(2 * np.pi))


def timepar(t, c, a, U, V, y, T=32):
    N, J = U.shape
    K = (N + T - 1) // T
    # K1: composite maps of chunks 0 .. K-2 (rows s..e-1, decays towards rows s+1..e)
    maps = []
    for k in range(K - 1):
        s, e = k * T, min(N, (k + 1) * T)
        R = np.eye(2 * J)
        for n in range(s, e):
            u, v = U[n], V[n]; kap = a[n] - u @ v
            x = np.concatenate([v, u]) / kap; yv = np.concatenate([-u, v])
            R = R + np.outer(x, yv @ R)
            p = np.exp(-c * (t[n + 1] - t[n]))
            R *= np.concatenate([p, 1.0 / p])[:, None]
        maps.append(R)
    # K2: chunk-start states
    Ss = [np.zeros((J, J))]
    for k in range(K - 1):
        XY = maps[k] @ np.vstack([Ss[-1], np.eye(J)])
        Sn = np.linalg.solve(XY[J:].T, XY[:J].T).T
        Ss.append(0.5 * (Sn + Sn.T))
    # K3: per chunk, sequential recursion from the chunk-start state; affine data for F
    outs = []
    for k in range(K):
        s, e = k * T, min(N, (k + 1) * T)
        S = Ss[k].copy(); G = np.eye(J); g = np.zeros(J)
        logdet = 0.0; q0 = 0.0; q1 = np.zeros(J); Q2 = np.zeros((J, J))
        for n in range(s, e):
            u, v = U[n], V[n]
            tmp = u @ S
            d = a[n] - tmp @ u; w = (v - tmp) / d
            r = u @ G; z0 = y[n] - u @ g
            logdet += np.log(d); q0 += z0 * z0 / d; q1 += z0 * r / d; Q2 += np.outer(r, r) / d
            if n + 1 < N:
                p = np.exp(-c * (t[n + 1] - t[n]))
                S = (S + d * np.outer(w, w)) * np.outer(p, p)
                G = p[:, None] * (G - np.outer(w, r))
                g = p * (g + w * z0)
        outs.append((S, G, g, logdet, q0, q1, Q2))
    # K4: chain F, verify the chunk-start states
    F = np.zeros(J); logdet = 0.0; quad = 0.0; mismatch = 0.0
    for k in range(K):
        S_end, G, g, ld, q0, q1, Q2 = outs[k]
        if k + 1 < K:
            mismatch = max(mismatch, np.abs(S_end - Ss[k + 1]).max() / max(np.abs(S_end).max(), 1e-300))
        logdet += ld; quad += q0 - 2 * q1 @ F + F @ Q2 @ F
        F = G @ F + g
    return -0.5 * (logdet + quad + N * np.log(2 * np.pi)), mismatch


if __name__ == "__main__":
    for J in (2, 4):
        t, c, a, U, V, y = dense.synthetic_batch(4, 4096, J)
        for b in range(4):
            ref = sequential(t[b], c[b], a[b], U[b], V[b], y[b])
            for T in (16, 32, 64):
                ll, mm = timepar(t[b], c[b], a[b], U[b], V[b], y[b], T)
                print("J %d series %d T %3d  rel err of ll %.2e  chunk-start mismatch %.2e" % (J, b, T, abs(ll - ref) / abs(ref), mm))
