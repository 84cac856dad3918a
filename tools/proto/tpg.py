"""Prototype (numpy, CPU) of the time-parallel GRADIENT of the fused log-likelihood: with d, W, z of every row known
(factor + solve), the states S, F obey linear recurrences, and the adjoint recursion is
    bF_n = A_n^T bF_{n+1} + ...,   bS_n = A_n^T bS_{n+1} A_n - (z_n/d_n) sym(bF_n u_n^T) + ...,   A_n = P_{n+1} (I - w_n u_n^T)
so a chunk of rows acts on the adjoint it receives as an affine map (Phi, C_k, g) found by J + 1 sweeps.
Checks the chunked result against the sequential oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import cpu, dense

def forward(t, c, a, U, V, y):
    N, J = U.shape
    S = np.zeros((J, J)); F = np.zeros(J)
    d = np.empty(N); W = np.empty((N, J)); z = np.empty(N); Sin = np.empty((N, J, J)); Fin = np.empty((N, J))
    for n in range(N):
        if n > 0:
            p = np.exp(-c * (t[n] - t[n - 1])); S = S * np.outer(p, p); F = F * p
        Sin[n] = S; Fin[n] = F
        tau = S @ U[n]; d[n] = a[n] - U[n] @ tau; W[n] = (V[n] - tau) / d[n]; z[n] = y[n] - U[n] @ F
        S = S + d[n] * np.outer(W[n], W[n]); F = F + W[n] * z[n]
    return d, W, z, Sin, Fin

def row_adjoint(n, N, t, c, u, w, d, z, bSn, bFn, src):
    """(bS', bF') of the state entering row n+1 -> (bS, bF) of the state entering row n; src: with the ll sources."""
    J = u.shape[0]
    if n + 1 < N:
        p = np.exp(-c * (t[n + 1] - t[n]))
    else:
        p = np.ones(J)
    bSp = bSn * np.outer(p, p); bFp = bFn * p
    g = bSp @ w
    bd = w @ g + (src * -0.5 * (1.0 / d - z * z / d ** 2))
    bw = 2 * d * g + z * bFp
    bz = w @ bFp + src * (-z / d)
    bF = bFp - bz * u
    bd -= (bw @ w) / d
    btau = -bw / d - bd * u
    bS = bSp + 0.5 * (np.outer(btau, u) + np.outer(u, btau))
    return bS, bF, dict(bd=bd, bw=bw, bz=bz, btau=btau, p=p)

def grad_sequential(t, c, a, U, V, y):
    N, J = U.shape
    d, W, z, Sin, Fin = forward(t, c, a, U, V, y)
    bS = np.zeros((J, J)); bF = np.zeros(J)
    bt = np.zeros(N); bc = np.zeros(J); ba = np.empty(N); bU = np.empty((N, J)); bV = np.empty((N, J)); by = np.empty(N)
    for n in range(N - 1, -1, -1):
        if n + 1 < N:   # gradient of the decay between rows n and n+1: uses the states entering row n+1
            pbp = 2 * np.sum(bS * Sin[n + 1], axis=1) + bF * Fin[n + 1]
            dt = t[n + 1] - t[n]
            bc += -dt * pbp
            bdt = -np.sum(c * pbp)
            bt[n + 1] += bdt; bt[n] -= bdt
        bS, bF, q = row_adjoint(n, N, t, c, U[n], W[n], d[n], z[n], bS, bF, 1.0)
        by[n] = q["bz"]; ba[n] = q["bd"]; bV[n] = q["bw"] / d[n]
        tau = V[n] - d[n] * W[n]
        bU[n] = -q["bz"] * Fin[n] - q["bd"] * tau + Sin[n] @ q["btau"]
    ll = -0.5 * np.sum(np.log(d) + z * z / d) - 0.5 * N * np.log(2 * np.pi)
    return ll, (bt, bc, ba, bU, bV, by)

def chunk_maps(lo, hi, N, t, c, U, W, d, z):
    """adjoint entering row hi (chunk end) -> adjoint entering row lo: Phi (bF_s = Phi^T bF_e, bS_s = Phi^T bS_e Phi
    + sum_k bF_e[k] C[k] + gS), gF."""
    J = U.shape[1]
    PhiT = np.empty((J, J)); C = np.empty((J, J, J))
    for k in range(J):
        bS = np.zeros((J, J)); bF = np.zeros(J); bF[k] = 1.0
        for n in range(hi - 1, lo - 1, -1):
            bS, bF, _ = row_adjoint(n, N, t, c, U[n], W[n], d[n], z[n], bS, bF, 0.0)
        PhiT[:, k] = bF; C[k] = bS
    bS = np.zeros((J, J)); bF = np.zeros(J)
    for n in range(hi - 1, lo - 1, -1):
        bS, bF, _ = row_adjoint(n, N, t, c, U[n], W[n], d[n], z[n], bS, bF, 1.0)
    return PhiT, C, bS, bF

def chunk_end_adjoints(t, c, U, W, d, z, L):
    N, J = U.shape
    K = (N + L - 1) // L
    ends = [None] * K
    bS = np.zeros((J, J)); bF = np.zeros(J)
    for k in range(K - 1, -1, -1):
        ends[k] = (bS.copy(), bF.copy())
        PhiT, C, gS, gF = chunk_maps(k * L, min(N, (k + 1) * L), N, t, c, U, W, d, z)
        Phi = PhiT.T
        bS, bF = PhiT @ bS @ Phi + np.tensordot(bF, C, axes=(0, 0)) + gS, PhiT @ bF + gF
    return ends

if __name__ == "__main__":
    for J, N, L in ((2, 50, 8), (4, 130, 16), (6, 100, 64), (8, 200, 64)):
        t, c, a, U, V, y = [x[0] for x in dense.synthetic_batch(1, N, J)]
        llo, go, _ = cpu.loglik_grad(t, c, a, U, V, y)
        ll, g = grad_sequential(t, c, a, U, V, y)
        err = max(float(np.abs(x - e).max() / max(np.abs(e).max(), 1e-300)) for x, e in zip(g, go))
        print("J %d N %d: sequential numpy vs oracle: ll %.1e grads %.1e" % (J, N, abs(ll - llo) / abs(llo), err))
        d, W, z, Sin, Fin = forward(t, c, a, U, V, y)
        ends = chunk_end_adjoints(t, c, U, W, d, z, L)
        # true adjoints at the chunk ends from the sequential sweep
        bS = np.zeros((J, J)); bF = np.zeros(J); worst = 0.0
        for n in range(N - 1, -1, -1):
            if (n + 1) % L == 0 or n == N - 1:
                k = n // L
                eS, eF = ends[k]
                sc = max(np.abs(bS).max(), np.abs(bF).max(), 1e-300)
                worst = max(worst, np.abs(eS - bS).max() / sc, np.abs(eF - bF).max() / sc)
            bS, bF, _ = row_adjoint(n, N, t, c, U[n], W[n], d[n], z[n], bS, bF, 1.0)
        print("   chunk-end adjoints from the chained maps vs sequential: %.1e" % worst)
