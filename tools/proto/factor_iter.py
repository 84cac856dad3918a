"""Prototype of the chunk-pass factor: how fast do the chunk end states converge?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import dense
J, N, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
t, c, a, U, V, y = [x[0] for x in dense.synthetic_batch(1, N, J)]
K = (N + L - 1) // L
ends = np.zeros((K, J, J))
print("c", c, "dt mean", np.diff(t).mean())
for p in range(1, 30):
    new = np.empty_like(ends); worst = 0.0; wk = -1
    for k in range(K):
        S = ends[k - 1].copy() if k > 0 else np.zeros((J, J))
        for n in range(k * L, min(N, (k + 1) * L)):
            tau = S @ U[n]; d = a[n] - U[n] @ tau; w = (V[n] - tau) / d
            S = S + d * np.outer(w, w)
            if n + 1 < N:
                pdec = np.exp(-c * (t[n + 1] - t[n])); S = S * np.outer(pdec, pdec)
        new[k] = S
        if k + 1 < K:
            r = np.abs(S - ends[k]).max() / max(np.abs(S).max(), 1e-300)
            if r > worst: worst, wk = r, k
    ends = new
    print("pass %2d: worst rel change %.2e (chunk %d)" % (p, worst, wk))
    if worst == 0: break
