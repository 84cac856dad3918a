"""Prototype: Newton iteration on the chunk start states of the factor recursion (multiple shooting).  Unknowns X_k
(state entering chunk k), equations X_{k+1} = f_k(X_k); the Jacobian of f_k is the congruence with Phi_k = prod A_n,
A_n = P (I - w u^T); the linearised recurrence delta_{k+1} = Phi_k delta_k Phi_k^T + (f_k(X_k) - X_{k+1}) is walked
sequentially (cheap)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import dense
J, N, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
t, c, a, U, V, y = [x[0] for x in dense.synthetic_batch(1, N, J)]
K = (N + L - 1) // L
def chunk(k, S):
    M = np.eye(J); S = S.copy(); dmin = np.inf
    for n in range(k * L, min(N, (k + 1) * L)):
        tau = S @ U[n]; d = a[n] - U[n] @ tau; w = (V[n] - tau) / d; dmin = min(dmin, d)
        S = S + d * np.outer(w, w); A = np.eye(J) - np.outer(w, U[n])
        if n + 1 < N:
            p = np.exp(-c * (t[n + 1] - t[n])); S = S * np.outer(p, p); A = p[:, None] * A
        M = A @ M
    return S, M, dmin
# exact
X_true = np.zeros((K, J, J)); S = np.zeros((J, J))
for k in range(K - 1):
    S, _, _ = chunk(k, S); X_true[k + 1] = S
X = np.zeros((K, J, J))
for it in range(1, 16):
    E = np.empty_like(X); Phi = np.empty_like(X); dmin = np.inf
    for k in range(K):
        E[k], Phi[k], dm = chunk(k, X[k]); dmin = min(dmin, dm)
    delta = np.zeros((J, J)); worst = 0.0
    for k in range(K - 1):
        newX = E[k] + Phi[k] @ delta @ Phi[k].T
        delta = newX - X[k + 1]
        worst = max(worst, np.abs(delta).max() / max(np.abs(newX).max(), 1e-300))
        X[k + 1] = newX
    err = max(np.abs(X[k] - X_true[k]).max() / max(np.abs(X_true[k]).max(), 1e-300) for k in range(1, K))
    print("iteration %2d: largest relative update %.2e, error of the start states %.2e, min d %.3e" % (it, worst, err, dmin))
    if worst < 1e-14: break
