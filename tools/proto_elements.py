# -*- coding: utf-8 -*-
"""numpy prototype of the chunk ELEMENTS of c2_timepar.hip (round 5): python tools/proto_elements.py [J] [rows per chunk].

The factor recursion (forward.hpp:105-134) and solve_lower (internal.hpp:135-145) over a span of rows, run from the ZERO state,
give the span's element (A, G, H = -Q, g, q1 = h, q0, log prod d).  `apply` pushes a state (T, F) through an element, `combine`
merges the elements of two consecutive spans; both are checked here against the sequential recursion on the bench's series:
  * apply on the true chunk-start states: T, F, sum z^2 / d and sum log d of every chunk (1e-14 .. 1e-15),
  * a Kogge-Stone scan over the chunks: every chunk-start state and the totals (sum log d, sum z^2 / d) of the series
    (1e-15 .. 1e-16), and the conditioning of the only matrix that is inverted, I + G1 H2 (~1e2 here: a / d of the problem).
(The kernels factor that matrix through the symmetric positive definite Ks = I - L^T Q2 L, G1 = L L^T; the prototype inverts it
directly.)  CPU only: the oracle builds the matrices."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celerite2_amd import synth
from oracle import cpu
J = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = 4096; R = int(sys.argv[2]) if len(sys.argv) > 2 else 64
t, diag, y, ac, bc, cc, dc = synth.host_inputs(0, 3, N, J)[:7]
def mats(i):
    z = np.zeros(0)
    a = np.empty(N); U = np.empty((N, J)); V = np.empty((N, J))
    cpu.get_celerite_matrices(z, ac[i], bc[i], dc[i], t[i], diag[i], a, U, V)
    return np.repeat(cc[i], 2), a, U, V
I = np.eye(J)
def chunk(c, a, U, V, t, y, s0, s1, T, F, elems=False):
    """rows s0..s1-1 from (T, F); returns end state, sums, and (if elems) the from-this-start transition elements"""
    A = I.copy(); H = np.zeros((J, J)); q1 = np.zeros(J); q0 = 0.0; ld = 0.0
    for n in range(s0, s1):
        u = U[n]; v = V[n]
        tau = u @ T; d = a[n] - tau @ u; w = (v - tau) / d
        z = y[n] - u @ F
        r = u @ A
        ld += np.log(d); q0 += z * z / d
        q1 += z / d * r; H -= np.outer(r, r) / d
        p = np.exp(-c * (t[n + 1] - t[n])) if n + 1 < len(t) else np.ones(J)
        T = (T + d * np.outer(w, w)) * np.outer(p, p)
        F = p * (F + w * z)
        A = p[:, None] * (A - np.outer(w, r))
    return dict(A=A, G=T, H=H, g=F, q1=q1, q0=q0, ld=ld)
def apply(e, T, F):
    K = I + e['H'] @ T          # (I + H T)
    Ki = np.linalg.inv(K)
    Tn = e['G'] + e['A'] @ T @ Ki @ e['A'].T
    KiT = np.linalg.inv(I + T @ e['H'])
    Fn = e['g'] + e['A'] @ KiT @ (F - T @ e['q1'])
    Q2 = -e['H']
    rr = e['q1'] - Q2 @ F
    quad = e['q0'] - 2 * e['q1'] @ F + F @ Q2 @ F + rr @ T @ Ki @ rr   # guess
    ld = e['ld'] + np.log(np.linalg.det(K))
    return Tn, Fn, quad, ld
def combine(e1, e2):
    G1, H2 = e1['G'], e2['H']
    M = np.linalg.inv(I + G1 @ H2)       # (I + G1 H2)^-1
    Mt = M.T                              # (I + H2 G1)^-1
    A = e2['A'] @ M @ e1['A']
    G = e2['G'] + e2['A'] @ G1 @ Mt @ e2['A'].T
    H = e1['H'] + e1['A'].T @ H2 @ M @ e1['A']
    g = e2['g'] + e2['A'] @ M @ (e1['g'] - G1 @ e2['q1'])
    q1 = e1['q1'] + e1['A'].T @ Mt @ (e2['q1'] + H2 @ e1['g'])
    _, _, quad, ld = apply(e2, G1, e1['g'])
    return dict(A=A, G=0.5 * (G + G.T), H=0.5 * (H + H.T), g=g, q1=q1, q0=e1['q0'] + quad, ld=e1['ld'] + ld)
for i in range(3):
    c, a, U, V = mats(i)
    K = N // R
    # sequential truth
    T = np.zeros((J, J)); F = np.zeros(J); truth = []; ld = 0; q = 0
    for k in range(K):
        truth.append((T.copy(), F.copy()))
        e = chunk(c, a, U, V, t[i], y[i], k * R, (k + 1) * R, T, F)
        T, F = e['G'], e['g']; ld += e['ld']; q += e['q0']
    els = [chunk(c, a, U, V, t[i], y[i], k * R, (k + 1) * R, np.zeros((J, J)), np.zeros(J)) for k in range(K)]
    # check apply() on true starts
    worst = np.zeros(4)
    for k in range(1, K - 1):
        Tn, Fn, quad, ldk = apply(els[k], *truth[k])
        ek = chunk(c, a, U, V, t[i], y[i], k * R, (k + 1) * R, *truth[k])
        worst = np.maximum(worst, [np.abs(Tn - truth[k + 1][0]).max() / np.abs(Tn).max(), np.abs(Fn - truth[k + 1][1]).max() / np.abs(Fn).max(),
                                   abs(quad - ek['q0']) / abs(ek['q0']), abs(ldk - ek['ld']) / abs(ek['ld'])])
    print(i, "apply to true starts: T %.1e F %.1e quad %.1e logdet %.1e" % tuple(worst))
    # Kogge-Stone inclusive scan
    pre = list(els); off = 1
    while off < K:
        new = list(pre)
        for k in range(off, K):
            new[k] = combine(pre[k - off], pre[k])
        pre = new; off *= 2
    wT = wF = 0
    for k in range(1, K):
        wT = max(wT, np.abs(pre[k - 1]['G'] - truth[k][0]).max() / np.abs(truth[k][0]).max())
        wF = max(wF, np.abs(pre[k - 1]['g'] - truth[k][1]).max() / np.abs(truth[k][1]).max())
    print(i, "scan starts vs sequential: T %.1e F %.1e; totals: ld %.2e q0 %.2e  (ld %.3f quad %.3f)" % (wT, wF, abs(pre[K - 1]['ld'] - ld) / abs(ld), abs(pre[K - 1]['q0'] - q) / abs(q), ld, q))
    print("   cond(I+G1H2) at last level:", np.linalg.cond(I + pre[K // 2 - 1]['G'] @ els[K // 2]['H']), "|A| total", np.abs(pre[K-1]['A']).max())
