# -*- coding: utf-8 -*-
"""CPU checks of the 2-D (multi-band) test infrastructure: the CPU oracle's 1-D recursion on the interleaved series
 against the dense Kronecker fixtures, and the gradient fold of oracle/dense.py against the
dense finite differences.  No GPU, no product code."""
import numpy as np
import pytest

from oracle import dense

CASES = [(16, 3, 2), (40, 4, 4), (64, 2, 6), (7, 1, 2)]


@pytest.mark.parametrize("N,M,J", CASES)
def test_interleaved_oracle_matches_dense_kronecker(oracle, golden_kron, N, M, J):
    key = "N%d_M%d_J%d_" % (N, M, J)
    g = {k[len(key):]: v for k, v in golden_kron.items() if k.startswith(key)}
    for b in range(len(g["y"])):
        t2, c2, a2, U2, V2 = dense.kron_interleaved(g["c"][b], g["a"][b], g["U"][b], g["V"][b], g["t"][b],
                                                    g["alpha"][b], g["diag"][b])
        assert np.all(np.diff(t2) >= 0)
        ll, grads2, flag = oracle.loglik_grad(t2, c2, a2, U2, V2, np.ascontiguousarray(g["y"][b].ravel()))
        assert flag == 0
        assert abs(ll - g["loglik_dense"][b]) <= 1e-10 * abs(g["loglik_dense"][b])
        bt, bc, ba, bU, bV, balpha, bdiag, by = dense.kron_fold_gradients(grads2, g["a"][b], g["U"][b], g["V"][b],
                                                                          g["alpha"][b])
        np.testing.assert_allclose(balpha, g["fd_balpha"][b], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(bdiag, g["fd_bdiag"][b], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(by, g["fd_by"][b], rtol=2e-6, atol=2e-6)


def test_kron_synthetic_is_a_kronecker_model():
    """a = k(0) = U_n . V_n for the generated celerite matrices: the condition under which the interleaved series
    IS the Kronecker model (same-epoch cross-band terms go through U_n . V_n)."""
    for J in (1, 2, 3, 6):
        t, c, a, U, V, alpha, diag, y, cos = dense.kron_synthetic(2, 20, 3, J)
        np.testing.assert_allclose(a, np.einsum("bnj,bnj->bn", U, V), rtol=1e-13)
        K = dense.kron_dense(cos[0], t[0], alpha[0], diag[0])
        assert np.allclose(K, K.T) and np.linalg.eigvalsh(K).min() > 0
