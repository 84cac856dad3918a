# -*- coding: utf-8 -*-
"""bench.py's `parity_sample` leg on the host: the checker that ties the TIMED batches to the CPU oracle (no GPU: the
sampled "device outputs" here are the oracle's own numbers, once exact and once perturbed)."""
import numpy as np

import bench
from oracle import dense


def _sample(oracle, B, N, J, scale=0.0):
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    ll, grads, flag = oracle.loglik_grad_batched(t, c, a, U, V, y, nthreads=2)
    assert int(np.abs(flag).sum()) == 0
    grads = [g * (1.0 + scale) for g in grads]
    return {"index": np.arange(B), "inputs": [t, c, a, U, V, y], "ll": ll, "grads": grads}


def _coeff_sample(oracle, B, N, Jc):
    rng = np.random.default_rng(3)
    x = np.sort(rng.uniform(0, N / 10.0, (B, N)), axis=1)
    diag = rng.uniform(0.1, 0.3, (B, N)); y = np.sin(x) + 0.1 * rng.standard_normal((B, N))
    co = [dense.sho_sum_coeffs(2 * Jc, xi) for xi in rng.uniform(-1, 1, B)]
    ac, bc, cc, dc = (np.stack([getattr(k, n) for k in co]) for n in ("ac", "bc", "cc", "dc"))
    z = np.zeros(0)
    want = [dense.coefficient_chain(oracle, z, z, ac[i], bc[i], cc[i], dc[i], x[i], diag[i], y[i]) for i in range(B)]
    grads = [np.stack([np.atleast_1d(w[1][k]) for w in want]) for k in range(9)]
    return {"index": np.arange(B), "inputs": [x, diag, y, ac, bc, cc, dc], "ll": np.array([w[0] for w in want]), "grads": grads}


def test_parity_sample_exact_and_perturbed(oracle):
    ok = bench.parity_sample({"step_batch": _sample(oracle, 3, 64, 8)}, _coeff_sample(oracle, 2, 48, 4))
    assert ok["within_1e-10"] and ok["worst"] == 0.0
    assert ok["step_batch"]["series"] == 3 and ok["coefficient_level"]["series"] == 2
    assert set(bench.PARITY_NAMES) <= set(ok["step_batch"])
    bad = bench.parity_sample({"step_batch": _sample(oracle, 3, 64, 8, scale=1e-8)}, None)
    assert not bad["within_1e-10"] and 0.5e-8 < bad["worst"] < 2e-8


def test_rel_errors_is_the_tests_criterion():
    want = np.array([[1.0, 1e-9, -2.0]]); got = want + np.array([[1e-12, 1e-12, 0.0]])
    to_largest, mixed = bench.rel_errors(got, want)
    assert abs(to_largest - 0.5e-12) < 1e-15
    assert abs(mixed - 1e-12 / (1e-9 + 2e-2)) < 1e-15   # a cancelled entry is measured against the floor term, not against itself
