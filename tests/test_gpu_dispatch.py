# -*- coding: utf-8 -*-
"""Self-check of the dispatch table ON THE BOX THE SUITE RUNS ON (celerite2_amd/csrc/c2_dispatch.hpp; tools/crossovers.py
is the long form): at the BASELINE shapes every automatic choice must be within 10 % of the best forced alternative, so a
threshold that is mis-tuned for this box fails a test instead of silently costing 1.5x.  Times are HIP events, median of 3,
the alternatives forced through c2_set_option (the same kernels the parity tests cover on both sides of every switch)."""
import pytest

pytestmark = pytest.mark.gpu
SLACK = 1.10


@pytest.fixture(scope="module")
def env():
    import torch
    from celerite2_amd import _lib, ops, synth
    assert torch.cuda.is_available()
    return torch, _lib, ops, synth


def _timed(torch, fn, reps=3):
    """median of `reps` at the steady clock (the device ramps for ~25 ms after a pause: profiles/r05_clock_ramp.md)"""
    from celerite2_amd import synth
    return synth.timed_steady(fn, reps=reps, warm_ms=30.0)


class _forced:
    def __init__(self, lib, **kw): self.lib, self.kw = lib, kw
    def __enter__(self):
        for k, v in self.kw.items(): self.lib.set_option(k, v)
    def __exit__(self, *a):
        for k in self.kw: self.lib.set_option(k, None)


@pytest.mark.parametrize("B", [65536, 32768, 24576, 16384, 8192])
def test_lane_mapping_choice_at_the_shards_of_configs2(env, B):
    """configs[2] (65536 x 4096 x 8, forward + gradient) and its 2 / 4 / 8-GPU shards: the lane mapping c2_loglik_grad and
    c2_loglik pick by themselves against every mapping forced."""
    torch, lib, ops, synth = env
    dev = torch.device("cuda:0")
    N, J = 4096, 8
    args = synth.device_batch_fast(0, B, N, J, dev)
    work = ops.loglik_grad_workspace(B, N, J, dev)
    out = ops.loglik_grad(*args, work=work)[1]
    auto_g = _timed(torch, lambda: ops.loglik_grad(*args, work=work, out=out))
    auto_f = _timed(torch, lambda: ops.loglik(*args))
    del work
    grad, fwd = {}, {}
    for lanes in (8, 4, 2, 1):
        if lanes == 2 and B > 32768:   # (two rounds of wavefronts: 27 ms at 34816 series, nothing to learn at 65536)
            continue
        with _forced(lib, lanes=lanes):
            fwd[lanes] = _timed(torch, lambda: ops.loglik(*args))
            if lanes != 4:   # (the two-columns-per-lane gradient pair is an A/B kernel, never the automatic choice)
                w = ops.loglik_grad_workspace(B, N, J, dev)
                grad[lanes] = _timed(torch, lambda: ops.loglik_grad(*args, work=w, out=out))
                del w
    # second round, every candidate again in the same order (re-timing only the automatic choice biased the
    # comparison towards passing): the minimum of two on BOTH sides
    w = ops.loglik_grad_workspace(B, N, J, dev)
    auto_g = min(auto_g, _timed(torch, lambda: ops.loglik_grad(*args, work=w, out=out)))
    auto_f = min(auto_f, _timed(torch, lambda: ops.loglik(*args)))
    del w
    for lanes in list(fwd):
        with _forced(lib, lanes=lanes):
            fwd[lanes] = min(fwd[lanes], _timed(torch, lambda: ops.loglik(*args)))
            if lanes in grad:
                w = ops.loglik_grad_workspace(B, N, J, dev)
                grad[lanes] = min(grad[lanes], _timed(torch, lambda: ops.loglik_grad(*args, work=w, out=out)))
                del w
    print("B = %d: fwd+grad auto %.2f ms, forced %s | fwd auto %.2f ms, forced %s" % (B, auto_g, grad, auto_f, fwd))
    assert auto_g <= SLACK * min(grad.values()), (B, auto_g, grad)
    assert auto_f <= SLACK * min(fwd.values()), (B, auto_f, fwd)


def test_time_parallel_choice_at_configs1(env):
    """configs[1] (1024 x 4096, J = 4, forward): parallel along time or row by row."""
    torch, lib, ops, synth = env
    dev = torch.device("cuda:0")
    args = synth.device_batch_fast(0, 1024, 4096, 4, dev)
    auto = _timed(torch, lambda: ops.loglik(*args))
    alt = {}
    for v in (0, 1):
        with _forced(lib, timepar=v):
            alt[v] = _timed(torch, lambda: ops.loglik(*args))
    # (a light load keeps speeding up for longer than a warm-up: EVERY candidate is timed a second time, minimum of two)
    auto = min(auto, _timed(torch, lambda: ops.loglik(*args)))
    for v in (0, 1):
        with _forced(lib, timepar=v):
            alt[v] = min(alt[v], _timed(torch, lambda: ops.loglik(*args)))
    print("configs[1]: auto %.3f ms, forced %s" % (auto, alt))
    assert auto <= SLACK * min(alt.values()), (auto, alt)


@pytest.mark.parametrize("B,nrhs", [(1, 256), (64, 64), (64, 1024), (2048, 64)])
def test_many_rhs_solve_choice(env, B, nrhs):
    """solve_lower with many right-hand sides: chunk maps over the columns or row by row (solve_cols_shape)."""
    torch, lib, ops, synth = env
    dev = torch.device("cuda:0")
    t, c, a, U, V, y = synth.device_batch_fast(0, B, 4096, 8, dev)
    d, W, _ = ops.factor(t, c, a, U, V)
    Y = torch.randn((B, 4096, nrhs), dtype=torch.float64, device=dev)
    Z = torch.empty_like(Y)
    auto = _timed(torch, lambda: ops.solve_lower(t, c, U, W, Y, Z=Z))
    alt = {}
    for v in (0, 1):
        with _forced(lib, solve_cols=v):
            alt[v] = _timed(torch, lambda: ops.solve_lower(t, c, U, W, Y, Z=Z))
    auto = min(auto, _timed(torch, lambda: ops.solve_lower(t, c, U, W, Y, Z=Z)))   # (second round: every candidate, min of two)
    for v in (0, 1):
        with _forced(lib, solve_cols=v):
            alt[v] = min(alt[v], _timed(torch, lambda: ops.solve_lower(t, c, U, W, Y, Z=Z)))
    print("B = %d, nrhs = %d: auto %.3f ms, forced %s" % (B, nrhs, auto, alt))
    assert auto <= SLACK * min(alt.values()), (auto, alt)
