"""Width 8 on small batches: factor / loglik / loglik_grad row by row, with the Newton iterations on the chunk start states, and
with the scanned chunk elements (round 6: C2_FACTOR_SCAN8, C2_E8_GROUP_CHUNKS) -- ms, HIP events at the steady clock.
    python tools/scan8_grid.py [factor|loglik|grad ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import _lib, ops, synth
dev = torch.device("cuda:0")
which = [a for a in sys.argv[1:]] or ["factor", "loglik", "grad"]
def timed(fn): return synth.timed_steady(fn, reps=5, warm_ms=20.0)
class forced:
    def __init__(self, **kw): self.kw = kw
    def __enter__(self):
        for k, v in self.kw.items(): _lib.set_option(k, v)
    def __exit__(self, *a):
        for k in self.kw: _lib.set_option(k, None)
GRID = [(1, n) for n in (256, 384, 512, 768, 1024, 2048, 4096, 20000, 100000)] + [(8, n) for n in (512, 1024, 4096)] + \
       [(64, n) for n in (256, 512, 1024, 2048, 4096)] + [(256, n) for n in (512, 1024, 2048, 4096)] + \
       [(512, n) for n in (1024, 2048, 4096)] + [(1024, n) for n in (1024, 2048, 4096)] + [(2048, 4096)]
for op in which:
    print("== %s, J = 8: rows / Newton / scan / default (ms)" % op)
    for B, N in GRID:
        t, c, a, U, V, y = synth.device_batch_fast(0, B, N, 8, dev)
        if op == "factor":
            d = torch.empty_like(a); W = torch.empty_like(V)
            fn = lambda: ops.factor(t, c, a, U, V, d=d, W=W)
            modes = {"rows": dict(factor_iter=0), "newton": dict(factor_iter=1, factor_scan8=0), "scan": dict(factor_iter=1), "default": {}}
        elif op == "loglik":
            fn = lambda: ops.loglik(t, c, a, U, V, y)
            modes = {"rows": dict(timepar=0), "newton": None, "scan": dict(timepar=1), "default": {}}
        else:
            work = ops.loglik_grad_workspace(B, N, 8, dev)
            fn = lambda: ops.loglik_grad(t, c, a, U, V, y)
            modes = {"rows": dict(timepar_grad=0), "newton": dict(timepar_grad=1, factor_scan8=0), "scan": dict(timepar_grad=1), "default": {}}
        r = {}
        for k, kw in modes.items():
            if kw is None: r[k] = float("nan"); continue
            with forced(**kw): r[k] = timed(fn)
        best = min(v for k, v in r.items() if k != "default" and v == v)
        print("  B %5d N %7d: %8.3f %8.3f %8.3f | default %8.3f %s" % (B, N, r["rows"], r["newton"], r["scan"], r["default"],
              "" if r["default"] <= 1.1 * best else "  <-- default %.0f %% off the best" % (100 * (r["default"] / best - 1))), flush=True)
