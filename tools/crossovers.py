# -*- coding: utf-8 -*-
"""Re-derive the measured crossovers of the dispatch table (celerite2_amd/csrc/c2_dispatch.hpp) on THIS box and print the
c2_set_option() calls that would move them.  The defaults were measured on one MI355X with the synthetic bench series
(mean spacing 0.1, rates up to 0.21); other data or another box may want other thresholds.

    python tools/crossovers.py [--quick]

Nothing is changed: the script forces each alternative through the options, times it (HIP events, median of 5) and reports.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from celerite2_amd import _lib, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
QUICK = "--quick" in sys.argv


def timed(fn, reps=5):
    return synth.timed_steady(fn, reps=reps, warm_ms=30.0)   # (steady clock: profiles/r05_clock_ramp.md)


class forced:
    def __init__(self, **kw): self.kw = kw
    def __enter__(self):
        for k, v in self.kw.items(): _lib.set_option(k, v)
    def __exit__(self, *a):
        for k in self.kw: _lib.set_option(k, None)


def lanes():
    print("== lane mapping of the fused log-likelihood (+ gradient), N = 4096, J = 8")
    best_f = best_g = None
    for B in ([16384, 32768] if QUICK else [8192, 16384, 24576, 32768, 49152]):
        args = synth.device_batch_fast(0, B, 4096, 8, dev)
        row = {}
        for lanes_ in (8, 4, 1):
            with forced(lanes=lanes_):
                row[("fwd", lanes_)] = timed(lambda: ops.loglik(*args))
                if lanes_ != 4:
                    work = ops.loglik_grad_workspace(B, 4096, 8, dev)
                    out = ops.loglik_grad(*args, work=work)[1]
                    row[("grad", lanes_)] = timed(lambda: ops.loglik_grad(*args, work=work, out=out))
                    del work, out
        print("  B %6d  fwd: 8 lanes %.2f  4 lanes %.2f  1 lane %.2f ms | fwd+grad: 8 lanes %.2f  1 lane %.2f ms"
              % (B, row[("fwd", 8)], row[("fwd", 4)], row[("fwd", 1)], row[("grad", 8)], row[("grad", 1)]))
        if best_f is None and row[("fwd", 1)] < min(row[("fwd", 8)], row[("fwd", 4)]): best_f = B
        if best_g is None and row[("grad", 1)] < row[("grad", 8)]: best_g = B
        del args
        torch.cuda.empty_cache()
    print("  -> c2_set_option(\"lanes1_min_batch_fwd\", \"%s\"); c2_set_option(\"lanes1_min_batch_grad\", \"%s\")"
          % (best_f or "beyond the sizes tried", best_g or "beyond the sizes tried"))


def timepar_grad():
    print("== gradient parallel along time vs row by row, one series, J = 8")
    first = None
    for N in ([256, 512, 1024, 2048] if QUICK else [128, 192, 256, 384, 512, 768, 1024, 1536, 2048, 4096]):
        args = synth.device_batch_fast(0, 1, N, 8, dev)
        with forced(timepar_grad=1): tp = timed(lambda: ops.loglik_grad(*args))   # (the factor inside: the dispatch's own choice)
        with forced(timepar_grad=0, factor_iter=0): rr = timed(lambda: ops.loglik_grad(*args))
        print("  N %5d  time-parallel %.3f ms  row by row %.3f ms" % (N, tp, rr))
        if first is None and tp < rr: first = N
    print("  -> one series draws level at ~%s rows (table: 256 for a handful of series, timepar_grad_min_rows = 1024 beyond)" % first)


def solve_cost_model():
    print("== cost model of the chunk-map solves (ms): fixed cost + cost per 64-row chunk, against microseconds per row")
    J = 8
    pts = []
    for B, N in ((64, 4096), (512, 4096)):
        t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
        d, W, _ = ops.factor(t, c, a, U, V)
        Y = y.unsqueeze(-1).contiguous()
        with forced(timepar=1, long_min_rows=128): ms = timed(lambda: ops.solve_lower(t, c, U, W, Y))
        pts.append((B * ((N + 63) // 64), ms))
        if B == 64:
            Y8 = torch.randn((B, N, 8), dtype=torch.float64, device=dev)
            with forced(timepar=0): r1 = timed(lambda: ops.solve_lower(t, c, U, W, Y)); r8 = timed(lambda: ops.solve_lower(t, c, U, W, Y8))
    (k0, m0), (k1, m1) = pts
    per = (m1 - m0) / (k1 - k0)
    fixed = m0 - per * k0
    row_us, rhs_us = 1e3 * (8 * r1 - r8) / (7 * 4096), 1e3 * (r8 - r1) / (7 * 4096)
    print("  chunk maps: %.3f ms at %d chunks, %.3f ms at %d  ->  fixed %.3f ms + %.2e ms per chunk" % (m0, k0, m1, k1, fixed, per))
    print("  row by row at 4096 rows: %.3f ms (1 rhs), %.3f ms (8 rhs)  ->  %.3f us per row + %.4f us per row and rhs" % (r1, r8, row_us, rhs_us))
    print("  -> c2_set_option(\"solve_chunk_col_ms\", \"%.3g\"); (\"solve_chunk_ms\", \"%.3g\"); (\"solve_row_us\", \"%.3g\"); (\"solve_row_rhs_us\", \"%.3g\")"
          % (max(fixed, 0.0), per, row_us, rhs_us))


def terms():
    """Coefficient-level log-likelihood (+ gradient): composed chain against the kernels that form the rows in the lanes, per lane
    mapping; the first batch size at which a mapping leads is the threshold to set."""
    import numpy as np
    print("== coefficient-level log-likelihood (+ gradient), N = 4096: composed / group of J lanes / four / two / one lane per series (ms)")
    modes = {"composed": dict(terms_fused=0, terms_two_lanes=0, terms_eight_lanes=0, terms_four_lanes=0),
             "group": dict(terms_fused=0, terms_two_lanes=0, terms_eight_lanes=1, terms_four_lanes=0),
             "four": dict(terms_fused=0, terms_two_lanes=0, terms_eight_lanes=0, terms_four_lanes=1),
             "two": dict(terms_fused=0, terms_two_lanes=1, terms_eight_lanes=0, terms_four_lanes=0),
             "one": dict(terms_fused=1, terms_two_lanes=0, terms_eight_lanes=0, terms_four_lanes=0)}
    N = 4096
    for J in ((8,) if QUICK else (8, 4, 2)):
        th, dgh, yh, ach, bch, cch, dch = synth.host_inputs(0, 8, N, J)
        lead = {}
        for B in ([4096, 16384] if QUICK else [1024, 2048, 3072, 4096, 6144, 8192, 10240, 12288, 16384, 24576, 32768, 49152]):
            f = lambda x: torch.from_numpy(np.ascontiguousarray(np.tile(x, ((B + 7) // 8,) + (1,) * (x.ndim - 1))[:B])).to(dev)
            e = torch.zeros((B, 0), dtype=torch.float64, device=dev)
            args = (e, e, f(ach), f(bch), f(cch), f(dch), f(th), f(dgh), f(yh))
            row = {}
            for name, kw in modes.items():
                if (name in ("two", "four") and J != 8) or (name == "group" and B * J > 65536) or (name == "four" and B > 16384):
                    continue
                with forced(**kw):
                    row[("fwd", name)] = timed(lambda: ops.loglik_terms(*args), reps=3)
                    row[("grad", name)] = timed(lambda: ops.loglik_terms_grad(*args), reps=3)
            for kind in ("fwd", "grad"):
                best = min((v, k[1]) for k, v in row.items() if k[0] == kind)[1]
                lead.setdefault((kind, best), B)
            print("  J %d  B %6d  fwd: %s | fwd+grad: %s" % (J, B, "  ".join("%s %.2f" % (k[1], v) for k, v in row.items() if k[0] == "fwd"),
                                                             "  ".join("%s %.2f" % (k[1], v) for k, v in row.items() if k[0] == "grad")))
            del args
            torch.cuda.empty_cache()
        print("  -> J = %d: first batch size each mapping leads at: %s" % (J, ", ".join("%s %s from %d" % (k[0], k[1], v) for k, v in sorted(lead.items()))))
    print("  (options: terms_eight_lanes_min_batch_grad / _fwd in series AT J = 8, terms_group_min_batch_fwd_j4, terms_four_lanes_min/max_batch_grad,")
    print("   terms_two_lanes_min/max_batch_*, terms_fused_min_batch_*, terms_group_max_batch_grad_j2)")


if __name__ == "__main__":
    t0 = time.time()
    print("dispatch table of the loaded library: %d options; device %s" % (len(_lib.options()), torch.cuda.get_device_name(0)))
    lanes()
    timepar_grad()
    solve_cost_model()
    terms()
    print("(%.0f s)" % (time.time() - t0))
