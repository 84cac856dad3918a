"""Width 8, `factor` from the scanned chunk elements (C2_FACTOR_SCAN8, c2_timepar.hip run8_states): the device-side check word
(largest mismatch between a chunk's end state and its successor's start state, relative to sqrt(X_ii X_jj)), kappa, and the
distance of d, W from the CPU oracle -- on the bench generator and on the draws of the time-parallel fuzz that have width 8.
    python tools/e8_words.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from celerite2_amd import _lib, ops, synth
from oracle import cpu
lib = _lib.load()
sink = torch.zeros(64, dtype=torch.float64, device="cuda")
lib.c2_internal_set_debug_sink.argtypes = [ctypes.c_void_p]; lib.c2_internal_set_debug_sink.restype = None
lib.c2_internal_set_debug_sink(ctypes.c_void_p(sink.data_ptr()))
TOL = 1e-12
def one(name, t, c, a, U, V):
    sink.zero_()
    dev = [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (t, c, a, U, V)]
    d, W, fl = ops.factor(*dev)
    torch.cuda.synchronize()
    w = sink.cpu().numpy()
    mism, kap = w[8 + 1] * TOL / 2, w[8 + 11]
    B, N = a.shape
    ed = ew = 0.0
    for b in range(B):
        do = np.empty(N); Wo = np.empty((N, U.shape[2]))
        f = cpu.factor_flag(t[b], c[b], a[b], U[b], V[b], do, Wo)
        if f or int(fl[b]): continue
        ed = max(ed, float(np.max(np.abs(d[b].cpu().numpy() - do) / do)))
        ew = max(ew, float(np.max(np.abs(W[b].cpu().numpy() - Wo)) / np.max(np.abs(Wo))))
    print("%-34s B=%-4d N=%-7d word: mismatch %.2e (kappa %.3g)   d rel %.1e   W / max|W| %.1e   flags %d" % (name, B, N, mism, kap, ed, ew, int((fl != 0).sum())))
os.environ["C2_FACTOR_ITER"] = "1"
for B, N in ((1, 4096), (8, 4096), (1, 100000), (64, 1000), (3, 390), (2, 20000)):
    t, c, a, U, V, y = [x.cpu().numpy() for x in synth.device_batch_fast(0, B, N, 8, torch.device("cuda:0"))]
    one("bench generator", t, c, a, U, V)
from test_gpu_fuzz import _tpg_draw
n = 0
for seed in list(range(400)) + [6564, 1896, 2731, 2107]:
    rows, t, c, a, U, V, y, st, sc = _tpg_draw(seed)
    if U.shape[2] != 8 or U.shape[1] < 32: continue
    if rows: os.environ["C2_TPG_ROWS"] = rows
    else: os.environ.pop("C2_TPG_ROWS", None)
    one("fuzz draw %d (rows %s)" % (seed, rows), t, c, a, U, V)
    n += 1
    if n >= 40 and seed < 400: continue
