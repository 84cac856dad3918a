import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from celerite2_amd import _lib, ops, synth
dev = torch.device("cuda:0"); N = 4096
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
for J in (8, 16, 4):
    for B in (16, 64, 128, 256):
        t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, dev)
        d, W, flag = ops.factor(t, c, a, U, V)
        for nrhs in (64, 128, 256, 512, 1024):
            Y = torch.randn((B, N, nrhs), dtype=torch.float64, device=dev)
            r = {}
            for v in (0, 1):
                _lib.set_option("solve_cols", v)
                r[v] = timed(lambda: ops.solve_lower(t, c, U, W, Y, Z=Y))
            _lib.set_option("solve_cols", None)
            auto = timed(lambda: ops.solve_lower(t, c, U, W, Y, Z=Y))
            print("J=%d B=%d nrhs=%d rows %.3f cols %.3f auto %.3f %s" % (J, B, nrhs, r[0], r[1], auto, "MISS" if auto > 1.08 * min(r.values()) else ""), flush=True)
            del Y
