"""the headline step (65536 x 4096 x 8, forward + gradient) as P parts on P streams, one-lane kernels
forced (a part of 32768 series leaves half the SIMDs free, so the forward pass of one part can run beside the reverse sweep of
another: read-heavy beside write-heavy).  Prints ms per whole step, median of 7, for P = 1 (the plain call), 2, 4.
    python tools/overlap_probe.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import _lib, ops, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N, J = 4096, 8
dev = torch.device("cuda:0")
args = synth.device_batch_fast(0, B, N, J, dev)
_lib.set_option("lanes", 1)
res = {}
for P in (1, 2, 4, 1, 2, 4):
    parts = [tuple(a[i * (B // P):(i + 1) * (B // P)] for a in args) for i in range(P)]
    works = [ops.loglik_grad_workspace(B // P, N, J, dev) for _ in range(P)]
    outs = [None] * P
    streams = [torch.cuda.Stream() for _ in range(P)]
    def step():
        for i in range(P):
            with torch.cuda.stream(streams[i]):
                _, outs[i], _ = ops.loglik_grad(*parts[i], work=works[i], out=outs[i])
    for _ in range(2): step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for s in streams: s.wait_event(e0)
        step()
        for s in streams: torch.cuda.current_stream().wait_stream(s)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    res.setdefault(P, []).append(sorted(ts)[3])
    del works, outs, parts
    torch.cuda.empty_cache()
print({"B": B, "ms_per_step_by_parts": {p: [round(x, 2) for x in v] for p, v in res.items()}})
