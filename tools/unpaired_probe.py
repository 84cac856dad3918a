"""Fused gradient pair with rates that are NOT pairwise equal (a real term next to complex ones): the general-exponential
instances of the lane mappings, N = 4096, J = 8."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
N, J = 4096, 8
for B in [int(x) for x in os.environ.get("UP_B", "8192,32768,65536").split(",")]:
    t, c, a, U, V, y = synth.device_batch_fast(0, B, N, J, torch.device("cuda:0"))
    out = {}
    for name, cc in (("paired", c), ("unpaired", (c * torch.tensor([1.0, 1.01, 1.0, 0.98, 1.0, 1.0, 1.03, 1.0], dtype=c.dtype, device=c.device)).contiguous())):
        for lanes in ("", "1", "2", "8"):
            if lanes == "2" and B > 32768: continue
            if lanes: os.environ["C2_LANES"] = lanes
            else: os.environ.pop("C2_LANES", None)
            work = ops.loglik_grad_workspace(B, N, J, a.device)
            ops.loglik_grad(t, cc, a, U, V, y, work=work); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(3):
                e0.record(); ops.loglik_grad(t, cc, a, U, V, y, work=work); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            out[name + ":" + (lanes or "auto")] = round(min(ts), 2)
            del work
    print(json.dumps({"B": B, "ms": out}), flush=True)
    del t, c, a, U, V, y; torch.cuda.empty_cache()
