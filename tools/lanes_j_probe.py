"""Group mapping (C2_LANES=8) against one lane per series (C2_LANES=1) for the fused gradient pair at widths 4 and 2,
N = 4096: where the crossover lies now that the group mapping's reverse sweep runs the recursion backward."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
N = 4096
for J in [int(x) for x in os.environ.get("LJ_J", "4,2").split(",")]:
    for B in [int(x) for x in os.environ.get("LJ_B", "16384,24576,32768,49152,65536").split(",")]:
        args = synth.device_batch_fast(0, B, N, J, torch.device("cuda:0"))
        out = {}
        for lanes in ("", "8", "1"):
            if lanes: os.environ["C2_LANES"] = lanes
            else: os.environ.pop("C2_LANES", None)
            work = ops.loglik_grad_workspace(B, N, J, args[2].device)
            ops.loglik_grad(*args, work=work); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(4):
                e0.record(); ops.loglik_grad(*args, work=work); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            out[lanes or "auto"] = round(min(ts), 3)
            del work
        print(json.dumps({"J": J, "B": B, "grad_ms": out}), flush=True)
        del args; torch.cuda.empty_cache()
