"""Two lanes per series (C2_LANES=2) against the dispatch's own choice: time of the fused gradient pair and of the
forward-only kernel at N = 4096, J = 8, and the largest element-relative difference between the two."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth, _lib

N, J = 4096, 8
out = []
for B in [int(x) for x in os.environ.get("K2_B", "16384,24576,32768,49152,65536").split(",")]:
    args = synth.device_batch_fast(0, B, N, J, torch.device("cuda:0"))
    res = {}
    for lanes in ["", "1", "2", "8"]:
        if lanes: os.environ["C2_LANES"] = lanes
        else: os.environ.pop("C2_LANES", None)
        work = ops.loglik_grad_workspace(B, N, J, args[2].device)
        ll, grads, flag = ops.loglik_grad(*args, work=work)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record(); ops.loglik_grad(*args, work=work); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        tf = []
        for _ in range(5):
            e0.record(); ops.loglik(*args); e1.record(); torch.cuda.synchronize()
            tf.append(e0.elapsed_time(e1))
        res[lanes or "auto"] = (min(ts), min(tf), ll.clone(), [g.clone() for g in grads], float(work[0]))
        del work, ll, grads, flag
    ref = res["1"]
    k2 = res["2"]
    diff = float(((k2[2] - ref[2]).abs() / ref[2].abs()).max())
    gd = max(float(((a - b).abs().max() / b.abs().max())) for a, b in zip(k2[3], ref[3]))
    line = {"B": B, "grad_ms": {k: round(v[0], 3) for k, v in res.items()}, "fwd_ms": {k: round(v[1], 3) for k, v in res.items()},
            "ll_rel_diff_k2_vs_one_lane": diff, "grad_rel_diff": gd, "guard_k2": k2[4]}
    print(json.dumps(line), flush=True)
    del res, ref, k2, args
    torch.cuda.empty_cache()
