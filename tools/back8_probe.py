"""The group mappings' reverse sweep by the backward recursion (C2_LOGLIK_BACK, default 1) against the replay (=0): time of
c2_loglik_grad at N = 4096 for batches below the two-lane window, and the largest difference between the two."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth

N = int(os.environ.get("B8_N", "4096"))
for J in [int(x) for x in os.environ.get("B8_J", "8").split(",")]:
    for B in [int(x) for x in os.environ.get("B8_B", "1024,4096,8192,12288").split(",")]:
        args = synth.device_batch_fast(0, B, N, J, torch.device("cuda:0"))
        res = {}
        for back in ["0", "1"]:
            os.environ["C2_LOGLIK_BACK"] = back
            work = ops.loglik_grad_workspace(B, N, J, args[2].device)
            ll, grads, flag = ops.loglik_grad(*args, work=work)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(5):
                e0.record(); ops.loglik_grad(*args, work=work); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            res[back] = (min(ts), ll.clone(), [g.clone() for g in grads])
            del work, ll, grads, flag
        os.environ.pop("C2_LOGLIK_BACK")
        gd = max(float(((a - b).abs().max() / b.abs().max())) for a, b in zip(res["1"][2], res["0"][2]))
        print(json.dumps({"J": J, "B": B, "N": N, "replay_ms": round(res["0"][0], 3), "backward_ms": round(res["1"][0], 3),
                          "ll_rel_diff": float(((res["1"][1] - res["0"][1]).abs() / res["0"][1].abs()).max()), "grad_rel_diff": gd}), flush=True)
        del res, args
        torch.cuda.empty_cache()
