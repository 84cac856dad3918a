"""A/B of one dispatch option on c2_loglik_grad inside one process (alternating, median of 7 each), with the largest
difference between the two results:
    python tools/ab_grad.py <option> [B,B,...] [J,J,...] [N]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import _lib, ops, synth

opt = sys.argv[1]
Bs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1024,4096,8192,16384").split(",")]
Js = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "8").split(",")]
N = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
dev = torch.device("cuda:0")
for J in Js:
    for B in Bs:
        args = synth.device_batch_fast(0, B, N, J, dev)
        res, out = {0: [], 1: []}, {}
        for rep in range(8):
            for v in (1, 0):
                _lib.set_option(opt, v)
                work = ops.loglik_grad_workspace(B, N, J, dev)
                ll, grads, flag = ops.loglik_grad(*args, work=work)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ll, grads, flag = ops.loglik_grad(*args, work=work); e1.record(); torch.cuda.synchronize()
                if rep: res[v].append(e0.elapsed_time(e1))
                else: out[v] = (ll.clone(), [g.clone() for g in grads])
                del work, ll, grads, flag
        _lib.set_option(opt, None)
        med = lambda x: sorted(x)[len(x) // 2]
        gd = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(out[1][1], out[0][1]))
        print(json.dumps({"option": opt, "J": J, "B": B, "N": N, "on_ms": round(med(res[1]), 3), "off_ms": round(med(res[0]), 3),
                          "ll_rel_diff": float(((out[1][0] - out[0][0]).abs() / out[0][0].abs()).max()), "grad_rel_diff": gd}), flush=True)
        del args, out
        torch.cuda.empty_cache()
