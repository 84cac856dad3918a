"""A/B of one dispatch option on c2_loglik_grad inside one process (alternating, median of 7 each), with the largest
difference between the results:
    python tools/ab_grad.py <option> [B,B,...] [J,J,...] [N] [values: e.g. 1,0 or 0,2,3,1]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import _lib, ops, synth

opt = sys.argv[1]
Bs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1024,4096,8192,16384").split(",")]
Js = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "8").split(",")]
N = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
vals = [int(x) for x in (sys.argv[5] if len(sys.argv) > 5 else "1,0").split(",")]
dev = torch.device("cuda:0")
for J in Js:
    for B in Bs:
        args = synth.device_batch_fast(0, B, N, J, dev)
        res, out = {v: [] for v in vals}, {}
        for rep in range(8):
            for v in vals:
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
        ref = out[vals[-1]]
        gd = max(max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(out[v][1], ref[1])) for v in vals)
        print(json.dumps({"option": opt, "J": J, "B": B, "N": N, "ms": {str(v): round(med(res[v]), 3) for v in vals},
                          "ll_rel_diff": max(float(((out[v][0] - ref[0]).abs() / ref[0].abs()).max()) for v in vals),
                          "grad_rel_diff": gd}), flush=True)
        del args, out
        torch.cuda.empty_cache()
