"""Lane mapping vs batch size (J = 8, N = 4096): forward-only and fused gradient."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
def timeit(fn, n=4):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for Bb in (16384, 24576, 32768, 40960, 49152, 65536):
    args = synth.device_batch_fast(0, Bb, 4096, 8, dev)
    row = ["B %6d" % Bb]
    for lanes in ("8", "4", "1"):
        os.environ["C2_LANES"] = lanes
        row.append("fwd L%s %.2f ms" % (lanes, 1e3 * timeit(lambda: ops.loglik(*args))))
    for lanes in ("8", "1"):
        os.environ["C2_LANES"] = lanes
        work = ops.loglik_grad_workspace(Bb, 4096, 8, dev)
        ll, out, fl = ops.loglik_grad(*args, work=work)
        row.append("grad L%s %.2f ms" % (lanes, 1e3 * timeit(lambda: ops.loglik_grad(*args, work=work, out=out))))
        del work, out
    print("  ".join(row), flush=True)
    del args
