"""Lane mapping vs batch size (N = 4096; J = 8 by default, `python tools/lanes_sweep.py 4` for another width): forward-only
and fused gradient."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
J = int(sys.argv[1]) if len(sys.argv) > 1 else 8
BS = (8192, 12288, 16384, 24576, 32768, 49152, 65536) if J != 8 else (16384, 24576, 32768, 40960, 49152, 65536)
def timeit(fn, n=4):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for Bb in BS:
    args = synth.device_batch_fast(0, Bb, 4096, J, dev)
    row = ["B %6d" % Bb]
    for lanes in (("8", "4", "1") if J == 8 else ("8", "1")):
        os.environ["C2_LANES"] = lanes
        row.append("fwd L%s %.2f ms" % (lanes, 1e3 * timeit(lambda: ops.loglik(*args))))
    for lanes in ("8", "1"):
        os.environ["C2_LANES"] = lanes
        work = ops.loglik_grad_workspace(Bb, 4096, J, dev)
        ll, out, fl = ops.loglik_grad(*args, work=work)
        row.append("grad L%s %.2f ms" % (lanes, 1e3 * timeit(lambda: ops.loglik_grad(*args, work=work, out=out))))
        del work, out
    print("  ".join(row), flush=True)
    del args
