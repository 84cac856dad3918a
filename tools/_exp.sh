mkdir -p gpurun_out/r5s2o
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dispatch.py -q -x 2>&1 | tail -3 > gpurun_out/r5s2o/test.txt
python - > gpurun_out/r5s2o/fwd16384.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
def run(lib):
    import subprocess
    code = '''
import torch, sys, os
sys.path.insert(0, os.getcwd())
from celerite2_amd import ops, synth
dev = torch.device("cuda:0")
for B in (16384,):
    t, c, a, U, V, y = synth.device_batch_fast(0, B, 4096, 8, dev)
    def timed(fn, reps=9):
        fn(); fn(); torch.cuda.synchronize(); ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2]
    print(B, "loglik fwd ms", timed(lambda: ops.loglik(t, c, a, U, V, y)))
'''
    env = dict(os.environ)
    if lib: env["C2_LIB_PATH"] = os.path.join(os.getcwd(), lib)
    print(lib or "new", subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip())
for _ in range(2):
    run("celerite2_amd/libcelerite2_amd_base.so"); run(None)
PY
cat gpurun_out/r5s2o/test.txt gpurun_out/r5s2o/fwd16384.txt
