# -*- coding: utf-8 -*-
"""bench.py -- headline benchmark: batched float64 GP log-likelihood + gradient per second at
N=4096, J=8 (BASELINE.json metric).  Workload = BASELINE.json configs[2]: a batch of 65536 independent GPs,
forward + reverse-mode gradient.  It fits one MI355X (41 GB inputs + 41 GB gradients + 33 GB replay records), so
N=1 runs all 65536 series on one GPU.  The path partitions over independent series with no data-path collective.
N GPUs SHARD that one batch (strong scaling: the literal "batch 65536 sharded across 8 x MI355X", 8192 series per
GPU at N = 8 -- `value`); the same line carries a `weak_scaling` object, 65536 series PER GPU measured in the same run
(per-GPU work fixed: what the kernels are priced on) -- never under the configs[2] label.  --batch-per-gpu B makes the
weak-scaling variant the line's `value` instead (labelled as such).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (fused log-lik + reverse-mode gradient w.r.t. t, c, a, U, V, y)
over the rank's shard of independent series, inputs already resident in HBM, followed for N>1 by the one
real exchange of the path: an RCCL all-gather of the per-rank log-likelihood vector.  The W warm-up steps are preceded by
untimed steps until the device has had ~60 ms of work (its clock ramps for ~25 ms after a pause: `config.clock_prewarm_steps`,
--no-clock-prewarm), and nothing but the barrier and the synchronisations separates warm-up and timed steps.  Rank 0 prints ONE
JSON line.  `roofline` prices the step against the HBM roofline with the ALGORITHMIC bytes of SURVEY.md
section 8(d) (each input read once, each output written once: 16(3+2J) B per time step per series);
`cpu_baseline` times the CPU restatement (oracle/, "port" -- the Eigen reference is unbuildable here) on a
bounded sample of the same workload on the host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes_per_gp(N, J, grad):
    """fwd N*8(3+2J)+8J+8 ; fwd+grad N*16(3+2J)+16J+8."""
    return N * 16 * (3 + 2 * J) + 16 * J + 8 if grad else N * 8 * (3 + 2 * J) + 8 * J + 8


def measured_traffic(grad, Bp, N, J):
    """HBM bytes per step from the committed rocprofv3 PMC measurement of this exact workload (profiles/, newest
    round first), or None."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                for w in json.load(f)["workloads"]:
                    if (w["mode"] == ("grad" if grad else "fwd") and w["batch_per_gpu"] == Bp and w["N"] == N
                            and w["J"] == J):
                        return w["traffic_bytes_per_step"], os.path.relpath(path, ROOT)
        except Exception:
            pass
    return None, None


def self_launch(args):
    """`python bench.py --gpus N` without torchrun: re-exec this script under torch.distributed.run, one rank per
    GPU.  Refuses (rc 2) when fewer than N devices are visible -- never a silent one-GPU run."""
    import socket
    import subprocess

    import torch

    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    oversub = os.environ.get("C2_DIST_BACKEND", "nccl") != "nccl"  # test mode: several gloo ranks on one device
    if ndev < 1 or (ndev < args.gpus and not oversub):
        sys.stderr.write("bench.py: --gpus %d requested but %d HIP device(s) visible; refusing to run a smaller job\n"
                         % (args.gpus, ndev))
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_baseline(N, J, grad, seconds):
    """Time the CPU restatement on a bounded sample of the same synthetic workload."""
    import numpy as np

    from oracle import cpu, dense

    cpu.build_native()  # -march=native, compiled on the host that times it; portable build if that fails
    cores = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))   # what this process may run on (cgroup / affinity mask of the GPU box's lease)
    except AttributeError:
        usable = cores
    omp_default = min(cores, cpu.num_threads()) or 1   # omp_get_max_threads() as the environment sets it (OMP_NUM_THREADS / the runtime)
    # build matrices on the host with the numpy recipe (this is the checker side)
    def mats(first, count):
        T, C, A, U, V, Y = dense.synthetic_batch(count, N, J, seed0=721 + first)
        return T, C, A, U, V, Y
    fn = cpu.loglik_grad_batched if grad else cpu.loglik_batched
    T, C, A, U, V, Y = mats(0, 2)
    t0 = time.perf_counter(); fn(T, C, A, U, V, Y, nthreads=1); per = (time.perf_counter() - t0) / 2
    # sample sized for ~`seconds` of all-core work, at least 4 series per thread
    nmax = max(omp_default, usable)
    count = int(max(4 * nmax, min(64 * nmax, seconds * omp_default / max(per, 1e-6))))
    T, C, A, U, V, Y = mats(0, count)
    # every thread count worth trying: the runtime's default and one thread per logical CPU the process may use (on the
    # GPU boxes 128 and 256: the second hardware thread of a core adds little to a cache-bound walk); the faster is `value`
    tried = {}
    for nt in sorted({omp_default, usable}):
        fn(T[:nt], C[:nt], A[:nt], U[:nt], V[:nt], Y[:nt], nthreads=nt)  # warm
        t0 = time.perf_counter(); fn(T, C, A, U, V, Y, nthreads=nt); tried[nt] = count / (time.perf_counter() - t0)
    nthreads = max(tried, key=tried.get)
    n1 = max(2, min(count, int(0.3 * seconds / max(per, 1e-6))))
    t0 = time.perf_counter(); fn(T[:n1], C[:n1], A[:n1], U[:n1], V[:n1], Y[:n1], nthreads=1); dt_1 = time.perf_counter() - t0
    return {
        "value": tried[nthreads], "unit": "GP/s", "cores": nthreads, "kind": "port",
        "sample": "%d series of N=%d J=%d (%s), CPU restatement of celerite2 recursions (Eigen unavailable; a "
                  "baseline, not a target: the 2 MiB/series S workspace thrashes the caches with all cores busy), "
                  "g++ %s, OpenMP over the batch; thread counts tried (GP/s): %s -- the runtime's default is %d, the process may "
                  "run on %d of the host's %d logical CPUs"
                  % (count, N, J, "fwd+grad" if grad else "fwd", cpu.build_flags(),
                     ", ".join("%d: %.0f" % kv for kv in sorted(tried.items())), omp_default, usable, cores),
        "single_thread_value": n1 / dt_1,
    }


PARITY_NAMES = ("bt", "bc", "ba", "bU", "bV", "by")


def take_sample(inputs, ll, grads, count, seed):
    """Copy `count` series drawn at random from a batch the hot path has just processed -- inputs AND the outputs of that
    very step -- to the host (untimed; the checker runs later, in the cpu_baseline leg)."""
    import numpy as np
    import torch

    B = int(ll.shape[0])
    idx = np.sort(np.random.default_rng(seed).choice(B, size=min(count, B), replace=False))
    sel = torch.from_numpy(idx).to(ll.device)
    host = lambda x: x.index_select(0, sel).cpu().numpy()
    return {"index": idx, "inputs": [host(x) for x in inputs], "ll": host(ll), "grads": [host(g) for g in grads]}


def rel_errors(got, want):
    """Per sampled series and array: (1) the largest |x - x_oracle| relative to the largest |x_oracle| of that series'
    array; (2) the criterion every parity test of tests/ applies element by element, |x - x_o| <= 1e-10 |x_o| + 1e-12
    max|x_o|, as one number: max_i |x_i - x_o,i| / (|x_o,i| + 1e-2 max|x_o|), which that test
    holds to 1e-10.  (A bare element-by-element ratio means nothing on the entries that are themselves cancellations:
    the oracle in double against its own long-double evaluation differs by 5e-10 .. 8e-10 there.)  Maximum over the sample."""
    import numpy as np

    g = got.reshape(got.shape[0], -1); w = want.reshape(want.shape[0], -1)
    big = np.abs(w).max(axis=1, keepdims=True)
    big = np.where(big > 0.0, big, 1.0)
    diff = np.abs(g - w)
    return float((diff / big).max()), float((diff / (np.abs(w) + 1e-2 * big)).max())


def parity_all(inputs, ll, grads, chunk=2048, nthreads=0):
    """EVERY series of a batch the hot path has just processed against the CPU oracle (`--parity-sample all`,
    tests/test_gpu_ops.py::test_bench_generator_full_batch_vs_oracle): inputs stream to the host `chunk` series at a time,
    the oracle (all host threads) computes log-likelihood + six gradients, its results go back to the device and are
    compared there.  Per array the worst value over ALL series of
      `criterion`: max_i |x_i - x_o,i| / (1e-10 |x_o,i| + 1e-12 max(1, max|x_o|))  -- the tests' element-wise criterion
                   (tests/test_gpu_ops.py::close, max taken per series and array) as a ratio: <= 1 passes;
      `rel_to_largest`: max|x - x_o| / max|x_o| per series and array."""
    import numpy as np
    import torch

    from oracle import cpu

    B = int(ll.shape[0])
    dev = ll.device
    crit = {nm: 0.0 for nm in ("ll",) + PARITY_NAMES}
    rel = {nm: 0.0 for nm in PARITY_NAMES}
    worst_series = {nm: -1 for nm in PARITY_NAMES}
    failed = 0
    for s in range(0, B, chunk):
        e = min(B, s + chunk)
        host = [x[s:e].cpu().numpy() if x.dim() > 1 and x.shape[0] == B else x.cpu().numpy() for x in inputs]
        llo, go, flago = cpu.loglik_grad_batched(*host, nthreads=nthreads)
        failed += int(np.abs(flago).sum())
        lo = torch.from_numpy(llo).to(dev)
        crit["ll"] = max(crit["ll"], float(((ll[s:e] - lo).abs() / (1e-10 * lo.abs())).max()))
        for nm, g, w in zip(PARITY_NAMES, grads, go):
            w = torch.from_numpy(w).to(dev).reshape(e - s, -1)
            d = (g[s:e].reshape(e - s, -1) - w).abs()
            big = w.abs().amax(dim=1, keepdim=True)
            r = (d / (1e-10 * w.abs() + 1e-12 * big.clamp(min=1.0))).amax(dim=1)
            k = int(r.argmax())
            if float(r[k]) > crit[nm]:
                crit[nm], worst_series[nm] = float(r[k]), s + k
            rel[nm] = max(rel[nm], float((d.amax(dim=1, keepdim=True) / torch.where(big > 0, big, torch.ones_like(big))).max()))
            del w, d
    return {"series": B, "oracle_failed": failed, "criterion": crit, "rel_to_largest": rel, "worst_series": worst_series,
            "worst_criterion": max(crit.values()), "passes": bool(max(crit.values()) <= 1.0 and failed == 0)}


def parity_sample(samples, coeff_sample):
    """The timed batches against the CPU oracle: the series `take_sample` set aside from the step's own batch (and from
    the `gappy_all` batch), log-likelihood and the six gradients; the coefficient-level leg's sample through the oracle
    chain (oracle/dense.py: the reverse of get_celerite_matrices in numpy).  Checker only -- after every timed region."""
    import numpy as np

    from oracle import cpu, dense

    out = {"criterion": "per array, max over the sampled series: the tests' element-wise criterion |x - x_o| <= 1e-10 |x_o| + "
                        "1e-12 max|x_o| as one number, max_i |x_i - x_o,i| / (|x_o,i| + 1e-2 max|x_o|) <= 1e-10 (the figure "
                        "listed per array); `rel_to_largest`: max|x - x_o| / max|x_o|; ll: relative"}
    worst = 0.0
    for name, smp in samples.items():
        llo, go, flago = cpu.loglik_grad_batched(*smp["inputs"])   # (every thread OpenMP has: the setting is global to the library)
        e = {"series": int(len(smp["index"])), "oracle_failed": int(np.abs(flago).sum()),
             "ll": float(np.max(np.abs(smp["ll"] - llo) / np.abs(llo)))}
        if "kappa_max" in smp:
            e["kappa_max"] = smp["kappa_max"]
        el = {}
        for nm, g, w in zip(PARITY_NAMES, smp["grads"], go):
            el[nm], e[nm] = rel_errors(g, w)
        e["rel_to_largest"] = el
        worst = max([worst, e["ll"]] + [e[nm] for nm in PARITY_NAMES])
        out[name] = e
    if coeff_sample is not None:
        x, diag, y, ac, bc, cc, dc = coeff_sample["inputs"]
        z = np.zeros(0)
        names = ("bac", "bbc", "bcc", "bdc", "bx", "bdiag", "by")
        want = [dense.coefficient_chain(cpu, z, z, ac[i], bc[i], cc[i], dc[i], x[i], diag[i], y[i]) for i in range(len(x))]
        e = {"series": int(len(x)), "oracle_failed": int(sum(w[2] != 0 for w in want)),
             "ll": float(np.max(np.abs(coeff_sample["ll"] - np.array([w[0] for w in want])) / np.abs(np.array([w[0] for w in want]))))}
        el = {}
        for k, nm in enumerate(names):   # (the nine gradients of c2_loglik_terms_grad; no real terms here: bar, bcr are empty)
            el[nm], e[nm] = rel_errors(coeff_sample["grads"][k + 2], np.stack([np.atleast_1d(w[1][k + 2]) for w in want]))
        e["rel_to_largest"] = el
        worst = max([worst, e["ll"]] + [e[nm] for nm in names])
        out["coefficient_level"] = e
    out["worst"] = worst
    out["within_1e-10"] = bool(worst <= 1e-10)
    return out


def coefficient_level(first, Bp, N, J, dev, ll_matrix, steps, want_sample=False):
    import torch

    from celerite2_amd import ops, synth

    x, diag, y, ac, bc, cc, dc = synth.device_coeffs_fast(first, Bp, N, J, dev)
    e = torch.zeros((Bp, 0), dtype=torch.float64, device=dev)
    work = ops.loglik_terms_workspace(Bp, N, 0, J // 2, dev, grad=True)
    outs = None
    for _ in range(2):
        ll, outs, flag = ops.loglik_terms_grad(e, e, ac, bc, cc, dc, x, diag, y, work=work, out=outs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        ll, outs, flag = ops.loglik_terms_grad(e, e, ac, bc, cc, dc, x, diag, y, work=work, out=outs)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    sample = take_sample((x, diag, y, ac, bc, cc, dc), ll, outs, 8, 3) if want_sample else None
    # bytes the entry point has to move: x, diag, y in; bx, bdiag, by out (the coefficients and their gradients are O(J))
    nbytes = Bp * N * 6 * 8
    # Roofline of this entry point: fp64 VECTOR arithmetic (48 B per step of memory traffic leave HBM far from the wall).
    # Flops per series-step: the ~1900 of the matrix-level forward + reverse pair (SURVEY.md section 8d, rows A, H, I at
    # J = 8, incl. 2 x 8 exponentials at ~20) + the rows formed in the lane: one sincos per complex term forward and one in
    # the reverse sweep (~40 flops each with its reduction) + the contraction of bU, bV into the coefficient gradients
    # (~12 per complex term).  Counted, not measured: the kernels issue MORE (packed-triangle bookkeeping, accvgpr moves).
    Jc = J // 2
    flops_per_gp = N * (1900.0 * (J / 8.0) ** 2 + Jc * (2 * 40.0 + 12.0))
    # TFLOP/s, MI355X fp64 vector: the vendor's published figure as SURVEY.md section 8d quotes it (256 CUs x 4 SIMDs x 16
    # lanes x 2 flops x 2.4 GHz); /opt/skills/guides/MI355X_MICROARCH.md carries no fp64 number of its own
    PEAK_F64_VECTOR = 78.6
    return {"entry": "c2_loglik_terms_grad", "value": Bp / ms * 1e3, "unit": "GP/s", "ms_per_step": ms, "steps": steps,
            "failed_factorizations": int((flag != 0).sum()),
            "ll_max_rel_diff_vs_matrix_level": float(((ll - ll_matrix).abs() / ll_matrix.abs()).max()),
            "algorithmic_bytes": nbytes, "bound": "valu (f64)",
            "roofline": {"bound": "valu_f64", "flops_per_gp": flops_per_gp, "achieved": Bp / ms * 1e3 * flops_per_gp / 1e12,
                         "peak": PEAK_F64_VECTOR, "unit": "TFLOP/s", "frac": Bp / ms * 1e3 * flops_per_gp / 1e12 / PEAK_F64_VECTOR,
                         "counting": "SURVEY.md 8d flops/step of rows A, H, I (~1900 at J = 8) + one sincos per complex "
                                     "term in each sweep (~40) + the coefficient-gradient contraction (~12 per term)"},
            "note": "informational: same series as `value`, gradient w.r.t. the celerite coefficients instead of U, V rows"}, sample


def gappy(first, Bp, N, J, dev, work, out, steps, clean_ms, every_series=False, samples=None):
    """Informational: the same workload with 5 % of the series carrying one gap of 100 mean spacings (a night, a season) at
    a row of their own.  The one-lane reverse sweep cannot invert a decay across such a gap; the forward pass re-anchors it
    there with an extra wavefront-uniform checkpoint (c2_loglik_t.hip), so the batch stays on the fast kernels -- `guard`
    (first double of the workspace) <= 2 says so; before round 3 ONE such series sent all 65536 to the replay kernels."""
    import torch

    from celerite2_amd import ops, synth

    if every_series:   # `gappy_all`: EVERY series three gaps at rows of its own (192 re-anchoring checkpoints per wavefront)
        t, c, a, U, V, y = synth.device_batch_fast(first, Bp, N, J, dev, gap_fraction=1.0, gap=10.0, gaps_per_series=3)
        what = "the step's batch with THREE gaps of 10 time units at rows of its own in every series (%d of %d)"
    else:
        t, c, a, U, V, y = synth.device_batch_fast(first, Bp, N, J, dev, gap_fraction=0.05, gap=10.0)
        what = "the step's batch with a gap of 10 time units (100 mean spacings) in 5 %% of the series (%d of %d)"
    ngap = int(((t[:, 1:] - t[:, :-1]).max(dim=1).values > 5.0).sum())
    for _ in range(2):
        ll, _, flag = ops.loglik_grad(t, c, a, U, V, y, work=work, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        ll, _, flag = ops.loglik_grad(t, c, a, U, V, y, work=work, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    guard, nfall = float(work[0]), int(work[:2].view(torch.int64)[1])
    if samples is not None:
        samples["gappy_all" if every_series else "gappy"] = take_sample((t, c, a, U, V, y), ll, out, 8, 2)
    return {"workload": what % (ngap, Bp),
            "ms_per_step": ms, "value": Bp / ms * 1e3, "unit": "GP/s", "steps": steps, "ratio_to_gap_free_step": ms / clean_ms,
            "guard": guard, "wavefronts_on_the_replay_kernels": nfall, "wavefronts": (Bp + 63) // 64,
            "path": "one lane per series, re-anchored at the gaps" if nfall == 0 else
                    "one lane per series; %d wavefronts of 64 series ran out of extra checkpoints and were replayed" % nfall,
            "failed_factorizations": int((flag != 0).sum())}


def long_series(J, dev, N=100_000, steps=3):
    """Informational: ONE series of 1e5 rows, log-likelihood + gradient -- the small-batch end of the same entry point,
    which runs parallel along time; row by row for comparison (C2_TIMEPAR_GRAD=0, C2_FACTOR_ITER=0)."""
    import torch

    from celerite2_amd import ops, synth

    args = synth.device_batch_fast(0, 1, N, J, dev)
    out = {}
    for name, env in (("ms", {}), ("row_by_row_ms", {"C2_TIMEPAR_GRAD": "0", "C2_FACTOR_ITER": "0"})):
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            work = ops.loglik_grad_workspace(1, N, J, dev)
            ll, g, flag = ops.loglik_grad(*args, work=work)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                ll, g, flag = ops.loglik_grad(*args, work=work, out=g)
            e1.record()
            torch.cuda.synchronize()
            out[name] = e0.elapsed_time(e1) / steps
            out["ll_" + name] = float(ll[0])
            out["g_" + name] = [x.clone() for x in g]
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    # the reverse-mode chain of the drop-in ops on the same series (device pointers): factor with S, solve_lower with F,
    # their reverses -- parallel along time as well; informational, never fatal
    ops_ms = {}
    try:
        t, c, a, U, V, y = args
        Y = y.unsqueeze(-1).contiguous()

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / steps

        d, W, S, _ = ops.factor(t, c, a, U, V, workspace=True)
        Z, F = ops.solve_lower(t, c, U, W, Y, workspace=True)
        bd, bW, bZ = torch.randn_like(d), torch.randn_like(W), torch.randn_like(Z)
        ops_ms["factor_with_S"] = timed(lambda: ops.factor(t, c, a, U, V, d=d, W=W, S=S))
        ops_ms["solve_lower_with_F"] = timed(lambda: ops.solve_lower(t, c, U, W, Y, Z=Z, F=F))
        ops_ms["solve_lower_rev"] = timed(lambda: ops.solve_lower_rev(t, c, U, W, Y, Z, F, bZ))
        ops_ms["factor_rev"] = timed(lambda: ops.factor_rev(t, c, a, U, V, d, W, S, bd, bW))
    except Exception as e:  # noqa: BLE001 -- informational only
        ops_ms["error"] = repr(e)[:200]
    # the gradients of the two evaluation orders against each other, per gradient array: relative to the array's largest
    # entry (the max-norm the time-parallel form is held to) and by the element-wise criterion of the
    # parity tests as one number (rel_errors: |x - x'| / (|x'| + 1e-2 max|x'|), held to 1e-10).  A bare element-by-element
    # ratio says nothing on entries that are cancellations (bt_n = f_{n+1} - f_n): there the float64 oracle differs from its
    # own long-double evaluation by 5e-10 .. 8e-10 as well.
    gdiff = {}
    for nm, gt, gr in zip(("bt", "bc", "ba", "bU", "bV", "by"), out["g_ms"], out["g_row_by_row_ms"]):
        mx, mixed = rel_errors(gt.cpu().numpy(), gr.cpu().numpy())
        gdiff[nm] = {"max_norm": mx, "tests_criterion": mixed}
    return {"entry": "c2_loglik_grad", "workload": "1 series, N=%d, J=%d, forward + reverse-mode grad" % (N, J),
            "ms": out["ms"], "row_by_row_ms": out["row_by_row_ms"], "drop_in_ops_ms": ops_ms,
            "ll_rel_diff": abs(out["ll_ms"] - out["ll_row_by_row_ms"]) / abs(out["ll_row_by_row_ms"]),
            "grad_max_rel_diff_vs_row_by_row": {"max_norm": max(v["max_norm"] for v in gdiff.values()),
                                                "tests_criterion": max(v["tests_criterion"] for v in gdiff.values()),
                                                "per_array": gdiff},
            "note": "informational: latency-bound regime, gradient parallel along time (c2_timepar_grad.hip)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--global-batch", type=int, default=65536,
                    help="total series, sharded over the GPUs (strong scaling; 65536 = the literal configs[2], the default)")
    ap.add_argument("--batch-per-gpu", type=int, default=0,
                    help="if > 0: series PER GPU (weak scaling) as the line's value -- a variant, not the literal configs[2]")
    ap.add_argument("--no-clock-prewarm", action="store_true",
                    help="do not run extra untimed steps in front of the W warm-up steps (by default the device gets ~60 ms of "
                         "work before the timed region: its clock ramps for ~25 ms after a pause, profiles/r05_clock_ramp.md)")
    ap.add_argument("--no-weak-object", action="store_true",
                    help="N > 1: skip the extra `weak_scaling` measurement (65536 series per GPU) the strong-scaling line carries")
    ap.add_argument("--weak-batch-per-gpu", type=int, default=65536,
                    help="series per GPU of that `weak_scaling` measurement (65536: the one-GPU workload on every GPU; tests "
                         "that put several ranks on one device pass less)")
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--J", type=int, default=8)
    ap.add_argument("--mode", choices=["grad", "fwd"], default="grad")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-sample", action="store_true",
                    help="skip the `parity_sample` object (series of the timed batches against the CPU oracle, after the timed regions)")
    ap.add_argument("--parity-sample", default="32",
                    help="how many series of the timed batch `parity_sample.step_batch` checks against the CPU oracle: a count "
                         "(default 32, drawn at random) or `all` -- every series of the batch, streamed through the oracle "
                         "(65536 series: ~1 min on the box's host cores; adds `parity_sample.step_batch_all`)")
    ap.add_argument("--exact-synth", action="store_true",
                    help="per-series numpy recipe (identical series whatever the sharding) instead of the device generator")
    ap.add_argument("--placement-search", type=int, default=1,
                    help="k > 1: before the warm-up, time ONE step on k placements of the whole job (each allocated while the "
                         "best so far is held) and run on the fastest; every candidate is listed in config.placement_search.  "
                         "Diagnostic: WHERE in HBM the step's arrays lie moves it by 6 - 10 % (profiles/r04_headline_spread.md), "
                         "but three candidates of one process can all be slow, so this is not the default (1 = the "
                         "allocator's own placement, what any caller gets)")
    ap.add_argument("--no-gappy", action="store_true", help="skip the informational gappy-batch measurement (`gappy` object)")
    ap.add_argument("--no-long-series", action="store_true",
                    help="skip the informational single-long-series measurement (`long_series` object)")
    ap.add_argument("--no-coefficient-level", action="store_true",
                    help="skip the extra (informational) measurement of c2_loglik_terms_grad on the same series")
    ap.add_argument("--dump-ll", default="", help="rank 0 saves the gathered log-likelihood vector here (.npy)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch
    import torch.distributed as dist

    from celerite2_amd import _lib, ops, parallel, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d must equal WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # one process per GPU; C2_DIST_BACKEND=gloo (+ several ranks on one device) exists only to exercise this
    # multi-rank path on a single-GPU box -- the real runs use RCCL ("nccl" backend on ROCm)
    backend = os.environ.get("C2_DIST_BACKEND", "nccl")
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit("bench.py: %d ranks but %d HIP device(s) visible" % (world, torch.cuda.device_count()))
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # C2_FORCE_DIST=1: initialise the process group and run the gather even with ONE rank -- lets a single-GPU box
    # execute the real RCCL communicator setup + all-gather on device tensors (tests/test_gpu_bench.py)
    dist_on = world > 1 or os.environ.get("C2_FORCE_DIST", "0") == "1"
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    _lib.load()

    weak = args.batch_per_gpu > 0
    N, J = args.N, args.J
    grad = args.mode == "grad"
    make = synth.device_batch if args.exact_synth else synth.device_batch_fast

    def measure(Btot, search):
        """W warm-up + K timed steps of the hot path on this rank's contiguous shard of a batch of Btot series (generated
        directly on the owning GPU), barrier + synchronize on both sides, MAX over ranks."""
        first, Bp = parallel.shard_range(Btot, rank, world)
        # WHERE in the 288 GB of HBM the dozen arrays of a chip-filling step lie moves its time by 6 - 10 %, deterministically:
        # the same job allocated behind a spacer of 16 or 60 GiB runs 27.9 ms where it runs 31.1 without, process after
        # process on one box, and the other way round on the next (profiles/r04_headline_spread.md; nothing in the kernels
        # or in shifts of MiB changes it, a plain copy does not show it).  build(S) allocates the step's arrays -- inputs,
        # workspace, gradients -- while a spacer of S GiB is held, and frees the spacer.
        base_gb = float(os.environ.get("C2_BENCH_SPACER_GB", "0") or 0)   # (diagnostic: shifts every candidate)

        def build(gib):
            gib = gib + base_gb
            spacer = torch.empty(int(gib * 2**30), dtype=torch.uint8, device=dev) if gib > 0 else None
            ins = make(first, Bp, N, J, dev)
            w_ = o_ = None
            if grad:
                U_ = ins[3]
                w_ = ops.loglik_grad_workspace(Bp, N, J, dev)
                o_ = (torch.empty((Bp, N), dtype=torch.float64, device=dev), torch.empty((Bp, J), dtype=torch.float64, device=dev),
                      torch.empty((Bp, N), dtype=torch.float64, device=dev), torch.empty_like(U_), torch.empty_like(U_),
                      torch.empty((Bp, N), dtype=torch.float64, device=dev))
            del spacer
            torch.cuda.empty_cache()
            return ins, w_, o_

        placement = None
        if grad and search > 1 and Bp * N * J >= 2**28:
            # Setup, not timed (--placement-search k; off by default): ONE step on each of k placements of the whole job; the
            # fastest set of arrays is KEPT (a job re-allocated after a free does not come back where it was: 27.9 -> 29.0,
            # 28.3 -> 31.7 ms in three trials), each further candidate allocated while the best so far is still held --
            # which is what moves it elsewhere in HBM.  Two sets at most at a time (230 of 288 GB at the bench shape).
            # What an application that reuses its buffers over thousands of steps can do once; every candidate is reported.
            cands, best = [], None
            for i in range(search):
                try:
                    ins, w_, o_ = build(0.0)
                except RuntimeError:   # (no room for a second set: keep what we have)
                    torch.cuda.empty_cache()
                    break
                for _ in range(2):
                    ops.loglik_grad(*ins, work=w_, out=o_)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.loglik_grad(*ins, work=w_, out=o_); e1.record()
                torch.cuda.synchronize()
                ms_ = e0.elapsed_time(e1)
                cands.append({"candidate": i, "ms": ms_, "U_at": hex(ins[3].data_ptr())})
                if best is None or ms_ < best[3]:
                    best = (ins, w_, o_, ms_, i)
                del ins, w_, o_
                torch.cuda.empty_cache()
            if best is None:   # (not even the first candidate fitted: say so instead of failing on `best[4]` below)
                raise SystemExit("bench.py --placement-search: out of device memory on the first candidate placement")
            placement = {"candidates": cands, "chosen": best[4],
                         "note": "setup, untimed: one step per candidate placement of the WHOLE job (inputs, workspace, "
                                 "gradients), each candidate allocated while the best so far is held; the fastest set is the "
                                 "one the warm-up and the timed steps run on (profiles/r04_headline_spread.md)"}
            (t, c, a, U, V, y), work, out = best[0], best[1], best[2]
            del best
        else:
            (t, c, a, U, V, y), work, out = build(0.0)

        def step():
            if grad:
                ll, _, flag = ops.loglik_grad(t, c, a, U, V, y, work=work, out=out)
            else:
                ll, flag = ops.loglik(t, c, a, U, V, y)
            if dist_on:  # the path's only exchange: B/n_gpu log-liks per rank
                if backend == "nccl":
                    ll = parallel.gather_loglik(ll, Btot, world, force=True)
                else:   # (gloo moves host tensors only: test mode, several ranks on one device)
                    ll = parallel.gather_loglik(ll.cpu(), Btot, world, force=True).to(dev)
            return ll, flag

        # Nothing but the barrier and the synchronisations sits between the warm-up and the timed steps: the events exist
        # beforehand and the failure count is taken afterwards.  (A host-side pause of a millisecond there -- creating the
        # events, a reduction + .item() -- lets the device's clock fall back, and the next ~25 ms of kernels run up to 35 %
        # slower while it ramps: visible on the 4 - 16 ms steps of the multi-GPU shards, profiles/r05_clock_ramp.md.)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        # The W warm-up steps bring the CLOCK up only if they last ~30 ms; short steps (the 4 - 16 ms of the multi-GPU shards)
        # get untimed steps in front of them until 60 ms of device time have gone by -- counted in config.clock_prewarm_steps.
        # (the count comes from the shape alone -- ~0.43 us per series and 4096 rows with the gradient, a quarter without --,
        # so every rank runs the same number of steps and of all-gathers)
        pre = 0
        if args.warmup > 0 and not args.no_clock_prewarm:
            est_ms = max(0.05, (Btot / world) * (N / 4096.0) * (0.43e-3 if grad else 0.11e-3))
            pre = max(0, min(64, int(60.0 / est_ms + 0.999) - args.warmup))
            for _ in range(pre):
                ll, flag = step()
        for _ in range(args.warmup):
            ll, flag = step()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            ev[i][0].record()
            ll, flag = step()
            ev[i][1].record()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        nfail = int((flag != 0).sum())
        if dist_on:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax[0])
        kernel_ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
        return dict(Btot=Btot, first=first, Bp=Bp, elapsed=elapsed, kernel_ms=kernel_ms, ll=ll, nfail=nfail, prewarm=pre,
                    placement=placement, work=work, out=out, inputs=(t, c, a, U, V, y))

    Btot = args.batch_per_gpu * world if weak else args.global_batch
    m = measure(Btot, args.placement_search)
    first, Bp, elapsed, kernel_ms, ll, nfail, placement = m["first"], m["Bp"], m["elapsed"], m["kernel_ms"], m["ll"], m["nfail"], m["placement"]
    prewarm = m["prewarm"]
    work, out = m["work"], m["out"]
    t, c, a, U, V, y = m["inputs"]
    m = None   # (the names above own the buffers now: the informational legs below free them one by one)
    kernel_ms_avg = sum(kernel_ms) / len(kernel_ms)
    # parity of the TIMED work: 32 series drawn from the batch the timed steps ran on, with the outputs of the last timed
    # step, set aside now and checked against the CPU oracle in the cpu_baseline leg (after every timed region)
    samples, coeff_sample = {}, None
    want_parity = rank == 0 and grad and not args.no_cpu_baseline and not args.no_parity_sample
    # conditioning of the timed batch (c2_condition: kappa = max a_n / d_n per series; untimed, after the timed region):
    # any float64 evaluation order carries ~0.4 eps kappa^2 of the largest gradient entry -- what the 1e-10 claim rests on
    kappa_obj = kap = None
    if rank == 0 and J <= 32:
        try:
            kap, _kf = ops.condition(t, c, a, U, V)
            fin = kap[torch.isfinite(kap)]
            kmax = float(fin.max()) if fin.numel() else float("inf")
            kappa_obj = {"max": kmax, "median": float(fin.median()) if fin.numel() else None,
                         "float64_floor_0.4_eps_kappa2": 0.4 * 2.220446049250313e-16 * kmax * kmax,
                         "tolerance_1e-10_attainable": bool(0.4 * 2.220446049250313e-16 * kmax * kmax <= 1e-10),
                         "series": int(kap.numel())}
        except Exception as e:  # noqa: BLE001 -- informational
            kappa_obj = {"error": repr(e)[:200]}
    parity_all_obj = None
    if want_parity:
        ll_mine = ll[first:first + Bp] if ll.shape[0] != Bp else ll
        nsmp = 32 if args.parity_sample == "all" else int(args.parity_sample)
        samples["step_batch"] = take_sample((t, c, a, U, V, y), ll_mine, out, nsmp, 1)
        if kap is not None:
            samples["step_batch"]["kappa_max"] = float(kap.index_select(0, torch.from_numpy(samples["step_batch"]["index"]).to(kap.device)).max())
        if args.parity_sample == "all":   # (the outputs of the last timed step, before anything reuses their arrays)
            from oracle import cpu as _cpu
            _cpu.build_native()
            parity_all_obj = parity_all((t, c, a, U, V, y), ll_mine, out)
    # who ran: one entry per rank (device, PCI bus id) + the collective library -- self-evidencing multi-GPU lines
    me = {"rank": rank, "device": dev_index, "name": torch.cuda.get_device_name(dev_index),
          "pci_bus_id": "%04x:%02x:%02x.0" % tuple(getattr(torch.cuda.get_device_properties(dev_index), k, 0)
                                                   for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))}
    ranks = [me]
    if dist_on:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    weak_obj = None
    if world > 1 and not weak and not args.no_weak_object and args.N == 4096:
        work = out = t = c = a = U = V = y = None
        torch.cuda.empty_cache()
        mw = measure(args.weak_batch_per_gpu * world, 1)
        kw = mw["kernel_ms"]
        if rank == 0:
            bpg = algorithmic_bytes_per_gp(N, J, grad)
            weak_obj = {"workload": "weak-scaling variant (NOT the literal configs[2]): %d series per GPU, %d in total" % (mw["Bp"], mw["Btot"]),
                        "value": mw["Btot"] * args.steps / mw["elapsed"], "unit": "GP/s", "ms_per_step": 1e3 * mw["elapsed"] / args.steps,
                        "global_batch": mw["Btot"], "batch_per_gpu": mw["Bp"], "scaling": "weak",
                        "roofline_frac_per_gpu": mw["Bp"] * bpg / (sum(kw) / len(kw) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "failed_factorizations": mw["nfail"]}
        del mw
        torch.cuda.empty_cache()

    if rank == 0 and args.dump_ll:
        import numpy as np

        np.save(args.dump_ll, ll.detach().cpu().numpy())
    if rank == 0:
        total_gps = Btot * args.steps
        value = total_gps / elapsed
        bytes_per_gp = algorithmic_bytes_per_gp(N, J, grad)
        achieved = Bp * bytes_per_gp / (kernel_ms_avg * 1e-3) / 1e9  # per GPU, HIP-event time of the hot path
        traffic, traffic_src = measured_traffic(grad, Bp, N, J)
        line = {
            "metric": "float64 GP log-lik+grad/sec at N=%d J=%d, batched" % (N, J) if grad
                      else "float64 GP log-lik/sec at N=%d J=%d, batched" % (N, J),
            "value": value, "unit": "GP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak" if weak else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("%s: batch of %d independent GPs (%d per GPU), N=%d, J=%d (sum of %d SHO terms), %s, "
                                    "inputs resident in HBM") % (
                                        "configs[2]" if (Btot == 65536 and N == 4096 and J == 8 and grad) else
                                        ("weak-scaling variant of configs[2] (NOT the literal config)" if weak and world > 1 else "variant of configs[2]"),
                                        Btot, Bp, N, J, J // 2, "forward + reverse-mode grad" if grad else "forward"),
                       "global_batch": Btot, "batch_per_gpu": Bp, "N": N, "J": J,
                       "parallelism": "batch-sharded x%d, all-gather of log-liks" % world,
                       "failed_factorizations": nfail, "clock_prewarm_steps": prewarm, "kappa": kappa_obj, "ranks": ranks},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": ("%s (rocprofv3 PMC, 2 x FETCH_SIZE + WRITE_SIZE in separate passes of this "
                                            "workload -- read from the committed file, NOT measured in this run)" % traffic_src)
                                           if traffic is not None else None,
                         "traffic_ratio": traffic / (Bp * bytes_per_gp) if traffic is not None else None,
                         "algorithmic_bytes_per_gp": bytes_per_gp, "kernel_ms_avg": kernel_ms_avg,
                         "kernel_ms_median": kernel_ms[len(kernel_ms) // 2]},
        }
        if dist_on:
            v = torch.cuda.nccl.version() if backend == "nccl" else None
            line["rccl"] = {"backend": backend, "nranks": world,
                            "version": ".".join(str(x) for x in v) if isinstance(v, tuple) else (str(v) if v is not None else None),
                            "distinct_devices": len({r["pci_bus_id"] for r in ranks})}
        if weak_obj is not None:
            line["weak_scaling"] = weak_obj
        if placement is not None:
            line["config"]["placement_search"] = placement
        if world == 1 and grad and not args.exact_synth and not args.no_gappy:
            del t, c, a, U, V, y   # (regenerated with gaps; the workspace and the gradient arrays are reused)
            torch.cuda.empty_cache()
            ll_keep = ll[:Bp].clone()
            try:
                line["gappy"] = gappy(first, Bp, N, J, dev, work, out, min(args.steps, 5), kernel_ms_avg, samples=samples if want_parity else None)
            except Exception as e:  # noqa: BLE001 -- informational only
                line["gappy"] = {"error": repr(e)[:200]}
            try:   # every series with three gaps of its own: what a wavefront's re-anchoring checkpoints cost at worst
                line["gappy_all"] = gappy(first, Bp, N, J, dev, work, out, min(args.steps, 5), kernel_ms_avg, every_series=True,
                                          samples=samples if want_parity else None)
            except Exception as e:  # noqa: BLE001 -- informational only
                line["gappy_all"] = {"error": repr(e)[:200]}
            t = c = a = U = V = y = None
            ll = ll_keep
        if world == 1 and grad and J == 8 and not args.exact_synth and not args.no_coefficient_level:
            # Informational, not `value`: the same series through the coefficient-level entry point,
            # log-likelihood + gradient w.r.t. (ac, bc, cc, dc, x, diag, y) with U, V formed inside the kernels.
            ll_matrix = ll[:Bp].clone()
            del t, c, a, U, V, y, out, work
            torch.cuda.empty_cache()
            line["coefficient_level"], coeff_sample = coefficient_level(first, Bp, N, J, dev, ll_matrix, min(args.steps, 5), want_parity)
        if world == 1 and grad and J in (2, 4, 6, 8) and not args.no_long_series:
            line["long_series"] = long_series(J, dev)
        if want_parity:
            try:
                from oracle import cpu as _cpu
                _cpu.build_native()   # (the timing build of the cpu_baseline leg must be chosen before the oracle is first loaded)
                line["parity_sample"] = parity_sample(samples, coeff_sample)
                if parity_all_obj is not None:
                    line["parity_sample"]["step_batch_all"] = parity_all_obj
            except Exception as e:  # noqa: BLE001 -- the checker must not take the line down; its absence is visible
                line["parity_sample"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N, J, grad, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
