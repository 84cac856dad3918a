# -*- coding: utf-8 -*-
"""bench.py as the driver calls it (`python bench.py --gpus N ...`, no torchrun): it must launch N ranks by
itself, refuse when it cannot, and produce the same gathered log-likelihoods whatever the sharding.

On the single-GPU test box two ranks share the one device over gloo (C2_DIST_BACKEND=gloo: RCCL refuses two
ranks on one device); the RCCL communicator itself is exercised with ONE rank (C2_FORCE_DIST=1: process-group
init with device_id + all_gather_into_tensor on device tensors), and with two when two devices are visible."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--global-batch", "64", "--N", "256", "--steps", "2", "--warmup", "1", "--exact-synth", "--no-cpu-baseline"]


def run_bench(args, env_extra, timeout=600):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def test_bench_self_launches_two_ranks(tmp_path):
    one, two = str(tmp_path / "ll1.npy"), str(tmp_path / "ll2.npy")
    rc, line1, err = run_bench(["--gpus", "1", "--dump-ll", one] + SMALL, {})
    assert rc == 0 and line1["n_gpus"] == 1, err
    rc, line2, err = run_bench(["--gpus", "2", "--dump-ll", two] + SMALL, {"C2_DIST_BACKEND": "gloo"})
    assert rc == 0, err
    assert line2["n_gpus"] == 2 and line2["config"]["batch_per_gpu"] == 32 and line2["config"]["global_batch"] == 64
    assert line2["scaling"] == "strong" and "roofline" in line2
    # self-evidencing: who ran (one entry per rank) and through which collective library
    assert [r["rank"] for r in line2["config"]["ranks"]] == [0, 1] and all("pci_bus_id" in r for r in line2["config"]["ranks"])
    assert line2["rccl"]["nranks"] == 2 and line2["rccl"]["backend"] == "gloo"
    a, b = np.load(one), np.load(two)
    assert a.shape == (64,) and np.array_equal(a, b)      # same series -> same bits, whichever rank computed them


def test_bench_weak_scaling_variant(tmp_path):
    """--batch-per-gpu: every rank owns that many series (weak scaling, labelled as a variant -- the default shards ONE
    batch, the literal configs[2]): two ranks process the 64 series the one-rank run of 64 does, in the same order."""
    weak = ["--batch-per-gpu", "32"] + SMALL[2:]
    two, ref = str(tmp_path / "ll2.npy"), str(tmp_path / "ll1.npy")
    rc, line, err = run_bench(["--gpus", "2", "--dump-ll", two] + weak, {"C2_DIST_BACKEND": "gloo"})
    assert rc == 0, err
    assert line["scaling"] == "weak" and line["n_gpus"] == 2
    assert line["config"]["batch_per_gpu"] == 32 and line["config"]["global_batch"] == 64
    assert not line["config"]["workload"].startswith("configs[2]:")
    rc, _, err = run_bench(["--gpus", "1", "--dump-ll", ref] + SMALL, {})
    assert rc == 0, err
    assert np.array_equal(np.load(two), np.load(ref))


def test_bench_refuses_more_ranks_than_devices():
    import torch
    n = torch.cuda.device_count() + 1
    rc, line, err = run_bench(["--gpus", str(n)] + SMALL, {})
    assert rc != 0 and line is None and "refusing" in err


def test_bench_rccl_communicator_one_rank(tmp_path):
    """The real `nccl` (= RCCL) backend: communicator bound to the device, all-gather of the device-resident
    log-likelihood vector, barrier, max-reduce of the step time -- on the one GPU this box has."""
    out = str(tmp_path / "ll.npy")
    rc, line, err = run_bench(["--gpus", "1", "--dump-ll", out] + SMALL, {"C2_FORCE_DIST": "1"})
    assert rc == 0 and line["n_gpus"] == 1, err
    assert line["rccl"]["backend"] == "nccl" and line["rccl"]["nranks"] == 1 and line["rccl"]["version"]
    ref = str(tmp_path / "ref.npy")
    rc, _, err = run_bench(["--gpus", "1", "--dump-ll", ref] + SMALL, {})
    assert rc == 0, err
    assert np.array_equal(np.load(out), np.load(ref))


def test_bench_rccl_two_ranks_if_two_devices(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two MI355X")
    one, two = str(tmp_path / "ll1.npy"), str(tmp_path / "ll2.npy")
    rc, _, err = run_bench(["--gpus", "1", "--dump-ll", one] + SMALL, {})
    assert rc == 0, err
    rc, line, err = run_bench(["--gpus", "2", "--dump-ll", two] + SMALL, {})
    assert rc == 0 and line["n_gpus"] == 2, err
    assert np.array_equal(np.load(one), np.load(two))
    # two RCCL ranks on two distinct devices, the gather on device tensors (no host bounce)
    assert line["rccl"] == {"backend": "nccl", "nranks": 2, "version": line["rccl"]["version"], "distinct_devices": 2}
    assert line["rccl"]["version"] and len({r["pci_bus_id"] for r in line["config"]["ranks"]}) == 2
    assert line["scaling"] == "strong" and line["config"]["global_batch"] == 64


def test_bench_default_flags_two_ranks_dry_run():
    """The line the driver's scaling run asks for -- `python bench.py --gpus N` with the DEFAULT steps / warm-up / shape (N = 4096,
    J = 8, 10 timed steps, the `weak_scaling` object measured in the same run) -- as a two-rank dry run over gloo on this
    box's one device, with a global batch (and a weak batch per GPU) small enough for two ranks to share it."""
    rc, line, err = run_bench(["--gpus", "2", "--global-batch", "4096", "--weak-batch-per-gpu", "4096"],
                              {"C2_DIST_BACKEND": "gloo"}, timeout=900)
    assert rc == 0 and line is not None, err
    assert line["n_gpus"] == 2 and line["steps"] == 10 and line["warmup"] == 3 and line["scaling"] == "strong"
    assert line["config"]["global_batch"] == 4096 and line["config"]["batch_per_gpu"] == 2048 and line["config"]["N"] == 4096
    assert line["config"]["failed_factorizations"] == 0
    assert line["rccl"]["nranks"] == 2 and [r["rank"] for r in line["config"]["ranks"]] == [0, 1]
    w = line["weak_scaling"]
    assert w["scaling"] == "weak" and w["batch_per_gpu"] == 4096 and w["global_batch"] == 8192 and w["failed_factorizations"] == 0
    assert w["value"] > 0 and line["value"] > 0 and 0 < line["roofline"]["frac"] < 1
    assert "parity_sample" not in line or line["parity_sample"].get("within_1e-10", True)
