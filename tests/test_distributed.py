# -*- coding: utf-8 -*-
"""Multi-process (gloo, world_size 2, CPU) test of the N>1 path of bench.py / celerite2_amd.parallel:
the batch is sharded contiguously across ranks with NO data-path collective; the only exchange is an
all-gather of the per-rank log-likelihood vectors.  The per-shard compute is stood in for by the CPU oracle
(there is no GPU here); what is under test is the sharding / gather logic that the RCCL run uses verbatim."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, N, J, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from celerite2_amd import parallel
    from oracle import cpu, dense

    first, count = parallel.shard_range(B, rank, world)
    t, c, a, U, V, y = dense.synthetic_batch(count, N, J, seed0=721 + first)
    ll, flag = cpu.loglik_batched(t, c, a, U, V, y, nthreads=1)      # stand-in for ops.loglik on the shard
    full = parallel.gather_loglik(torch.from_numpy(ll), B, world)       # the path's one exchange
    flags = parallel.gather_loglik(torch.from_numpy(flag.astype(np.float64)), B, world)
    if rank == 0:
        np.save(os.path.join(out_dir, "ll.npy"), full.numpy())
        np.save(os.path.join(out_dir, "flag.npy"), flags.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 11])
def test_sharded_loglik_gather(tmp_path, B):
    from oracle import cpu, dense

    N, J, world = 64, 4, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, N, J, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "ll.npy")
    t, c, a, U, V, y = dense.synthetic_batch(B, N, J)
    want, _ = cpu.loglik_batched(t, c, a, U, V, y, nthreads=1)
    assert got.shape == (B,)
    np.testing.assert_array_equal(got, want)          # same series -> same bits, wherever they were computed
    assert np.all(np.load(tmp_path / "flag.npy") == 0)


def test_shard_range_covers_batch():
    from celerite2_amd import parallel

    for B in (1, 7, 8, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == B
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
