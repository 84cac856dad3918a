# -*- coding: utf-8 -*-
"""Multi-GPU plumbing for the batched path.

Series are independent, so the batch shards with NO data-path collective: rank r owns the contiguous block
`shard_range(B, r, world)` (inputs generated / loaded directly on the owning GPU, gradients stay sharded).
The one exchange of the path is an all-gather of the per-rank log-likelihood vectors -- B/n_gpu float64 per
rank (64 KiB at B=65536 on 8 GPUs), latency-bound, RCCL over xGMI (`backend="nccl"` on ROCm) or gloo on CPU.
"""
import torch
import torch.distributed as dist


def shard_range(B, rank, world):
    """(first, count) of the contiguous shard of `rank`: sizes differ by at most one."""
    base, rem = divmod(B, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def gather_loglik(ll_local, B, world=None, force=False):
    """All-gather the per-rank (count_r,) vectors into the full (B,) vector on every rank.  `force` runs the
    collective even for a single rank (communicator smoke test on a one-GPU box)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1 and not (force and dist.is_initialized()):
        return ll_local
    counts = [shard_range(B, r, world)[1] for r in range(world)]
    if len(set(counts)) == 1:
        out = torch.empty(B, dtype=ll_local.dtype, device=ll_local.device)
        dist.all_gather_into_tensor(out, ll_local.contiguous())
        return out
    pad = max(counts)
    buf = torch.zeros(pad, dtype=ll_local.dtype, device=ll_local.device)
    buf[: ll_local.numel()] = ll_local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    return torch.cat([p[:n] for p, n in zip(parts, counts)])
