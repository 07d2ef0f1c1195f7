"""Multi-GPU sharding of the encode -> fuse -> score path: one process per GPU, RCCL over xGMI.

The reference is single-process / single-GPU (no torch.distributed anywhere).  What
shards naturally: encode+fuse is independent per drug, so rank r encodes a contiguous
block of drugs; ONE exchange step -- an all-gather of the per-rank embedding blocks [N/G,128]
(<= 6.4 MB per rank at N=100k: launch-latency bound on xGMI, not link-bandwidth bound, so a single
one-shot all-gather and no bucketing) -- gives every rank z[N,128]; the head is then embarrassingly
parallel by outcome (or by head row), each rank writing only its own slab of the score tensor.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo,hi) of ``n`` items owned by ``rank``; the first n % world ranks get one more."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n: int, world: int) -> List[int]:
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def all_gather_rows(local: torch.Tensor, n_total: int, rank: int, world: int, group=None) -> torch.Tensor:
    """Concatenate every rank's row block (sizes from ``shard_range``) in rank order -> [n_total, ...].

    Uneven blocks are padded to the largest block so that one ``all_gather_into_tensor`` suffices."""
    if world == 1:
        return local
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # functional rehearsal of the multi-rank path without RCCL (several ranks sharing one card): stage via the host
        return all_gather_rows(local.cpu(), n_total, rank, world, group).to(local.device)
    sizes = shard_sizes(n_total, world)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank}: expected {sizes[rank]} rows, got {local.shape[0]}")
    m = max(sizes)
    if min(sizes) == m:
        out = torch.empty((n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    buf = torch.empty((world * m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * m: r * m + sizes[r]] for r in range(world)], dim=0)
