"""Multi-GPU sharding of the encode -> fuse -> score path: one process per GPU, RCCL over xGMI.

The reference is single-process / single-GPU (no torch.distributed anywhere).  What
shards naturally: encode+fuse is independent per drug, so rank r encodes a contiguous
block of drugs; ONE exchange step -- an all-gather of the per-rank embedding blocks [N/G,128]
(<= 6.4 MB per rank at N=100k: launch-latency bound on xGMI, not link-bandwidth bound, so a single
one-shot all-gather and no bucketing) -- gives every rank z[N,128]; the head is then embarrassingly
parallel by outcome (or by head row), each rank writing only its own slab of the score tensor.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist
from torch.autograd import Function


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo,hi) of ``n`` items owned by ``rank``; the first n % world ranks get one more."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n: int, world: int) -> List[int]:
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def all_gather_rows(local: torch.Tensor, n_total: int, rank: int, world: int, group=None, sizes: Optional[List[int]] = None) -> torch.Tensor:
    """Concatenate every rank's row block in rank order -> [n_total, ...].  Block sizes come from ``shard_range`` or, when
    the blocks are not the even split (rows per drug vary), from ``sizes`` (known on every rank).

    Uneven blocks are padded to the largest block so that one ``all_gather_into_tensor`` suffices."""
    if world == 1:
        return local
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # functional rehearsal of the multi-rank path without RCCL (several ranks sharing one card): stage via the host
        return all_gather_rows(local.cpu(), n_total, rank, world, group, sizes).to(local.device)
    sizes = shard_sizes(n_total, world) if sizes is None else [int(v) for v in sizes]
    if len(sizes) != world or sum(sizes) != n_total:
        raise ValueError(f"block sizes {sizes} do not add up to {n_total} rows over {world} ranks")
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


# ------------------------------------------------------------------------------------------- data-parallel finetuning
# Drug-sharded encoders (SyncBatchNorm statistics over all ranks), all-gather of the embeddings with a reduce-scatter
# gradient, the labelled triples sharded over ranks for the gathered head, one flat all-reduce of the parameter
# gradients: every rank then applies the same AdamW update.  (madrigal_amd/train.py: FinetuneStep(world > 1))

def all_reduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over ranks.  gloo + GPU tensors (several ranks rehearsing on one card) are staged through the host."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.detach().cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
        return t
    dist.all_reduce(t, group=group)
    return t


class _AllGatherRowsGrad(Function):
    """Every rank's row block concatenated in rank order; the gradient of a rank's block is the sum over ranks of the
    corresponding slice of their gradients (reduce-scatter; the [N,128] embedding matrix is a few MB, so it is done
    as one all-reduce + slice)."""

    @staticmethod
    def forward(ctx, local, n_total, rank, world, group, sizes):
        ctx.meta = (n_total, rank, world, group, sizes)
        return all_gather_rows(local.contiguous(), n_total, rank, world, group, sizes)

    @staticmethod
    def backward(ctx, dfull):
        n_total, rank, world, group, sizes = ctx.meta
        if sizes is None:
            lo, hi = shard_range(n_total, rank, world)
        else:
            lo = sum(sizes[:rank])
            hi = lo + sizes[rank]
        g = all_reduce_sum_(dfull.contiguous().clone(), group)
        return g[lo:hi].contiguous(), None, None, None, None, None


def all_gather_rows_grad(local: torch.Tensor, n_total: int, rank: int, world: int, group=None, sizes: Optional[List[int]] = None) -> torch.Tensor:
    return local if world == 1 else _AllGatherRowsGrad.apply(local, n_total, rank, world, group, None if sizes is None else tuple(int(v) for v in sizes))


def allreduce_gradients(params: Iterable[torch.nn.Parameter], group=None, bucket_bytes: int = 256 << 20) -> None:
    """Sum the gradients of ``params`` over ranks in place, in flat buckets (one collective per ~256 MB: xGMI rings are
    per-link bound, a few large messages beat hundreds of small ones).

    A parameter that has a gradient on SOME rank but not on this one (e.g. outcomes absent from the rank's triple shard)
    contributes zeros.  A parameter without a gradient on ANY rank keeps ``grad=None`` -- the single-process step skips
    such parameters (no weight decay, no step count), and so must every rank here: a has-gradient bitmap is summed over
    ranks first (one small collective and one host read per step)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    ps = [p for p in params if p.requires_grad]
    if not ps:
        return
    dev = ps[0].device
    have = torch.tensor([0 if p.grad is None else 1 for p in ps], dtype=torch.int32).to(dev)
    all_reduce_sum_(have, group)
    have = have.cpu().tolist()
    live = []
    for p, h in zip(ps, have):
        if h == 0:
            continue
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        live.append(p)
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in bucket])
        all_reduce_sum_(flat, group)
        off = 0
        for p in bucket:
            n = p.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
        bucket, size = [], 0
    for p in live:
        bucket.append(p)
        size += p.numel() * p.element_size()
        if size >= bucket_bytes:
            flush()
    flush()


_BUCKET_GROUPS = {}          # (ranks, backend) -> the bucket communicator (never re-created per step object)


def destroy_bucket_groups() -> None:
    """Release the bucket communicators (call before dist.destroy_process_group() in a long-lived process)."""
    for g in _BUCKET_GROUPS.values():
        try:
            dist.destroy_process_group(g)
        except Exception:
            pass
    _BUCKET_GROUPS.clear()


class GradientBuckets:
    """Bucketed gradient all-reduce that overlaps the backward pass (the reference is single-GPU; torch DDP's scheme, restated
    for this step: parameters in REVERSE registration order -- roughly the order their gradients become final -- are cut
    into flat buckets of ``bucket_bytes``; a post-accumulate hook marks a gradient ready; bucket k is all-reduced
    (``async_op``) as soon as its gradients are all ready AND buckets 0..k-1 have been issued, so every rank issues the
    same sequence of collectives whatever the arrival order.  ``finish()`` -- after ``backward()`` -- issues the buckets
    that never became ready on this rank (a parameter without a local gradient contributes zeros), then sums a
    has-gradient bitmap over the ranks, waits, copies the sums back, and resets to ``None`` the gradients of parameters
    that received a gradient on NO rank (the single-process step skips those: no weight decay, no step count).

    xGMI rings are per-link bound (about 153 GB/s per link): buckets default to 64 MB -- 20 to 80 M fp32 parameters make
    2 to 5 collectives per step, each long enough to run at link rate, the last ones hidden under the encoders' backward."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, bucket_bytes: int = 64 << 20, overlap: Optional[bool] = None):
        # A communicator of its own: bucket collectives are issued from gradient hooks BETWEEN the collectives the backward
        # pass itself issues (SyncBatchNorm sums, the embedding reduce-scatter); on a rank that lacks some gradient a bucket
        # moves to finish(), i.e. to another position in that rank's stream of collectives.  NON-BLOCKING collectives of
        # different communicators need no common order, so the buckets (fixed order among themselves) cannot pair up with the
        # wrong partner.  (Constructed on every rank, like the step objects that own it; one communicator per set of ranks and
        # backend is kept for the life of the process and shared by every GradientBuckets over it.)
        self.params = [p for p in params if p.requires_grad]
        blocking = False
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            backend = dist.get_backend(group)
            ranks = tuple(range(dist.get_world_size())) if group is None else tuple(dist.get_process_group_ranks(group))
            key = (ranks, backend)
            if key not in _BUCKET_GROUPS:
                _BUCKET_GROUPS[key] = dist.new_group(ranks=list(ranks), backend=backend)
            group = _BUCKET_GROUPS[key]
            # gloo on device tensors is staged through the host and BLOCKS: a bucket issued from a hook on one rank while
            # another rank (which lacks one of the bucket's gradients and defers it to finish()) blocks in a SyncBatchNorm
            # all-reduce of the default group would leave each waiting for the other.  A blocking transport issues every
            # bucket in finish(), after the backward pass, in the one fixed order.
            blocking = backend == "gloo" and any(p.is_cuda for p in self.params)
        self.overlap = (not blocking) if overlap is None else (bool(overlap) and not blocking)
        self.group = group
        self.bucket_bytes = bucket_bytes
        self.cold = frozenset()                          # parameters that had a gradient on NO rank in the last step
        self._hooks = []
        self._armed = False
        self._build()

    def _build(self) -> None:
        """Buckets over the parameters in reverse registration order; parameters that received no gradient on any rank in
        the previous step (modules outside the step's path: e.g. the fusion transformer under raw-encoder-output
        pretraining) go last, in buckets of their own -- a bucket that can never become ready would otherwise hold back
        every bucket behind it (buckets are issued strictly in order so that all ranks issue the same sequence).  The
        has-gradient bitmap is summed over ranks, so every rank rebuilds the same buckets."""
        self.buckets: List[List[int]] = []
        for group_idx in ([i for i in reversed(range(len(self.params))) if i not in self.cold],
                          [i for i in reversed(range(len(self.params))) if i in self.cold]):
            cur, size = [], 0
            for i in group_idx:
                cur.append(i)
                size += self.params[i].numel() * self.params[i].element_size()
                if size >= self.bucket_bytes:
                    self.buckets.append(cur)
                    cur, size = [], 0
            if cur:
                self.buckets.append(cur)
        self.bucket_of = {i: b for b, idxs in enumerate(self.buckets) for i in idxs}
        self._reset()

    def _reset(self):
        self.ready = [0] * len(self.buckets)
        self.seen = set()
        self.next_bucket = 0
        self.inflight = []                              # (bucket, flat tensor, work handle | None)

    def arm(self) -> None:
        """Call before the (single) backward pass of a step whose gradients start from None."""
        if not self._hooks:
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(lambda _p, i=i: self._on_grad(i)))
        self._reset()
        self._armed = True

    def _on_grad(self, i: int) -> None:
        if not self._armed or i in self.seen:
            return
        self.seen.add(i)
        b = self.bucket_of[i]
        self.ready[b] += 1
        if not self.overlap:
            return
        while self.next_bucket < len(self.buckets) and self.ready[self.next_bucket] == len(self.buckets[self.next_bucket]):
            self._issue(self.next_bucket)
            self.next_bucket += 1

    def _issue(self, b: int) -> None:
        ps = [self.params[i] for i in self.buckets[b]]
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ps])
        work = None
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            if flat.is_cuda and dist.get_backend(self.group) == "gloo":
                all_reduce_sum_(flat, self.group)         # rehearsal on one card: staged through the host, synchronous
            else:
                work = dist.all_reduce(flat, group=self.group, async_op=True)
        self.inflight.append((b, flat, work))

    def finish(self) -> None:
        """After ``backward()``: complete every bucket (same order on every rank) and write the summed gradients back."""
        self._armed = False
        if not self.params:
            return
        dev = self.params[0].device
        have = torch.tensor([0 if p.grad is None else 1 for p in self.params], dtype=torch.int32).to(dev)
        while self.next_bucket < len(self.buckets):      # the buckets this rank could not complete, in order
            self._issue(self.next_bucket)
            self.next_bucket += 1
        all_reduce_sum_(have, self.group)                # AFTER the last bucket on every rank: one sequence of collectives
        have = have.cpu().tolist()
        for b, flat, work in self.inflight:
            if work is not None:
                work.wait()
            off = 0
            for i in self.buckets[b]:
                p = self.params[i]
                n = p.numel()
                if have[i]:
                    if p.grad is None:
                        p.grad = flat[off:off + n].view_as(p).clone()
                    else:
                        p.grad.copy_(flat[off:off + n].view_as(p.grad))
                else:
                    p.grad = None
                off += n
        self.inflight = []
        cold = frozenset(i for i, h in enumerate(have) if not h)
        if cold != self.cold:
            self.cold = cold
            self._build()
