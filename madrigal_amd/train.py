"""The finetune step of train_ddi_batch.py:231-354 on the HIP path.

The reference runs, per epoch (full-batch training): ``optimizer.zero_grad()``; ``sigmoid(model(batch_head, batch_tail,
masks_head, masks_tail, batch_kg))[labels, heads, tails]`` -> ``nn.BCELoss`` -> ``backward()`` (once, or three times
with different modality masks in the ``str_*`` finetune modes) -> ``optimizer.step()``.  ``FinetuneStep`` keeps that
shape; only the dense [L,N,N] intermediate is gone: ``NovelDDIMultilabel.score_triples`` computes exactly the gathered
entries (mdg_bilinear_gather), forward and backward.
"""
from __future__ import annotations

import torch

from . import autograd as ag
from . import ops
from .parallel import GradientBuckets, all_gather_rows_grad, all_reduce_sum_, allreduce_gradients, shard_range
from .pipeline import slice_batch


def _ensure_training(model) -> None:
    """``model.train()`` of the reference's loops (train_ddi_batch.py:231, pretrain.py:61: once per epoch there) without walking
    the ~200 submodules in every step: the walk happens when the model, or any submodule, is not in training mode."""
    if not model.training or not all(m.training for m in model.modules()):
        model.train()


def _refuse_rank_local_batchnorm(encoder, world: int) -> None:
    """Data-parallel steps synchronise BatchNorm statistics with collectives issued from inside the BatchNorm nodes, so every
    rank must run every BatchNorm-bearing module the same number of times.  GIN and the chemCPA encoder always run (every
    rank owns at least one drug, and training encodes every (drug, cell line) row); the uni-modal projector / fuser run only
    on ranks whose drug block has uni-modal rows (models.py:855-865) -- with ``proj_norm='bn'`` a rank without such rows
    would skip collectives the others issue (a hang under RCCL).  The shipped configs use LayerNorm there; 'bn' is refused."""
    if world <= 1:
        return
    for name in ("uni_projector", "uni_fuser"):
        mod = getattr(encoder, name, None)
        if mod is not None and any(isinstance(m, torch.nn.BatchNorm1d) for m in mod.modules()):
            raise NotImplementedError(f"data-parallel training with BatchNorm inside encoder.{name} (proj_norm='bn'): the module runs on a "
                                      "data-dependent subset of the ranks, which would mismatch the SyncBatchNorm collectives")


class FinetuneStep:
    """``step(...)`` = one optimizer step of the reference's 'full_full' / 'double_random' modes;
    ``accumulate(...)`` = one ``loss.backward()`` of the multi-pass modes (call it per mask pair, then ``apply()``)."""

    def __init__(self, model, optimizer, loss_readout: str = "mean", scheduler=None, rank: int = 0, world: int = 1, group=None,
                 shard_kg: bool = True):
        """``world > 1``: data-parallel step over one process per GPU (the reference is single-GPU).  Every rank
        holds the full model and the full batch description; rank r encodes a contiguous block of drugs on both sides
        (SyncBatchNorm statistics over all ranks; the KG encoder is per-graph and runs replicated, once per step), the
        embedding blocks are all-gathered (reduce-scatter gradient), the labelled triples are dealt round-robin to the ranks
        for the gathered head, and the parameter gradients are summed in flat buckets before the identical AdamW update."""
        self.model, self.optimizer, self.loss_readout, self.scheduler = model, optimizer, loss_readout, scheduler
        self.rank, self.world, self.group = rank, world, group
        # data-parallel: the KG encoder's convs run destination-partitioned (HGTConv._forward_train(shard=...)): every KG edge is
        # attended to on exactly one rank instead of on all of them; False keeps the KG encoder replicated
        self.shard_kg = bool(shard_kg) and world > 1
        model.encoder.kg_graph = False          # (the finetune step is GPU-bound: captured KG graphs cost it 1 ms; see NovelDDIEncoder._kg_graphed)
        _refuse_rank_local_batchnorm(model.encoder, world)
        # single-pass steps overlap the gradient all-reduce with the backward pass (buckets issued from gradient hooks);
        # the multi-pass modes (accumulate() ... apply()) sum the finished gradients in flat buckets afterwards
        self._buckets = GradientBuckets(model.parameters(), group) if world > 1 else None
        self._overlapped = False
        self._plan_key = None
        self._plan = None
        self._plans = {}                # key -> (plan, index tensors): the current and a prefetched set of triples
        self._plan_stream = None
        self._shards = {}               # side ('head' | 'tail') -> (key, sliced batch, the batch itself): ONE entry per side

    @staticmethod
    def _key_of(labels, heads, tails, n_head, n_tail):
        return tuple((t.data_ptr(), t._version, t.numel()) for t in (labels, heads, tails)) + (n_head, n_tail)

    def plan(self, labels: torch.Tensor, heads: torch.Tensor, tails: torch.Tensor, n_head: int, n_tail: int) -> dict:
        """Label-sorted tiling of the triples, rebuilt only when the index tensors change (they are fixed for a run)."""
        key = self._key_of(labels, heads, tails, n_head, n_tail)
        if self._plan_key != key:
            hit = self._plans.get(key)
            if hit is None or hit[1][0] is not labels:
                n_labels = int(self.model.decoder.out_features)
                hit = (ops.triple_plan(labels, heads, tails, n_labels, n_head, n_tail), (labels, heads, tails))
                while len(self._plans) >= 3:
                    self._plans.pop(next(iter(self._plans)))
                self._plans[key] = hit
            self._plan, self._pinned = hit
            self._plan_key = key
        return self._plan

    def _shard(self, batch, masks, side):
        """This rank's block of drugs.  The slice of the last batch seen on each side is kept (full-batch finetuning
        passes the same dict every step, train_ddi_batch.py:116); a different batch replaces it, nothing accumulates."""
        n = int(batch["drugs"].shape[0])
        lo, hi = shard_range(n, self.rank, self.world)
        key = (id(batch), lo, hi)
        hit = self._shards.get(side)
        if hit is None or hit[0] != key or hit[2] is not batch:
            hit = (key, slice_batch(batch, lo, hi), batch)                   # holds ``batch``: its id cannot be recycled meanwhile
            self._shards[side] = hit
        return hit[1], masks[lo:hi], n

    def _accumulate_sharded(self, batch_head, batch_tail, masks_head, masks_tail, batch_kg, labels, heads, tails, targets, **kwargs):
        model, rank, world, group = self.model, self.rank, self.world, self.group
        ag.set_batchnorm_sync(lambda t: all_reduce_sum_(t, group))
        try:
            kw = dict(kwargs, kg_share={}) if 'kg_share' not in kwargs else dict(kwargs)
            if self.shard_kg:
                kw['kg_shard'] = (rank, world, group)
            sides = []
            for side, batch, masks in (("head", batch_head, masks_head), ("tail", batch_tail, masks_tail)):
                b, m, n = self._shard(batch, masks, side)
                z = model.encoder(b["drugs"], m, b["strs"], batch_kg, b["cv"], b["tx"], **kw)
                z = all_gather_rows_grad(z, n, rank, world, group)
                sides.append(ag.l2_normalize(z) if model.normalize else z)
            T = int(labels.numel())
            key = tuple((t.data_ptr(), t._version, t.numel()) for t in (labels, heads, tails)) + (sides[0].shape[0], sides[1].shape[0], rank, world)
            if self._plan_key != key:
                mine = torch.arange(rank, T, world, device=labels.device)
                self._plan = (ops.triple_plan(labels[mine].contiguous(), heads[mine].contiguous(), tails[mine].contiguous(),
                                              int(model.decoder.out_features), sides[0].shape[0], sides[1].shape[0]), mine)
                self._plan_key = key
                self._pinned = (labels, heads, tails)
            plan, mine = self._plan
            scores = model.decoder.score_triples(sides[0], sides[1], plan)                    # label-sorted order of the local triples
            y = targets[mine][plan["perm"]]
            part = ag.bce_with_sigmoid(scores, y, "sum")
            loss = part * (1.0 / T) if self.loss_readout == "mean" else part
            loss.backward()
            total = loss.detach().reshape(1).clone()
            all_reduce_sum_(total, group)
            return total.reshape(())
        finally:
            ag.set_batchnorm_sync(None)

    def accumulate(self, batch_head, batch_tail, masks_head, masks_tail, batch_kg, labels, heads, tails, targets, **kwargs) -> torch.Tensor:
        if self.world > 1:
            return self._accumulate_sharded(batch_head, batch_tail, masks_head, masks_tail, batch_kg, labels, heads, tails, targets, **kwargs)
        n_head, n_tail = int(batch_head["drugs"].shape[0]), int(batch_tail["drugs"].shape[0])
        dev = labels.device
        if dev.type == "cuda" and self._plan_key != self._key_of(labels, heads, tails, n_head, n_tail):
            # A new set of triples (every batch of an epoch): its plan is a handful of sorts with host round trips for the
            # tile / pair counts.  Queue the encoders first, then build the plan on a side stream that waits only for what
            # was queued BEFORE this step (where the index tensors come from): the sorts run beside the encoders and the host
            # round trips wait for the sorts alone, not for the encoders.
            main = torch.cuda.current_stream(dev)
            start = torch.cuda.Event()
            start.record(main)
            z_head, z_tail = self.model.embed(batch_head, batch_tail, masks_head, masks_tail, batch_kg, **kwargs)
            if self._plan_stream is None:
                self._plan_stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(self._plan_stream):
                self._plan_stream.wait_event(start)
                plan = self.plan(labels, heads, tails, n_head, n_tail)
            main.wait_stream(self._plan_stream)
            scores = self.model.decoder.score_triples(z_head, z_tail, plan).index_select(0, plan["inv_perm"])
        else:
            plan = self.plan(labels, heads, tails, n_head, n_tail)
            scores = self.model.score_triples(batch_head, batch_tail, masks_head, masks_tail, batch_kg, plan, **kwargs)
        loss = ag.bce_with_sigmoid(scores, targets, self.loss_readout)
        loss.backward()
        return loss.detach()

    def apply(self) -> None:
        if self.world > 1:              # once per optimizer step, after every accumulate() of the step
            if self._overlapped:
                self._buckets.finish()
                self._overlapped = False
            else:
                allreduce_gradients(self.model.parameters(), self.group)
        self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()

    def step(self, batch_head, batch_tail, masks_head, masks_tail, batch_kg, labels, heads, tails, targets, **kwargs) -> torch.Tensor:
        _ensure_training(self.model)
        self.optimizer.zero_grad(set_to_none=True)
        if self._buckets is not None:
            self._buckets.arm()
            self._overlapped = True
        loss = self.accumulate(batch_head, batch_tail, masks_head, masks_tail, batch_kg, labels, heads, tails, targets, **kwargs)
        self.apply()
        return loss


class PretrainStep:
    """One iteration of pretrain.py:59-93 (train_epoch): ``optimizer.zero_grad()``; ``SimCLR_NovelDDI(drug_indices, mask1, mask2,
    too_hard_neg, batch_data)`` -> InfoNCE loss; ``backward()``; ``optimizer.step()``.

    ``world > 1`` (BASELINE configs[2]: contrastive pretraining, batch 2048, data-parallel with an all-gather of the features):
    rank r runs both views of a contiguous block of the batch through the encoder and the predictors (SyncBatchNorm), the
    predictor outputs are all-gathered (gradient = all-reduce + own slice), every rank evaluates the same [2R,2R] loss scaled
    by 1/world (so that the summed gradients are exact), and the parameter gradients are summed in flat buckets.  With
    ``raw_encoder_output=True`` (every shipped config: encoders -> uni_projector only, models.py:890-894) a view holds one row
    per AVAILABLE (drug, modality) pair in drug-major order, so the blocks of the ranks concatenate to the single-process row
    order; their sizes follow from the full mask tensors every rank holds ('str_center_uni' views: exactly one row per drug).
    With dropout off the sharded step reproduces the single-process step."""

    def __init__(self, model, optimizer, rank: int = 0, world: int = 1, group=None, scheduler=None, shard_kg: bool = True, kg_graph: bool = True):
        """``kg_graph`` (single process): the KG encoder's forward and backward replay as two captured hipGraphs
        (NovelDDIEncoder._kg_graphed: the contrastive step is bound by the host's launch rate, ~1 000 small launches per step, 40 %
        of them the KG pass; 17.7-19.8 -> 16.0 ms per 2048-drug step).  ``shard_kg`` (world > 1): destination-partitioned KG convs,
        as in ``FinetuneStep``.  ``scheduler``: called with the optimizer at the top of every step, before ``zero_grad`` -- pretrain.py:65 adjusts the
        learning rate per ITERATION (``optim.PretrainSchedule`` mirrors madrigal/utils.py:680-692); None keeps the rates."""
        self.model, self.optimizer, self.rank, self.world, self.group, self.scheduler = model, optimizer, rank, world, group, scheduler
        self.shard_kg = bool(shard_kg) and world > 1
        model.base_encoder.kg_graph = bool(kg_graph) and world == 1
        _refuse_rank_local_batchnorm(model.base_encoder, world)
        self._buckets = GradientBuckets(model.parameters(), group) if world > 1 else None
        self._last = None               # (key, sliced batch, batch_data): the slice of the LAST batch only (a DataLoader
        #                                 hands over a fresh batch every iteration: nothing may accumulate here)

    def _local(self, drug_indices, batch_data):
        mols, kg, cv, tx = batch_data
        B = int(drug_indices.shape[0])
        lo, hi = shard_range(B, self.rank, self.world)
        key = (id(mols), id(cv), lo, hi)
        if self._last is None or self._last[0] != key or self._last[2][0] is not mols:
            whole = {"drugs": drug_indices, "strs": mols, "cv": cv, "tx": tx, "masks": torch.zeros(B, 1, dtype=torch.bool, device=cv.device)}
            self._last = (key, slice_batch(whole, lo, hi), batch_data)
        return self._last[1], lo, hi, B

    def _row_blocks(self, masks: torch.Tensor, B: int):
        """Rows each rank contributes to a raw-encoder-output view: available (drug, modality) pairs of its drug block."""
        per_drug = (~masks).sum(dim=1).cpu()
        return [int(per_drug[slice(*shard_range(B, r, self.world))].sum()) for r in range(self.world)]

    def step(self, drug_indices, mask1, mask2, too_hard_neg, batch_data) -> torch.Tensor:
        model = self.model
        _ensure_training(model)
        if self.scheduler is not None:
            self.scheduler(self.optimizer)
        self.optimizer.zero_grad(set_to_none=True)
        if self.world == 1:
            _, _, (_, _, loss) = model(drug_indices, mask1, mask2, too_hard_neg, batch_data)
            loss.backward()
            self.optimizer.step()
            return loss.detach()
        from .models import _run_sequential_train
        group = self.group
        b, lo, hi, B = self._local(drug_indices, batch_data)
        kg = batch_data[1]
        raw = bool(model.raw_encoder_output)
        ag.set_batchnorm_sync(lambda t: all_reduce_sum_(t, group))
        try:
            p1 = model.predictor if model.shared_predictor else model.predictor_1
            p2 = model.predictor if model.shared_predictor else model.predictor_2
            share = {}
            views = []
            for masks, pred in ((mask1, p1), (mask2, p2)):
                sizes = self._row_blocks(masks, B) if raw else None
                e = model.base_encoder(b["drugs"], masks[lo:hi], b["strs"], kg, b["cv"], b["tx"], raw_encoder_output=raw, kg_share=share,
                                       **({"kg_shard": (self.rank, self.world, group)} if self.shard_kg else {}))
                views.append(all_gather_rows_grad(_run_sequential_train(pred, e), sum(sizes) if raw else B, self.rank, self.world, group, sizes))
            _, _, loss = model.contrastive_loss(views[0], views[1], too_hard_neg)
            self._buckets.arm()                     # gradient buckets are all-reduced from hooks while backward still runs
            (loss * (1.0 / self.world)).backward()
            self._buckets.finish()
        finally:
            ag.set_batchnorm_sync(None)
        self.optimizer.step()
        return loss.detach()
