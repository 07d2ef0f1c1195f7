"""The finetune step of train_ddi_batch.py:231-354 on the HIP path.

The reference runs, per epoch (full-batch training): ``optimizer.zero_grad()``; ``sigmoid(model(batch_head, batch_tail,
masks_head, masks_tail, batch_kg))[labels, heads, tails]`` -> ``nn.BCELoss`` -> ``backward()`` (once, or three times
with different modality masks in the ``str_*`` finetune modes) -> ``optimizer.step()``.  ``FinetuneStep`` keeps that
shape; only the dense [L,N,N] intermediate is gone: ``NovelDDIMultilabel.score_triples`` computes exactly the gathered
entries (mdg_bilinear_gather), forward and backward.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import autograd as ag
from . import ops


class FinetuneStep:
    """``step(...)`` = one optimizer step of the reference's 'full_full' / 'double_random' modes;
    ``accumulate(...)`` = one ``loss.backward()`` of the multi-pass modes (call it per mask pair, then ``apply()``)."""

    def __init__(self, model, optimizer, loss_readout: str = "mean", scheduler=None):
        self.model, self.optimizer, self.loss_readout, self.scheduler = model, optimizer, loss_readout, scheduler
        self._plan_key = None
        self._plan = None

    def plan(self, labels: torch.Tensor, heads: torch.Tensor, tails: torch.Tensor, n_head: int, n_tail: int) -> dict:
        """Label-sorted tiling of the triples, rebuilt only when the index tensors change (they are fixed for a run)."""
        key = tuple((t.data_ptr(), t._version, t.numel()) for t in (labels, heads, tails)) + (n_head, n_tail)
        if self._plan_key != key:
            n_labels = int(self.model.decoder.out_features)
            self._plan = ops.triple_plan(labels, heads, tails, n_labels, n_head, n_tail)
            self._plan_key = key
            self._pinned = (labels, heads, tails)
        return self._plan

    def accumulate(self, batch_head, batch_tail, masks_head, masks_tail, batch_kg, labels, heads, tails, targets, **kwargs) -> torch.Tensor:
        plan = self.plan(labels, heads, tails, int(batch_head["drugs"].shape[0]), int(batch_tail["drugs"].shape[0]))
        scores = self.model.score_triples(batch_head, batch_tail, masks_head, masks_tail, batch_kg, plan, **kwargs)
        loss = ag.bce_with_sigmoid(scores, targets, self.loss_readout)
        loss.backward()
        return loss.detach()

    def apply(self) -> None:
        self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()

    def step(self, batch_head, batch_tail, masks_head, masks_tail, batch_kg, labels, heads, tails, targets, **kwargs) -> torch.Tensor:
        self.model.train()
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.accumulate(batch_head, batch_tail, masks_head, masks_tail, batch_kg, labels, heads, tails, targets, **kwargs)
        self.apply()
        return loss
