"""The acceptance metric of the path: AUPRC (average precision) per outcome, macro-averaged
(madrigal/evaluate/metrics.py:60-191: sklearn ``average_precision_score`` per label, then the mean over labels with both
classes present).  Harness-side: sorting and prefix sums on the device (torch), no kernels of its own; sklearn is the
checker in tests/."""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def average_precision(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """sklearn.metrics.average_precision_score(target, pred) for one binary problem: sum over the DISTINCT thresholds of
    (recall_k - recall_{k-1}) * precision_k, scores sorted descending.  Returns a 0-dim float64 tensor (NaN without positives)."""
    if pred.shape != target.shape or pred.dim() != 1:
        raise ValueError("average_precision: 1-D pred / target of the same length")
    order = torch.argsort(pred, descending=True, stable=True)
    p, y = pred[order], target[order].to(torch.float64)
    tp = torch.cumsum(y, 0)
    n_pos = tp[-1] if y.numel() else torch.zeros((), dtype=torch.float64, device=pred.device)
    last = torch.ones_like(p, dtype=torch.bool)                 # last element of each run of equal scores = one threshold
    if p.numel() > 1:
        last[:-1] = p[1:] != p[:-1]
    tp_k = tp[last]
    k = torch.nonzero(last).flatten().to(torch.float64) + 1.0
    precision = tp_k / k
    recall = tp_k / n_pos
    prev = torch.cat([torch.zeros(1, dtype=torch.float64, device=pred.device), recall[:-1]])
    return ((recall - prev) * precision).sum()


def macro_auprc(pred: torch.Tensor, target: torch.Tensor, labels: torch.Tensor, n_labels: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-outcome AUPRC of labelled samples (pred / target / labels all [T]) and its mean over the outcomes that have
    both classes (get_metrics, metrics.py:129-191) -> (macro, per_label [L] with NaN where undefined)."""
    L = int(labels.max().item()) + 1 if n_labels is None else n_labels
    out = torch.full((L,), float("nan"), dtype=torch.float64, device=pred.device)
    order = torch.argsort(labels, stable=True)
    ls, ps, ys = labels[order], pred[order], target[order]
    counts = torch.bincount(ls, minlength=L)
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=pred.device), torch.cumsum(counts, 0)]).tolist()
    for l in range(L):
        lo, hi = ptr[l], ptr[l + 1]
        if hi > lo:
            y = ys[lo:hi]
            s = float(y.sum())
            if 0 < s < hi - lo:
                out[l] = average_precision(ps[lo:hi], y)
    return torch.nanmean(out), out
