"""Modality masks of the evaluation settings (madrigal/evaluate/eval_utils.py:39-305): host-side index logic of the
train / evaluate harness around the path.  A mask row is bool[19] (str, kg, cv, 16 tx cell lines), True = ABSENT.

An evaluation type is ``"<head>_<tail>"``; each side is either ``full`` (everything the finetune mode ever saw) or a
``+``-joined list of modalities.  A single modality means "exactly that modality, for every drug"; a list keeps the drug's
own availability among the listed modalities and hides the rest.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from .data import CELL_LINES, NON_TX_MODALITIES, NUM_MODALITIES, NUM_NON_TX_MODALITIES

_TX_COLS = [NUM_NON_TX_MODALITIES + i for i in range(len(CELL_LINES))]

# columns of each named modality (eval_utils.py:39-56)
MODALITY_COLUMNS: Dict[str, List[int]] = {m: [i] for i, m in enumerate(NON_TX_MODALITIES)}
MODALITY_COLUMNS.update({f"tx_{c}": [NUM_NON_TX_MODALITIES + i] for i, c in enumerate(CELL_LINES)})
MODALITY_COLUMNS["tx"] = list(_TX_COLS)


def _never_seen(finetune_mode: str) -> List[int]:
    """Columns an ablation finetune mode never trained on (eval_utils.py:112-135): everything outside the modalities named
    in the mode.  'ablation_<a>_<a>_…' names one modality; 'ablation_str_random_str+x+y_full_sample' names str plus x, y."""
    if finetune_mode == "ablation_str_str":
        named = {"str"}
    elif finetune_mode.startswith("ablation_str_random_"):
        named = set(finetune_mode[len("ablation_str_random_"):].split("_")[0].split("+"))
    else:
        named = {finetune_mode.split("_")[1]}
    unknown = named - set(NON_TX_MODALITIES) - {"tx"}
    if unknown:
        raise KeyError(finetune_mode)
    cols = [i for i, m in enumerate(NON_TX_MODALITIES) if m not in named]
    return cols + ([] if "tx" in named else list(_TX_COLS))


def full_mask_for_finetune_mode(finetune_mode: str, masks_base: torch.Tensor) -> torch.Tensor:
    """eval_utils.py:248-263: 'full' = the drug's own availability, minus what an ablation mode never saw; the
    single-modality ablations additionally force their modality present."""
    out = masks_base.clone()
    if "ablation" in finetune_mode:
        out[:, _never_seen(finetune_mode)] = True
        if "kg_kg" in finetune_mode:
            out[:, 1] = False
        elif "cv_cv" in finetune_mode:
            out[:, 2] = False
        elif "tx_tx" in finetune_mode:
            out[:, NUM_NON_TX_MODALITIES:] = False
    return out


def modality_mask(masks_base: torch.Tensor, modality: str) -> torch.Tensor:
    """eval_utils.py:266-284."""
    if "+" not in modality:
        out = torch.ones_like(masks_base)
        out[:, MODALITY_COLUMNS[modality]] = 0
        return out.bool()
    keep = set()
    for m in modality.split("+"):
        keep.update(MODALITY_COLUMNS[m])
    out = masks_base.clone()
    out[:, [c for c in range(NUM_MODALITIES) if c not in keep]] = 1
    return out.bool()


def get_evaluate_masks(head_masks_base: torch.Tensor, tail_masks_base: torch.Tensor, eval_type: str, finetune_mode: str,
                       device) -> Tuple[torch.Tensor, torch.Tensor]:
    """eval_utils.py:287-305: (head masks, tail masks) on ``device`` for one evaluation type."""
    parts = eval_type.split("_")
    if len(parts) != 2:
        raise AssertionError(f"eval_type must be '<head>_<tail>', got {eval_type!r}")
    sides = []
    for kind, base in zip(parts, (head_masks_base, tail_masks_base)):
        sides.append(full_mask_for_finetune_mode(finetune_mode, base) if kind == "full" else modality_mask(base, kind))
    return sides[0].to(device), sides[1].to(device)
