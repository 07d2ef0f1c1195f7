"""MI355X-native drop-in for ``madrigal.models.simclr`` (second-stage contrastive wrapper).

``SimCLR_NovelDDI`` keeps the reference's constructor, attribute names (``base_encoder``, ``predictor_1``,
``predictor_2`` / ``predictor``) and return value ``(aug_1, aug_2, (logits, labels, loss))``
(madrigal/models/simclr.py:11-140).  In training mode (pretrain.py) the encoder, the predictor MLPs (BatchNorm batch
statistics) and the InfoNCE loss record a tape through madrigal_amd.autograd; forward and backward run in HIP kernels.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import autograd as ag
from . import ops
from .models import _run_sequential, _run_sequential_train, _train_path, get_precision


class SimCLR_NovelDDI(nn.Module):
    def __init__(self, base_encoder, dim=256, mlp_dim=1024, T=1.0, raw_encoder_output=False, shared_predictor=False):
        super().__init__()
        self.base_encoder = base_encoder
        self.T = T
        self.raw_encoder_output = raw_encoder_output
        self.shared_predictor = shared_predictor
        hidden_dim = self.base_encoder.uni_projector.fc[-1].weight.shape[0]
        assert hidden_dim == dim, f"Hidden dim of the encoder ({hidden_dim}) should be the same as the dim of the projection head ({dim})."
        if shared_predictor:
            self.predictor = self._build_mlp(2, dim, mlp_dim, dim)
        else:
            self.predictor_1 = self._build_mlp(2, dim, mlp_dim, dim)
            self.predictor_2 = self._build_mlp(2, dim, mlp_dim, dim)

    @staticmethod
    def _build_mlp(num_layers, input_dim, mlp_dim, output_dim, last_bn=True):
        """simclr.py:46-62: Linear(no bias) + BN + ReLU, ..., Linear(no bias) + BN(affine=False)."""
        mlp = []
        for l in range(num_layers):
            d1 = input_dim if l == 0 else mlp_dim
            d2 = output_dim if l == num_layers - 1 else mlp_dim
            mlp.append(nn.Linear(d1, d2, bias=False))
            if l < num_layers - 1:
                mlp.append(nn.BatchNorm1d(d2))
                mlp.append(nn.ReLU(inplace=True))
            elif last_bn:
                mlp.append(nn.BatchNorm1d(d2, affine=False))
        return nn.Sequential(*mlp)

    def contrastive_loss(self, aug1, aug2, batch_too_hard_neg_mask):
        """simclr.py:74-108 -> (logits [2B,2B-1], labels [2B,2B-1], loss)."""
        if ag.needs_grad(aug1, aug2):
            return ag.info_nce(aug1, aug2, batch_too_hard_neg_mask, float(self.T), precision=get_precision())
        return ops.info_nce(aug1, aug2, batch_too_hard_neg_mask, float(self.T), precision=get_precision())

    def forward(self, drug_indices, batch_mask_1, batch_mask_2, batch_too_hard_neg_mask, batch_data, batch_extra_mols=None,
                batch_extra_masks=None):
        batch_mols, batch_kg, batch_cv, batch_tx_dict = batch_data
        p1 = self.predictor if self.shared_predictor else self.predictor_1
        p2 = self.predictor if self.shared_predictor else self.predictor_2
        # both views see the same KG and the KG encoder has neither dropout nor batch statistics: one pass serves both
        # (under autograd the two views' gradients meet in one backward pass, the sum the reference forms from two)
        share = {}
        enc = self.base_encoder
        # training, raw encoder output: the per-row stages with dropout that the reference runs once per view -- the cell-viability
        # encoder and the uni-modal projector (LayerNorm: per row) -- run ONCE over both views' rows (independent masks for the two
        # halves, as two passes draw them; half the launches and one gradient per parameter).  MDG_FUSE_VIEWS=0: two passes.
        fuse = (self.raw_encoder_output and _train_path(self) and torch.is_grad_enabled() and os.environ.get("MDG_FUSE_VIEWS", "1") != "0"
                and batch_cv.is_cuda and not any(isinstance(m, nn.modules.batchnorm._BatchNorm) for mod in (enc.cv_encoder, enc.uni_projector)
                                                 for m in mod.modules()))
        if fuse:
            n = batch_cv.shape[0]
            cv_both = enc.cv_encoder(torch.cat([batch_cv, batch_cv], dim=0))
            u1 = enc(drug_indices, batch_mask_1, batch_mols, batch_kg, batch_cv, batch_tx_dict, raw_encoder_output=True, kg_share=share,
                     cv_out=cv_both[:n], defer_projector=True)
            u2 = enc(drug_indices, batch_mask_2, batch_mols, batch_kg, batch_cv, batch_tx_dict, raw_encoder_output=True, kg_share=share,
                     cv_out=cv_both[n:], defer_projector=True)
            e = enc.uni_projector(torch.cat([u1, u2], dim=0))
            e1, e2 = e[:u1.shape[0]], e[u1.shape[0]:]
        else:
            e1 = enc(drug_indices, batch_mask_1, batch_mols, batch_kg, batch_cv, batch_tx_dict,
                     raw_encoder_output=self.raw_encoder_output, kg_share=share)
            e2 = enc(drug_indices, batch_mask_2, batch_mols, batch_kg, batch_cv, batch_tx_dict,
                     raw_encoder_output=self.raw_encoder_output, kg_share=share)
        run = _run_sequential_train if (_train_path(self) or ag.needs_grad(e1, e2)) else _run_sequential
        aug_1, aug_2 = run(p1, e1), run(p2, e2)
        return aug_1, aug_2, self.contrastive_loss(aug_1, aug_2, batch_too_hard_neg_mask)
