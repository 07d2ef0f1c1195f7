"""Checkpoint interop with the reference: the dict formats its drivers write and read.

finetune   ``{"epoch", "state_dict", "encoder_configs", "model_configs"}``        train_ddi_batch.py:393-412 (best val), and every
           100 epochs in train_ddi_batch_all_train.py:343-350; read back by ``NovelDDIEncoder(**encoder_configs)`` ->
           ``NovelDDIMultilabel(encoder, **model_configs)`` -> ``load_state_dict``   madrigal/evaluate/predict.py:178-205
pretrain   ``{"epoch", "state_dict", "optimizer", "encoder_configs", "kg_args"}``  pretrain.py:230-236 (keys prefixed
           ``base_encoder.`` + the predictors)
transfer   a pretraining checkpoint seeds a finetune encoder: ``encoder_configs`` with the fusion-side entries replaced,
           the state_dict stripped of ``base_encoder.`` and of everything fusion-related       madrigal/utils.py:246-311

The state_dict key sets of the mirrored classes equal the reference's (asserted in the tests), so the ``state_dict`` entries
travel both ways as they are.  ``encoder_configs['all_kg_data']`` is a PyG ``HeteroData`` in the reference's files; here it
is accepted duck-typed (``x_dict / edge_index_dict / metadata()``) and written either as this package's ``KGData`` or as a
plain dict of tensors (``kg_format='plain'``: no class of either code base inside the file).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import data as D

# the encoder_configs entries a finetune run may override on top of a pretraining checkpoint (utils.py:263-269)
TRANSFER_OVERRIDES = ('num_tx_bottlenecks', 'pos_emb_type', 'pos_emb_dropout', 'transformer_fusion_hparams', 'proj_hparams', 'fusion',
                      'normalize', 'use_tx_basal', 'adapt_before_fusion')


def make_encoder_configs(all_kg_data, feat_dim, str_encoder_name, str_encoder_hparams, kg_encoder_name, kg_encoder_hparams, cv_encoder_name,
                         cv_encoder_hparams, tx_encoder_name, tx_encoder_hparams, num_tx_bottlenecks, pos_emb_type, pos_emb_dropout,
                         transformer_fusion_hparams, proj_hparams, fusion, str_node_feat_dim=D.MOL_DIM, use_modality_pretrain=True,
                         normalize=False, adapt_before_fusion=False, tab_mod_encoder_hparams_dict=None, use_tx_basal=None) -> dict:
    """The keyword dict ``get_model`` assembles and every checkpoint carries (madrigal/utils.py:222-244): exactly the arguments
    of ``NovelDDIEncoder.__init__``."""
    cfg = {'all_kg_data': all_kg_data, 'feat_dim': feat_dim, 'str_encoder_name': str_encoder_name, 'str_encoder_hparams': str_encoder_hparams,
           'kg_encoder_name': kg_encoder_name, 'kg_encoder_hparams': kg_encoder_hparams, 'cv_encoder_name': cv_encoder_name,
           'cv_encoder_hparams': cv_encoder_hparams, 'tx_encoder_name': tx_encoder_name, 'tx_encoder_hparams': tx_encoder_hparams,
           'num_tx_bottlenecks': num_tx_bottlenecks, 'pos_emb_type': pos_emb_type, 'pos_emb_dropout': pos_emb_dropout,
           'transformer_fusion_hparams': transformer_fusion_hparams, 'proj_hparams': proj_hparams, 'fusion': fusion,
           'str_node_feat_dim': str_node_feat_dim, 'use_modality_pretrain': use_modality_pretrain, 'normalize': normalize,
           'adapt_before_fusion': adapt_before_fusion, 'tab_mod_encoder_hparams_dict': tab_mod_encoder_hparams_dict}
    if use_tx_basal is not None:
        cfg['use_tx_basal'] = use_tx_basal
    return cfg


def make_model_configs(feat_dim, prediction_dim, normalize=False, use_single_drug=False) -> dict:
    """madrigal/utils.py:335-341."""
    return {'feat_dim': feat_dim, 'prediction_dim': prediction_dim, 'normalize': normalize, 'use_single_drug': use_single_drug}


def _portable_configs(encoder_configs: dict, kg_format: str) -> dict:
    cfg = dict(encoder_configs)
    kg = cfg.get('all_kg_data')
    if kg is not None:
        if kg_format == 'plain':
            cfg['all_kg_data'] = D.kg_to_plain(kg)
        elif kg_format == 'object':
            cfg['all_kg_data'] = D.as_kg_data(kg).to('cpu')
        else:
            raise ValueError(f"kg_format must be 'object' or 'plain', got {kg_format!r}")
    return cfg


def _cpu_state_dict(module) -> dict:
    return {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}


def save_finetune_checkpoint(path: str, model, epoch: int, encoder_configs: dict, model_configs: dict, kg_format: str = 'object') -> dict:
    """train_ddi_batch.py:393-412 / train_ddi_batch_all_train.py:343-350."""
    ckpt = {"epoch": epoch, "state_dict": _cpu_state_dict(model), "encoder_configs": _portable_configs(encoder_configs, kg_format),
            "model_configs": dict(model_configs)}
    torch.save(ckpt, path)
    return ckpt


def save_pretrain_checkpoint(path: str, model, optimizer, epoch: int, encoder_configs: dict, kg_args=None, kg_format: str = 'object') -> dict:
    """pretrain.py:230-236 (``utils.save_checkpoint``): the SimCLR wrapper's state (``base_encoder.*`` + predictors), the
    optimizer state (resume, pretrain.py:185-194) and the encoder's constructor arguments."""
    ckpt = {"epoch": epoch, "state_dict": _cpu_state_dict(model), "optimizer": optimizer.state_dict(),
            "encoder_configs": _portable_configs(encoder_configs, kg_format), "kg_args": kg_args}
    torch.save(ckpt, path)
    return ckpt


def _read(path_or_dict, map_location='cpu') -> dict:
    if isinstance(path_or_dict, dict):
        return path_or_dict
    # the configs hold python objects (the KG container, hyper-parameter dicts): a pickle, as in the reference
    return torch.load(path_or_dict, map_location=map_location, weights_only=False)


def _encoder_from_configs(encoder_configs: dict):
    from .models import NovelDDIEncoder
    cfg = dict(encoder_configs)
    cfg['all_kg_data'] = D.as_kg_data(cfg['all_kg_data'])
    cfg.pop('finetune_mode', None)
    if cfg.get('tab_mod_encoder_hparams_dict') is None:
        cfg.pop('tab_mod_encoder_hparams_dict', None)
    return NovelDDIEncoder(**cfg), cfg


def load_finetune_checkpoint(path_or_dict, device=None, strict: bool = True):
    """madrigal/evaluate/predict.py:178-205 -> ``(model, checkpoint, incompatible_keys)``: rebuild encoder and model from the stored configs and
    load the weights.  ``use_modality_pretrain`` is switched off for the rebuild (the stored weights replace the
    unimodal ones the constructor would fetch from ENCODER_CKPT_DIR)."""
    from .models import NovelDDIMultilabel
    ckpt = _read(path_or_dict)
    enc_cfg = dict(ckpt['encoder_configs'])
    enc_cfg['use_modality_pretrain'] = False
    encoder, _ = _encoder_from_configs(enc_cfg)
    model = NovelDDIMultilabel(encoder, **ckpt['model_configs'])
    msg = model.load_state_dict(ckpt['state_dict'], strict=strict)
    if device is not None:
        model = model.to(device)
    return model, ckpt, msg


def filter_pretrained_state_dict(state_dict: dict, use_pretrained_adaptor: bool = True) -> dict:
    """madrigal/utils.py:281-295: what a finetune encoder takes over from a contrastive checkpoint -- the four modality
    encoders (and, optionally, the uni-modal projector), with the ``base_encoder.`` prefix removed.  Everything
    fusion-related (transformer, position encoding, learned tokens, a ``head``) and the predictors stay behind."""
    out = {}
    for k, v in state_dict.items():
        if not k.startswith('base_encoder'):
            continue
        if k.startswith('base_encoder.head') or k in ('base_encoder.tx_bottleneck_tokens', 'base_encoder.cls'):
            continue
        if k.startswith('base_encoder.pos_encoder') or k.startswith('base_encoder.transformer'):
            continue
        if not use_pretrained_adaptor and k.startswith('base_encoder.uni_projector'):
            continue
        out[k[len('base_encoder.'):]] = v
    return out


def load_pretrained_encoder(path_or_dict, overrides: Optional[dict] = None, use_pretrained_adaptor: bool = True, device=None):
    """madrigal/utils.py:246-311 -> ``(encoder, encoder_configs, incompatible_keys)``: the finetune encoder seeded from a
    contrastive-pretraining checkpoint.  ``overrides`` may replace the fusion-side constructor arguments
    (``TRANSFER_OVERRIDES``; None values keep the checkpoint's)."""
    ckpt = _read(path_or_dict)
    cfg = dict(ckpt['encoder_configs'])
    for k, v in (overrides or {}).items():
        if k not in TRANSFER_OVERRIDES:
            raise KeyError(f"{k!r} cannot be overridden on a pretrained encoder (allowed: {TRANSFER_OVERRIDES})")
        if v is not None:
            cfg[k] = v
    # utils.py:281-295 filters and renames UNCONDITIONALLY (whatever other entries the checkpoint holds)
    state = filter_pretrained_state_dict(ckpt['state_dict'], use_pretrained_adaptor)
    if not state:
        raise ValueError("load_pretrained_encoder: no 'base_encoder.*' modality-encoder weights in this checkpoint -- not a contrastive-"
                         "pretraining checkpoint (the reference would silently return a randomly initialised encoder here)")
    # The reference builds the encoder with the checkpoint's own `use_modality_pretrain` (its constructor then fetches the unimodal
    # files from ENCODER_CKPT_DIR) and overwrites those weights with the state dict.  When the state dict covers all four
    # modality encoders the fetch changes nothing, so it is skipped; the returned configs keep the checkpoint's value either way.
    wanted = bool(cfg.get('use_modality_pretrain', True))
    covered = all(any(k.startswith(m + '.') for k in state) for m in ('str_encoder', 'kg_encoder', 'cv_encoder', 'tx_encoder'))
    build_cfg = dict(cfg)
    build_cfg['use_modality_pretrain'] = wanted and not covered
    encoder, _ = _encoder_from_configs(build_cfg)
    msg = encoder.load_state_dict(state, strict=False)
    if device is not None:
        encoder = encoder.to(device)
    return encoder, cfg, msg
