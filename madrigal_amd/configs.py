"""The reference's shipped hyper-parameter sets that fix the kernel shapes of the path (feature_dim 128
everywhere), and a builder that instantiates the mirrored model classes from them.

Sources: configs/ddi_finetune/DrugBank/sweep_config_elated_sweep_163.yaml,
configs/ddi_finetune/TWOSIDES/sweep_config_{good_sweep_105,hardy_sweep_321}.yaml,
configs/chemcpa/chemcpa_finetune_configs.yaml, defaults from madrigal/parse_args.py:31-102.
"""
from __future__ import annotations

import copy

GIN = dict(gin_hidden_dims=[128, 128, 128], gin_edge_input_dim=18, gin_num_mlp_layer=3, gin_eps=0, gin_batch_norm=True,
           gin_actn="relu", gin_readout="mean")
HGT = dict(hgt_hidden_dim=128, hgt_num_layers=2, hgt_att_heads=4, hgt_group="sum")
CV = dict(cv_input_dim=559, cv_mlp_hidden_dims=[512, 256], cv_mlp_dropout=0.2, cv_mlp_norm=None, cv_mlp_actn="relu", cv_mlp_order="nd")
TX_CHEMCPA = {"model": {"hparams": {"dim": 128, "autoencoder_width": 512, "autoencoder_depth": 2},
                        "additional_params": {"decoder_activation": "linear", "doser_type": "amortized", "multi_task": False, "seed": 42},
                        "append_ae_layer": False, "pretrained_model_ckpt": None, "use_drugs": False}}
PROJ = dict(proj_hidden_dims=[512, 512], proj_dropout=0.2, proj_norm="ln", proj_actn="relu", proj_order="nd")


def _tf(heads, head_dim, ffn, layers, dropout, norm_first=True, agg="x-attn", actn="gelu"):
    return dict(transformer_num_layers=layers, transformer_att_heads=heads, transformer_head_dim=head_dim, transformer_ffn_dim=ffn,
                transformer_dropout=dropout, transformer_actn=actn, transformer_norm_first=norm_first,
                transformer_batch_first=False, transformer_agg=agg)


SHIPPED = {
    # name: (fusion, num_attention_bottlenecks, pos_emb_type, transformer hparams)
    "drugbank163": dict(fusion="transformer", nb=4, pos="sinusoidal", tf=_tf(8, 64, 256, 2, 0.3)),
    "twosides105": dict(fusion="transformer", nb=2, pos="learnable", tf=_tf(2, 256, 512, 2, 0.4)),
    "twosides321": dict(fusion="transformer_uni_proj", nb=2, pos="sinusoidal", tf=_tf(8, 256, 1024, 2, 0.2)),
}


def build_model(name: str, kg_data, n_outcomes: int, use_modality_pretrain: bool = False):
    """NovelDDIMultilabel(NovelDDIEncoder(...)) for one shipped configuration (random init unless the
    pre-trained unimodal checkpoints are available under ENCODER_CKPT_DIR)."""
    from . import models as M
    c = SHIPPED[name]
    enc = M.NovelDDIEncoder(all_kg_data=kg_data, feat_dim=128, str_encoder_name="gin", str_encoder_hparams=copy.deepcopy(GIN),
                            kg_encoder_name="hgt", kg_encoder_hparams=dict(HGT), cv_encoder_name="mlp", cv_encoder_hparams=copy.deepcopy(CV),
                            tx_encoder_name="chemcpa", tx_encoder_hparams=copy.deepcopy(TX_CHEMCPA), num_tx_bottlenecks=c["nb"],
                            pos_emb_dropout=0.2, transformer_fusion_hparams=dict(c["tf"]), proj_hparams=copy.deepcopy(PROJ),
                            fusion=c["fusion"], use_modality_pretrain=use_modality_pretrain, normalize=False,
                            pos_emb_type=c["pos"], adapt_before_fusion=False)
    return M.NovelDDIMultilabel(enc, 128, n_outcomes, normalize=False)
