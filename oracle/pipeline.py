"""Whole-path CPU oracle: the building blocks of oracle/madrigal_oracle.py chained the way
NovelDDIEncoder.encode + NovelDDIMultilabel.forward chain them (madrigal/models/models.py:717-953).

TEST INFRASTRUCTURE ONLY (see the header of oracle/madrigal_oracle.py)."""
from __future__ import annotations

import torch

from madrigal_amd import data as D          # containers / constants only; no arithmetic
from oracle import madrigal_oracle as O


def oracle_pipeline(case, p, batch, bkg, masks, kg_filler, label_slices=()):
    """CPU oracle of the whole path for one ENCODE_CASES-style configuration: GIN, HGT, cv MLP, chemCPA tx encoder,
    token assembly + fusion, bilinear head.  ``p`` = full state_dict (reference key names) as CPU tensors."""
    name, fusion, nb, pos, H, dh, ffn, nl, nf, agg, normalize, adapt = case
    n = int(batch["drugs"].shape[0])
    enc = O._sub(p, "encoder.")
    mols, kg = batch["strs"], bkg["data"]
    str_out = O.gin_forward(O._sub(enc, "str_encoder."), mols.node_feature, mols.edge_list, mols.edge_feature,
                            mols.node2graph, mols.batch_size, num_layers=4, num_mlp_layer=3)["graph_feature"]
    kg_valid = O.hgt_forward(O._sub(enc, "kg_encoder."), kg.x_dict, kg.edge_index_dict, kg.node_types, kg.edge_types,
                             num_layers=2, heads=4, hidden=128)["drug"]
    kg_out = O.place_kg_rows(kg_valid, bkg["drug_index_map"], batch["drugs"], kg_filler)
    cv_out = O.mlp_encoder_forward(O._sub(enc, "cv_encoder."), batch["cv"], 2, None, "relu", 0.2)
    sigs = torch.cat([batch["tx"][c]["sigs"] for c in D.CELL_LINES])
    cov = torch.arange(16).repeat_interleave(n)     # sklearn OneHotEncoder sorts categories; CELL_LINES is sorted
    _, _, _, treated = O.chemcpa_predict(O._sub(enc, "tx_encoder."), sigs, cov, 3, 3, with_decoder=False)
    all_embeds = torch.stack([str_out, kg_out, cv_out] + list(treated.split(n)), dim=1)
    if pos == "sinusoidal":
        max_len = (D.NUM_MODALITIES if nb == 0 else D.NUM_NON_TX_MODALITIES) + (1 if agg == "cls" else 0)
        enc["pos_encoder.pe"] = O.sinusoidal_pe_table(128, max_len, nb, agg)
    cfg = dict(fusion=fusion, normalize=normalize, adapt_before_fusion=adapt, pos_emb_type=pos, num_tx_bottlenecks=nb,
               agg=agg, num_layers=nl, num_heads=H, norm_first=nf, actn="gelu",
               proj=dict(n_hidden=2, norm="ln", actn="relu", dropout=0.2, order="nd"))
    z = O.fuse_modalities(enc, all_embeds, masks, cfg)
    z_raw = O.fuse_modalities(enc, all_embeds, masks, cfg, raw_encoder_output=True)
    w = p["decoder.parametrizations.weight.original"]
    out = dict(str_out=str_out, kg_out=kg_valid, cv_out=cv_out, z=z, z_raw=z_raw, scores=O.bilinear_scores(z, z, w))
    for lo, hi in label_slices:
        out[f"scores_{lo}_{hi}"] = O.bilinear_scores(z, z, w, (lo, hi))
    return out


def oracle_encoders(p, batch, bkg, kg_filler, use_tx_basal: bool = False):
    """The four modality encoders of NovelDDIEncoder.encode (models.py:717-775) -> all_embeds [n,19,128]."""
    n = int(batch["drugs"].shape[0])
    enc = O._sub(p, "encoder.")
    mols, kg = batch["strs"], bkg["data"]
    str_out = O.gin_forward(O._sub(enc, "str_encoder."), mols.node_feature, mols.edge_list, mols.edge_feature,
                            mols.node2graph, mols.batch_size, num_layers=4, num_mlp_layer=3)["graph_feature"]
    kg_valid = O.hgt_forward(O._sub(enc, "kg_encoder."), kg.x_dict, kg.edge_index_dict, kg.node_types, kg.edge_types,
                             num_layers=2, heads=4, hidden=128)["drug"]
    kg_out = O.place_kg_rows(kg_valid, bkg["drug_index_map"], batch["drugs"], kg_filler)
    cv_out = O.mlp_encoder_forward(O._sub(enc, "cv_encoder."), batch["cv"], 2, None, "relu", 0.2)
    sigs = torch.cat([batch["tx"][c]["sigs"] for c in D.CELL_LINES])
    cov = torch.arange(16).repeat_interleave(n)
    _, _, basal, treated = O.chemcpa_predict(O._sub(enc, "tx_encoder."), sigs, cov, 3, 3, with_decoder=False)
    return torch.stack([str_out, kg_out, cv_out] + list((basal if use_tx_basal else treated).split(n)), dim=1)


def oracle_simclr(p, batch, bkg, mask1, mask2, hard, temperature, kg_filler, shared_predictor=False, use_tx_basal=False,
                  normalize=False):
    """SimCLR_NovelDDI.forward with raw_encoder_output=True (madrigal/models/simclr.py:110-140 over models.py:890-894):
    per view, all four encoders -> the available (drug, modality) rows -> uni_projector -> predictor; then the InfoNCE
    loss.  ``p`` = the SimCLR state_dict (``base_encoder.*``, ``predictor_1.* predictor_2.*`` | ``predictor.*``).  The
    reference runs every encoder once per view; inside ``O.batch_statistics()`` this is the training-mode forward."""
    pe = {"encoder." + k: v for k, v in O._sub(p, "base_encoder.").items()}
    proj = dict(n_hidden=2, norm="ln", actn="relu", dropout=0.2, order="nd")
    views = []
    for masks, pred in ((mask1, "predictor." if shared_predictor else "predictor_1."), (mask2, "predictor." if shared_predictor else "predictor_2.")):
        all_embeds = oracle_encoders(pe, batch, bkg, kg_filler, use_tx_basal)
        uni = all_embeds[~masks]
        if normalize:
            uni = O.l2_normalize(uni)
        raw = O.mlp_encoder_forward(O._sub(pe, "encoder.uni_projector."), uni, proj["n_hidden"], proj["norm"], proj["actn"], proj["dropout"], proj["order"])
        views.append((raw, O.simclr_predictor(O._sub(p, pred), raw)))
    logits, labels, loss = O.info_nce(views[0][1], views[1][1], hard, temperature)
    return dict(raw1=views[0][0], raw2=views[1][0], aug1=views[0][1], aug2=views[1][1], logits=logits, labels=labels, loss=loss)
