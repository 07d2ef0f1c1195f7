#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own classes on CPU.

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (needs /root/reference;
the GPU box never sees it).  Usage:  python oracle/gen_golden.py [--ref /root/reference]

How the reference is imported: ``madrigal.chemcpa.chemCPA.model`` imports as is;
``madrigal.models.models`` / ``.simclr`` import after inert ``sys.modules`` entries are
registered for packages absent from the image (torch_geometric, torch_scatter,
torchdrug, dotenv, jsonpickle).  Those entries carry no arithmetic of their own except
two stand-in layer classes (GraphIsomorphismNetwork, HGTConv) that wrap THIS repo's
oracle restatement of the two un-vendored third-party layers -- so fixtures that pass
through them pin the reference's glue (token assembly, masks, fusion, head), not the
third-party layers themselves (those stay "parity unpinned", see the oracle header).

Fixtures hold inputs / seeds / expected outputs only.  Weights are regenerated from
(seed, name, shape) by oracle/params.py on both sides.
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import madrigal_oracle as O          # noqa: E402
from oracle.params import det_input, fill_module  # noqa: E402
from madrigal_amd import data as D               # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


# ----------------------------------------------------------------------------- stand-ins
class _StandInGIN(nn.Module):
    """Parameter layout of torchdrug GraphIsomorphismNetwork (per GIN_256x4_muv.pt);
    arithmetic = oracle.gin_forward."""

    def __init__(self, input_dim, hidden_dims, edge_input_dim, num_mlp_layer, eps, batch_norm, activation,
                 readout, **kw):
        super().__init__()
        dims = [input_dim] + list(hidden_dims)
        self.cfg = dict(num_layers=len(dims) - 1, num_mlp_layer=num_mlp_layer, batch_norm=batch_norm, readout=readout)
        self.layers = nn.ModuleList()
        for i in range(len(dims) - 1):
            layer = nn.Module()
            layer.register_buffer("eps", torch.tensor([float(eps)]))
            if batch_norm:
                layer.batch_norm = nn.BatchNorm1d(dims[i + 1])
            layer.mlp = nn.Module()
            md = [dims[i]] + [dims[i + 1]] * num_mlp_layer
            layer.mlp.layers = nn.ModuleList([nn.Linear(md[j], md[j + 1]) for j in range(num_mlp_layer)])
            layer.edge_linear = nn.Linear(edge_input_dim, dims[i])
            self.layers.append(layer)

    def forward(self, graph, node_input):
        return O.gin_forward(dict(self.state_dict()), node_input, graph.edge_list, graph.edge_feature,
                             graph.node2graph, graph.batch_size, edge_weight=graph.edge_weight, **self.cfg)


class _StandInHGTConv(nn.Module):
    """Parameter layout of PyG 2.3 HGTConv; arithmetic = oracle.hgt_conv_forward."""

    def __init__(self, in_channels, out_channels, metadata, heads=1, group="sum", **kw):
        super().__init__()
        self.node_types, self.edge_types = list(metadata[0]), [tuple(e) for e in metadata[1]]
        self.heads, self.out_channels = heads, out_channels
        D_ = out_channels // heads
        self.kqv_lin = nn.Module()
        self.kqv_lin.lins = nn.ModuleDict({t: nn.Linear(in_channels, 3 * out_channels) for t in self.node_types})
        self.out_lin = nn.Module()
        self.out_lin.lins = nn.ModuleDict({t: nn.Linear(out_channels, out_channels) for t in self.node_types})
        self.k_rel = nn.Module()
        self.k_rel.weight = nn.Parameter(torch.randn(heads * len(self.edge_types), D_, D_))
        self.v_rel = nn.Module()
        self.v_rel.weight = nn.Parameter(torch.randn(heads * len(self.edge_types), D_, D_))
        self.skip = nn.ParameterDict({t: nn.Parameter(torch.ones(1)) for t in self.node_types})
        self.p_rel = nn.ParameterDict({"__".join(e): nn.Parameter(torch.ones(1, heads)) for e in self.edge_types})

    def forward(self, x_dict, edge_index_dict):
        return O.hgt_conv_forward(dict(self.state_dict()), x_dict, edge_index_dict, self.node_types,
                                  self.edge_types, self.heads, self.out_channels)


def import_reference(ref_root: str):
    def shim(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Absent:
        def __init__(self, *a, **k):
            raise RuntimeError("third-party layer absent from this image")

    def scatter_mean(src, index, dim=0):
        n = int(index.max()) + 1
        s = torch.zeros(n, src.shape[1]).index_add_(0, index, src)
        c = torch.zeros(n).index_add_(0, index, torch.ones(index.shape[0]))
        return s / c.clamp_min(1).unsqueeze(-1)

    def scatter_add(src, index, dim=0):
        return torch.zeros(int(index.max()) + 1, src.shape[1]).index_add_(0, index, src)

    def scatter_max(src, index, dim=0):
        n = int(index.max()) + 1
        out = torch.full((n, src.shape[1]), float("-inf")).scatter_reduce(
            0, index.view(-1, 1).expand(-1, src.shape[1]), src, reduce="amax", include_self=True)
        return out, None

    tg = shim("torch_geometric")
    tg.nn = shim("torch_geometric.nn", HANConv=_Absent, RGCNConv=_Absent, HGTConv=_StandInHGTConv,
                 HeteroLinear=_Absent)
    tg.data = shim("torch_geometric.data", HeteroData=D.KGData, Data=_Absent)
    shim("torch_scatter", scatter_mean=scatter_mean, scatter_add=scatter_add, scatter_max=scatter_max)
    td = shim("torchdrug")
    td.models = shim("torchdrug.models", GraphIsomorphismNetwork=_StandInGIN, GraphAttentionNetwork=_Absent)
    td.data = shim("torchdrug.data", PackedMolecule=D.MoleculeBatch)
    shim("dotenv", load_dotenv=lambda *a, **k: None)
    shim("jsonpickle")
    sys.path.insert(0, ref_root)
    import madrigal.models.models as M
    import madrigal.models.simclr as S
    import madrigal.chemcpa.chemCPA.model as C
    return M, S, C


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"{name:28s} {os.path.getsize(path) / 1024:8.1f} KiB")


# ----------------------------------------------------------------------------- fixtures
FUSION_CASES = [
    # name, heads, head_dim, ffn, layers, norm_first, agg, nb, actn
    ("drugbank163", 8, 64, 256, 2, True, "x-attn", 4, "gelu"),
    ("twosides105", 2, 256, 512, 2, True, "x-attn", 2, "gelu"),
    ("twosides321", 8, 256, 1024, 2, True, "x-attn", 2, "gelu"),
    ("cl_default", 4, 128, 512, 3, False, "x-attn", 0, "gelu"),
    ("cls_small", 4, 32, 64, 1, True, "cls", 2, "relu"),
    ("mean_small", 4, 32, 64, 2, False, "mean", 0, "gelu"),
    ("max_small", 2, 64, 128, 1, True, "max", 2, "gelu"),
]


def fusion_masks(n, nb, agg, seed):
    masks = D.make_masks(n, seed, p_kg=0.6, p_cv=0.5, p_tx=0.3)
    masks[0, 1:] = True                      # structure only
    masks[1, :] = False                      # everything present
    return masks


def gen_head(M):
    torch.manual_seed(0)
    dec = M.BilinearDDIScorer(128, 128, 5)
    nn.utils.parametrize.register_parametrization(dec, "weight", M.Symmetric())
    w = det_input(11, "head.W", (5, 128, 128), 1 / math.sqrt(128))
    dec.parametrizations.weight.original.data.copy_(w)
    zh, zt = det_input(11, "head.zh", (24, 128)), det_input(11, "head.zt", (17, 128))
    with torch.no_grad():
        save("head", z_head=zh, z_tail=zt, w_original=w, scores=dec(zh, zt), scores_1_4=dec(zh, zt, (1, 4)),
             w_sym=dec.weight, scores_self=dec(zh, zh))


def gen_mlps(M):
    cases = {
        "cv": dict(cls=M.MLPEncoder, in_dim=559, hidden=[512, 256], out=128, p=0.2, norm=None, actn="relu", order="nd"),
        "proj": dict(cls=M.MLPAdaptor, in_dim=128, hidden=[512, 512], out=128, p=0.2, norm="ln", actn="relu", order="nd"),
        "bn_dn": dict(cls=M.MLPEncoder, in_dim=40, hidden=[64, 48, 32], out=16, p=0.1, norm="bn", actn="gelu", order="dn"),
        "one_hidden": dict(cls=M.MLPAdaptor, in_dim=32, hidden=[64], out=8, p=0.0, norm="ln", actn="tanh", order="nd"),
    }
    out = {}
    for name, c in cases.items():
        m = c["cls"](c["in_dim"], c["hidden"], c["out"], c["p"], c["norm"], c["actn"], c["order"]).eval()
        fill_module(m, 21)
        x = det_input(21, "mlp." + name, (9, c["in_dim"]))
        with torch.no_grad():
            out[name + "_x"], out[name + "_y"] = x, m(x)
        out[name + "_keys"] = np.array(sorted(m.state_dict().keys()))
    save("mlps", **out)


def gen_posenc(M):
    out = {}
    for nb, agg in [(0, "x-attn"), (4, "x-attn"), (2, "cls")]:
        max_len = (D.NUM_MODALITIES if nb == 0 else D.NUM_NON_TX_MODALITIES) + (1 if agg == "cls" else 0)
        S = D.NUM_MODALITIES + nb + (1 if agg == "cls" else 0)
        x = det_input(31, f"pe.{nb}.{agg}", (3, S, 128))
        sin = M.PositionEncodingSinusoidal(128, 0.2, max_len, nb, agg).eval()
        lrn = M.PositionEncodingLearnable(128, 0.2, max_len, nb, agg).eval()
        fill_module(lrn, 31)
        with torch.no_grad():
            out[f"sin_{nb}_{agg}_pe"], out[f"sin_{nb}_{agg}_y"] = sin.pe, sin(x.clone())
            out[f"lrn_{nb}_{agg}_y"] = lrn(x.clone())
        out[f"x_{nb}_{agg}"] = x
    save("posenc", **out)


def gen_fusion(M):
    for name, H, dh, ffn, nl, nf, agg, nb, actn in FUSION_CASES:
        m = M.TransformerFusion(128, nb, nl, H, dh, ffn, 0.3, actn, nf, False, agg).eval()
        fill_module(m, 41)
        n = 6
        masks = fusion_masks(n, nb, agg, 41)
        emb = det_input(41, "fusion." + name, (n, D.NUM_MODALITIES, 128))
        bt = det_input(41, "fusion.bt." + name, (nb, 128)) if nb else None
        cls = det_input(41, "fusion.cls." + name, (1, 128)) if agg == "cls" else None
        seq, kpm, src = O.assemble_fusion_inputs(emb, masks, bt, cls)
        captured = {}
        hook = m.transformer_encoder.layers[-1].self_attn.register_forward_hook(
            lambda mod, inp, outp: captured.__setitem__("w", outp[1]))
        with torch.no_grad():
            y = m(seq, kpm, src)
        hook.remove()
        save("fusion_" + name, seq=seq, kpm=kpm, src=np.zeros((0,)) if src is None else src, out=y,
             attn_last=captured["w"], cfg=np.array([H, dh, ffn, nl, int(nf), nb]), agg=np.array(agg),
             actn=np.array(actn))


def make_chemcpa(C, seed):
    hp = {"dim": 128, "autoencoder_width": 512, "autoencoder_depth": 2, "autoencoder_lr": 1e-3, "autoencoder_wd": 1e-6,
          "adversary_width": 128, "adversary_depth": 2, "adversary_lr": 3e-4, "adversary_wd": 1e-4,
          "dosers_width": 64, "dosers_depth": 2, "dosers_lr": 1e-3, "dosers_wd": 1e-7, "step_size_lr": 45,
          "embedding_encoder_width": 512, "embedding_encoder_depth": 0, "reg_adversary": 5, "penalty_adversary": 3,
          "adversary_steps": 3, "batch_size": 128}
    m = C.TxAdaptingComPert(num_genes=978, num_drugs=50, covariate_names_unique={"cell_iname": [c.upper() for c in D.CELL_LINES]},
                            hparams=hp, drug_embeddings=None, append_layer_width=None, use_drugs=False,
                            disable_adv=True, doser_type="logsigm", decoder_activation="linear").eval()
    fill_module(m, seed)
    return m


def gen_chemcpa(C):
    m = make_chemcpa(C, 51)
    n = 5
    genes = det_input(51, "cpa.genes", (16 * n, 978))
    genes[::7] = 0.0
    cov = torch.arange(16).repeat_interleave(n)
    onehot = torch.nn.functional.one_hot(cov, 16).long()
    with torch.no_grad():
        rec, emb, basal, treated = m.predict(genes=genes, drugs_idx=torch.zeros(16 * n, dtype=torch.long),
                                             dosages=torch.ones(16 * n), covariates=[onehot],
                                             return_latent_basal=True, return_latent_treated=True)
    save("chemcpa", genes=genes, cov_idx=cov, recon=rec, cell_emb=emb, basal=basal, treated=treated,
         keys=np.array(sorted(m.state_dict().keys())))


ENCODE_CASES = [
    # name, fusion, nb, pos, heads, head_dim, ffn, layers, norm_first, agg, normalize, adapt
    ("drugbank163", "transformer", 4, "sinusoidal", 8, 64, 256, 2, True, "x-attn", False, False),
    ("uniproj", "transformer_uni_proj", 2, "learnable", 2, 64, 128, 2, True, "x-attn", True, False),
    ("cls_adapt", "transformer", 2, "learnable", 4, 32, 64, 1, False, "cls", False, True),
    ("meanfuse", "mean", 0, "learnable", 4, 32, 64, 1, False, "x-attn", True, False),
]


def build_reference_model(M, C, kg, case, L, seed, **enc_kwargs):
    name, fusion, nb, pos, H, dh, ffn, nl, nf, agg, normalize, adapt = case
    enc = M.NovelDDIEncoder(
        all_kg_data=kg, feat_dim=128, str_encoder_name="gin",
        str_encoder_hparams=dict(gin_hidden_dims=[128, 128, 128], gin_edge_input_dim=18, gin_num_mlp_layer=3, gin_eps=0,
                                 gin_batch_norm=True, gin_actn="relu", gin_readout="mean"),
        kg_encoder_name="hgt", kg_encoder_hparams=dict(hgt_hidden_dim=128, hgt_num_layers=2, hgt_att_heads=4, hgt_group="sum"),
        cv_encoder_name="mlp", cv_encoder_hparams=dict(cv_input_dim=559, cv_mlp_hidden_dims=[512, 256], cv_mlp_dropout=0.2,
                                                      cv_mlp_norm=None, cv_mlp_actn="relu", cv_mlp_order="nd"),
        tx_encoder_name="mlp", tx_encoder_hparams=dict(tx_input_dim=978, tx_mlp_hidden_dims=[8], tx_mlp_dropout=0.0,
                                                      tx_mlp_norm=None, tx_mlp_actn="relu", tx_mlp_order="nd"),
        num_tx_bottlenecks=nb, pos_emb_dropout=0.2,
        transformer_fusion_hparams=dict(transformer_num_layers=nl, transformer_att_heads=H, transformer_head_dim=dh,
                                        transformer_ffn_dim=ffn, transformer_dropout=0.3, transformer_actn="gelu",
                                        transformer_norm_first=nf, transformer_batch_first=False, transformer_agg=agg),
        proj_hparams=dict(proj_hidden_dims=[512, 512], proj_dropout=0.2, proj_norm="ln", proj_actn="relu", proj_order="nd"),
        fusion=fusion, use_modality_pretrain=False, normalize=normalize, pos_emb_type=pos, adapt_before_fusion=adapt,
        **enc_kwargs)
    # the shipped runs use the chemCPA tx encoder; its constructor path inside NovelDDIEncoder reads two
    # external data files (models.py:271-273), so attach the reference's own TxAdaptingComPert directly.
    from sklearn.preprocessing import OneHotEncoder
    enc.tx_encoder_dict = None
    enc.tx_encoder = make_chemcpa(C, seed)
    enc.tx_cell_line_onehot_encoder = OneHotEncoder(sparse_output=False)
    enc.tx_cell_line_onehot_encoder.fit(np.array([c.lower() for c in D.CELL_LINES]).reshape(-1, 1))
    model = M.NovelDDIMultilabel(enc, 128, L, normalize=False).eval()
    skip = [k for k in model.state_dict() if k.endswith("pos_encoder.pe") and pos == "sinusoidal"]
    fill_module(model, seed, skip=skip)
    return model


def gen_encode(M, C):
    n, L, seed = 14, 6, 61
    for case in ENCODE_CASES:
        masks = D.make_masks(n, seed, p_kg=0.6, p_cv=0.5, p_tx=0.25)
        masks[0, 1:] = True
        masks[1, 1:] = True
        masks[1, 0] = False
        masks[2, :] = False
        masks[3, 1:] = True
        batch, bkg = D.make_batch(n, seed, kg_nodes=300, kg_edges=2500, masks=masks)
        model = build_reference_model(M, C, bkg["data"], case, L, seed)
        drugs = batch["drugs"]
        filler_shape = (max(int(drugs.max()) + 1, int(bkg["drug_index_map"].max()) + 1), 128)
        with torch.no_grad():
            torch.manual_seed(seed)
            filler = torch.randn(filler_shape)
            torch.manual_seed(seed)
            z = model.encoder(drugs, batch["masks"], batch["strs"], bkg, batch["cv"], batch["tx"])
            torch.manual_seed(seed)
            z_raw = model.encoder(drugs, batch["masks"], batch["strs"], bkg, batch["cv"], batch["tx"],
                                  raw_encoder_output=True)
            torch.manual_seed(seed)
            scores = model(batch, batch, batch["masks"], batch["masks"], bkg)
            scores_rng = model.decoder(z, z, (2, 5))
            str_out = model.encoder.str_encoder(batch["strs"], batch["strs"].node_feature.float())["graph_feature"]
            kg_out = model.encoder.kg_encoder(bkg["data"].x_dict, bkg["data"].edge_index_dict)["drug"]
            cv_out = model.encoder.cv_encoder(batch["cv"])
        save("encode_" + case[0], masks=batch["masks"], z=z, z_raw=z_raw, scores=scores, scores_2_5=scores_rng,
             kg_filler=filler, str_out=str_out, kg_out=kg_out, cv_out=cv_out,
             meta=np.array([n, L, seed]), keys=np.array(sorted(model.state_dict().keys())))


def gen_infonce(M, S):
    enc = types.SimpleNamespace(uni_projector=types.SimpleNamespace(fc=[nn.Linear(512, 128)]))
    B = 24
    a1, a2 = det_input(71, "nce.a1", (B, 128)), det_input(71, "nce.a2", (B, 128))
    hard = torch.from_numpy(np.random.default_rng(71).random((B, B)) < 0.1)
    hard = hard & ~torch.eye(B, dtype=torch.bool)
    sim = S.SimCLR_NovelDDI.__new__(S.SimCLR_NovelDDI)
    nn.Module.__init__(sim)
    sim.T = 0.1
    with torch.no_grad():
        lg, lb, loss = sim.contrastive_loss(a1, a2, hard.clone())
        lg0, lb0, loss0 = sim.contrastive_loss(a1, a2, None)
    model = S.SimCLR_NovelDDI(enc, dim=128, mlp_dim=512, T=0.1).eval()
    fill_module(model.predictor_1, 71)
    x = det_input(71, "nce.pred", (B, 128))
    with torch.no_grad():
        pred = model.predictor_1(x)
    save("infonce", aug1=a1, aug2=a2, hard=hard, logits=lg, labels=lb, loss=loss, logits_nomask=lg0, loss_nomask=loss0,
         T=np.array(0.1), pred_x=x, pred_y=pred, pred_keys=np.array(sorted(model.predictor_1.state_dict().keys())))


from oracle.gen_cases import CL_CASE             # noqa: E402


def gen_pretrain_views():
    """Host logic of pretrain.py:59-71: the reference's own get_pretrain_masks / pretrain_modality_subset_sampler
    (madrigal/utils.py:51-145, 360-390) on a seeded availability table, for every mode that constructs."""
    import madrigal.utils as U
    avail = D.make_masks(60, 7, p_kg=0.7, p_cv=0.5, p_tx=0.2).numpy().astype(np.int64)
    avail[:, 1] = np.where(avail[:, 1:].all(axis=1), 0, avail[:, 1])          # every drug owns a second modality
    drugs = list(range(100, 160))
    out = {"avail": avail, "drugs": np.array(drugs)}
    for mode in ("str_center_uni", "double_random", "str_kg", "str_center", "str_center_comb"):
        for unb in (False, True):
            tag = f"{mode}_{int(unb)}"
            try:
                bank = U.get_pretrain_masks(drugs, avail.copy(), mode, unb, 0.2)
            except Exception as e:                                             # modes whose bank the reference cannot file
                out[tag + "_error"] = np.array(type(e).__name__)
                continue
            np.random.seed(123)
            torch.manual_seed(123)
            order = [drugs[i] for i in np.random.default_rng(3).permutation(len(drugs))[:32]]
            a1, a2 = U.pretrain_modality_subset_sampler([bank[d] for d in order], pretrain_mode=mode, unbalanced=unb)
            b1, b2 = U.pretrain_modality_subset_sampler([bank[d] for d in order], pretrain_mode=mode, unbalanced=unb)
            out[tag + "_order"] = np.array(order)
            out[tag + "_aug1"], out[tag + "_aug2"], out[tag + "_aug1b"], out[tag + "_aug2b"] = a1, a2, b1, b2
            if mode == "str_center_uni" and not unb:
                out["uni_probs_d100"] = np.asarray(bank[100][1])
    save("pretrain_views", **out)


def gen_simclr_raw(M, S, C):
    """BASELINE configs[2] as the reference ships it: SimCLR_NovelDDI.forward (simclr.py:110-140) with
    raw_encoder_output=True (encoders -> uni_projector only, models.py:890-894) on views drawn by the reference's own
    'str_center_uni' sampler; eval mode, both predictor layouts, both tx latents (use_tx_basal)."""
    import madrigal.utils as U
    n, seed = 40, 71
    avail = D.make_masks(n, seed, p_kg=0.6, p_cv=0.5, p_tx=0.25)
    avail[:, 1] = torch.where(avail[:, 1:].all(dim=1), torch.zeros(n, dtype=torch.bool), avail[:, 1])
    batch, bkg = D.make_batch(n, seed, kg_nodes=300, kg_edges=2500, masks=avail)
    drugs = batch["drugs"]
    bank = U.get_pretrain_masks(drugs.tolist(), avail.numpy().astype(np.int64), "str_center_uni", False, 0.2)
    np.random.seed(seed)
    m1, m2 = U.pretrain_modality_subset_sampler([bank[d] for d in drugs.tolist()], pretrain_mode="str_center_uni", unbalanced=False)
    hard = torch.from_numpy(np.random.default_rng(seed).random((n, n)) < 0.05)
    hard = (hard | hard.T) & ~torch.eye(n, dtype=torch.bool)
    filler_shape = (max(int(drugs.max()) + 1, int(bkg["drug_index_map"].max()) + 1), 128)
    torch.manual_seed(seed)
    filler = torch.randn(filler_shape)
    out = dict(avail=avail, mask1=m1, mask2=m2, hard=hard, kg_filler=filler, meta=np.array([n, seed]), T=np.array(0.1))
    for shared in (False, True):
        for basal in (False, True):
            enc = build_reference_model(M, C, bkg["data"], CL_CASE, 4, seed, use_tx_basal=basal).encoder
            model = S.SimCLR_NovelDDI(enc, dim=128, mlp_dim=512, T=0.1, raw_encoder_output=True, shared_predictor=shared).eval()
            fill_module(model, seed, skip=[])
            tag = f"s{int(shared)}b{int(basal)}"
            with torch.no_grad():
                torch.manual_seed(seed)
                a1, a2, (lg, lb, loss) = model(drugs, m1, m2, hard.clone(), (batch["strs"], bkg, batch["cv"], batch["tx"]), None, None)
                torch.manual_seed(seed)
                raw1 = model.base_encoder(drugs, m1, batch["strs"], bkg, batch["cv"], batch["tx"], raw_encoder_output=True)
                torch.manual_seed(seed)
                raw2 = model.base_encoder(drugs, m2, batch["strs"], bkg, batch["cv"], batch["tx"], raw_encoder_output=True)
            out.update({f"{tag}_aug1": a1, f"{tag}_aug2": a2, f"{tag}_logits": lg, f"{tag}_loss": loss, f"{tag}_raw1": raw1, f"{tag}_raw2": raw2})
            out[f"{tag}_keys"] = np.array(sorted(model.state_dict().keys()))
    save("simclr_raw", **out)


def gen_checkpoint_fixtures(ref_root, M, S, C):
    """(i) The CL -> finetune key filter: the reference's own loop (madrigal/utils.py:281-295 -- inline code of get_model, taken
    from the file text at generation time like gen_ranks; nothing of it is stored) applied to the key list of its own
    SimCLR_NovelDDI state_dict (+ the optional entries the loop names).  (ii) Parameter names and shapes -- no values -- of the
    two weight files the reference ships (modality_pretraining/str/GIN_256x4_muv.pt, cv/cv_model_ae.pt)."""
    import textwrap
    src = open(os.path.join(ref_root, "madrigal", "utils.py")).read().splitlines()
    start = next(i for i, l in enumerate(src) if l.strip() == "for k in list(state_dict.keys()):")
    end = next(i for i in range(start, len(src)) if src[i].strip() == "del state_dict[k]")
    loop = textwrap.dedent("\n".join(src[start:end + 1]))
    batch, bkg = D.make_batch(12, 3, kg_nodes=200, kg_edges=900)
    out = {}
    for shared in (False, True):
        enc = build_reference_model(M, C, bkg["data"], CL_CASE, 4, 3).encoder
        model = S.SimCLR_NovelDDI(enc, dim=128, mlp_dim=512, T=0.1, raw_encoder_output=True, shared_predictor=shared)
        keys = list(model.state_dict().keys()) + ["base_encoder.cls", "base_encoder.head.weight", "base_encoder.head.bias"]
        for adaptor in (False, True):
            ns = {"state_dict": {k: k for k in keys}, "use_pretrained_adaptor": adaptor}
            exec(loop, ns)
            out[f"s{int(shared)}a{int(adaptor)}_out"] = np.array(sorted(ns["state_dict"].keys()))
            out[f"s{int(shared)}a{int(adaptor)}_src"] = np.array([ns["state_dict"][k] for k in sorted(ns["state_dict"].keys())])
        out[f"s{int(shared)}_in"] = np.array(keys)
    save("ckpt_filter", **out)
    lay = {}
    for name, rel in (("gin", "modality_pretraining/str/GIN_256x4_muv.pt"), ("cv", "modality_pretraining/cv/cv_model_ae.pt")):
        sd = torch.load(os.path.join(ref_root, rel), map_location="cpu", weights_only=True)
        lay[name + "_keys"] = np.array(list(sd.keys()))
        lay[name + "_shapes"] = np.array([",".join(str(d) for d in v.shape) for v in sd.values()])
        lay[name + "_dtypes"] = np.array([str(v.dtype) for v in sd.values()])
    save("pretrained_layouts", **lay)


def gen_ranks(ref_root):
    # notebooks/normalize_scores.py runs file I/O at import (:26); take the pure-numpy function from the
    # file text at generation time (nothing of it is stored).
    src = open(os.path.join(ref_root, "notebooks", "normalize_scores.py")).read().splitlines()
    start = next(i for i, l in enumerate(src) if l.startswith("def classwise_normalized_rank_3d_numpy"))
    end = next(i for i, l in enumerate(src) if l.startswith("def run_slice"))
    ns = {"np": np}
    exec("\n".join(src[start:end]), ns)
    fn = ns["classwise_normalized_rank_3d_numpy"]
    rng = np.random.default_rng(81)
    N, L = 37, 3
    scores = rng.standard_normal((L, N, N)).astype(np.float32)
    iu = np.vstack(np.triu_indices(N, k=0, m=N))
    out = np.zeros_like(scores)
    for l in range(L):                       # run_slice (:62-74), one outcome at a time
        s = scores[l:l + 1].copy()
        s[:, iu[0], iu[1]] = 1e7
        r = fn(s)
        r[:, iu[0], iu[1]] = 0
        out[l:l + 1] = r + r.swapaxes(1, 2)
    save("ranks", scores=scores, normalized=out)


def gen_bce():
    rng = np.random.default_rng(91)
    L, N = 7, 19
    S_ = torch.from_numpy(rng.standard_normal((L, N, N)).astype(np.float32) * 3)
    lab, h, t, y = D.make_labelled_triples(N, L, 40, 91)
    p = torch.sigmoid(S_)[lab, h, t]
    loss = torch.nn.BCELoss()(p, y)
    save("bce", scores=S_, labels=lab, heads=h, tails=t, y=y, pred=p, loss=loss)


def gen_eval_masks_and_schedule():
    """Harness-side host logic: the reference's own get_evaluate_masks (eval_utils.py:287-305) on a seeded
    availability table for every evaluation type x finetune mode the training script uses, and the learning rates of its
    LinearWarmupCosineDecaySchedule (utils.py:665-680)."""
    sys.modules.setdefault("umap", types.ModuleType("umap")).UMAP = object
    import madrigal.evaluate.eval_utils as EU
    import madrigal.utils as U
    base_h, base_t = D.make_masks(40, 5), D.make_masks(40, 6)
    eval_types = ["full_full", "str_str", "str_full", "kg_kg", "cv_cv", "tx_tx", "str+tx_full", "str+kg_str+cv", "str+kg+cv_tx",
                  "full_str+cv+tx", "kg+tx_cv"]
    modes = ["full_full", "double_random", "str_full", "ablation_str_str", "ablation_kg_kg_subset", "ablation_cv_cv_padded",
             "ablation_tx_tx_padded", "ablation_str_random_str+kg_full_sample", "ablation_str_random_str+tx_full_sample",
             "ablation_str_random_str+kg+cv_full_sample", "ablation_str_random_str+cv+tx_full_sample"]
    out = {"base_head": base_h, "base_tail": base_t, "eval_types": np.array(eval_types), "modes": np.array(modes)}
    for i, et in enumerate(eval_types):
        for j, fm in enumerate(modes):
            h, t = EU.get_evaluate_masks(base_h, base_t, et, fm, "cpu")
            out[f"h_{i}_{j}"], out[f"t_{i}_{j}"] = h, t
    save("eval_masks", **out)
    ps = [torch.nn.Parameter(torch.zeros(1)) for _ in range(2)]
    opt = torch.optim.AdamW([{"params": [ps[0]], "lr": 1e-3}, {"params": [ps[1]], "lr": 5e-5}])
    sch = U.LinearWarmupCosineDecaySchedule(opt, warmup_epochs=7, total_epochs=40, num_cycles=1.0)
    lrs = []
    for _ in range(40):
        lrs.append([g["lr"] for g in opt.param_groups])
        opt.step()
        sch.step()
    save("lr_schedule", lrs=np.asarray(lrs, dtype=np.float64), warmup=7, total=40)


def gen_param_groups(M, C):
    """The reference's create_optimizer (madrigal/utils.py:463-613) on its own model: per parameter name the learning rate
    and weight decay of the group it lands in (NaN = the parameter is in no group, i.e. never updated)."""
    import madrigal.utils as U
    hp = dict(optimizer="adamw", structure_encoder_lr=1e-3, kg_encoder_lr=2e-3, perturb_encoders_lr=3e-3, fusion_lr=4e-3, decoder_lr=5e-3,
              wd=0.25, beta1=0.9, beta2=0.999, eps=1e-8)
    n, L, seed = 14, 6, 61
    out = {}
    for case in (ENCODE_CASES[1], ENCODE_CASES[2]):
        batch, bkg = D.make_batch(n, seed, kg_nodes=300, kg_edges=2500, masks=D.make_masks(n, seed))
        model = build_reference_model(M, C, bkg["data"], case, L, seed)
        opt = U.create_optimizer(model, hp)
        where = {}
        for g in opt.param_groups:
            for p in g["params"]:
                assert id(p) not in where
                where[id(p)] = (g["lr"], g["weight_decay"])
        names = [k for k, _ in model.named_parameters()]
        vals = np.array([where.get(id(p), (np.nan, np.nan)) for _, p in model.named_parameters()], dtype=np.float64)
        out[f"{case[0]}_names"], out[f"{case[0]}_lr_wd"] = np.array(names), vals
        out[f"{case[0]}_n_groups"] = len(opt.param_groups)
    save("param_groups", **out)


def gen_pretrain_lr():
    """The contrastive loop's per-ITERATION learning rate (pretrain.py:65 calls madrigal/utils.py:680-692 adjust_learning_rate
    with cur_epoch = epoch + i / iters_per_epoch): the reference function itself on a two-group optimizer, over three epochs of
    7 iterations with one warm-up epoch, and over a run whose warm-up is zero epochs long."""
    import madrigal.utils as U
    out = {}
    for tag, (lr, warm, total, ipe) in {"a": (3e-4, 1, 3, 7), "b": (1e-3, 0, 2, 5), "c": (5e-5, 2, 10, 4)}.items():
        ps = [torch.nn.Parameter(torch.zeros(1)) for _ in range(2)]
        opt = torch.optim.AdamW([{"params": [ps[0]], "lr": 9.0}, {"params": [ps[1]], "lr": 7.0}])
        got = []
        for epoch in range(total):
            for i in range(ipe):
                ret = U.adjust_learning_rate(opt, epoch + i / ipe, lr, warm, total)
                got.append([ret] + [g["lr"] for g in opt.param_groups])
        out[f"{tag}_lrs"] = np.asarray(got, dtype=np.float64)
        out[f"{tag}_cfg"] = np.asarray([lr, warm, total, ipe], dtype=np.float64)
    save("pretrain_lr", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="", help="comma-separated fixture groups (default: all)")
    args = ap.parse_args()
    torch.set_num_threads(4)
    M, S, C = import_reference(args.ref)
    groups = {
        "head": lambda: gen_head(M), "mlps": lambda: gen_mlps(M), "posenc": lambda: gen_posenc(M), "fusion": lambda: gen_fusion(M),
        "chemcpa": lambda: gen_chemcpa(C), "encode": lambda: gen_encode(M, C), "infonce": lambda: gen_infonce(M, S),
        "pretrain_views": gen_pretrain_views, "simclr_raw": lambda: gen_simclr_raw(M, S, C),
        "checkpoints": lambda: gen_checkpoint_fixtures(args.ref, M, S, C), "ranks": lambda: gen_ranks(args.ref),
        "bce": gen_bce, "eval_masks": gen_eval_masks_and_schedule, "param_groups": lambda: gen_param_groups(M, C),
        "pretrain_lr": gen_pretrain_lr,
    }
    only = [g for g in args.only.split(",") if g]
    for name, fn in groups.items():
        if not only or name in only:
            fn()


if __name__ == "__main__":
    main()
