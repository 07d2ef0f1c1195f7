"""Checkpoint interop and the on-device batch bundle (8f-3): host-side logic, no GPU needed."""
import os
import types

import numpy as np
import pytest
import torch

from madrigal_amd import checkpoint as CK, configs, data as D, models as M
from madrigal_amd.simclr import SimCLR_NovelDDI


def _encoder_configs(kg, **over):
    c = configs.SHIPPED["twosides321"]
    cfg = CK.make_encoder_configs(all_kg_data=kg, feat_dim=128, str_encoder_name="gin", str_encoder_hparams=dict(configs.GIN),
                                  kg_encoder_name="hgt", kg_encoder_hparams=dict(configs.HGT), cv_encoder_name="mlp",
                                  cv_encoder_hparams=dict(configs.CV), tx_encoder_name="chemcpa", tx_encoder_hparams=configs.TX_CHEMCPA,
                                  num_tx_bottlenecks=c["nb"], pos_emb_type=c["pos"], pos_emb_dropout=0.2,
                                  transformer_fusion_hparams=dict(configs._tf(2, 32, 64, 1, 0.1)), proj_hparams=dict(configs.PROJ),
                                  fusion=c["fusion"], use_modality_pretrain=False)
    cfg.update(over)
    return cfg


def _build(kg, L=5, **over):
    cfg = _encoder_configs(kg, **over)
    enc = M.NovelDDIEncoder(**{k: v for k, v in cfg.items() if v is not None or k != "tab_mod_encoder_hparams_dict"})
    return M.NovelDDIMultilabel(enc, **CK.make_model_configs(128, L)), cfg


@pytest.fixture(scope="module")
def small():
    batch, bkg = D.make_batch(12, 3, kg_nodes=200, kg_edges=900)
    return batch, bkg


@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("adaptor", [False, True])
def test_cl_to_finetune_key_filter_matches_reference(golden, shared, adaptor):
    """filter_pretrained_state_dict against the reference's own loop (madrigal/utils.py:281-295) run on its own SimCLR key
    list.  The reference leaves the skipped projector entries in the dict under their prefixed names, where
    load_state_dict(strict=False) ignores them; what reaches the encoder is the un-prefixed set."""
    g = golden("ckpt_filter")
    keys = [str(k) for k in g[f"s{int(shared)}_in"]]
    mine = CK.filter_pretrained_state_dict({k: k for k in keys}, use_pretrained_adaptor=adaptor)
    tag = f"s{int(shared)}a{int(adaptor)}"
    ref = {str(k): str(v) for k, v in zip(g[tag + "_out"], g[tag + "_src"]) if not str(k).startswith("base_encoder")}
    assert mine == ref
    assert not any(k.startswith(("transformer.", "pos_encoder.", "head.", "predictor")) or k in ("cls", "tx_bottleneck_tokens") for k in mine)
    assert any(k.startswith("uni_projector.") for k in mine) == adaptor


@pytest.mark.parametrize("kg_format", ["object", "plain"])
def test_finetune_checkpoint_round_trip(small, tmp_path, kg_format):
    """train_ddi_batch.py:393-412 writes {epoch, state_dict, encoder_configs, model_configs}; predict.py:178-205 reads it."""
    batch, bkg = small
    torch.manual_seed(0)
    model, cfg = _build(bkg["data"], L=7)
    path = str(tmp_path / "best_model.pt")
    CK.save_finetune_checkpoint(path, model, epoch=41, encoder_configs=cfg, model_configs=CK.make_model_configs(128, 7), kg_format=kg_format)
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert set(raw) == {"epoch", "state_dict", "encoder_configs", "model_configs"} and raw["epoch"] == 41
    assert set(raw["model_configs"]) == {"feat_dim", "prediction_dim", "normalize", "use_single_drug"}
    assert list(raw["state_dict"]) == list(model.state_dict())
    if kg_format == "plain":                       # nothing but tensors, lists, tuples, strings under the KG entry
        assert isinstance(raw["encoder_configs"]["all_kg_data"], dict)
    loaded, ckpt, msg = CK.load_finetune_checkpoint(path)
    assert not msg.missing_keys and not msg.unexpected_keys
    assert isinstance(loaded, M.NovelDDIMultilabel) and loaded.decoder.out_features == 7
    for (k, a), (k2, b) in zip(model.state_dict().items(), loaded.state_dict().items()):
        assert k == k2 and torch.equal(a, b), k
    assert loaded.encoder.kg_encoder is not None and ckpt["encoder_configs"]["num_tx_bottlenecks"] == cfg["num_tx_bottlenecks"]
    # a reference-written file carries a HeteroData: anything with x_dict / edge_index_dict / metadata() is accepted
    kg = bkg["data"]
    hetero = types.SimpleNamespace(x_dict=kg.x_dict, edge_index_dict=kg.edge_index_dict, metadata=lambda: (kg.node_types, kg.edge_types))
    raw["encoder_configs"]["all_kg_data"] = hetero
    again, _, _ = CK.load_finetune_checkpoint(raw)
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), again.state_dict().values()))


def test_pretraining_checkpoint_seeds_a_finetune_encoder(small, tmp_path):
    """pretrain.py:230-236 -> madrigal/utils.py:246-311: modality encoders (and optionally the projector) carried over with the
    prefix stripped; fusion transformer, position encoding and learned tokens re-initialised under the finetune run's own
    hyper-parameters; the predictors dropped."""
    from madrigal_amd.optim import AdamW
    batch, bkg = small
    torch.manual_seed(1)
    model, cfg = _build(bkg["data"], num_tx_bottlenecks=1, pos_emb_type="learnable")
    sim = SimCLR_NovelDDI(model.encoder, dim=128, mlp_dim=64, T=0.1, raw_encoder_output=True)
    opt = AdamW(sim.parameters(), lr=1e-3)
    path = str(tmp_path / "checkpoint_49.pt")
    CK.save_pretrain_checkpoint(path, sim, opt, epoch=50, encoder_configs=cfg, kg_args={"kg_sampling_num_neighbors": None})
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert set(raw) == {"epoch", "state_dict", "optimizer", "encoder_configs", "kg_args"}
    assert all(k.startswith(("base_encoder.", "predictor_1.", "predictor_2.")) for k in raw["state_dict"])
    tf = dict(configs._tf(4, 32, 64, 2, 0.1))
    for adaptor in (True, False):
        torch.manual_seed(2)
        enc, new_cfg, msg = CK.load_pretrained_encoder(path, overrides={"num_tx_bottlenecks": 2, "transformer_fusion_hparams": tf, "fusion": None},
                                                       use_pretrained_adaptor=adaptor)
        assert new_cfg["num_tx_bottlenecks"] == 2 and new_cfg["fusion"] == cfg["fusion"] and enc.tx_bottleneck_tokens.shape == (2, 128)
        assert not msg.unexpected_keys
        src = sim.base_encoder.state_dict()
        loaded = enc.state_dict()
        for k, v in loaded.items():
            carried = k.startswith(("str_encoder.", "kg_encoder.", "cv_encoder.", "tx_encoder.", "uni_fuser.")) or (adaptor and k.startswith("uni_projector."))
            if carried:
                assert torch.equal(v, src[k]), k
            else:
                assert k in msg.missing_keys, k
    with pytest.raises(KeyError):
        CK.load_pretrained_encoder(path, overrides={"feat_dim": 64})
    # utils.py:281-295 filters whatever else the checkpoint holds: one without 'epoch' loads the same weights ...
    no_epoch = {k: v for k, v in raw.items() if k != "epoch"}
    enc2, cfg2, msg2 = CK.load_pretrained_encoder(no_epoch)
    assert not msg2.unexpected_keys and cfg2["use_modality_pretrain"] == cfg["use_modality_pretrain"]
    assert all(torch.equal(v, sim.base_encoder.state_dict()[k]) for k, v in enc2.state_dict().items() if k.startswith(("str_encoder.", "cv_encoder.")))
    # ... and a state dict with no base_encoder.* entry (a finetune checkpoint) is refused instead of returning random weights
    with pytest.raises(ValueError, match="base_encoder"):
        CK.load_pretrained_encoder({**no_epoch, "state_dict": {k[len("base_encoder."):]: v for k, v in raw["state_dict"].items() if k.startswith("base_encoder.")}})


def test_shipped_unimodal_weight_files_load_strictly(golden, tmp_path, monkeypatch):
    """modality_pretraining/str/GIN_256x4_muv.pt and cv/cv_model_ae.pt (the two weight files inside the reference checkout):
    files with exactly their parameter names, shapes and dtypes must load strictly through get_str_encoder /
    get_tabular_mod_encoder (madrigal/models/models.py:213-232, 250-259), values intact."""
    g = golden("pretrained_layouts")
    files = {}
    for name, rel in (("gin", "str/GIN_256x4_muv.pt"), ("cv", "cv/cv_model_ae.pt")):
        gen = torch.Generator().manual_seed(len(name))
        sd = {}
        for k, shp, dt in zip(g[name + "_keys"], g[name + "_shapes"], g[name + "_dtypes"]):
            shape = tuple(int(x) for x in str(shp).split(",") if x)
            sd[str(k)] = torch.zeros(shape, dtype=torch.int64) if "int64" in str(dt) else torch.rand(shape, generator=gen) + 0.5
        os.makedirs(tmp_path / os.path.dirname(rel), exist_ok=True)
        torch.save(sd, str(tmp_path / rel))
        files[name] = sd
    monkeypatch.setenv("ENCODER_CKPT_DIR", str(tmp_path) + "/")
    gin = M.get_str_encoder("gin", dict(configs.GIN), 128, D.MOL_DIM, use_modality_pretrain=True)
    got = gin.state_dict()
    assert set(got) == {k[len("model."):] if k.startswith("model.") else k for k in files["gin"] if k.startswith(("model.", "layer"))}
    for k, v in got.items():
        src = files["gin"].get(k, files["gin"].get("model." + k))
        assert v.shape == src.shape and torch.equal(v, src.to(v.dtype)), k
    cv = M.get_tabular_mod_encoder("mlp", dict(configs.CV), 128, use_modality_pretrain=True, mod="cv")
    for k, v in cv.state_dict().items():
        assert torch.equal(v, files["cv"][k]), k
    assert set(cv.state_dict()) == set(files["cv"])


def test_bundle_round_trip_and_duck_typed_sources(small, tmp_path):
    """The collator's output (madrigal/data/data.py:948-964) as plain tensors: save -> load reproduces every field; a
    torchdrug PackedMolecule / PyG HeteroData look-alike converts to the same containers."""
    batch, bkg = small
    trip = D.make_labelled_triples(12, 5, 30, 3)
    path = str(tmp_path / "batch.pt")
    D.save_bundle(path, batch, bkg, trip)
    b2, kg2, t2 = D.load_bundle(path)
    for k in ("drugs", "cv", "masks"):
        assert torch.equal(batch[k], b2[k])
    for f in ("node_feature", "edge_list", "edge_feature", "node2graph", "edge_weight"):
        assert torch.equal(getattr(batch["strs"], f), getattr(b2["strs"], f)), f
    assert b2["strs"].batch_size == batch["strs"].batch_size
    for c in D.CELL_LINES:
        for f in ("sigs", "drugs", "dosages"):
            assert torch.equal(batch["tx"][c][f], b2["tx"][c][f])
        assert list(batch["tx"][c]["cell_lines"]) == list(b2["tx"][c]["cell_lines"])
    assert kg2["data"].metadata() == bkg["data"].metadata() and torch.equal(kg2["drug_index_map"], bkg["drug_index_map"])
    for k, v in bkg["data"].x_dict.items():
        assert torch.equal(v, kg2["data"].x_dict[k])
    for k, v in bkg["data"].edge_index_dict.items():
        assert torch.equal(v, kg2["data"].edge_index_dict[k])
    assert all(torch.equal(a, b) for a, b in zip(trip, t2))
    # look-alikes of the reference's own objects (attributes only; neither package is imported)
    m = batch["strs"]
    packed = types.SimpleNamespace(node_feature=m.node_feature.numpy(), edge_list=m.edge_list, edge_feature=m.edge_feature.double(),
                                   node2graph=m.node2graph, num_nodes=torch.bincount(m.node2graph))
    conv = D.as_molecule_batch(packed)
    assert conv.batch_size == m.batch_size and torch.equal(conv.node_feature, m.node_feature) and torch.equal(conv.edge_feature, m.edge_feature)
    assert bool((conv.edge_weight == 1).all())
    kg = bkg["data"]
    hetero = types.SimpleNamespace(x_dict=kg.x_dict, edge_index_dict=kg.edge_index_dict, metadata=lambda: (kg.node_types, kg.edge_types))
    ck = D.as_kg_data(hetero)
    assert ck.metadata() == kg.metadata() and all(torch.equal(ck.edge_index_dict[e], kg.edge_index_dict[e]) for e in kg.edge_types)
    with pytest.raises(ValueError):
        D.batch_from_bundle({"format": "something else"})


def test_data_parallel_steps_refuse_rank_local_batchnorm(small):
    """proj_norm='bn' puts a BatchNorm inside modules that only ranks WITH uni-modal drugs would run: under SyncBatchNorm that
    is a collective mismatch, so the data-parallel steps refuse it up front (the single-process steps accept it)."""
    from madrigal_amd.train import FinetuneStep, PretrainStep
    batch, bkg = small
    model, _ = _build(bkg["data"], proj_hparams=dict(configs.PROJ, proj_norm="bn"))
    FinetuneStep(model, optimizer=None)
    with pytest.raises(NotImplementedError, match="proj_norm"):
        FinetuneStep(model, optimizer=None, rank=0, world=2)
    sim = SimCLR_NovelDDI(model.encoder, dim=128, mlp_dim=64)
    with pytest.raises(NotImplementedError, match="proj_norm"):
        PretrainStep(sim, optimizer=None, rank=1, world=2)
    ok, _ = _build(bkg["data"])
    FinetuneStep(ok, optimizer=None, rank=0, world=2)
