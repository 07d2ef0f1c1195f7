"""BASELINE configs[2] as the reference ships it (configs/cl_pretrain/*.yaml): SimCLR_NovelDDI with raw_encoder_output=True
(encoders -> uni_projector only, madrigal/models/models.py:890-894) on 'str_center_uni' views (madrigal/utils.py:97-117,
360-390) -- eval-mode outputs against the reference's own, the training-mode step against torch autograd over the oracle."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import rel_err, t                                   # noqa: E402
from oracle.gen_cases import CL_CASE                             # noqa: E402
from oracle.params import det_state_dict                         # noqa: E402

pytestmark = pytest.mark.gpu


def _build(M, kg, shared, basal, mlp_dim=512, T=0.1):
    from madrigal_amd.simclr import SimCLR_NovelDDI
    from test_models_gpu import build_model
    enc = build_model(M, CL_CASE, kg, 4, use_tx_basal=basal).encoder
    return SimCLR_NovelDDI(enc, dim=128, mlp_dim=mlp_dim, T=T, raw_encoder_output=True, shared_predictor=shared)


def _views(n, seed, p_kg=0.6, p_cv=0.5, p_tx=0.25):
    """A seeded availability table in which every drug owns a second modality, and one 'str_center_uni' draw from it."""
    from madrigal_amd import data as D, masks as MK
    avail = D.make_masks(n, seed, p_kg=p_kg, p_cv=p_cv, p_tx=p_tx)
    avail[:, 1] = torch.where(avail[:, 1:].all(dim=1), torch.zeros(n, dtype=torch.bool), avail[:, 1])
    bank = MK.get_pretrain_masks(list(range(n)), avail.numpy().astype(np.int64), "str_center_uni", False, 0.2)
    state = np.random.get_state()
    np.random.seed(seed)
    m1, m2 = MK.pretrain_modality_subset_sampler([bank[d] for d in range(n)], "str_center_uni", False)
    np.random.set_state(state)
    return avail, m1, m2


@pytest.mark.parametrize("prec,tol", [("f32", 3e-5), ("bf16x3", 1e-4)])
@pytest.mark.parametrize("shared,basal", [(False, False), (False, True), (True, False), (True, True)])
def test_simclr_raw_forward_matches_reference(golden, shared, basal, prec, tol):
    from madrigal_amd import data as D, models as M
    g = golden("simclr_raw")
    n, seed = (int(v) for v in g["meta"])
    batch, bkg = D.make_batch(n, seed, kg_nodes=300, kg_edges=2500, masks=t(g["avail"]))
    tag = f"s{int(shared)}b{int(basal)}"
    model = _build(M, bkg["data"], shared, basal)
    sd = model.state_dict()
    assert sorted(sd.keys()) == list(g[tag + "_keys"])                     # the reference's own SimCLR state_dict layout
    model.load_state_dict(det_state_dict(seed, {k: tuple(v.shape) for k, v in sd.items()}))
    model = model.cuda().eval()
    b = D.batch_to(batch, "cuda")
    kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
    m1, m2, hard = t(g["mask1"]).cuda(), t(g["mask2"]).cuda(), t(g["hard"]).cuda()
    with torch.no_grad(), M.precision(prec):
        a1, a2, (lg, lb, loss) = model(b["drugs"], m1, m2, hard, (b["strs"], kgc, b["cv"], b["tx"]))
        raw1 = model.base_encoder(b["drugs"], m1, b["strs"], kgc, b["cv"], b["tx"], raw_encoder_output=True)
    assert rel_err(raw1.cpu(), g[tag + "_raw1"]) < tol
    assert rel_err(a1.cpu(), g[tag + "_aug1"]) < tol and rel_err(a2.cpu(), g[tag + "_aug2"]) < tol
    keep = np.abs(g[tag + "_logits"]) < 1e8
    assert rel_err(lg.cpu().numpy()[keep], g[tag + "_logits"][keep]) < tol
    assert abs(float(loss) - float(g[tag + "_loss"])) < 10 * tol * abs(float(g[tag + "_loss"]))


def _no_dropout(model):
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    return model


@pytest.mark.parametrize("shared,basal", [(False, True), (True, False)], ids=["drugbank_basal", "shared_predictor"])
def test_simclr_raw_training_step_matches_oracle_autograd(shared, basal):
    """pretrain.py:59-93 on the shipped path, training mode: loss, every parameter gradient and the BatchNorm running
    statistics after the step, against torch autograd over the CPU oracle run with batch statistics (dropout off on both
    sides: masks are not comparable across implementations).  The reference runs all four encoders once per view, so GIN
    and the tx encoder take TWO running-statistics updates per step."""
    from madrigal_amd import data as D, models as M
    from oracle import madrigal_oracle as O
    from oracle.pipeline import oracle_simclr
    from helpers import smooth_relu
    n, T = 72, 0.1

    def run(seed):
        avail, m1, m2 = _views(n, seed)
        batch, bkg = D.make_batch(n, seed, kg_nodes=500, kg_edges=5000, masks=avail)
        hard = torch.rand(n, n, generator=torch.Generator().manual_seed(3)) < 0.04
        hard = (hard | hard.T) & ~torch.eye(n, dtype=torch.bool)
        torch.manual_seed(seed)
        model = smooth_relu(_no_dropout(_build(M, bkg["data"], shared, basal, mlp_dim=256, T=T)))
        p0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        filler = torch.zeros(max(int(batch["drugs"].max()) + 1, int(bkg["drug_index_map"].max()) + 1), 128)

        pr = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in p0.items()}
        record = {}
        with O.batch_statistics(record), O.relu_as("gelu"):
            ref = oracle_simclr(pr, batch, bkg, m1, m2, hard, T, filler, shared_predictor=shared, use_tx_basal=basal)
        ref["loss"].backward()

        model = model.cuda().train()
        b = D.batch_to(batch, "cuda")
        kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
        with M.precision("f32"):                                   # ReLU networks: exact-fp32 products keep derivative flips rare
            a1, a2, (lg, lb, loss) = model(b["drugs"], m1.cuda(), m2.cuda(), hard.cuda(), (b["strs"], kgc, b["cv"], b["tx"]))
            loss.backward()
        assert a1.shape == (n, 128) and a2.shape == (n, 128)
        assert abs(float(loss) - float(ref["loss"])) < 1e-4 * abs(float(ref["loss"]))
        assert rel_err(a1.detach().cpu(), ref["aug1"].detach()) < 2e-4 and rel_err(a2.detach().cpu(), ref["aug2"].detach()) < 2e-4
        named = dict(model.named_parameters())
        gmax = max(float(v.grad.abs().max()) for v in pr.values() if torch.is_tensor(v) and v.grad is not None)
        checked, errs = 0, []
        for k, v in pr.items():
            if k not in named or not (torch.is_tensor(v) and v.requires_grad):       # buffers (GIN eps, running statistics)
                continue
            if v.grad is None or not bool(v.grad.any()):
                # the fusion transformer, uni_fuser, learned tokens and (under str_center_uni) nothing else: unused by this path
                if k in named and named[k].grad is not None:
                    assert float(named[k].grad.abs().max()) <= 1e-6 * gmax, k
                continue
            assert named[k].grad is not None, f"{k}: no gradient on the HIP path"
            a, r = named[k].grad.cpu().double(), v.grad.double()
            errs.append((float((a - r).abs().max()) / max(float(r.abs().max()), 1e-2 * gmax), k))
            checked += 1
        assert checked > 60, checked
        # parameters outside the path get no gradient at all (the reference's optimizer skips them: grad is None there)
        for k, q in named.items():
            if k.startswith(("base_encoder.transformer.", "base_encoder.uni_fuser.", "base_encoder.tx_bottleneck_tokens", "base_encoder.pos_encoder.")):
                assert q.grad is None, k
        # BatchNorm running statistics: momentum updates replayed from the oracle's batch statistics, in call order
        sd = model.state_dict()
        n_bn = 0
        for k in p0:
            if not k.endswith("running_mean"):
                continue
            stem = k[: -len("running_mean")]
            calls = record.get(id(pr[k]), [])
            rm, rv = p0[k].clone(), p0[stem + "running_var"].clone()
            for mean, var_unbiased in calls:
                rm = 0.9 * rm + 0.1 * mean
                rv = 0.9 * rv + 0.1 * var_unbiased
            assert rel_err(sd[k].cpu(), rm) < 2e-5, k
            assert rel_err(sd[stem + "running_var"].cpu(), rv) < 2e-4, k
            assert int(sd[stem + "num_batches_tracked"]) == int(p0[stem + "num_batches_tracked"]) + len(calls), k
            n_bn += len(calls) > 0
        assert n_bn >= 4 + 2 + 2                                    # GIN x4, chemCPA encoder x2, the predictors' BatchNorms
        return errs

    # The comparison runs on the network with GELU in place of every ReLU (helpers.smooth_relu / oracle.relu_as; the ReLU kernels have
    # their own forward / backward tests): every gradient entry of every tensor within 5e-4 of its tensor's scale, on EVERY seed
    for sd in (33, 34, 35, 36):
        worst = max(run(sd))
        print("seed", sd, "worst (error, tensor):", worst)
        assert worst[0] < 5e-4, (sd, worst)


def test_simclr_raw_pretraining_steps_reduce_loss_and_are_reproducible():
    """pretrain.py's loop on the shipped path: a fresh 'str_center_uni' draw per iteration (utils.py:360-390), AdamW; the
    loss falls and a seeded run repeats bit for bit."""
    from madrigal_amd import data as D, masks as MK, models as M
    from madrigal_amd.optim import AdamW
    from madrigal_amd.train import PretrainStep
    n, seed = 96, 12

    def run(steps, host_inputs=False):
        torch.manual_seed(seed)
        np.random.seed(seed)
        avail, _, _ = _views(n, seed)
        batch, bkg = D.make_batch(n, seed, kg_nodes=600, kg_edges=6000, masks=avail)
        model = _build(M, bkg["data"], False, True, mlp_dim=256, T=0.5).cuda().train()
        bank = MK.get_pretrain_masks(list(range(n)), avail.numpy().astype(np.int64), "str_center_uni", False, 0.2)
        b = D.batch_to(batch, "cuda")
        kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
        step = PretrainStep(model, AdamW(model.parameters(), lr=3e-4, weight_decay=1e-2))
        losses = []
        for _ in range(steps):
            m1, m2 = MK.pretrain_modality_subset_sampler([bank[d] for d in range(n)], "str_center_uni", False)
            if host_inputs:             # the loader's CPU tensors as they are: index work on the host, pinned asynchronous uploads
                loss = step.step(batch["drugs"], m1, m2, None, (b["strs"], kgc, b["cv"], b["tx"]))
            else:
                loss = step.step(b["drugs"], m1.cuda(), m2.cuda(), None, (b["strs"], kgc, b["cv"], b["tx"]))
            losses.append(float(loss))
        return losses
    l1, l2 = run(10), run(10)
    assert all(np.isfinite(l1)) and min(l1[-3:]) < l1[0], l1
    assert l1 == l2
    assert run(10, host_inputs=True) == l1          # same rows in the same order, same kernels: bit-identical


@pytest.mark.parametrize("prec,tol", [("f32", 1e-4), ("bf16x3", 3e-4)])
def test_contrastive_forward_at_the_stated_batch_against_the_oracle(prec, tol):
    """BASELINE configs[2] at its stated size -- batch 2048 over the bench's KG (130 000 nodes / 8 000 000 edges), 'str_center_uni'
    views, separate predictors, T = 0.1, mlp_dim 512 -- training-mode forward (BatchNorm batch statistics over the 2048 drugs /
    their 53 000 atoms / the 32 768 tx rows; dropout off on both sides) against the CPU oracle, stage by stage at FULL size:
    view 1 = uni_projector(GIN(all molecules)) for every drug; view 2 for every drug whose drawn modality is cv or a tx cell line
    (the oracle's chemCPA encoder over all 16 x 2048 rows with batch statistics); the KG-view rows are the one stage taken from the
    HIP side (the oracle's HGT over 8e6 edges takes minutes: it is pinned at fixture size, tests above and test_models_gpu); then
    predictors (batch statistics over 2048 rows) and the [4096, 4095] InfoNCE loss on the HIP side's views against the oracle's on
    the same views: every aug row and the loss."""
    from madrigal_amd import data as D, models as M
    from oracle import madrigal_oracle as O
    free, _ = torch.cuda.mem_get_info()
    if free < 30 * 2 ** 30:
        pytest.skip("needs 30 GB of free HBM")
    B, seed, T = 2048, 5, 0.1
    avail, m1, m2 = _views(B, seed, p_kg=0.6, p_cv=0.3, p_tx=0.1)            # SURVEY 8(d) availability rates
    batch, bkg = D.make_batch(B, seed, kg_nodes=130_000, kg_edges=8_000_000, masks=avail)
    torch.manual_seed(seed)
    model = _no_dropout(_build(M, bkg["data"], False, False, mlp_dim=512, T=T))
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.cuda().train()
    b = D.batch_to(batch, "cuda")
    kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
    data = (b["strs"], kgc, b["cv"], b["tx"])
    with torch.no_grad(), M.precision(prec):
        e1 = model.base_encoder(b["drugs"], m1.cuda(), *data, raw_encoder_output=True).cpu()
        e2 = model.base_encoder(b["drugs"], m2.cuda(), *data, raw_encoder_output=True).cpu()
        a1, a2, (lg, lb, loss) = model(b["drugs"], m1.cuda(), m2.cuda(), None, data)
    assert e1.shape == (B, 128) and e2.shape == (B, 128) and a1.shape == (B, 128)
    assert int((~m1).sum()) == B and bool((~m1)[:, 0].all()) and int((~m2).sum()) == B      # one row per drug and view
    enc = O._sub({"encoder." + k: v for k, v in O._sub(p, "base_encoder.").items()}, "encoder.")
    proj = lambda x: O.mlp_encoder_forward(O._sub(enc, "uni_projector."), x, 2, "ln", "relu", 0.2, "nd")
    mols = batch["strs"]
    with O.batch_statistics():
        str_out = O.gin_forward(O._sub(enc, "str_encoder."), mols.node_feature, mols.edge_list, mols.edge_feature, mols.node2graph,
                                mols.batch_size, num_layers=4, num_mlp_layer=3)["graph_feature"]
        assert rel_err(e1, proj(str_out)) < tol                                   # view 1, all 2048 drugs
        cv_out = O.mlp_encoder_forward(O._sub(enc, "cv_encoder."), batch["cv"], 2, None, "relu", 0.2)
        sigs = torch.cat([batch["tx"][c]["sigs"] for c in D.CELL_LINES])
        _, _, _, treated = O.chemcpa_predict(O._sub(enc, "tx_encoder."), sigs, torch.arange(16).repeat_interleave(B), 3, 3, with_decoder=False)
        col = (~m2).float().argmax(dim=1)                                         # the one modality view 2 shows of each drug
        mods = torch.stack([str_out, torch.zeros_like(str_out), cv_out] + list(treated.split(B)), dim=1)      # [B, 19, 128]; KG column unused
        keep = col != 1
        assert int(keep.sum()) > B // 4 and int((col == 2).sum()) > 50 and int((col >= 3).sum()) > 50
        want2 = proj(mods[torch.arange(B), col][keep])
        assert rel_err(e2[keep], want2) < tol                                     # view 2, every cv / tx drug
        # predictors + InfoNCE at B = 2048 on the HIP side's views
        r1 = O.simclr_predictor(O._sub(p, "predictor_1."), e1)
        r2 = O.simclr_predictor(O._sub(p, "predictor_2."), e2)
    assert rel_err(a1.cpu(), r1) < tol and rel_err(a2.cpu(), r2) < tol
    logits, labels, loss_ref = O.info_nce(r1, r2, None, T)
    assert tuple(lg.shape) == (2 * B, 2 * B - 1)
    assert abs(float(loss) - float(loss_ref)) < 1e-4 * abs(float(loss_ref)), (float(loss), float(loss_ref))
    assert rel_err(lg.cpu(), logits) < 10 * tol
