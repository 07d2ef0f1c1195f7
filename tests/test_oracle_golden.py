"""The CPU oracle against golden vectors produced by the imported reference
(oracle/gen_golden.py).  No GPU, no HIP: this pins the checker itself."""
import numpy as np
import pytest
import torch

from oracle import madrigal_oracle as O
from oracle.params import det_state_dict
from madrigal_amd import data as D
from helpers import (ENCODE_CASES, FUSION_CASES, chemcpa_shapes, fusion_params, mlp_shapes, model_shapes_for_case,
                     oracle_pipeline, rel_err, t)

TOL = 2e-5


def test_head(golden):
    g = golden("head")
    zh, zt, w = t(g["z_head"]), t(g["z_tail"]), t(g["w_original"])
    assert np.array_equal(O.symmetric(w).numpy(), g["w_sym"])
    assert rel_err(O.bilinear_scores(zh, zt, w), g["scores"]) < TOL
    assert rel_err(O.bilinear_scores(zh, zt, w, (1, 4)), g["scores_1_4"]) < TOL
    s = O.bilinear_scores(zh, zh, w)
    assert rel_err(s, g["scores_self"]) < TOL
    assert rel_err(s, s.transpose(1, 2)) < TOL        # symmetric W + same drugs => symmetric scores


@pytest.mark.parametrize("name,in_dim,hidden,out,p,norm,actn,order", [
    ("cv", 559, [512, 256], 128, 0.2, None, "relu", "nd"),
    ("proj", 128, [512, 512], 128, 0.2, "ln", "relu", "nd"),
    ("bn_dn", 40, [64, 48, 32], 16, 0.1, "bn", "gelu", "dn"),
    ("one_hidden", 32, [64], 8, 0.0, "ln", "tanh", "nd"),
])
def test_mlps(golden, name, in_dim, hidden, out, p, norm, actn, order):
    g = golden("mlps")
    shapes = mlp_shapes(in_dim, hidden, out, p, norm, order)
    assert sorted(shapes) == list(g[name + "_keys"])          # state_dict keys match the reference's
    params = det_state_dict(21, shapes)
    y = O.mlp_encoder_forward(params, t(g[name + "_x"]), len(hidden), norm, actn, p, order)
    assert rel_err(y, g[name + "_y"]) < TOL


@pytest.mark.parametrize("nb,agg", [(0, "x-attn"), (4, "x-attn"), (2, "cls")])
def test_posenc(golden, nb, agg):
    g = golden("posenc")
    max_len = (D.NUM_MODALITIES if nb == 0 else D.NUM_NON_TX_MODALITIES) + (1 if agg == "cls" else 0)
    pe = O.sinusoidal_pe_table(128, max_len, nb, agg)
    assert rel_err(pe, g[f"sin_{nb}_{agg}_pe"]) < 1e-6
    x = t(g[f"x_{nb}_{agg}"])
    assert rel_err(O.apply_pos_enc(x, pe, "sinusoidal"), g[f"sin_{nb}_{agg}_y"]) < 1e-6
    lp = det_state_dict(31, {"pe": (1, max_len, 128)})["pe"]
    assert rel_err(O.apply_pos_enc(x, lp, "learnable"), g[f"lrn_{nb}_{agg}_y"]) < 1e-6


@pytest.mark.parametrize("case", FUSION_CASES, ids=[c[0] for c in FUSION_CASES])
def test_fusion(golden, case):
    name, H, dh, ffn, nl, nf, agg, nb, actn = case
    g = golden("fusion_" + name)
    p = fusion_params(41, H, dh, ffn, nl, agg)
    src = t(g["src"]) if g["src"].size else None
    out, probs = O.transformer_fusion_forward(p, t(g["seq"]), t(g["kpm"]), src, num_layers=nl, num_heads=H,
                                              norm_first=nf, actn=actn, agg=agg, num_tx_bottlenecks=nb,
                                              return_probs=True)
    assert rel_err(out, g["out"]) < TOL
    # attention weights of the last layer (the reference's forward-hook target, predict.py:643)
    assert rel_err(probs, g["attn_last"]) < TOL


def test_chemcpa(golden):
    g = golden("chemcpa")
    shapes = chemcpa_shapes()
    assert sorted(shapes) == list(g["keys"])
    p = det_state_dict(51, shapes)
    rec, emb, basal, treated = O.chemcpa_predict(p, t(g["genes"]), t(g["cov_idx"]), 3, 3)
    for a, b in ((rec, "recon"), (emb, "cell_emb"), (basal, "basal"), (treated, "treated")):
        assert rel_err(a, g[b]) < TOL


def encode_with_oracle(case, g):
    """Restated NovelDDIEncoder.encode + NovelDDIMultilabel.forward for one golden case."""
    n, L, seed = (int(v) for v in g["meta"])
    masks = t(g["masks"])
    batch, bkg = D.make_batch(n, seed, kg_nodes=300, kg_edges=2500, masks=masks)
    shapes = model_shapes_for_case(case, bkg["data"], L)
    assert sorted(shapes) == list(g["keys"])
    skip = [k for k in shapes if k.endswith("pos_encoder.pe") and case[3] == "sinusoidal"]
    p = det_state_dict(seed, shapes, skip)
    return oracle_pipeline(case, p, batch, bkg, masks, t(g["kg_filler"]), label_slices=[(2, 5)])


@pytest.mark.parametrize("case", ENCODE_CASES, ids=[c[0] for c in ENCODE_CASES])
def test_encode_glue(golden, case):
    g = golden("encode_" + case[0])
    got = encode_with_oracle(case, g)
    for k, v in got.items():
        assert rel_err(v, g[k]) < 5e-5, k


def test_infonce(golden):
    g = golden("infonce")
    lg, lb, loss = O.info_nce(t(g["aug1"]), t(g["aug2"]), t(g["hard"]), float(g["T"]))
    assert rel_err(lg, g["logits"]) < TOL and np.array_equal(lb.numpy(), g["labels"])
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    lg0, _, loss0 = O.info_nce(t(g["aug1"]), t(g["aug2"]), None, float(g["T"]))
    assert rel_err(lg0, g["logits_nomask"]) < TOL
    assert abs(float(loss0) - float(g["loss_nomask"])) < 1e-5 * abs(float(g["loss_nomask"]))


def test_ranks(golden):
    g = golden("ranks")
    out = O.rank_normalize(g["scores"])
    assert np.array_equal(out, g["normalized"])         # distinct scores: rank ordering is bit-exact
    r = O.lower_triangle_ranks(g["scores"])
    N = g["scores"].shape[1]
    il = np.tril_indices(N, k=-1)
    assert np.array_equal((r[0] / (N * (N - 1) / 2)).astype(np.float32), g["normalized"][0][il])


def test_bce(golden):
    g = golden("bce")
    p, loss = O.gathered_bce_loss(t(g["scores"]), t(g["labels"]), t(g["heads"]), t(g["tails"]), t(g["y"]))
    assert rel_err(p, g["pred"]) < 1e-6
    assert abs(float(loss) - float(g["loss"])) < 1e-6


# ---------------------------------------------------------------------------------------------- harness-side host logic
def test_evaluate_masks_match_reference(golden):
    """madrigal_amd.masks against the reference's own get_evaluate_masks (eval_utils.py:287-305), every evaluation type x
    finetune mode of the fixture ({masks_base, eval_type} -> masks)."""
    import torch
    from madrigal_amd.masks import get_evaluate_masks
    g = golden("eval_masks")
    bh, bt = torch.from_numpy(g["base_head"]), torch.from_numpy(g["base_tail"])
    n = 0
    for i, et in enumerate(g["eval_types"]):
        for j, fm in enumerate(g["modes"]):
            h, t = get_evaluate_masks(bh, bt, str(et), str(fm), "cpu")
            assert h.dtype == torch.bool and t.dtype == torch.bool
            assert torch.equal(h, torch.from_numpy(g[f"h_{i}_{j}"])), (et, fm, "head")
            assert torch.equal(t, torch.from_numpy(g[f"t_{i}_{j}"])), (et, fm, "tail")
            n += 1
    assert n == 121
    import pytest
    with pytest.raises(AssertionError):
        get_evaluate_masks(bh, bt, "str", "full_full", "cpu")
    with pytest.raises(KeyError):
        get_evaluate_masks(bh, bt, "str_nosuch", "full_full", "cpu")


def test_warmup_cosine_schedule_matches_reference(golden):
    import numpy as np
    import torch
    from madrigal_amd.optim import LinearWarmupCosineDecaySchedule
    g = golden("lr_schedule")
    ps = [torch.nn.Parameter(torch.zeros(1)) for _ in range(2)]
    opt = torch.optim.AdamW([{"params": [ps[0]], "lr": 1e-3}, {"params": [ps[1]], "lr": 5e-5}])
    sch = LinearWarmupCosineDecaySchedule(opt, warmup_epochs=int(g["warmup"]), total_epochs=int(g["total"]))
    lrs = []
    for _ in range(int(g["total"])):
        lrs.append([q["lr"] for q in opt.param_groups])
        opt.step()
        sch.step()
    assert np.allclose(np.asarray(lrs), g["lrs"], rtol=1e-12, atol=0)


def test_pretrain_lr_schedule_matches_reference_per_iteration(golden):
    """pretrain.py:65 + madrigal/utils.py:680-692: the contrastive loop sets ONE rate on every parameter group before every
    iteration, from the fractional epoch.  ``optim.PretrainSchedule`` as the hook of ``train.PretrainStep`` against the rates the
    reference function produced (tests/golden/pretrain_lr.npz), through the step object itself."""
    import numpy as np
    import torch
    from madrigal_amd.optim import PretrainSchedule, adjust_learning_rate
    from madrigal_amd.train import PretrainStep
    g = golden("pretrain_lr")

    class Toy(torch.nn.Module):                    # the step only needs model(...) -> (_, _, (_, _, loss)) and .base_encoder
        def __init__(self):
            super().__init__()
            self.base_encoder = torch.nn.Linear(3, 3)
            self.head = torch.nn.Linear(3, 1)

        def forward(self, drug_indices, m1, m2, hard, data):
            return None, None, (None, None, self.head(self.base_encoder(data)).pow(2).mean())
    for tag in "abc":
        lr, warm, total, ipe = (float(v) for v in g[f"{tag}_cfg"])
        want = g[f"{tag}_lrs"]
        model = Toy()
        opt = torch.optim.AdamW([{"params": model.base_encoder.parameters(), "lr": 9.0}, {"params": model.head.parameters(), "lr": 7.0}])
        step = PretrainStep(model, opt, scheduler=PretrainSchedule(lr, warm, total, int(ipe)))
        seen = []
        for _ in range(int(total) * int(ipe)):
            step.step(None, None, None, None, torch.ones(2, 3))
            seen.append([step.scheduler.last_lr] + [q["lr"] for q in opt.param_groups])
        assert np.allclose(np.asarray(seen), want, rtol=1e-13, atol=0), tag
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    assert adjust_learning_rate(opt, 0.5, 2e-3, 1, 4) == 1e-3 and opt.param_groups[0]["lr"] == 1e-3


def test_average_precision_matches_sklearn():
    import numpy as np
    import torch
    from sklearn.metrics import average_precision_score
    from madrigal_amd.metrics import average_precision, macro_auprc
    rng = np.random.default_rng(0)
    for n, ties in ((1, False), (50, False), (5000, False), (5000, True)):
        s = rng.standard_normal(n).astype(np.float32)
        if ties:
            s = np.round(s, 1)
        y = (rng.random(n) < 0.3).astype(np.float32)
        if y.sum() == 0:
            y[0] = 1
        got = float(average_precision(torch.from_numpy(s), torch.from_numpy(y)))
        assert abs(got - average_precision_score(y, s)) < 1e-12, (n, ties)
    T, L = 20000, 12
    lab = rng.integers(0, L, T)
    s = rng.standard_normal(T).astype(np.float32)
    y = (rng.random(T) < 0.2).astype(np.float32)
    y[lab == 3] = 1.0                                            # an outcome with one class only: skipped, as the reference does
    macro, per = macro_auprc(torch.from_numpy(s), torch.from_numpy(y), torch.from_numpy(lab), L)
    want = [average_precision_score(y[lab == l], s[lab == l]) for l in range(L) if l != 3]
    assert abs(float(macro) - float(np.mean(want))) < 1e-12 and bool(torch.isnan(per[3]))


def test_parameter_groups_match_reference_create_optimizer(golden):
    """Learning rate and weight decay of every parameter as the reference's create_optimizer assigns them on its own model
    (same state_dict names), including the parameters it leaves out of every group."""
    import math
    import numpy as np
    import torch
    from madrigal_amd import data as D, models as M
    from madrigal_amd.optim import parameter_groups
    from helpers import ENCODE_CASES
    from test_models_gpu import build_model
    g = golden("param_groups")
    hp = dict(optimizer="adamw", structure_encoder_lr=1e-3, kg_encoder_lr=2e-3, perturb_encoders_lr=3e-3, fusion_lr=4e-3, decoder_lr=5e-3,
              wd=0.25, beta1=0.9, beta2=0.999, eps=1e-8)
    for case in (ENCODE_CASES[1], ENCODE_CASES[2]):
        batch, bkg = D.make_batch(14, 61, kg_nodes=300, kg_edges=2500, masks=D.make_masks(14, 61))
        model = build_model(M, case, bkg["data"], 6)
        want = {str(n): tuple(v) for n, v in zip(g[case[0] + "_names"], g[case[0] + "_lr_wd"])}
        groups = parameter_groups(model, hp)
        got = {}
        name_of = {id(p): k for k, p in model.named_parameters()}
        for grp in groups:
            for p in grp["params"]:
                assert name_of[id(p)] not in got
                got[name_of[id(p)]] = (grp["lr"], grp["weight_decay"])
        assert set(want) == set(name_of.values()), set(want) ^ set(name_of.values())
        for k, (lr, wd) in want.items():
            if math.isnan(lr):
                assert k not in got, k                       # the reference never updates these (cls / bottleneck tokens)
            else:
                assert got[k] == (lr, wd), (k, got.get(k), (lr, wd))
        with_tokens = {name_of[id(p)] for grp in parameter_groups(model, hp, include_learned_tokens=True) for p in grp["params"]}
        assert with_tokens == set(name_of.values())


# ---------------------------------------------------------------------------------------------- contrastive pretraining as shipped
def test_pretrain_view_sampler_matches_reference(golden):
    """madrigal_amd.masks.get_pretrain_masks / pretrain_modality_subset_sampler against the reference's own functions
    (madrigal/utils.py:51-145, 360-390) under the same numpy / torch seeds; the two modes whose bank the reference itself
    cannot construct are refused."""
    import torch
    from madrigal_amd import masks as MK
    g = golden("pretrain_views")
    avail, drugs = g["avail"], g["drugs"].tolist()
    for mode in ("str_center_uni", "double_random", "str_kg"):
        for unb in (False, True):
            tag = f"{mode}_{int(unb)}"
            bank = MK.get_pretrain_masks(drugs, avail.copy(), mode, unb, 0.2)
            np.random.seed(123)
            torch.manual_seed(123)
            order = g[tag + "_order"].tolist()
            a1, a2 = MK.pretrain_modality_subset_sampler([bank[d] for d in order], mode, unb)
            b1, b2 = MK.pretrain_modality_subset_sampler([bank[d] for d in order], mode, unb)
            for got, key in ((a1, "_aug1"), (a2, "_aug2"), (b1, "_aug1b"), (b2, "_aug2b")):
                assert got.dtype == torch.bool and np.array_equal(got.numpy(), g[tag + key]), (tag, key)
    bank = MK.get_pretrain_masks(drugs, avail.copy(), "str_center_uni", False, 0.2)
    assert np.allclose(bank[100][1], g["uni_probs_d100"], rtol=1e-12)
    # the batched sampler consumes numpy's global stream exactly like the reference's per-drug loop
    fast = MK.StrCenterUniSampler(bank)
    np.random.seed(123)
    order = g["str_center_uni_0_order"].tolist()
    for tag in ("", "b"):
        a1, a2 = fast(order)
        assert np.array_equal(a1.numpy(), g["str_center_uni_0_aug1" + tag]) and np.array_equal(a2.numpy(), g["str_center_uni_0_aug2" + tag])
    # 'str_center_uni': the structure alone / exactly one other modality the drug owns
    a1, a2 = t(g["str_center_uni_0_aug1"]), t(g["str_center_uni_0_aug2"])
    assert bool((~a1).sum(1).eq(1).all()) and bool((~a1[:, 0]).all()) and bool((~a2).sum(1).eq(1).all()) and bool(a2[:, 0].all())
    for mode in ("str_center", "str_center_comb"):
        for unb in (0, 1):
            assert f"{mode}_{unb}_error" in g.files                      # the reference raises there
            with pytest.raises(NotImplementedError):
                MK.get_pretrain_masks(drugs, avail.copy(), mode, bool(unb), 0.2)


def simclr_shapes(kg, shared, mlp_dim=512):
    from helpers import model_shapes_for_case
    from oracle.gen_cases import CL_CASE
    s = {"base_encoder." + k[len("encoder."):]: v for k, v in model_shapes_for_case(CL_CASE, kg, 4).items() if k.startswith("encoder.")}
    for pred in (("predictor",) if shared else ("predictor_1", "predictor_2")):
        s.update({f"{pred}.0.weight": (mlp_dim, 128), f"{pred}.1.weight": (mlp_dim,), f"{pred}.1.bias": (mlp_dim,),
                  f"{pred}.1.running_mean": (mlp_dim,), f"{pred}.1.running_var": (mlp_dim,), f"{pred}.1.num_batches_tracked": (),
                  f"{pred}.3.weight": (128, mlp_dim), f"{pred}.4.running_mean": (128,), f"{pred}.4.running_var": (128,),
                  f"{pred}.4.num_batches_tracked": ()})
    return s


@pytest.mark.parametrize("shared,basal", [(False, False), (False, True), (True, False), (True, True)])
def test_simclr_raw_encoder_output_as_shipped(golden, shared, basal):
    """BASELINE configs[2] as the reference ships it (configs/cl_pretrain/*.yaml: raw_encoder_output, str_center_uni):
    the restated SimCLR forward against the reference's own SimCLR_NovelDDI.forward outputs."""
    from oracle.pipeline import oracle_simclr
    g = golden("simclr_raw")
    n, seed = (int(v) for v in g["meta"])
    batch, bkg = D.make_batch(n, seed, kg_nodes=300, kg_edges=2500, masks=t(g["avail"]))
    tag = f"s{int(shared)}b{int(basal)}"
    shapes = simclr_shapes(bkg["data"], shared)
    assert sorted(shapes) == list(g[tag + "_keys"])
    p = det_state_dict(seed, shapes)
    got = oracle_simclr(p, batch, bkg, t(g["mask1"]), t(g["mask2"]), t(g["hard"]), float(g["T"]), t(g["kg_filler"]),
                        shared_predictor=shared, use_tx_basal=basal)
    assert got["raw1"].shape == (n, 128) and got["raw2"].shape == (n, 128)          # one row per drug and view
    for k in ("raw1", "raw2", "aug1", "aug2"):
        assert rel_err(got[k], g[f"{tag}_{k}"]) < 5e-5, k
    keep = np.abs(g[tag + "_logits"]) < 1e8
    assert rel_err(got["logits"].numpy()[keep], g[tag + "_logits"][keep]) < 5e-5
    assert abs(float(got["loss"]) - float(g[tag + "_loss"])) < 2e-5 * abs(float(g[tag + "_loss"]))


def test_gin_edge_bias_enters_once_per_atom():
    """torchdrug 0.2.1 runs GraphIsomorphismConv.message_and_aggregate: bond features summed per atom, edge_linear applied
    once (bias once, isolated atoms included).  The per-edge reading (rounds 1-3) differs by exactly (deg - 1) b."""
    from helpers import gin_bias_case
    mols, p, b, deg = gin_bias_case()
    kw = dict(num_layers=1, num_mlp_layer=1, batch_norm=False, readout="sum")
    a = O.gin_forward(p, mols.node_feature, mols.edge_list, mols.edge_feature, mols.node2graph, 1, **kw)["node_feature"]
    e = O.gin_forward(p, mols.node_feature, mols.edge_list, mols.edge_feature, mols.node2graph, 1, edge_bias="per_edge", **kw)["node_feature"]
    assert float(deg[0]) == 3 and float(deg[4]) == 0
    assert torch.allclose(e - a, (deg - 1).unsqueeze(1) * b.unsqueeze(0), atol=1e-6)
    # the isolated atom: (1 + eps) h + W_e 0 + b
    assert torch.allclose(a[4], mols.node_feature[4] + b, atol=1e-6)
    # the degree-3 atom by hand
    src, dst = mols.edge_list[:, 0], mols.edge_list[:, 1]
    inc = dst == 0
    want = mols.node_feature[0] + mols.node_feature[src[inc]].sum(0) + p["layers.0.edge_linear.weight"] @ mols.edge_feature[inc].sum(0) + b
    assert torch.allclose(a[0], want, atol=1e-6)
