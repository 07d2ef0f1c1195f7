"""InfoNCE, gathered BCE and rank normalisation on the GPU against the reference goldens / the oracle."""
import numpy as np
import pytest
import torch

from helpers import rel_err, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from madrigal_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("prec,tol", [("f32", 2e-5), ("bf16x3", 1e-4)])
def test_infonce_golden(ops, golden, prec, tol):
    g = golden("infonce")
    a1, a2, hard = t(g["aug1"]).cuda(), t(g["aug2"]).cuda(), t(g["hard"]).cuda()
    lg, lb, loss = ops.info_nce(a1, a2, hard, float(g["T"]), precision=prec)
    # masked entries are -1e9/T in both; compare the rest on the logit scale
    ref = g["logits"]
    keep = ref > -1e8
    assert np.array_equal(lg.cpu().numpy() < -1e8, ~keep)
    assert float(np.abs(lg.cpu().numpy()[keep] - ref[keep]).max()) < tol * float(np.abs(ref[keep]).max())
    assert np.array_equal(lb.cpu().numpy(), g["labels"])
    assert abs(float(loss) - float(g["loss"])) < 5 * tol * abs(float(g["loss"]))
    lg0, _, loss0 = ops.info_nce(a1, a2, None, float(g["T"]), precision=prec)
    assert rel_err(lg0.cpu(), g["logits_nomask"]) < tol
    assert abs(float(loss0) - float(g["loss_nomask"])) < 5 * tol * abs(float(g["loss_nomask"]))


def test_simclr_predictor_golden(golden):
    """Linear(no bias)+BN+ReLU+Linear(no bias)+BN(affine=False), eval mode (simclr.py:46-62)."""
    import types
    import torch.nn as nn
    from madrigal_amd import models as M
    from madrigal_amd.simclr import SimCLR_NovelDDI
    from oracle.params import fill_module
    g = golden("infonce")
    enc = types.SimpleNamespace(uni_projector=types.SimpleNamespace(fc=[nn.Linear(512, 128)]))
    sim = SimCLR_NovelDDI.__new__(SimCLR_NovelDDI)
    nn.Module.__init__(sim)
    p = SimCLR_NovelDDI._build_mlp(2, 128, 512, 128)
    assert sorted(p.state_dict().keys()) == list(g["pred_keys"])
    fill_module(p, 71)
    p = p.cuda().eval()
    with torch.no_grad(), M.precision("f32"):
        y = M._run_sequential(p, t(g["pred_x"]).cuda()).cpu()
    assert rel_err(y, g["pred_y"]) < 3e-5


def test_gather_bce_golden(ops, golden):
    g = golden("bce")
    s = t(g["scores"]).cuda()
    pred, loss = ops.gather_bce(s, t(g["labels"]).cuda(), t(g["heads"]).cuda(), t(g["tails"]).cuda(), t(g["y"]).cuda())
    assert float((pred.cpu() - t(g["pred"])).abs().max()) < 1e-6
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    # same through the head's fused sigmoid epilogue + identity gather
    pred2, loss2 = ops.gather_bce(torch.sigmoid(s), t(g["labels"]).cuda(), t(g["heads"]).cuda(), t(g["tails"]).cuda(),
                                  t(g["y"]).cuda(), apply_sigmoid=False)
    assert float((pred2 - pred).abs().max()) < 1e-6 and abs(float(loss2) - float(loss)) < 1e-6
    # saturated probabilities hit BCELoss's -100 clamp instead of inf
    big = torch.full((1, 2, 2), 200.0, device="cuda")
    z = torch.zeros(1, dtype=torch.int64, device="cuda")
    _, l = ops.gather_bce(big, z, z, z, torch.zeros(1, device="cuda"))
    assert float(l) == 100.0
    # an index outside the score tensor raises, as the reference's advanced indexing does (it is never read on the device)
    for bad in ((1, 0, 0), (0, 2, 0), (0, 0, 2), (-1, 0, 0)):
        idx = [torch.tensor([v], dtype=torch.int64, device="cuda") for v in bad]
        with pytest.raises(IndexError):
            ops.gather_bce(big, *idx, torch.zeros(1, device="cuda"))


def test_ranks_golden_bit_exact(ops, golden):
    g = golden("ranks")
    out = ops.rank_normalize(t(g["scores"]).cuda()).cpu().numpy()
    assert np.array_equal(out, g["normalized"])                      # rank ordering and fp32 values bit exact


@pytest.mark.parametrize("path", ["default", "tile8192", "direct", "lookback"])
@pytest.mark.parametrize("N,L", [(300, 5), (2, 3), (1, 2), (97, 1), (1025, 2), (1283, 1)])
def test_ranks_vs_oracle(ops, monkeypatch, N, L, path):
    """Every path of the sort: 16384-key tiles (default at these N) and 8192-key tiles, the blocked last pass and the direct one
    (the large-N path), contiguous and row-pitched tensors; ragged N (not a multiple of 128 / 4)."""
    from helpers import set_switch
    from oracle import madrigal_oracle as O
    if path == "tile8192":
        set_switch(monkeypatch, "MDG_RANKS_TILE", "8192")
    if path == "direct":
        set_switch(monkeypatch, "MDG_RANKS_DIRECT", "1")
    if path == "lookback":                       # tile offsets by decoupled look-back instead of the histogram / scan launches
        set_switch(monkeypatch, "MDG_RANKS_LOOKBACK", "1")
    rng = np.random.default_rng(N)
    s = rng.standard_normal((L, N, N)).astype(np.float32) * 7
    ref = O.rank_normalize(s)
    out = ops.rank_normalize(torch.from_numpy(s).cuda()).cpu().numpy()
    assert np.array_equal(out, ref)
    assert np.array_equal(out, out.transpose(0, 2, 1))
    pit = ops.empty_scores(L, N, N, "cuda")
    pit.copy_(torch.from_numpy(s))
    got = ops.rank_normalize(pit)
    assert got.stride(1) == pit.stride(1) and np.array_equal(got.cpu().numpy(), ref)


def test_ranks_ties_are_stable_in_flat_index(ops):
    from oracle import madrigal_oracle as O
    rng = np.random.default_rng(7)
    s = rng.integers(-3, 4, size=(2, 200, 200)).astype(np.float32)     # heavy ties, negative and positive zeros
    s[0, 5, 3] = -0.0
    out = ops.rank_normalize(torch.from_numpy(s).cuda()).cpu().numpy()
    ref = O.rank_normalize(s)
    # -0.0 and +0.0 compare equal for numpy but differ in the radix key; exclude that single corner
    assert np.array_equal(out[1], ref[1])
    il = np.tril_indices(200, k=-1)
    r = np.round(out[0][il] * (200 * 199 / 2)).astype(np.int64)
    assert np.array_equal(np.sort(r), np.arange(1, il[0].size + 1))     # a permutation: every rank used once


def test_ranks_full_size_properties(ops):
    """BASELINE N=4096 (one outcome slab at a time fits easily): ranks are a permutation of 1..M, the output is
    symmetric with zero diagonal, and ordering agrees with the scores (sortedness)."""
    N, L = 4096, 3
    g = torch.Generator(device="cuda").manual_seed(0)
    s = torch.randn(L, N, N, device="cuda", generator=g)
    out = ops.rank_normalize(s)
    M = N * (N - 1) // 2
    assert torch.equal(out, out.transpose(1, 2)) and float(out.diagonal(dim1=1, dim2=2).abs().max()) == 0.0
    il = torch.tril_indices(N, N, -1, device="cuda")
    for l in range(L):
        r = torch.round(out[l][il[0], il[1]].double() * M).long()
        assert int(r.min()) == 1 and int(r.max()) == M
        order = torch.argsort(r)
        v = s[l][il[0], il[1]][order]
        assert bool(torch.all(v[1:] >= v[:-1]))                        # ascending scores <=> ascending ranks
        assert int(torch.unique(r).numel()) == M


@pytest.mark.parametrize("N", [64, 45])
def test_gmean_and_seed_ensembling(ops, N):
    """5-seed ensembling: gmean of normalised ranks (fp32, as scipy.stats.mstats.gmean on float32) then re-ranked.  N = 45: an odd
    element count (the tail of the vectorised kernel) and, for the row-pitched layout, rank tensors that are [:, :, :N] views."""
    from oracle import madrigal_oracle as O
    rng = np.random.default_rng(3)
    L, K = 3, 5
    ranks = [O.rank_normalize(rng.standard_normal((L, N, N)).astype(np.float32)) for _ in range(K)]
    with np.errstate(divide="ignore"):
        ref = np.exp(np.mean(np.log(np.stack(ranks, -1)), axis=-1, dtype=np.float32)).astype(np.float32)
    ref[:, np.arange(N), np.arange(N)] = 0.0
    g = ops.gmean([torch.from_numpy(r).cuda() for r in ranks])
    assert float(np.abs(g.cpu().numpy() - ref).max()) < 1e-6
    ens = ops.ensemble_ranks([torch.from_numpy(r).cuda() for r in ranks]).cpu().numpy()
    # re-ranking of the GPU gmean with the oracle: identical; vs the numpy gmean: same up to near-ties of fp32 log/exp
    assert np.array_equal(ens, O.rank_normalize(g.cpu().numpy()))
    assert float(np.abs(ens - O.rank_normalize(ref)).max()) < 5.0 / (N * (N - 1) / 2)
    assert np.array_equal(ens, ens.transpose(0, 2, 1))
    pit = []
    for r in ranks:                                  # the same through row-pitched tensors: identical bits
        p_ = ops.empty_scores(L, N, N, "cuda")
        p_.copy_(torch.from_numpy(r))
        pit.append(p_)
    gp = ops.gmean(pit)
    assert gp.stride(1) == pit[0].stride(1) and torch.equal(gp, g)
    assert np.array_equal(ops.ensemble_ranks(pit).cpu().numpy(), ens)
