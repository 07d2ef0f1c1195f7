"""madrigal_amd.ops.triple_plan (csrc/plan.hip: the index plumbing of the gathered head, rebuilt for every batch of labelled triples the
reference's loop feeds a step, train_ddi_batch.py:231-354) against its earlier construction from torch index / sort / scan calls
(tests/helpers.triple_plan_torch): every table of the plan, entry for entry."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _same(a, b, path="plan"):
    if isinstance(a, dict):
        assert isinstance(b, dict) and sorted(a) == sorted(b), (path, sorted(a), sorted(b) if isinstance(b, dict) else b)
        for k in a:
            _same(a[k], b[k], f"{path}.{k}")
    elif isinstance(a, (tuple, list)):
        assert isinstance(b, (tuple, list)) and len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f"{path}[{i}]")
    elif torch.is_tensor(a):
        assert torch.is_tensor(b) and a.dtype == b.dtype and a.shape == b.shape, (path, a.dtype, b.dtype if torch.is_tensor(b) else b, a.shape)
        assert torch.equal(a, b), path
    else:
        assert a == b, (path, a, b)


def _triples(T, L, n_head, n_tail, seed, skew=False):
    g = torch.Generator().manual_seed(seed)
    lab = torch.randint(0, L, (T,), generator=g)
    if skew:                                                    # a few drugs own most triples (lists beyond 256 entries: the two-level sums)
        hd = (torch.rand(T, generator=g) ** 4 * n_head).long().clamp(max=n_head - 1)
        tl = (torch.rand(T, generator=g) ** 4 * n_tail).long().clamp(max=n_tail - 1)
    else:
        hd = torch.randint(0, n_head, (T,), generator=g)
        tl = torch.randint(0, n_tail, (T,), generator=g)
    return lab.cuda(), hd.cuda(), tl.cuda()


@pytest.mark.parametrize("T,L,n_head,n_tail,skew", [
    (5000, 7, 50, 61, False),          # labels of a few hundred triples: several tiles and chunks each
    (3000, 40, 300, 300, False),       # some labels empty, most drugs with one or two triples
    (20000, 5, 10, 12, True),          # drug lists of thousands of entries: head / tail pieces
    (60000, 400, 3, 3, True),          # ... and more than 256 (label, head) pairs per head drug: the pair table's drug pieces
    (1, 3, 4, 4, False),
    (0, 3, 4, 4, False),
    (300000, 896, 4096, 4096, False),  # the bench step's shape at a twentieth of its triples
])
def test_triple_plan_matches_its_torch_construction(T, L, n_head, n_tail, skew):
    from madrigal_amd import ops
    from helpers import triple_plan_torch
    lab, hd, tl = _triples(T, L, n_head, n_tail, seed=T + L, skew=skew)
    if T >= 3000 and not skew:
        lab[lab == 2] = 3                                       # an empty label in the middle
    got = ops.triple_plan(lab, hd, tl, L, n_head, n_tail)
    ref = triple_plan_torch(lab, hd, tl, L, n_head, n_tail)
    _same(ref, got)
    if skew:
        assert got["head_pieces"] is not None and got["tail_pieces"] is not None      # (the two-level sums are exercised)


def test_triple_plan_rejects_indices_outside_the_tables():
    from madrigal_amd import ops
    lab, hd, tl = _triples(1000, 5, 20, 20, seed=1)
    bad = lab.clone(); bad[17] = 5
    with pytest.raises(ValueError, match="labels"):
        ops.triple_plan(bad, hd, tl, 5, 20, 20)
    bad = lab.clone(); bad[3] = -1
    with pytest.raises(ValueError, match="labels"):
        ops.triple_plan(bad, hd, tl, 5, 20, 20)
    bad = hd.clone(); bad[0] = 20
    with pytest.raises(ValueError, match="heads / tails"):
        ops.triple_plan(lab, bad, tl, 5, 20, 20)
    bad = tl.clone(); bad[999] = -3
    with pytest.raises(ValueError, match="heads / tails"):
        ops.triple_plan(lab, hd, bad, 5, 20, 20)
