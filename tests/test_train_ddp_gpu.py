"""Data-parallel finetune step: two ranks (gloo, both on the one card of the test box) against the single-process step.

With dropout switched off the sharded step is the same function as the single-GPU step: SyncBatchNorm reproduces the
full-batch statistics, the all-gather / reduce-scatter pair and the flat gradient all-reduce reproduce the full gradient.
(RCCL itself is exercised by bench.py --gpus N on the multi-GPU node; here the collectives run over gloo.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CASE = ("twosides321", "transformer_uni_proj", 2, "sinusoidal", 4, 64, 256, 2, True, "x-attn", False, False)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(seed, n, L):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from madrigal_amd import data as D, models as M
    from test_models_gpu import build_model
    masks = D.make_masks(n, seed)
    batch, bkg = D.make_batch(n, seed, kg_nodes=700, kg_edges=9000, masks=masks)
    torch.manual_seed(seed)
    from helpers import smooth_relu
    # GELU in place of every ReLU (helpers.smooth_relu): under SyncBatchNorm's other summation order a ReLU at rounding distance of zero
    # flips its derivative and moves every gradient upstream of it by percents -- not what these comparisons are about
    model = smooth_relu(build_model(M, CASE, bkg["data"], L)).cuda().train()
    for mod in model.modules():                       # no dropout: the sharded and the full step are the same function
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    b = D.batch_to(batch, "cuda")
    kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
    trip = tuple(t.cuda() for t in D.make_labelled_triples(n, L, 400, seed))
    filler = torch.randn(n, 128, generator=torch.Generator().manual_seed(1)).cuda()
    return model, b, kgc, trip, filler


def _worker(rank, world, port, ret, shard_kg=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from madrigal_amd import models as M
        from madrigal_amd.optim import AdamW
        from madrigal_amd.train import FinetuneStep
        n, L = 101, 12                                 # odd drug count: uneven shards
        M.set_precision("f32")
        tried = []
        # every seed of a fixed range (helpers.assert_tensors_agree explains the rule; the loop lives inside the ranks so that the
        # process group is set up once)
        for seed in (21, 22, 23):
            model, b, kgc, (lab, hd, tl, y), filler = _build(seed, n, L)
            fs = FinetuneStep(model, AdamW(model.parameters(), lr=1e-4, weight_decay=0.0), rank=rank, world=world, shard_kg=shard_kg)
            model.zero_grad(set_to_none=True)
            loss = fs.accumulate(b, b, b["masks"], b["masks"], kgc, lab, hd, tl, y, kg_filler=filler)
            from madrigal_amd.parallel import allreduce_gradients
            allreduce_gradients(model.parameters())
            grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
            bufs = {k: v.detach().clone() for k, v in model.named_buffers() if "running" in k}
            # the same step in one process (fresh identical model)
            ref, b2, kgc2, _, _ = _build(seed, n, L)
            fs1 = FinetuneStep(ref, AdamW(ref.parameters(), lr=1e-4, weight_decay=0.0))
            ref.zero_grad(set_to_none=True)
            loss1 = fs1.accumulate(b2, b2, b2["masks"], b2["masks"], kgc2, lab, hd, tl, y, kg_filler=filler)
            emax, el2 = [], []
            gmax = max(float(p.grad.abs().max()) for p in ref.parameters() if p.grad is not None)
            for k, p in ref.named_parameters():
                if p.grad is None:
                    continue
                emax.append((float((grads[k] - p.grad).abs().max()) / max(float(p.grad.abs().max()), 1e-2 * gmax), k))
                # per tensor in the 2-norm: a ReLU whose pre-activation sits at rounding distance of zero flips its derivative when the
                # SyncBatchNorm sums are formed in another order (three ranks instead of one) and moves a handful of entries by percents
                el2.append((float((grads[k] - p.grad).norm()) / max(float(p.grad.norm()), 1e-3 * gmax * p.grad.numel() ** 0.5), k))
            berr = max(float((bufs[k] - v).abs().max()) / max(float(v.abs().max()), 1e-6) for k, v in ref.named_buffers() if "running" in k)
            kg_top = max(float(p.grad.norm()) for k, p in ref.named_parameters() if p.grad is not None and "kg_encoder" in k)
            kg_l2 = max((float((grads[k] - p.grad).norm()) / max(float(p.grad.norm()), 1e-2 * kg_top), k) for k, p in ref.named_parameters() if p.grad is not None and "kg_encoder" in k)
            tried.append((seed, abs(float(loss) - float(loss1)) / abs(float(loss1)), emax, el2, kg_l2, berr, len(grads)))
        ret[rank] = tried
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shard_kg", [(2, False), (2, True), (3, False), (3, True)],
                         ids=["2ranks_kg_replicated", "2ranks_kg_partitioned", "3ranks_kg_replicated", "3ranks_kg_partitioned"])
def test_two_rank_finetune_step_equals_single_process_step(world, shard_kg):
    """``shard_kg``: the KG encoder's convs destination-partitioned over the ranks in TRAINING (every KG edge attended to on one
    rank; one all-gather per conv forward, the reverse exchange backward) against the single-process step, as the replicated
    KG encoder is: same loss, same summed gradients, same BatchNorm statistics."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret, shard_kg)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    for r in range(world):
        tried = ret[r]
        assert len(tried) == 3
        for seed, lerr, emax, el2, kg_l2, berr, n_grads in tried:
            assert lerr < 1e-5, (r, seed, lerr)
            # EVERY seed, at any world size: every gradient tensor within 2e-3 (max norm) / 5e-3 (2-norm) of the single-process step
            assert max(emax)[0] < 2e-3 and max(el2)[0] < 5e-3, (r, seed, max(emax), max(el2))
            assert kg_l2[0] < 1e-4, (r, seed, kg_l2)       # the KG encoder's own gradients (per tensor, floored at 1 % of the largest): 4e-6 in either variant
            assert berr < 1e-4, (r, seed, berr)            # BatchNorm running statistics = full-batch statistics on every rank
            assert n_grads > 150


def _pretrain_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from madrigal_amd import data as D, models as M
        from madrigal_amd.optim import AdamW
        from madrigal_amd.simclr import SimCLR_NovelDDI
        from madrigal_amd.train import PretrainStep
        from test_models_gpu import build_model
        M.set_precision("f32")
        case = ("twosides105", "transformer_uni_proj", 2, "learnable", 2, 64, 128, 1, True, "x-attn", True, False)
        n, seed = 75, 12

        def build():
            torch.manual_seed(seed)
            masks = D.make_masks(n, seed)
            batch, bkg = D.make_batch(n, seed, kg_nodes=600, kg_edges=6000, masks=masks)
            enc = build_model(M, case, bkg["data"], 4).encoder
            model = SimCLR_NovelDDI(enc, dim=128, mlp_dim=256, T=0.5, raw_encoder_output=False).cuda().train()
            for mod in model.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
                if isinstance(mod, torch.nn.MultiheadAttention):
                    mod.dropout = 0.0
            b = D.batch_to(batch, "cuda")
            kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
            m2 = b["masks"].clone()
            m2[:, 1:] = True
            hard = torch.rand(n, n, generator=torch.Generator().manual_seed(3)) < 0.03
            hard = ((hard | hard.T) & ~torch.eye(n, dtype=torch.bool)).cuda()
            return model, b, kgc, b["masks"].clone(), m2, hard
        model, b, kgc, m1, m2, hard = build()
        before = {k: p.detach().clone() for k, p in model.named_parameters()}
        step = PretrainStep(model, AdamW(model.parameters(), lr=1e-3, weight_decay=0.0), rank=rank, world=world)
        loss = step.step(b["drugs"], m1, m2, hard, (b["strs"], kgc, b["cv"], b["tx"]))
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        ref, b2, kgc2, m1b, m2b, hard2 = build()
        step1 = PretrainStep(ref, AdamW(ref.parameters(), lr=1e-3, weight_decay=0.0))
        loss1 = step1.step(b2["drugs"], m1b, m2b, hard2, (b2["strs"], kgc2, b2["cv"], b2["tx"]))
        gmax = max(float(p.grad.abs().max()) for p in ref.parameters() if p.grad is not None)
        worst = (0.0, "")
        for k, p in ref.named_parameters():
            if p.grad is None:
                continue
            err = float((grads[k] - p.grad).abs().max()) / max(float(p.grad.abs().max()), 1e-2 * gmax)
            worst = max(worst, (err, k))
        ret[rank] = (abs(float(loss) - float(loss1)) / abs(float(loss1)), worst, len(grads))
    finally:
        dist.destroy_process_group()


def test_two_rank_pretraining_step_equals_single_process_step():
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_pretrain_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    for r in range(2):
        lerr, worst, n_grads = ret[r]
        assert lerr < 1e-5, (r, lerr)
        assert worst[0] < 2e-3, (r, worst)
        assert n_grads > 100


def _pretrain_raw_worker(rank, world, port, ret):
    """BASELINE configs[2] as shipped (raw_encoder_output=True, 'str_center_uni' views), data-parallel: four iterations with a
    FRESH batch and a fresh view draw each (what pretrain.py's DataLoader hands over), against the single-process steps."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        import numpy as np
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from madrigal_amd import data as D, masks as MK, models as M
        from madrigal_amd.optim import AdamW
        from madrigal_amd.train import PretrainStep
        from test_pretrain_gpu import _build, _no_dropout, _views
        M.set_precision("f32")
        n = 67                                            # odd batch: uneven drug blocks

        def compare(seed):
            avail, _, _ = _views(n, seed)
            bank = MK.get_pretrain_masks(list(range(n)), avail.numpy().astype(np.int64), "str_center_uni", False, 0.2)
            _, bkg0 = D.make_batch(n, seed, kg_nodes=600, kg_edges=6000, masks=avail)

            def run(rank_, world_):
                torch.manual_seed(seed)
                np.random.seed(seed)
                from helpers import smooth_relu
                model = smooth_relu(_no_dropout(_build(M, bkg0["data"], False, True, mlp_dim=256, T=0.5))).cuda().train()     # (GELU for ReLU: see _build)
                # eps well above the gradients' rounding noise: Adam's first steps otherwise move every entry by +-lr whatever
                # its size, so entries whose gradient is noise would take opposite steps in the two runs
                step = PretrainStep(model, AdamW(model.parameters(), lr=1e-3, weight_decay=1e-2, eps=1e-3), rank=rank_, world=world_)
                kg_dev = bkg0["data"].to("cuda")
                losses, mem = [], []
                for it in range(6):
                    batch, bkg = D.make_batch(n, seed + it, kg=bkg0["data"], masks=avail)          # a different batch every iteration
                    b = D.batch_to(batch, "cuda")
                    kgc = {"data": kg_dev, "drug_index_map": bkg["drug_index_map"].cuda()}
                    m1, m2 = MK.pretrain_modality_subset_sampler([bank[d] for d in range(n)], "str_center_uni", False)
                    hard = torch.rand(n, n, generator=torch.Generator().manual_seed(it)) < 0.03
                    hard = ((hard | hard.T) & ~torch.eye(n, dtype=torch.bool)).cuda()
                    losses.append(float(step.step(b["drugs"], m1.cuda(), m2.cuda(), hard, (b["strs"], kgc, b["cv"], b["tx"]))))
                    if it == 0:                       # gradients of the FIRST step: same weights in both runs
                        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
                    del batch, b, kgc, m1, m2, hard
                    torch.cuda.synchronize()
                    mem.append(torch.cuda.memory_allocated())
                params = {k: p.detach().clone() for k, p in model.named_parameters()}
                return losses, mem, grads, params
            l2, mem2, g2, p2 = run(rank, world)
            l1, _, g1, p1 = run(0, 1)
            gmax = max(float(v.abs().max()) for v in g1.values())
            # per tensor: relative L2 error (a ReLU whose pre-activation sits within fp32 rounding of zero may take the other branch
            # under SyncBatchNorm's different summation order: one atom's term moves in a few entries) and the max-norm error
            gl2 = max(float(v.norm()) for v in g1.values())
            el2 = [(float((g2[k] - v).norm()) / max(float(v.norm()), 1e-2 * gl2), k) for k, v in g1.items()]
            emax = [(float((g2[k] - v).abs().max()) / max(float(v.abs().max()), 1e-2 * gmax), k) for k, v in g1.items()]
            worst = (el2, emax)
            # the same set of parameters received a gradient (the fusion transformer etc. stay grad=None on every rank, so
            # weight decay leaves them untouched exactly as in the single-process step)
            untouched = max(float((p2[k] - p1[k]).abs().max()) for k in p1 if k not in g1)
            return ([abs(a - b) / abs(b) for a, b in zip(l2, l1)], worst, set(g2) == set(g1), untouched, mem2)

        # every seed of a fixed range, judged by helpers.assert_tensors_agree in the parent (no search)
        ret[rank] = [compare(seed) for seed in (12, 13, 14)]
    finally:
        dist.destroy_process_group()


def test_two_rank_shipped_pretraining_steps_equal_single_process_and_hold_no_batches():
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_pretrain_raw_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
        assert p.exitcode == 0
    for r in range(2):
      for si, (lerr, worst, same_set, untouched, mem) in enumerate(ret[r]):
        # the first step is the same function of the same weights (fp32 summation order only); later losses also carry the
        # three AdamW updates in between, whose per-entry normalisation amplifies rounding-level gradient differences
        # (iterations 5 and 6 only feed the memory check below).  EVERY seed: the loss bounds, and every gradient tensor of the first
        # step within 5e-3 (2-norm) / 5e-2 (max norm) of the single-process step
        assert lerr[0] < 1e-5 and max(lerr[:4]) < 2e-3, (r, si, lerr)
        assert max(worst[0])[0] < 5e-3 and max(worst[1])[0] < 5e-2, (r, si, max(worst[0]), max(worst[1]))
        assert same_set and untouched == 0.0, (r, same_set, untouched)
        # nothing of an earlier iteration's batch stays allocated.  A rank's share of one batch is > 2 MB (tx signatures alone:
        # 34 drugs x 16 x 978 floats), so holding batches would add > 8 MB between iterations 2 and 6; what is allowed to move
        # is the one-entry plan caches (sized by the LAST batch's atom count) and the allocator's block rounding (~1 MB)
        assert abs(mem[5] - mem[1]) < (3 << 20) and max(mem[1:]) - min(mem[1:]) < (4 << 20), (r, mem)


def _kg_shard_worker(rank, world, port, ret):
    """Multi-GPU inference encode (pipeline.generate_embeddings): drugs sharded by rank, the KG encoder destination-partitioned
    (every rank computes its block of every node type, one all-gather per conv) -- against the single-process encode."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from madrigal_amd import data as D, models as M
        from madrigal_amd.pipeline import generate_embeddings
        from test_models_gpu import build_model
        n, seed = 83, 5                                              # uneven drug blocks; node-type sizes not divisible by 3
        batch, bkg = D.make_batch(n, seed, kg_nodes=1501, kg_edges=30000)
        torch.manual_seed(seed)
        model = build_model(M, CASE, bkg["data"], 4).cuda().eval()
        b = D.batch_to(batch, "cuda")
        kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
        filler = torch.randn(n, 128, generator=torch.Generator().manual_seed(1)).cuda()
        with torch.no_grad(), M.precision("bf16x3"):
            z_ref = generate_embeddings(model, b, kgc, kg_filler=filler)                       # one process, everything
            z = generate_embeddings(model, b, kgc, rank=rank, world=world, kg_filler=filler)
            os.environ["MDG_SHARD_KG"] = "0"
            z_rep = generate_embeddings(model, b, kgc, rank=rank, world=world, kg_filler=filler)  # KG encoder replicated
            # the conv on its own: partitioned rows == unpartitioned rows, every node type
            conv = model.encoder.kg_encoder.convs[0]
            full = conv(kgc["data"].x_dict, kgc["data"].edge_index_dict)
            part = conv(kgc["data"].x_dict, kgc["data"].edge_index_dict, shard=(rank, world, None))
        same_conv = all(torch.equal(full[t], part[t]) for t in full) and set(full) == set(part)
        # sharded drugs change the row counts of the per-drug GEMMs (tile shapes, hence fp32 summation grouping): against the
        # single-process encode the comparison is at rounding level; partitioned vs replicated KG encoder is bit for bit
        err = float((z - z_ref).abs().max() / z_ref.abs().max())
        ret[rank] = (bool(torch.equal(z, z_rep)), err, same_conv, tuple(z.shape))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_destination_partitioned_kg_encoder_equals_replicated(world):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_kg_shard_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    for r in range(world):
        same_as_replicated, err, same_conv, shape = ret[r]
        assert shape == (83, 128)
        assert same_conv and same_as_replicated and err < 1e-5, (r, ret[r])
