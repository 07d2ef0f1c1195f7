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
    model = build_model(M, CASE, bkg["data"], L).cuda().train()
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


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from madrigal_amd import models as M
        from madrigal_amd.optim import AdamW
        from madrigal_amd.train import FinetuneStep
        n, L, seed = 101, 12, 21                       # odd drug count: uneven shards
        M.set_precision("f32")
        model, b, kgc, (lab, hd, tl, y), filler = _build(seed, n, L)
        fs = FinetuneStep(model, AdamW(model.parameters(), lr=1e-4, weight_decay=0.0), rank=rank, world=world)
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
        worst = (0.0, "")
        gmax = max(float(p.grad.abs().max()) for p in ref.parameters() if p.grad is not None)
        for k, p in ref.named_parameters():
            if p.grad is None:
                continue
            err = float((grads[k] - p.grad).abs().max()) / max(float(p.grad.abs().max()), 1e-2 * gmax)
            worst = max(worst, (err, k))
        berr = max(float((bufs[k] - v).abs().max()) / max(float(v.abs().max()), 1e-6) for k, v in ref.named_buffers() if "running" in k)
        ret[rank] = (abs(float(loss) - float(loss1)) / abs(float(loss1)), worst, berr, len(grads))
    finally:
        dist.destroy_process_group()


def test_two_rank_finetune_step_equals_single_process_step():
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    for r in range(2):
        lerr, worst, berr, n_grads = ret[r]
        assert lerr < 1e-5, (r, lerr)
        assert worst[0] < 2e-3, (r, worst)           # fp32 summation order + ReLU flips at rounding distance
        assert berr < 1e-4, (r, berr)                # BatchNorm running statistics = full-batch statistics on every rank
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
