"""World-size-2 gloo tests of the multi-GPU sharding logic (CPU; the head itself needs the GPU, so
the per-rank compute here is the oracle -- test infrastructure standing in for the kernel)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from madrigal_amd.parallel import shard_range, shard_sizes


def test_shard_ranges_cover():
    for n in (0, 1, 7, 8, 4096, 4003):
        for w in (1, 2, 3, 8):
            ranges = [shard_range(n, r, w) for r in range(w)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            assert max(shard_sizes(n, w)) - min(shard_sizes(n, w)) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from madrigal_amd.parallel import all_gather_rows
    from oracle import madrigal_oracle as O
    g = torch.Generator().manual_seed(0)
    z = torch.randn(n, 128, generator=g)
    lo, hi = shard_range(n, rank, world)
    z_full = all_gather_rows(z[lo:hi].clone(), n, rank, world)
    ok = torch.equal(z_full, z)
    # outcome-sharded head: rank r scores its own outcomes against all pairs
    L = 4
    gw = torch.Generator().manual_seed(7)
    w = torch.randn(L, 128, 128, generator=gw)
    llo, lhi = shard_range(L, rank, world)
    mine = O.bilinear_scores(z_full, z_full, w[llo:lhi])
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    full = torch.cat(parts, dim=0)
    ok = ok and torch.allclose(full, O.bilinear_scores(z, z, w), rtol=1e-6, atol=1e-5)
    ret[rank] = ok
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [64, 37])
def test_gloo_world2_allgather_and_outcome_shards(n):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret[0] and ret[1]


def _enc_worker(rank, world, port, n, ret):
    """Drug-sharded generate_embeddings == full-batch result.  The per-drug 'encoder' here is the oracle's GIN
    read-out (CPU test infrastructure standing in for the HIP encoder): it exercises slice_batch /
    slice_molecules re-indexing and the uneven all-gather."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from madrigal_amd import data as D
    from madrigal_amd.pipeline import generate_embeddings
    from oracle import madrigal_oracle as O
    from oracle.params import det_state_dict
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__)))
    from helpers import gin_shapes
    p = det_state_dict(3, gin_shapes())

    class Enc(torch.nn.Module):
        def forward(self, drugs, masks, mols, kg, cv, tx, kg_filler=None):
            g = O.gin_forward(p, mols.node_feature, mols.edge_list, mols.edge_feature, mols.node2graph, mols.batch_size,
                              num_layers=4, num_mlp_layer=3)["graph_feature"]
            return g + cv[:, :128] + tx["a375"]["sigs"][:, :128] + drugs.float().unsqueeze(1) * 1e-3

    class Model:
        encoder = Enc()
    batch, bkg = D.make_batch(n, 5, kg_nodes=100, kg_edges=300)
    full = Model.encoder(batch["drugs"], batch["masks"], batch["strs"], bkg, batch["cv"], batch["tx"])
    z = generate_embeddings(Model, batch, bkg, rank=rank, world=world)
    ret[rank] = bool(torch.allclose(z, full, rtol=1e-5, atol=1e-5)) and z.shape == full.shape
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [10, 7])
def test_gloo_world2_sharded_encode_matches_full_batch(n):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_enc_worker, args=(r, 2, port, n, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret[0] and ret[1]


def _grad_worker(rank, world, port, ret):
    """Collective autograd helpers of the data-parallel finetune step, on CPU tensors over gloo."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from madrigal_amd.parallel import all_gather_rows_grad, all_reduce_sum_, allreduce_gradients
    n = 11
    g = torch.Generator().manual_seed(0)
    z = torch.randn(n, 8, generator=g)
    wts = torch.randn(world, n, 8, generator=g)              # rank r's loss = sum(w_r * z_full)
    lo, hi = shard_range(n, rank, world)
    local = z[lo:hi].clone().requires_grad_(True)
    full = all_gather_rows_grad(local, n, rank, world)
    ok = torch.equal(full.detach(), z)
    (full * wts[rank]).sum().backward()
    ok = ok and torch.allclose(local.grad, wts.sum(0)[lo:hi])          # reduce-scatter: every rank's gradient of my rows
    # flat-bucket all-reduce of parameter gradients, with a parameter that has no gradient on one rank
    ps = [torch.nn.Parameter(torch.zeros(s)) for s in ((3, 5), (7,), (2, 2, 2))]
    for i, p in enumerate(ps):
        if not (i == 1 and rank == 1):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    allreduce_gradients(ps, bucket_bytes=64)
    for i, p in enumerate(ps):
        want = sum(float(r + 1) * (i + 1) for r in range(world) if not (i == 1 and r == 1))
        ok = ok and torch.allclose(p.grad, torch.full_like(p, want))
    t = torch.tensor([float(rank + 1)])
    ok = ok and float(all_reduce_sum_(t)) == sum(range(1, world + 1))
    ret[rank] = ok
    dist.destroy_process_group()


def test_gloo_world2_gradient_collectives():
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(2)), dict(ret)
