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
        def forward(self, drugs, masks, mols, kg, cv, tx, kg_filler=None, **kwargs):     # (the encoder API takes **kwargs, models.py:898)
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


def _bucket_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from madrigal_amd.parallel import GradientBuckets, allreduce_gradients
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(40, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8))
        only_rank0 = torch.nn.Linear(40, 8)            # receives a gradient on rank 0 only
        nobody = torch.nn.Linear(8, 8)                  # receives a gradient on no rank
        params = list(net.parameters()) + list(only_rank0.parameters()) + list(nobody.parameters())
        x = torch.randn(16, 40, generator=torch.Generator().manual_seed(10 + rank))

        def loss():
            y = net(x)
            if rank == 0:
                y = y + only_rank0(x)
            return (y ** 2).mean()
        # reference: flat all-reduce after the backward pass
        for p in params:
            p.grad = None
        loss().backward()
        allreduce_gradients(params)
        want = [None if p.grad is None else p.grad.clone() for p in params]
        # buckets issued from the hooks (tiny buckets: several collectives, one of them incomplete on rank 1)
        for p in params:
            p.grad = None
        gb = GradientBuckets(params, bucket_bytes=3000)
        gb.arm()
        loss().backward()
        gb.finish()
        same = all((a is None and p.grad is None) or (a is not None and p.grad is not None and torch.allclose(a, p.grad, rtol=0, atol=1e-7))
                   for a, p in zip(want, params))
        # a second step reuses the hooks
        for p in params:
            p.grad = None
        gb.arm()                                        # (the parameters without any gradient now sit in trailing buckets)
        loss().backward()
        issued_early = gb.next_bucket
        gb.finish()
        same2 = all((a is None and p.grad is None) or (a is not None and torch.allclose(a, p.grad, rtol=0, atol=1e-7)) for a, p in zip(want, params))
        ret[rank] = (same, same2, len(gb.buckets), issued_early, [p.grad is None for p in nobody.parameters()], [p.grad is None for p in only_rank0.parameters()])
    finally:
        dist.destroy_process_group()


def test_gradient_buckets_issued_from_hooks_equal_the_flat_allreduce():
    """GradientBuckets: same sums as allreduce_gradients, buckets issued during backward in one fixed order on every rank;
    a parameter with a gradient on one rank only is summed with zeros, one with a gradient nowhere stays None."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(2):
        same, same2, n_buckets, early, nobody_none, only0_none = ret[r]
        assert same and same2, (r, ret[r])
        assert n_buckets >= 3 and early < n_buckets
        # rank 0 holds every "hot" gradient: collectives issued before backward() returned; rank 1 lacks the gradients of the
        # first bucket (the module only rank 0 ran), so everything waits for finish() there -- correct, only not overlapped
        assert early >= 1 if r == 0 else early == 0
        assert all(nobody_none) and not any(only0_none)


class _SumOverRanksInBackward(torch.autograd.Function):
    """A blocking default-group collective inside the backward pass (what a SyncBatchNorm layer's backward issues)."""

    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        g = g.clone()
        dist.all_reduce(g)
        return g


def _bucket_sync_worker(rank, world, port, overlap, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from madrigal_amd.parallel import GradientBuckets, allreduce_gradients, destroy_bucket_groups
        torch.manual_seed(0)
        l1, l2, l3 = torch.nn.Linear(12, 16), torch.nn.Linear(16, 16), torch.nn.Linear(16, 4)
        side = torch.nn.Linear(16, 16)                 # MID-order parameters with a gradient on rank 0 only (rank 1 has no such rows)
        params = list(l1.parameters()) + list(side.parameters()) + list(l2.parameters()) + list(l3.parameters())
        x = torch.randn(8, 12, generator=torch.Generator().manual_seed(3 + rank))

        def loss():
            h = torch.relu(l1(x))
            h = _SumOverRanksInBackward.apply(h)        # blocking collective of the default group between the buckets
            h2 = l2(h)
            if rank == 0:
                h2 = h2 + side(h)
            return (l3(torch.relu(h2)) ** 2).mean()
        for p in params:
            p.grad = None
        loss().backward()
        allreduce_gradients(params)
        want = [p.grad.clone() for p in params]
        outs = []
        for _step in range(2):
            for p in params:
                p.grad = None
            gb = GradientBuckets(params, bucket_bytes=600, overlap=overlap)
            gb.arm()
            loss().backward()
            early = gb.next_bucket
            gb.finish()
            outs.append(all(torch.allclose(a, p.grad, rtol=0, atol=1e-7) for a, p in zip(want, params)))
        ret[rank] = (outs, early, len(gb.buckets), id(gb.group))
        destroy_bucket_groups()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_gradient_buckets_beside_a_blocking_collective_and_a_rank_without_a_mid_order_gradient(overlap):
    """one rank lacks the gradient of a mid-order parameter (no uni-modal rows) and the backward pass itself issues a
    blocking all-reduce (SyncBatchNorm): with a blocking bucket transport every bucket must wait for finish() (overlap=False is
    what GradientBuckets selects by itself for gloo + device tensors); the non-blocking transport may overlap.  Either way the
    sums equal the flat all-reduce, twice (the second step object reuses the process-wide bucket communicator)."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_sync_worker, args=(r, 2, port, overlap, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, "deadlock or crash"
    for r in range(2):
        outs, early, n_buckets, _ = ret[r]
        assert all(outs) and n_buckets >= 3
        if not overlap:
            assert early == 0


class _LateWork:
    """The work handle of a non-blocking transport (RCCL): the sum lands in the buffer only when ``wait()`` is called, and until then
    the buffer holds poison -- a reader that touches a bucket before waiting for it fails the comparison."""
    issued = []                                         # every handle created in this process, in issue order
    waited = []

    def __init__(self, tensor, group):
        self.tensor, self.group, self.saved = tensor, group, tensor.clone()
        tensor.fill_(float("nan"))
        _LateWork.issued.append(self)

    def wait(self):
        self.tensor.copy_(self.saved)
        _REAL_ALL_REDUCE(self.tensor, group=self.group)
        _LateWork.waited.append(self)
        return True


_REAL_ALL_REDUCE = dist.all_reduce


def _late_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from madrigal_amd import parallel as P
        torch.manual_seed(0)
        layers = [torch.nn.Linear(12, 16), torch.nn.Linear(16, 16), torch.nn.Linear(16, 16), torch.nn.Linear(16, 4)]
        params = [p for l in layers for p in l.parameters()]
        x = torch.randn(8, 12, generator=torch.Generator().manual_seed(3 + rank))

        def loss():
            h = x
            for l in layers[:-1]:
                h = torch.relu(l(h))
            return (layers[-1](h) ** 2).mean()
        loss().backward()
        P.allreduce_gradients(params)
        want = [p.grad.clone() for p in params]
        for p in params:
            p.grad = None

        def fake_all_reduce(tensor, op=dist.ReduceOp.SUM, group=None, async_op=False):
            if not async_op:
                return _REAL_ALL_REDUCE(tensor, op=op, group=group)
            return _LateWork(tensor, group)
        P.dist.all_reduce = fake_all_reduce               # the module's own reference to torch.distributed
        try:
            gb = P.GradientBuckets(params, bucket_bytes=500, overlap=True)
            gb.arm()
            loss().backward()
            issued_in_backward = len(_LateWork.issued)
            poisoned = all(bool(torch.isnan(w.tensor).all()) for w in _LateWork.issued)       # nothing has landed yet
            gb.finish()
        finally:
            P.dist.all_reduce = _REAL_ALL_REDUCE
        same = all(torch.allclose(a, p.grad, rtol=0, atol=1e-7) for a, p in zip(want, params))
        in_order = [id(w) for w in _LateWork.waited] == [id(w) for w in _LateWork.issued]
        ret[rank] = (same, issued_in_backward, len(_LateWork.issued), poisoned, in_order, len(gb.buckets))
        P.destroy_bucket_groups()
    finally:
        dist.destroy_process_group()


def test_gradient_buckets_with_a_transport_whose_sums_land_late():
    """What RCCL does and gloo on CPU tensors does not: ``all_reduce(async_op=True)`` returns at once and the sum is in the buffer only
    after ``wait()``.  The bucket collectives are replaced by handles that poison the buffer until waited for: every bucket is issued
    from the hooks (during backward), none is read before its wait, the waits come in issue order, the gradients equal the flat
    all-reduce's."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_late_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(2):
        same, early, total, poisoned, in_order, n_buckets = ret[r]
        assert same and poisoned and in_order, (r, ret[r])
        assert n_buckets >= 3 and total == n_buckets and early == n_buckets, (r, ret[r])      # all of them issued before backward() returned
