"""Backward / training-mode parity of the HIP path (finetune step, train_ddi_batch.py:231-354).

The reference's backward pass is torch autograd over its nn.Modules; the checker here is exactly that: the same
module structure (a CPU float64 deep copy run through torch's own layers) differentiated by torch.  The GPU side
runs forward and backward through libmadrigal_hip.so (madrigal_amd/autograd.py).  Tolerance: 1e-4 relative to the
largest gradient entry in fp32 mode and in the default split-bf16 mode (north_star's fp32 tolerance).
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _close(a, b, tol=1e-4, what="", floor=1e-6):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(b.abs().max()), floor)
    err = float((a - b).abs().max()) / scale
    assert err <= tol, f"{what}: rel err {err:.3e} > {tol}"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


# ---------------------------------------------------------------------------------------------- building blocks
@pytest.mark.parametrize("R,C", [(1, 1), (63, 65), (64, 64), (257, 130), (1000, 978), (4096, 128)])
def test_transpose(R, C):
    from madrigal_amd import ops
    x = _rand(R, C, seed=R + C).to(DEV)
    y = ops.transpose(x)
    assert y.shape == (C, (R + 3) // 4 * 4)
    assert torch.equal(y[:, :R], x.t())
    assert not y[:, R:].any()
    xs = _rand(R, C + 4, seed=1).to(DEV)[:, :C]            # strided input view
    assert torch.equal(ops.transpose(xs, pad_inner=False), xs.t())


@pytest.mark.parametrize("R,C", [(1, 5), (255, 64), (256, 65), (1031, 130), (5000, 384)])
def test_colsum(R, C):
    from madrigal_amd import ops
    x = _rand(R, C, seed=R).to(DEV)
    s = ops.colsum(x)
    _close(s, x.double().sum(0), 2e-6, "colsum")
    assert torch.equal(s, ops.colsum(x))                     # fixed summation order: bit-identical re-run
    acc = torch.ones(C, device=DEV)
    ops.colsum(x, out=acc, beta=0.5)
    _close(acc, 0.5 + x.double().sum(0), 2e-6, "colsum beta")


@pytest.mark.parametrize("act", ["relu", "gelu", "sigmoid", "tanh", "leakyrelu", "softplus", "selu"])
def test_activation_fwd_bwd(act):
    from madrigal_amd import ops
    fn = {"relu": torch.relu, "gelu": torch.nn.functional.gelu, "sigmoid": torch.sigmoid, "tanh": torch.tanh,
          "leakyrelu": torch.nn.functional.leaky_relu, "softplus": torch.nn.functional.softplus, "selu": torch.selu}[act]
    x = _rand(777, 33, seed=3, scale=2.0)
    dy = _rand(777, 33, seed=4)
    xr = x.double().requires_grad_(True)
    yr = fn(xr)
    yr.backward(dy.double())
    y = ops.activation_fwd(x.to(DEV), act)
    _close(y, yr, 2e-6, act + " fwd")
    dx = ops.activation_bwd(dy.to(DEV), x.to(DEV), act)
    _close(dx, xr.grad, 3e-6, act + " bwd")
    if act == "relu":                                        # relu' may be taken from the output
        _close(ops.activation_bwd(dy.to(DEV), y, act), xr.grad, 1e-7, "relu bwd from y")


def test_dropout_mask_statistics_and_replay():
    from madrigal_amd import ops
    x = torch.ones(1 << 20, device=DEV)
    for p in (0.1, 0.5, 0.9):
        y = ops.dropout(x, p, seed=1234)
        kept = (y != 0)
        assert abs(float(kept.float().mean()) - (1 - p)) < 4e-3
        assert torch.allclose(y[kept], torch.full_like(y[kept], 1 / (1 - p)))
        assert torch.equal(y, ops.dropout(x, p, seed=1234))          # replay = backward mask
        assert not torch.equal(y, ops.dropout(x, p, seed=1235))
    # no visible structure between neighbouring elements / seeds
    a = ops.dropout(x, 0.5, seed=7) != 0
    b = ops.dropout(x, 0.5, seed=8) != 0
    assert abs(float((a & b).float().mean()) - 0.25) < 4e-3
    assert abs(float((a[1:] & a[:-1]).float().mean()) - 0.25) < 4e-3
    assert torch.equal(ops.dropout(x, 0.0, seed=1), x)


@pytest.mark.parametrize("R,d", [(1, 128), (77, 128), (4096, 256), (130, 100), (65, 1024)])
def test_layernorm_backward(R, d):
    from madrigal_amd import autograd as ag
    x, dy = _rand(R, d, seed=1, scale=3.0), _rand(R, d, seed=2)
    w, b = _rand(d, seed=3) + 1.0, _rand(d, seed=4)
    xr, wr, br = (v.double().requires_grad_(True) for v in (x, w, b))
    torch.nn.functional.layer_norm(xr, (d,), wr, br, 1e-5).backward(dy.double())
    xg, wg, bg = (v.to(DEV).requires_grad_(True) for v in (x, w, b))
    y = ag.layernorm(xg, wg, bg, 1e-5)
    y.backward(dy.to(DEV))
    _close(xg.grad, xr.grad, 2e-5, "ln dx")
    _close(wg.grad, wr.grad, 2e-5, "ln dgamma")
    _close(bg.grad, br.grad, 2e-5, "ln dbeta")


@pytest.mark.parametrize("R,C,act", [(2, 8, None), (300, 130, "relu"), (4096, 512, "relu"), (513, 64, "gelu")])
def test_batchnorm_train_forward_backward_and_running_stats(R, C, act):
    from madrigal_amd import autograd as ag
    x, dy = _rand(R, C, seed=5, scale=2.0) + 0.5, _rand(R, C, seed=6)
    ref = torch.nn.BatchNorm1d(C).double()
    with torch.no_grad():
        ref.weight.copy_(_rand(C, seed=7).double() + 1.0)
        ref.bias.copy_(_rand(C, seed=8).double())
        ref.running_mean.copy_(_rand(C, seed=9).double())
        ref.running_var.copy_(_rand(C, seed=10).double().abs() + 0.5)
    mine = copy.deepcopy(ref).float().to(DEV)
    fn = {None: lambda v: v, "relu": torch.relu, "gelu": torch.nn.functional.gelu}[act]
    xr = x.double().requires_grad_(True)
    fn(ref(xr)).backward(dy.double())
    xg = x.to(DEV).requires_grad_(True)
    y = ag.batchnorm_act(xg, mine, act)
    _close(y, fn(ref(x.double())), 1e-5, "bn fwd")           # (second ref call also moves the running stats: redo below)
    y.backward(dy.to(DEV))
    # with two rows xhat = +-1 and dx cancels to O(eps): measure that case against the size of dy instead
    _close(xg.grad, xr.grad, 5e-5, "bn dx", floor=1.0 if R == 2 else 1e-6)
    _close(mine.weight.grad, ref.weight.grad, 5e-5, "bn dgamma")
    _close(mine.bias.grad, ref.bias.grad, 5e-5, "bn dbeta")
    # running statistics after exactly one training call
    ref2 = torch.nn.BatchNorm1d(C).double()
    with torch.no_grad():
        ref2.running_mean.copy_(_rand(C, seed=9).double())
        ref2.running_var.copy_(_rand(C, seed=10).double().abs() + 0.5)
    ref2(x.double())
    _close(mine.running_mean, ref2.running_mean, 1e-5, "running_mean")
    _close(mine.running_var, ref2.running_var, 1e-5, "running_var")
    assert int(mine.num_batches_tracked) == 1


@pytest.mark.parametrize("prec,tol", [("f32", 2e-5), ("bf16x3", 1e-4)])
@pytest.mark.parametrize("M,K,N,act", [(4096, 128, 512, "gelu"), (333, 978, 512, "relu"), (5, 7, 3, None), (1024, 512, 130, "tanh")])
def test_linear_autograd(M, K, N, act, prec, tol):
    from madrigal_amd import autograd as ag
    x, w, b, dy = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3), _rand(M, N, seed=4)
    fn = {None: lambda v: v, "relu": torch.relu, "gelu": torch.nn.functional.gelu, "tanh": torch.tanh}[act]
    xr, wr, br = (v.double().requires_grad_(True) for v in (x, w, b))
    fn(torch.nn.functional.linear(xr, wr, br)).backward(dy.double())
    xg, wg, bg = (v.to(DEV).requires_grad_(True) for v in (x, w, b))
    y = ag.linear(xg, wg, bg, act, prec)
    y.backward(dy.to(DEV))
    _close(y, fn(torch.nn.functional.linear(x.double(), w.double(), b.double())), tol, "fwd")
    _close(xg.grad, xr.grad, tol, "dx")
    _close(wg.grad, wr.grad, tol, "dW")
    _close(bg.grad, br.grad, tol, "db")


# ---------------------------------------------------------------------------------------------- modules
def _module_grads(mod_gpu, ref_cpu, run_gpu, run_ref, tol, names=None):
    """Forward + every parameter gradient of the HIP module against torch autograd over the CPU float64 copy.
    Gradients that are analytically zero (a bias in front of a BatchNorm) are measured against the largest gradient
    of the module.  ReLU networks are compared in the exact-fp32 mode: a pre-activation within rounding distance of
    zero flips relu' (a discontinuity of the reference too), which is not what these tests are about."""
    out_g, out_r = run_gpu(mod_gpu), run_ref(ref_cpu)
    _close(out_g, out_r, tol, "module forward")
    dy = _rand(*out_r.shape, seed=99)
    out_g.backward(dy.to(DEV))
    out_r.backward(dy.double())
    gmax = max(float(p.grad.abs().max()) for p in ref_cpu.parameters() if p.grad is not None)
    checked = 0
    for (n, pg), (_, pr) in zip(mod_gpu.named_parameters(), ref_cpu.named_parameters()):
        if pr.grad is None:
            assert pg.grad is None or not pg.grad.any(), n
            continue
        assert pg.grad is not None, f"{n}: no gradient on the HIP path"
        _close(pg.grad, pr.grad, tol, n, floor=1e-3 * gmax)
        checked += 1
    assert checked > 0


@pytest.mark.parametrize("norm,order,actn,prec,tol", [
    ("ln", "nd", "relu", "f32", 2e-5), ("bn", "nd", "relu", "f32", 2e-5), (None, "nd", "relu", "f32", 2e-5),
    ("ln", "dn", "gelu", "bf16x3", 1e-4), ("bn", "nd", "tanh", "bf16x3", 1e-4), ("ln", "nd", "selu", "bf16x3", 1e-4)])
def test_mlp_encoder_training_step_matches_torch(norm, order, actn, prec, tol):
    from madrigal_amd import models as M
    torch.manual_seed(0)
    m = M.MLPEncoder(50, [128, 256, 64], 128, 0.0, norm, actn, order)
    ref = copy.deepcopy(m).double().train()
    m = m.to(DEV).train()
    x = _rand(300, 50, seed=11)
    xg = x.to(DEV).requires_grad_(True)
    xr = x.double().requires_grad_(True)
    with M.precision(prec):
        _module_grads(m, ref, lambda mm: mm(xg), lambda rr: rr.fc(xr), tol)
    _close(xg.grad, xr.grad, tol, "dx")
    if norm == "bn":
        for (n, bg), (_, br) in zip(m.named_buffers(), ref.named_buffers()):
            _close(bg.float(), br.float(), 1e-5, n)


def test_mlp_dropout_is_active_only_in_training_and_backward_replays_mask():
    from madrigal_amd import models as M
    torch.manual_seed(1)
    m = M.MLPAdaptor(64, [128, 128], 32, 0.5, "ln", "gelu").to(DEV)
    x = _rand(512, 64, seed=3).to(DEV)
    m.eval()
    with torch.no_grad():
        e1, e2 = m(x), m(x)
    assert torch.equal(e1, e2)
    m.train()
    t1, t2 = m(x), m(x)
    assert not torch.equal(t1, t2)                                      # fresh mask per call
    torch.manual_seed(5)
    a = m(x)
    torch.manual_seed(5)
    b = m(x)
    assert torch.equal(a, b)                                            # masks follow torch.manual_seed
    # finite-difference check through the dropout mask: directional derivative along a parameter
    w = m.fc[0].weight
    torch.manual_seed(9)
    loss = m(x).sum()
    loss.backward()
    g = w.grad.clone()
    d = torch.randn_like(w)
    d /= d.norm()
    eps = 1e-2
    with torch.no_grad():
        w.add_(eps * d)
        torch.manual_seed(9)
        lp = m(x).sum()
        w.sub_(2 * eps * d)
        torch.manual_seed(9)
        lm = m(x).sum()
        w.add_(eps * d)
    fd = float(lp - lm) / (2 * eps)
    an = float((g * d).sum())
    assert abs(fd - an) <= 2e-2 * max(abs(an), 1.0), (fd, an)


def test_chemcpa_mlp_training_matches_torch():
    from madrigal_amd import models as M
    torch.manual_seed(2)
    m = M.ChemCPAMLP([978, 256, 256, 64], batch_norm=True)
    ref = copy.deepcopy(m).double().train()
    m = m.to(DEV).train()
    x = _rand(200, 978, seed=4)
    res = _rand(200, 64, seed=5)
    with M.precision("f32"):
        _module_grads(m, ref, lambda mm: mm(x.to(DEV), residual=res.to(DEV)), lambda rr: rr.network(x.double()) + res.double(), 2e-5)
