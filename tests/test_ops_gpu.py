"""Op-level parity of the HIP kernels (through the C ABI) against the CPU oracle's building blocks."""
import math

import numpy as np
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu
TOL = {"f32": 2e-5, "bf16x3": 1e-4, "bf16": 3e-2}


@pytest.fixture(scope="module")
def ops():
    from madrigal_amd import ops as _ops
    return _ops


def _rand(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("prec", ["f32", "bf16x3", "bf16"])
@pytest.mark.parametrize("M,N,K", [(300, 128, 128), (1, 5, 4), (129, 257, 68), (1000, 512, 560), (77, 1536, 512), (515, 128, 980)])
def test_linear_plain(ops, prec, M, N, K):
    x, w, b = _rand((M, K), 1), _rand((N, K), 2, 1 / math.sqrt(K)), _rand((N,), 3)
    ref = x @ w.t() + b
    got = ops.linear(x.cuda(), w.cuda(), b.cuda(), precision=prec).cpu()
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) < TOL[prec] * max(float(ref.abs().max()), 1.0)


@pytest.mark.parametrize("act", ["relu", "gelu", "tanh", "sigmoid", "leakyrelu", "softplus", "selu", None])
def test_linear_epilogue(ops, act):
    from oracle import madrigal_oracle as O
    M, N, K = 200, 96, 64
    x, w, b = _rand((M, K), 4), _rand((N, K), 5, 0.2), _rand((N,), 6)
    scale, shift, res = _rand((N,), 7).abs() + 0.5, _rand((N,), 8), _rand((M, N), 9)
    ref = 0.7 * O._act(act, (x @ w.t() + b) * scale + shift) + 0.3 * res
    got = ops.linear(x.cuda(), w.cuda(), b.cuda(), scale=scale.cuda(), shift=shift.cuda(), act=act, residual=res.cuda(),
                     alpha=0.7, beta=0.3, precision="f32").cpu()
    assert rel_err(got, ref) < 2e-5


def test_linear_unpadded_k_strided_views_and_broadcast_residual(ops):
    x, w = _rand((50, 559), 10), _rand((40, 559), 11, 0.05)
    got = ops.linear(x.cuda(), w.cuda(), None, precision="f32").cpu()          # K = 559 is zero-padded to 560
    assert rel_err(got, x @ w.t()) < 2e-5
    big = _rand((30, 384), 12).cuda()
    w2 = _rand((128, 128), 13, 0.1)
    out = torch.zeros(30, 256, device="cuda")
    ops.linear(big[:, 128:256], w2.cuda(), None, precision="f32", out=out[:, 128:])
    assert rel_err(out[:, 128:].cpu(), big[:, 128:256].cpu() @ w2.t()) < 2e-5
    assert float(out[:, :128].abs().max()) == 0.0
    r = _rand((128,), 14)
    got = ops.linear(big[:, :128], w2.cuda(), None, residual=r.cuda(), precision="f32").cpu()
    assert rel_err(got, big[:, :128].cpu() @ w2.t() + r) < 2e-5


@pytest.mark.parametrize("prec,tol", [("f32", 2e-5), ("bf16x3", 1e-4)])
def test_linear_epilogue_on_misaligned_destinations_and_in_place_residual(ops, prec, tol):
    """The row-major epilogue writes 16 bytes per lane when the destination allows it and falls back to single floats when
    it does not: a destination view starting at an odd column, a residual with an odd row stride, 300 x 130 outputs (a ragged
    last column group), and y aliasing the residual (each lane reads its own four residual values before it stores)."""
    x, w, b = _rand((300, 128), 31), _rand((130, 128), 32, 0.1), _rand((130,), 33)
    want = x @ w.t() + b
    wide = torch.zeros(300, 133, device="cuda")
    ops.linear(x.cuda(), w.cuda(), b.cuda(), precision=prec, out=wide[:, 3:])
    assert rel_err(wide[:, 3:].cpu(), want) < tol and float(wide[:, :3].abs().max()) == 0.0
    res_wide = _rand((300, 131), 34).cuda()
    got = ops.linear(x.cuda(), w.cuda(), b.cuda(), residual=res_wide[:, 1:], precision=prec).cpu()
    assert rel_err(got, want + res_wide[:, 1:].cpu()) < tol
    y = _rand((300, 130), 35).cuda()
    y0 = y.cpu().clone()
    ops.linear(x.cuda(), w.cuda(), b.cuda(), residual=y, precision=prec, out=y)
    assert rel_err(y.cpu(), want + y0) < tol


def test_pinned_ring_uploads_keep_their_contents():
    """hostio.upload: more uploads under one tag than the ring has buffers, queued without any synchronisation in between."""
    from madrigal_amd import hostio
    src = [torch.arange(1000, dtype=torch.int64) * (i + 1) for i in range(11)]
    dev = [hostio.upload(t, "cuda", "test-ring", depth=3) for t in src]
    torch.cuda.synchronize()
    for a, b in zip(src, dev):
        assert torch.equal(a, b.cpu())
    m = torch.rand(64, 19) < 0.5
    assert torch.equal(hostio.upload(m, "cuda", "test-ring-mask").cpu(), m)


@pytest.mark.parametrize("prec", ["f32", "bf16x3", "bf16"])
def test_grouped_linear_equals_one_launch_per_group(ops, prec):
    """mdg_linear_grouped: groups of different row counts and widths (a group smaller than a tile, one with a ragged last tile,
    an empty one), outputs into blocks of different row strides, per-group gated residual -- bit-identical to one mdg_linear per
    group (same tiles, same k order)."""
    K = 128
    rows = [300, 17, 0, 1000]
    widths = [384, 128, 256, 640]
    xs = [_rand((r, K), 40 + i).cuda() for i, r in enumerate(rows)]
    ws = [_rand((n, K), 50 + i, 0.1).cuda() for i, n in enumerate(widths)]
    bs = [_rand((n,), 60 + i).cuda() for i, n in enumerate(widths)]
    x_all, w_all, b_all = torch.cat(xs), torch.cat(ws), torch.cat(bs)
    groups, m, n, off = [], 0, 0, 0
    for r, wd in zip(rows, widths):
        groups.append(dict(m_base=m, rows=r, n_base=n, n=wd, y_off=off, ldy=wd))
        m, n, off = m + r, n + wd, off + r * wd
    y = torch.full((off,), float("nan"), device="cuda")
    ops.linear_grouped(x_all, w_all, b_all, ops.group_tile_table(groups, "cuda"), y, act="gelu", precision=prec)
    for g, x, w, b in zip(groups, xs, ws, bs):
        want = ops.linear(x, w, b, act="gelu", precision=prec, cache_weight=False)
        got = y[g["y_off"]: g["y_off"] + g["rows"] * g["ldy"]].view(g["rows"], g["ldy"])
        assert torch.equal(got, want)
    # 128 -> 128 layers with x itself as the gated residual (the output projection of the KG conv)
    ws2 = [_rand((128, K), 70 + i, 0.1).cuda() for i in range(4)]
    gates = [0.3, 0.9, 0.5, 0.1]
    groups2, m = [], 0
    for i, r in enumerate(rows):
        groups2.append(dict(m_base=m, rows=r, n_base=128 * i, n=128, y_off=m * 128, ldy=128, res_off=m * 128, ldr=128, alpha=gates[i], beta=1 - gates[i]))
        m += r
    y2 = torch.empty(m, 128, device="cuda")
    ops.linear_grouped(x_all, torch.cat(ws2), None, ops.group_tile_table(groups2, "cuda"), y2, residual=x_all, precision=prec)
    for g, x, w, a in zip(groups2, xs, ws2, gates):
        want = ops.linear(x, w, None, residual=x, alpha=a, beta=1 - a, precision=prec, cache_weight=False)
        assert torch.equal(y2[g["m_base"]: g["m_base"] + g["rows"]], want)


@pytest.mark.parametrize("prec", ["bf16x3", "bf16", "f32"])
def test_layernorm_emits_the_operand_image_of_the_next_dense_block(ops, prec):
    """layernorm_packed + linear_packed == layernorm + linear bit for bit (the image is what linear's own pre-pass writes);
    the fp32 rows can be skipped; fp32 mode / widths that are no multiple of 32 fall back to no image."""
    x = _rand((333, 256), 80).cuda()
    g, b = _rand((256,), 81).cuda(), _rand((256,), 82).cuda()
    w, bias = _rand((192, 256), 83, 0.1).cuda(), _rand((192,), 84).cuda()
    y_ref = ops.layernorm(x, g, b, 1e-5)
    want = ops.linear(y_ref, w, bias, act="gelu", precision=prec)
    y, img = ops.layernorm_packed(x, g, b, 1e-5, prec)
    assert torch.equal(y, y_ref)
    if prec == "f32":
        assert img is None
        return
    assert torch.equal(ops.linear_packed(img, 333, w, bias, act="gelu", precision=prec), want)
    y2, img2 = ops.layernorm_packed(x, g, b, 1e-5, prec, want_fp32=False)
    assert y2 is None and torch.equal(img2, img)
    wide = torch.zeros(333, 400, device="cuda")
    ops.linear_packed(img, 333, w, bias, act="gelu", precision=prec, out=wide[:, 8:200])
    assert torch.equal(wide[:, 8:200], want) and float(wide[:, 200:].abs().max()) == 0.0
    assert ops.layernorm_packed(x[:, :100].contiguous(), g[:100].contiguous(), b[:100].contiguous(), 1e-5, prec)[1] is None


def test_linear_errors(ops):
    x, w = torch.zeros(4, 8, device="cuda"), torch.zeros(3, 12, device="cuda")
    with pytest.raises(ValueError):
        ops.linear(x, w)
    with pytest.raises(ValueError):
        ops.linear(x, torch.zeros(3, 8, device="cuda"), act="swish")
    with pytest.raises(RuntimeError, match="forward-only"):
        ops.linear(x.requires_grad_(), torch.zeros(3, 8, device="cuda"))


@pytest.mark.parametrize("rows,d", [(37, 128), (5, 512), (3, 2048), (1000, 256), (9, 36)])
def test_layernorm(ops, rows, d):
    from oracle import madrigal_oracle as O
    x, g, b = _rand((rows, d), 20, 3.0) + 1.5, _rand((d,), 21).abs() + 0.5, _rand((d,), 22)
    got = ops.layernorm(x.cuda(), g.cuda(), b.cuda()).cpu()
    assert float((got - O.layer_norm(x, g, b)).abs().max()) < 2e-5


@pytest.mark.parametrize("S,H,dh", [(23, 8, 64), (21, 2, 256), (19, 4, 128), (22, 4, 32), (32, 1, 32), (1, 2, 64)])
def test_fusion_attention(ops, S, H, dh):
    n, d = 9, H * dh
    qkv = _rand((n * S, 3 * d), 30)
    kpm = torch.rand(n, S, generator=torch.Generator().manual_seed(31)) < 0.3
    kpm[:, min(3, S - 1)] = False                        # at least one key everyone may attend
    src = torch.zeros(S, S, dtype=torch.bool)
    if S > 4:
        src[:2, -2:] = True
        src[-2:, :2] = True
    q, k, v = (t.view(n, S, H, dh).transpose(1, 2) for t in qkv.view(n, S, 3 * d).split(d, dim=2))
    logits = (q / math.sqrt(dh)) @ k.transpose(-1, -2)
    logits = logits.masked_fill(src.view(1, 1, S, S), float("-inf")).masked_fill(kpm.view(n, 1, 1, S), float("-inf"))
    p = torch.softmax(logits, dim=-1)
    ref = (p @ v).transpose(1, 2).reshape(n * S, d)
    out, probs = ops.fusion_attention(qkv.cuda(), n, S, H, dh, ops.mask_bits(kpm.cuda()), ops.mask_bits(src.cuda()), want_probs=True)
    assert rel_err(probs.cpu(), p) < 2e-5
    assert rel_err(out.cpu(), ref) < 2e-5
    out2, none = ops.fusion_attention(qkv.cuda(), n, S, H, dh)          # no masks
    p2 = torch.softmax((q / math.sqrt(dh)) @ k.transpose(-1, -2), dim=-1)
    assert none is None and rel_err(out2.cpu(), (p2 @ v).transpose(1, 2).reshape(n * S, d)) < 2e-5


def test_fusion_attention_fully_masked_row_is_nan_like_torch(ops):
    n, S, H, dh = 2, 5, 1, 32
    qkv = _rand((n * S, 3 * H * dh), 33)
    kpm = torch.zeros(n, S, dtype=torch.bool)
    kpm[1, :] = True
    out, _ = ops.fusion_attention(qkv.cuda(), n, S, H, dh, ops.mask_bits(kpm.cuda()), None)
    out = out.cpu().view(n, S, -1)
    assert torch.isfinite(out[0]).all() and torch.isnan(out[1]).all()


@pytest.mark.parametrize("Tk,H,dh", [(4, 8, 64), (2, 2, 256), (19, 4, 128), (1, 4, 32)])
def test_xattn_pool(ops, Tk, H, dh):
    n, d = 11, H * dh
    q, kv = _rand((d,), 40), _rand((n * Tk, 2 * d), 41)
    k, v = (t.view(n, Tk, H, dh) for t in kv.split(d, dim=1))
    lg = torch.einsum("hd,nthd->nht", q.view(H, dh) / math.sqrt(dh), k)
    ref = torch.einsum("nht,nthd->nhd", torch.softmax(lg, -1), v).reshape(n, d)
    assert rel_err(ops.xattn_pool(q.cuda(), kv.cuda(), n, Tk, H, dh).cpu(), ref) < 2e-5


def test_assemble_tokens_and_pools(ops):
    from oracle import madrigal_oracle as O
    n, D = 7, 128
    s, k, c = _rand((n, D), 50), _rand((n, D), 51), _rand((n, D), 52)
    tx = _rand((16 * n, D), 53)
    bt, cls = _rand((4, D), 54), _rand((1, D), 55)
    all_embeds = torch.stack([s, k, c] + list(tx.split(n)), dim=1)
    masks = torch.rand(n, 19, generator=torch.Generator().manual_seed(56)) < 0.4
    masks[:, 0] = False
    for nb, use_cls, norm in [(4, False, False), (2, True, True), (0, False, True)]:
        seq, kpm, src = O.assemble_fusion_inputs(all_embeds, masks, bt[:nb] if nb else None, cls if use_cls else None)
        pe = _rand((seq.shape[1] - 5, D), 57)
        ref = O.l2_normalize(seq) if norm else seq.clone()
        ref[:, :pe.shape[0]] += pe
        got = ops.assemble_tokens(s.cuda(), k.cuda(), c.cuda(), tx.cuda(), bottleneck=bt[:nb].cuda() if nb else None,
                                  cls=cls.cuda() if use_cls else None, pe=pe.cuda(), normalize=norm).cpu()
        assert float((got - ref).abs().max()) < 2e-6
    rows = torch.tensor([5, 0, 3])
    got = ops.assemble_tokens(s.cuda(), k.cuda(), c.cuda(), tx.cuda(), rows=rows.cuda()).cpu()
    assert torch.equal(got, all_embeds[rows])
    bits = ops.mask_bits(masks.cuda())
    keep = (~masks).unsqueeze(-1)
    assert float((ops.token_pool(all_embeds.cuda(), bits, "sum").cpu() - (all_embeds * keep).sum(1)).abs().max()) < 1e-5
    assert float((ops.token_pool(all_embeds.cuda(), bits, "mean").cpu() - (all_embeds * keep).sum(1) / keep.sum(1)).abs().max()) < 1e-5
    mx = all_embeds.masked_fill(~keep, float("-inf")).max(1).values
    assert torch.equal(ops.token_pool(all_embeds.cuda(), bits, "max").cpu(), mx)
    assert float((ops.l2_normalize(all_embeds.cuda()).cpu() - O.l2_normalize(all_embeds)).abs().max()) < 1e-6


@pytest.mark.parametrize("F", [128, 68, 20, 256])
def test_csr_aggregate(ops, F):
    g = torch.Generator().manual_seed(60)
    n_src, n_dst, E = 500, 300, 4000
    x = _rand((n_src, F), 61)
    dst = torch.randint(0, n_dst, (E,), generator=g).sort().values
    dst[dst == 7] = 8                                    # an empty row
    dst = dst.sort().values
    col = torch.randint(0, n_src, (E,), generator=g)
    w = torch.rand(E, generator=g)
    rowptr = torch.zeros(n_dst + 1, dtype=torch.int64)
    rowptr[1:] = torch.bincount(dst, minlength=n_dst).cumsum(0)
    xs = _rand((n_dst, F), 62)
    eps = torch.tensor([0.25])
    ref = torch.zeros(n_dst, F).index_add_(0, dst, x[col] * w[:, None]) + 1.25 * xs
    got = ops.csr_aggregate(x.cuda(), rowptr.cuda(), col.cuda(), edge_weight=w.cuda(), x_self=xs.cuda(), self_coef_dev=eps.cuda(),
                            self_coef_add=1.0).cpu()
    assert float((got - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    # contiguous segments + mean (molecule read-out)
    seg = torch.tensor([0, 3, 3, 10, 500])
    ref2 = torch.stack([x[a:b].mean(0) if b > a else torch.zeros(F) for a, b in zip(seg[:-1], seg[1:])])
    got2 = ops.csr_aggregate(x.cuda(), seg.cuda(), None, mean=True).cpu()
    assert float((got2 - ref2).abs().max()) < 1e-5


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("M,N,K", [(22016, 1024, 512), (2700, 6144, 256), (16897, 2048, 128)])
def test_dense_block_row_split_is_bit_identical(ops, monkeypatch, prec, M, N, K):
    """A product whose 256-tiles fill whole rounds plus a small remainder runs as two launches (whole rounds on the 256-tile kernel,
    the remaining rows on the 128-tile kernel: linear.hip launch_linear_core).  Same operand images, same k order per element: the
    result equals the single launch bit for bit -- bias, activation, residual, ragged last row tile included."""
    from helpers import set_switch
    x, w, b = _rand((M, K), 1).cuda(), _rand((N, K), 2, 0.05).cuda(), _rand((N,), 3).cuda()
    res = _rand((M, N), 4).cuda()
    tiles = -(-M // 256) * -(-N // 256)
    assert tiles > 256 and 0 < tiles % 256 <= 128                      # the shapes take the split on a 256-CU card
    set_switch(monkeypatch, "MDG_LINEAR_TAIL128", "0")
    one = ops.linear(x, w, b, act="gelu", residual=res, precision=prec, cache_weight=False)
    set_switch(monkeypatch, "MDG_LINEAR_TAIL128", "1")
    two = ops.linear(x, w, b, act="gelu", residual=res, precision=prec, cache_weight=False)
    assert torch.equal(one, two)
    ref = torch.nn.functional.gelu(x[-300:].double() @ w.double().t() + b.double()) + res[-300:].double()
    assert float((two[-300:].double() - ref).abs().max()) < (3e-2 if prec == "bf16" else 1e-4) * float(ref.abs().max())


@pytest.mark.parametrize("heads", [4, 1, 8])
def test_hgt_attention_heavy_tail(ops, heads):
    """Edge softmax + aggregation incl. a destination with > CHUNK edges (split into work items), an
    isolated destination and single-edge destinations."""
    from madrigal_amd.graph_plans import HGT_CHUNK, hgt_plan
    g = torch.Generator().manual_seed(70)
    n_src, n_dst = 900, 40
    e_heavy = 3 * HGT_CHUNK + 17
    e_hub = 20 * HGT_CHUNK + 5                             # > 16 work items: merged by the whole workgroup of the combine kernel
    dst = torch.cat([torch.zeros(e_heavy, dtype=torch.int64), torch.randint(2, n_dst, (600,), generator=g), torch.tensor([1]),
                     torch.full((e_hub,), 7, dtype=torch.int64)])
    src = torch.randint(0, n_src, (dst.numel(),), generator=g)
    dst[dst == 5] = 6                                    # destination 5 has no edges
    ei = {("a", "r", "b"): torch.stack([src, dst])}
    plan = hgt_plan({k: v.cuda() for k, v in ei.items()}, [("a", "r", "b")], {"a": n_src, "b": n_dst}, torch.device("cuda"))
    # projection layout of source type "a": rows of width 128 + 256 = [q | k' v']
    assert plan["width"]["a"] == 384 and plan["base"]["a"] == 0
    q = _rand((n_dst, 128), 71)
    proj_a = _rand((n_src, 384), 72)
    kv = proj_a[:, 128:]                                     # [n_src, 256] = k' | v' (reference view for the check below)
    buf = torch.zeros(plan["total_floats"])
    buf[: n_src * 384] = proj_a.flatten()
    D = 128 // heads
    a = (q[dst].view(-1, heads, D) * kv[src, :128].view(-1, heads, D)).sum(-1)
    amax = torch.full((n_dst, heads), float("-inf")).scatter_reduce(0, dst[:, None].expand(-1, heads), a, reduce="amax")
    e = torch.exp(a - amax[dst])
    den = torch.zeros(n_dst, heads).index_add_(0, dst, e)
    alpha = e / (den[dst] + 1e-16)
    agg = torch.zeros(n_dst, heads, D).index_add_(0, dst, kv[src, 128:].view(-1, heads, D) * alpha[..., None]).reshape(n_dst, 128)
    from oracle import madrigal_oracle as O
    kvd = buf.cuda().view(-1, 128)
    got = ops.hgt_attention(q.cuda(), kvd, plan["per_dst"]["b"], heads, apply_gelu=False).cpu()
    assert float((got - agg).abs().max()) < 2e-5
    got_g = ops.hgt_attention(q.cuda(), kvd, plan["per_dst"]["b"], heads, apply_gelu=True).cpu()
    assert float((got_g - O._act("gelu", agg)).abs().max()) < 2e-5
    assert float(got[5].abs().max()) == 0.0


@pytest.mark.parametrize("heads", [4, 8])
def test_hgt_attention_backward_heavy_tail(ops, heads):
    """dq / dk' / dv' of the edge attention against float64 autograd of the same formula: a destination with > CHUNK edges and a
    key row with > CHUNK outgoing edges (several work items: partials summed in item order), rows with exactly one item (written
    by the gather kernels themselves), an isolated destination, key rows nobody attends to (their gradient rows stay zero)."""
    from madrigal_amd.graph_plans import HGT_CHUNK, hgt_plan, hgt_reverse_plan
    g = torch.Generator().manual_seed(80)
    n_src, n_dst = 500, 300
    hub = 2 * HGT_CHUNK + 9
    big = 18 * HGT_CHUNK + 3                            # > 16 work items: summed by the whole workgroup (hgt_sum_items_kernel / hgt_combine_kernel)
    dst = torch.cat([torch.zeros(hub, dtype=torch.int64), torch.randint(2, n_dst, (hub,), generator=g), torch.randint(2, n_dst, (900,), generator=g),
                     torch.tensor([1]), torch.full((big,), 9, dtype=torch.int64), torch.randint(2, n_dst, (big,), generator=g)])
    src = torch.cat([torch.randint(0, n_src - 50, (hub,), generator=g), torch.full((hub,), 7), torch.randint(0, n_src - 50, (901,), generator=g),
                     torch.randint(0, n_src - 50, (big,), generator=g), torch.full((big,), 11)])
    dst[dst == 5] = 6                                    # destination 5 has no edges; key rows n_src-50.. have none either
    plan = hgt_plan({("a", "r", "b"): torch.stack([src, dst]).cuda()}, [("a", "r", "b")], {"a": n_src, "b": n_dst}, torch.device("cuda"))
    pd = plan["per_dst"]["b"]
    rev = hgt_reverse_plan(pd)
    assert int((pd["item_ptr"][1:] - pd["item_ptr"][:-1]).max()) == 19 and int((rev["item_ptr"][1:] - rev["item_ptr"][:-1]).max()) == 19
    q = _rand((n_dst, 128), 81)
    proj = _rand((n_src, 384), 82)
    dout = _rand((n_dst, 128), 83)
    D = 128 // heads
    q64 = q.double().requires_grad_(True)
    kv64 = proj[:, 128:].double().requires_grad_(True)
    a = (q64[dst].view(-1, heads, D) * kv64[src, :128].view(-1, heads, D)).sum(-1)
    amax = torch.full((n_dst, heads), float("-inf"), dtype=torch.float64).scatter_reduce(0, dst[:, None].expand(-1, heads), a.detach(), reduce="amax")
    e = torch.exp(a - amax[dst])
    den = torch.zeros(n_dst, heads, dtype=torch.float64).index_add(0, dst, e)
    alpha = e / (den[dst] + 1e-16)
    agg = torch.zeros(n_dst, heads, D, dtype=torch.float64).index_add(0, dst, kv64[src, 128:].view(-1, heads, D) * alpha[..., None]).reshape(n_dst, 128)
    (agg * dout.double()).sum().backward()
    buf = torch.zeros(plan["total_floats"])
    buf[: n_src * 384] = proj.flatten()
    kvd = buf.cuda().view(-1, 128)
    qd = q.cuda()
    out, stats = ops.hgt_attention_stats(qd, kvd, pd, heads)
    assert float((out.cpu() - agg.detach().float()).abs().max()) < 2e-5
    dkv = torch.zeros_like(kvd)
    dq = ops.hgt_attention_bwd(qd, kvd, pd, rev, heads, dout.cuda(), out, stats, dkv)
    scale = float(q64.grad.abs().max())
    assert float((dq.cpu() - q64.grad.float()).abs().max()) < 2e-5 * max(scale, 1.0)
    got = dkv.view(-1)[: n_src * 384].view(n_src, 384).cpu()
    assert float(got[:, :128].abs().max()) == 0.0                                  # the query slots are not this call's to write
    assert float((got[:, 128:] - kv64.grad.float()).abs().max()) < 2e-5 * max(float(kv64.grad.abs().max()), 1.0)
    assert float(got[n_src - 50:].abs().max()) == 0.0 and float(dq[5].abs().max()) == 0.0
    dkv2 = torch.zeros_like(kvd)
    dq2 = ops.hgt_attention_bwd(qd, kvd, pd, rev, heads, dout.cuda(), out, stats, dkv2)
    assert torch.equal(dq, dq2) and torch.equal(dkv, dkv2)                         # no atomics: bit-reproducible
    # reduced-precision mode: k' | v' gathered from the bf16 mirror of the projection buffer == the fp32 kernels run on the rounded rows
    kv16 = ops.f32_to_bf16(kvd)
    assert torch.equal(kv16.cpu(), buf.view(-1, 128).to(torch.bfloat16))           # round to nearest even, as torch rounds
    kvr = kv16.float()
    out_r, stats_r = ops.hgt_attention_stats(qd, kvr, pd, heads)
    out_h, stats_h = ops.hgt_attention_stats(qd, kvd, pd, heads, kv16=kv16)
    assert torch.equal(out_r, out_h) and torch.equal(stats_r, stats_h)
    dkv_r, dkv_h = torch.zeros_like(kvd), torch.zeros_like(kvd)
    dq_r = ops.hgt_attention_bwd(qd, kvr, pd, rev, heads, dout.cuda(), out_r, stats_r, dkv_r)
    dq_h = ops.hgt_attention_bwd(qd, kvd, pd, rev, heads, dout.cuda(), out_h, stats_h, dkv_h, kv16=kv16)
    assert torch.equal(dq_r, dq_h) and torch.equal(dkv_r, dkv_h)
    assert float((out_h.cpu() - agg.detach().float()).abs().max()) < 3e-2 * max(float(agg.abs().max()), 1.0)
