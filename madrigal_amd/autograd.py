"""Autograd nodes over the HIP kernels: the finetune step (train_ddi_batch.py:231-354) backpropagates through the
same modules as inference, with every forward AND backward product computed by ``libmadrigal_hip.so``.

``torch.autograd.Function`` is used for what it is — the tape; no torch arithmetic runs inside these nodes.  Each
node's backward is checked in tests/test_train_gpu.py against torch's own autograd over the reference module
structure on the CPU (the reference's backward *is* torch autograd).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops

# The training steps run the KG encoder and the structure encoder on side streams of their own (NovelDDIEncoder.encode: their forward, and
# therefore -- torch runs a node's backward on the stream of its forward -- their backward, overlap the other encoders').  The gradients
# of those encoders' parameters are thus produced on the side stream, while a parameter's AccumulateGrad node keeps the stream it was first
# used on (the main stream, in the first steps' plan-building passes): torch orders the two with an event wait, which is exactly the join the
# step needs before AdamW reads the gradients, and warns once per process that this "may incur unnecessary synchronization".  Intended here;
# the warning is switched off (tests/test_train_gpu.py::test_training_steps_emit_no_stream_warnings).
if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)


def needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


def next_seed() -> int:
    """A fresh dropout seed from torch's CPU generator (follows torch.manual_seed; no device sync)."""
    return int(torch.randint(0, 2 ** 62, (1,), device="cpu").item())


class _Linear(Function):
    """y = drop(act(x W^T + b)) + residual;  dX = g W,  dW = g^T X,  db = colsum(g),  g = drop'(dy) * act'(pre),  d residual = dy.
    ``drop`` = inverted dropout (p, seed) or the identity (p = 0); with it the block is nn.TransformerEncoderLayer's
    ``x + dropout(linear(...))`` / ``dropout(activation(linear(x)))`` in one node: the mask is applied in the GEMM's epilogue (no
    activation) or together with the activation, and its backward while the incoming gradient is packed for the two backward GEMMs."""

    @staticmethod
    def _is_parameter(w) -> bool:
        """A leaf whose storage address identifies it across the step (the image caches are keyed by address + in-place version)."""
        return w.is_leaf and w.requires_grad and w.dim() == 2 and w.shape[1] % 4 == 0 and w.is_contiguous() and \
            not (w.is_cuda and torch.cuda.is_current_stream_capturing())

    @staticmethod
    def forward(ctx, x, w, b, act, precision, p=0.0, seed=0, residual=None, x_image=None):
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        ctx.k_in = x2.shape[1]
        if x_image is not None and (precision not in ("bf16", "bf16x3") or ctx.k_in % 64 != 0):
            x_image = None              # (the producer's operand image of x: layernorm_packed; only the 16-bit modes read images)
        if ctx.k_in % 4:            # 978 genes, 559 viability features, 67 atom features: zero columns up to a multiple of 4, once, so that
            x2 = ops._pad_last(x2)  # the forward GEMM and the weight gradient both read 16-byte aligned rows
        # a parameter's operand image is kept per in-place version (both sides / views of a step use it); anything else is packed
        # inside the call (a temporary must not enter the per-storage cache)
        keep = _Linear._is_parameter(w)
        w_img = None
        if keep and ctx.needs_input_grad[0] and precision in ("bf16", "bf16x3"):
            both = ops.parameter_images(w, precision)          # W's image and W^T's (for dx) from one pass over W, per parameter version
            w_img = None if both is None else both[0]
        N = w.shape[0]
        res2 = None
        if residual is not None:
            if tuple(residual.shape) != tuple(lead) + (N,):
                raise ValueError("linear: residual must have the shape of the output")
            res2 = residual.reshape(-1, N)
            res2 = res2 if res2.is_contiguous() else res2.contiguous()
        plain = act in (None, "none")

        def gemm(act_=None, residual_=None):             # the block's GEMM without dropout: from the producer's image of x when there is one
            if x_image is not None:
                return ops.linear_packed(x_image, x2.shape[0], w, b, act=act_, residual=residual_, precision=precision, cache_weight=keep,
                                         weight_image=w_img)
            return ops.linear(x2, w, b, act=act_, precision=precision, cache_weight=keep, residual=residual_, weight_image=w_img)
        if p > 0.0 and not plain:
            if res2 is not None:
                raise ValueError("linear: activation + dropout + residual in one block is not a layer of the reference")
            pre = gemm()
            y = ops.activation_dropout_fwd(pre, act, p, seed)
        elif act in (None, "none", "relu"):
            if p > 0.0:
                y = ops.linear(x2, w, b, act=act, precision=precision, cache_weight=keep, residual=res2, dropout_p=p, dropout_seed=seed,
                               weight_image=w_img)
            else:
                y = gemm(act, res2)
            pre = y if act == "relu" else None
            if act == "relu" and res2 is not None:
                raise ValueError("linear: relu + residual in one block is not a layer of the reference")
        else:
            pre = gemm()
            y = ops.activation_fwd(pre, act)
            if res2 is not None:
                y = ops.axpby(y, res2)
        ctx.save_for_backward(x2, w, pre)
        ctx.act, ctx.precision, ctx.has_bias, ctx.lead, ctx.p, ctx.seed, ctx.has_res = act, precision, b is not None, lead, p, seed, residual is not None
        return y.view(*lead, N)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, w, pre = ctx.saved_tensors
        g = dy.reshape(-1, w.shape[0])
        g = g if g.is_contiguous() else g.contiguous()
        d_res = dy if (ctx.has_res and ctx.needs_input_grad[7]) else None
        mask_p = ctx.p                                   # dropout backward still to be applied to g
        if pre is not None:
            if ctx.p > 0.0:
                g, mask_p = ops.activation_dropout_bwd(g, pre, ctx.act, ctx.p, ctx.seed), 0.0
            else:
                g = ops.activation_bwd(g, pre, ctx.act)
        dx = dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        N, K = w.shape[0], ctx.k_in
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and ctx.precision in ("bf16", "bf16x3") and N % 64 == 0 and K % 4 == 0 \
                and ops.wide_weight_gradient(N, K) and g.shape[0] > 0 and g.data_ptr() % 16 == 0 and x2.stride(1) == 1:
            # wide block: ONE pass over g makes its operand image (dx GEMM), the image of its transpose (dW GEMM) and the bias gradient
            row_img, t_img, db = ops.linear_backward_pack(g, ctx.precision, want_bias=want_db, dropout_p=mask_p, dropout_seed=ctx.seed)
            wt, wt_img = ops.transposed_weight_image(w, ctx.precision) if _Linear._is_parameter(w) else (ops.transpose(w), None)
            dx = ops.linear_packed(row_img, g.shape[0], wt, precision=ctx.precision, weight_image=wt_img, cache_weight=False)[:, :K]
            dw = ops.linear_tn_packed_g(t_img, x2, N, ctx.precision)
            return dx.reshape(*ctx.lead, K), dw, db, None, None, None, None, d_res, None
        if mask_p > 0.0:
            g = ops.dropout(g, mask_p, ctx.seed)
        if ctx.needs_input_grad[0]:
            if _Linear._is_parameter(w):
                wt, wt_img = ops.transposed_weight_image(w, ctx.precision)
            elif w.is_leaf and w.requires_grad:
                wt, wt_img = ops.weight_transposed(w), None
            else:
                wt, wt_img = ops.transpose(w), None
            dx = ops.linear(g, wt, precision=ctx.precision, cache_weight=False, weight_image=wt_img)[:, :K]
            dx = dx.reshape(*ctx.lead, K)
        if ctx.needs_input_grad[1]:
            dw = ops.grad_weight(g, x2, ctx.precision, want_bias=want_db)
            if want_db:
                dw, db = dw
            if dw.shape[1] != K:                          # the zero columns x was padded with
                dw = dw[:, :K].contiguous()
        elif want_db:
            db = ops.colsum(g)
        return dx, dw, db, None, None, None, None, d_res, None


def linear(x, w, b=None, act=None, precision="bf16x3", x_image=None):
    """``x_image``: the operand image of ``x`` its producer wrote on the side (``layernorm(..., image_precision=...)``): the forward GEMM
    reads it instead of packing ``x`` again; ``x`` itself is what the backward pass differentiates."""
    return _Linear.apply(x, w, b, act, precision, 0.0, 0, None, x_image)


def linear_dropout(x, w, b, act, precision, p: float, training: bool = True, residual=None, seed: Optional[int] = None, x_image=None):
    """``residual + dropout(act(x W^T + b), p)`` as ONE node (dropout inside the dense block's epilogue / activation pass, its backward
    inside the gradient's packing pass); identical, bit for bit, to ``add(residual, dropout(linear(x, w, b, act), p, seed=seed))``."""
    if not training or p == 0.0:
        return _Linear.apply(x, w, b, act, precision, 0.0, 0, residual, x_image)
    if p >= 1.0:
        raise ValueError("dropout p must be < 1")
    return _Linear.apply(x, w, b, act, precision, float(p), next_seed() if seed is None else seed, residual, x_image)


class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x = x if x.is_contiguous() else x.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.eps = eps
        return ops.layernorm(x, weight, bias, eps)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx, dg, db = ops.layernorm_bwd(dy if dy.is_contiguous() else dy.contiguous(), x, weight, ctx.eps)
        return dx, dg, db, None


class _LayerNormImage(Function):
    """LayerNorm whose kernel also writes y as the operand image of the dense block that consumes it (ops.layernorm_packed): the
    image is a side output without a gradient; y carries the graph."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, precision):
        ctx.save_for_backward(x, weight)
        ctx.eps = eps
        y, img = ops.layernorm_packed(x, weight, bias, eps, precision, want_fp32=True)
        ctx.mark_non_differentiable(img)
        return y, img

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, _dimg):
        x, weight = ctx.saved_tensors
        dx, dg, db = ops.layernorm_bwd(dy if dy.is_contiguous() else dy.contiguous(), x, weight, ctx.eps)
        return dx, dg, db, None, None


class _LayerNormFork(Function):
    """x feeds a LayerNorm AND a residual connection (pre-norm transformer blocks): -> (LayerNorm(x), its operand image | None, x as
    the residual branch).  One node instead of two consumers of x: the backward pass adds the residual branch's gradient while the
    norm's dx is written (mdg_layernorm_bwd_add) instead of leaving the sum to a pass of the autograd engine."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, precision):
        ctx.save_for_backward(x, weight)
        ctx.eps = eps
        ctx.set_materialize_grads(False)
        img = None
        if precision in ("bf16", "bf16x3") and x.dim() == 2 and x.is_contiguous() and x.shape[1] % 64 == 0 and x.shape[0] > 0:
            y, img = ops.layernorm_packed(x, weight, bias, eps, precision, want_fp32=True)
            ctx.mark_non_differentiable(img)
        else:
            y = ops.layernorm(x if x.is_contiguous() else x.contiguous(), weight, bias, eps)
        return y, img, x.view_as(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, _dimg, dres):
        x, weight = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None, None
        dy = dy if dy.is_contiguous() else dy.contiguous()
        if dres is not None and not dres.is_contiguous():
            dres = dres.contiguous()
        dx, dg, db = ops.layernorm_bwd(dy, x if x.is_contiguous() else x.contiguous(), weight, ctx.eps, extra=dres)
        return dx, dg, db, None, None


def layernorm_fork(x, weight, bias, eps=1e-5, image_precision=None):
    """-> (LayerNorm(x), operand image of it | None, x for the residual connection): see _LayerNormFork."""
    return _LayerNormFork.apply(x, weight, bias, eps, image_precision)


def layernorm(x, weight, bias, eps=1e-5, image_precision=None):
    """``image_precision`` ("bf16" / "bf16x3"): -> (y, operand image of y | None) -- the image for ``linear(..., x_image=...)`` of the
    block that follows, written by the LayerNorm kernel itself (2-D contiguous x whose width is a multiple of 64)."""
    if image_precision is None:
        return _LayerNorm.apply(x, weight, bias, eps)
    if image_precision in ("bf16", "bf16x3") and x.dim() == 2 and x.is_contiguous() and x.shape[1] % 64 == 0 and x.shape[0] > 0:
        return _LayerNormImage.apply(x, weight, bias, eps, image_precision)
    return _LayerNorm.apply(x, weight, bias, eps), None


_bn_sync = {"reduce": None}


def set_batchnorm_sync(reduce_fn) -> None:
    """Data-parallel training: ``reduce_fn(t)`` sums a small device tensor over ranks in place; BatchNorm batch statistics
    (and their backward sums) are then taken over every rank's rows (SyncBatchNorm).  None switches it off."""
    _bn_sync["reduce"] = reduce_fn


class _SyncBatchNormAct(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, act, reduce_):
        x = x if x.is_contiguous() else x.contiguous()
        fused = act if act in (None, "none", "relu") else None
        y, stats, count = ops.sync_batchnorm_train_fwd(x, gamma, beta, running_mean, running_var, eps, momentum, fused, reduce_)
        pre = None
        if act == "relu":
            pre = y
        elif fused is None and act not in (None, "none"):
            pre, y = y, ops.activation_fwd(y, act)
        ctx.save_for_backward(x, stats, pre, count)
        ctx.act, ctx.affine, ctx.reduce_ = act, gamma is not None, reduce_
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, stats, pre, count = ctx.saved_tensors
        g = dy if dy.is_contiguous() else dy.contiguous()
        if pre is not None:
            g = ops.activation_bwd(g, pre, ctx.act)
        dx, dg, db = ops.sync_batchnorm_train_bwd(g, x, stats, count, ctx.reduce_)
        return dx, (dg if ctx.affine else None), (db if ctx.affine else None), None, None, None, None, None, None


class _BatchNormAct(Function):
    """nn.BatchNorm1d with batch statistics, fused with the activation that follows it."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, act):
        x = x if x.is_contiguous() else x.contiguous()
        fused = act if act in (None, "none", "relu") else None
        y, stats = ops.batchnorm_train_fwd(x, gamma, beta, running_mean, running_var, eps, momentum, act=fused)
        _BatchNormAct.last_stats = stats              # (read by record_batchnorm right after the call)
        pre = None
        if act == "relu":
            pre = y
        elif fused is None and act not in (None, "none"):
            pre, y = y, ops.activation_fwd(y, act)
        ctx.save_for_backward(x, stats, pre)
        ctx.act = act
        ctx.affine = gamma is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, stats, pre = ctx.saved_tensors
        g = dy if dy.is_contiguous() else dy.contiguous()
        if pre is not None:
            g = ops.activation_bwd(g, pre, ctx.act)
        dx, dg, db = ops.batchnorm_train_bwd(g, x, stats)
        return dx, (dg if ctx.affine else None), (db if ctx.affine else None), None, None, None, None, None


_bn_log = {"active": None}


class record_batchnorm:
    """``with record_batchnorm() as log:`` every training-mode BatchNorm forward inside appends (module, batch statistics,
    rows) to ``log``, so that ``replay_batchnorm(log)`` can later apply the same running-statistics update again without
    re-running the layers (a deterministic encoder that the reference runs twice on the same rows in one step)."""

    def __enter__(self):
        self.prev = _bn_log["active"]
        self.log = []
        _bn_log["active"] = self.log
        return self.log

    def __exit__(self, *exc):
        _bn_log["active"] = self.prev


def replay_batchnorm(log) -> None:
    with torch.no_grad():
        for bn, stats, rows in log:
            if bn.track_running_stats and bn.running_mean is not None:
                if bn.num_batches_tracked is not None:
                    bn.num_batches_tracked.add_(1)
                ops.batchnorm_replay_update(stats, bn.running_mean, bn.running_var, rows, bn.eps, bn.momentum)


def batchnorm_act(x, bn: torch.nn.BatchNorm1d, act=None):
    """Training-mode BatchNorm1d (+ activation) of a [rows, C] tensor; updates bn.running_* and num_batches_tracked."""
    if bn.momentum is None:
        raise NotImplementedError("BatchNorm1d(momentum=None) (cumulative average) is not used by the reference")
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    if _bn_sync["reduce"] is not None:
        # (synchronised statistics are not logged: encode() never shares passes under data parallelism)
        return _SyncBatchNormAct.apply(x, bn.weight, bn.bias, rm, rv, bn.eps, bn.momentum, act, _bn_sync["reduce"])
    y = _BatchNormAct.apply(x, bn.weight, bn.bias, rm, rv, bn.eps, bn.momentum, act)
    if _bn_log["active"] is not None:
        _bn_log["active"].append((bn, _BatchNormAct.last_stats, int(x.shape[0])))
    return y


class _Dropout(Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        return ops.dropout(x if x.is_contiguous() else x.contiguous(), p, seed)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return ops.dropout(dy if dy.is_contiguous() else dy.contiguous(), ctx.p, ctx.seed), None, None


def dropout(x, p: float, training: bool = True, seed: Optional[int] = None):
    if not training or p == 0.0:
        return x
    if p >= 1.0:
        raise ValueError("dropout p must be < 1")
    return _Dropout.apply(x, float(p), next_seed() if seed is None else seed)


class _AffineAct(Function):
    """Eval-mode BatchNorm (constant per-column scale / shift) + activation inside a differentiated graph."""

    @staticmethod
    def forward(ctx, x, scale, shift, act):
        x = x if x.is_contiguous() else x.contiguous()
        if act in (None, "none", "relu"):
            y = ops.affine_act(x, scale, shift, act)
            pre = y if act == "relu" else None
        else:
            pre = ops.affine_act(x, scale, shift, None)
            y = ops.activation_fwd(pre, act)
        ctx.save_for_backward(scale, pre)
        ctx.act = act
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        scale, pre = ctx.saved_tensors
        g = dy if dy.is_contiguous() else dy.contiguous()
        if pre is not None:
            g = ops.activation_bwd(g, pre, ctx.act)
        return ops.affine_act(g, scale, None, None), None, None, None


def affine_act(x, scale, shift, act=None):
    return _AffineAct.apply(x, scale, shift, act)


class _Axpby(Function):
    """alpha * a + beta * b with b broadcast over the leading dims of a (residual adds)."""

    @staticmethod
    def forward(ctx, a, b, alpha, beta):
        ctx.alpha, ctx.beta, ctx.bshape, ctx.same = alpha, beta, b.shape, b.numel() == a.numel()
        return ops.axpby(a, b, alpha, beta)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dy = dy if dy.is_contiguous() else dy.contiguous()
        da = db = None
        if ctx.needs_input_grad[0]:
            da = dy if ctx.alpha == 1.0 else ops.axpby(dy, dy, ctx.alpha, 0.0)
        if ctx.needs_input_grad[1]:
            if ctx.same:
                db = (dy if ctx.beta == 1.0 else ops.axpby(dy, dy, ctx.beta, 0.0)).view(ctx.bshape)
            else:
                nb = 1
                for s in ctx.bshape:
                    nb *= s
                db = ops.colsum(dy.reshape(-1, nb))
                db = (db if ctx.beta == 1.0 else ops.axpby(db, db, ctx.beta, 0.0)).view(ctx.bshape)
        return da, db, None, None


def add(a, b, alpha: float = 1.0, beta: float = 1.0):
    return _Axpby.apply(a, b, alpha, beta)


# ------------------------------------------------------------------------------------------- fusion
class _FusionAttention(Function):
    """Self-attention over the tokens of each tile (dense drug or packed live-token tile), dropout on the weights."""

    @staticmethod
    def forward(ctx, qkv, n, S, H, dh, kpm_bits, src_bits, row_start, row_bits, p_drop, seed):
        qkv = qkv if qkv.is_contiguous() else qkv.contiguous()
        out, _ = ops.fusion_attention(qkv, n, S, H, dh, kpm_bits, src_bits, row_start=row_start, row_bits=row_bits, p_drop=p_drop,
                                      seed=seed)
        ctx.save_for_backward(qkv, kpm_bits, src_bits, row_start, row_bits)
        ctx.meta = (n, S, H, dh, p_drop, seed)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        qkv, kpm_bits, src_bits, row_start, row_bits = ctx.saved_tensors
        n, S, H, dh, p_drop, seed = ctx.meta
        dqkv = ops.fusion_attention_bwd(qkv, dout if dout.is_contiguous() else dout.contiguous(), n, S, H, dh, kpm_bits, src_bits,
                                        row_start, row_bits, p_drop, seed)
        return (dqkv,) + (None,) * 10


def fusion_attention(qkv, n, S, H, dh, kpm_bits=None, src_bits=None, row_start=None, row_bits=None, p_drop=0.0, seed=None):
    p_drop = float(p_drop)
    return _FusionAttention.apply(qkv, n, S, H, dh, kpm_bits, src_bits, row_start, row_bits, p_drop,
                                  (next_seed() if p_drop > 0 else 0) if seed is None else seed)


class _XAttnPool(Function):
    @staticmethod
    def forward(ctx, q_proj, kv_proj, n, Tk, H, dh, p_drop, seed):
        kv_proj = kv_proj if kv_proj.is_contiguous() else kv_proj.contiguous()
        ctx.save_for_backward(q_proj, kv_proj)
        ctx.meta = (n, Tk, H, dh, p_drop, seed)
        return ops.xattn_pool(q_proj, kv_proj, n, Tk, H, dh, p_drop, seed)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        q_proj, kv_proj = ctx.saved_tensors
        n, Tk, H, dh, p_drop, seed = ctx.meta
        dq, dkv = ops.xattn_pool_bwd(q_proj, kv_proj, dout if dout.is_contiguous() else dout.contiguous(), n, Tk, H, dh, p_drop, seed)
        return dq.view(q_proj.shape), dkv, None, None, None, None, None, None


def xattn_pool(q_proj, kv_proj, n, Tk, H, dh, p_drop=0.0, seed=None):
    p_drop = float(p_drop)
    return _XAttnPool.apply(q_proj, kv_proj, n, Tk, H, dh, p_drop, (next_seed() if p_drop > 0 else 0) if seed is None else seed)


class _AssembleTokens(Function):
    @staticmethod
    def forward(ctx, str_emb, kg_emb, cv_emb, tx_emb, bottleneck, cls, pe, normalize, token_index):
        ctx.save_for_backward(str_emb, kg_emb, cv_emb, tx_emb, bottleneck, cls, token_index)
        ctx.pe_shape = None if pe is None else pe.shape
        ctx.normalize = normalize
        return ops.assemble_tokens(str_emb, kg_emb, cv_emb, tx_emb, bottleneck=bottleneck, cls=cls, pe=pe, normalize=normalize,
                                   token_index=token_index)

    @staticmethod
    @once_differentiable
    def backward(ctx, dseq):
        str_emb, kg_emb, cv_emb, tx_emb, bottleneck, cls, token_index = ctx.saved_tensors
        pe_len = 0 if ctx.pe_shape is None else int(ctx.pe_shape[-2])
        g = ops.assemble_tokens_bwd(dseq if dseq.is_contiguous() else dseq.contiguous(), str_emb, kg_emb, cv_emb, tx_emb,
                                    bottleneck=bottleneck, cls=cls, pe_len=pe_len, normalize=ctx.normalize, token_index=token_index)
        dpe = None if ctx.pe_shape is None else g["pe"].reshape(ctx.pe_shape)
        dcls = None if cls is None else g["cls"].reshape(cls.shape)
        dbn = None if bottleneck is None else g["bottleneck"].reshape(bottleneck.shape)
        return g["str"], g["kg"], g["cv"], g["tx"], dbn, dcls, dpe, None, None


def assemble_tokens(str_emb, kg_emb, cv_emb, tx_emb, bottleneck=None, cls=None, pe=None, normalize=False, token_index=None):
    return _AssembleTokens.apply(str_emb, kg_emb, cv_emb, tx_emb, bottleneck, cls, pe, bool(normalize), token_index)


class _L2Normalize(Function):
    @staticmethod
    def forward(ctx, x):
        x = x if x.is_contiguous() else x.contiguous()
        ctx.save_for_backward(x)
        return ops.l2_normalize(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.l2_normalize_bwd(dy if dy.is_contiguous() else dy.contiguous(), x)


def l2_normalize(x):
    return _L2Normalize.apply(x)


# ------------------------------------------------------------------------------------------- decoder
class _Symmetrize(Function):
    @staticmethod
    def forward(ctx, w):
        return ops.symmetrize(w if w.is_contiguous() else w.contiguous())

    @staticmethod
    @once_differentiable
    def backward(ctx, dws):
        return ops.symmetrize_bwd(dws if dws.is_contiguous() else dws.contiguous())


def symmetrize(w):
    return _Symmetrize.apply(w)


class _BilinearGather(Function):
    """Scores of the plan's (label, head, tail) triples only (plan order = sorted by label), through the (label, head drug)
    pairs: one 128 x 128 product per pair forward, one backward (ops.bilinear_gather_pairs).  ``w_sym`` is symmetric."""

    @staticmethod
    def forward(ctx, z_head, z_tail, w_sym, plan, precision="f32"):
        score, V = ops.bilinear_gather_pairs(z_head, z_tail, w_sym, plan, precision=precision)
        ctx.save_for_backward(z_head, z_tail, w_sym, V)
        ctx.plan, ctx.precision = plan, precision
        return score

    @staticmethod
    @once_differentiable
    def backward(ctx, ds):
        z_head, z_tail, w_sym, V = ctx.saved_tensors
        dzh, dzt, dw = ops.bilinear_gather_pairs_bwd(z_head, z_tail, w_sym, ctx.plan, ds if ds.is_contiguous() else ds.contiguous(), V,
                                                     need_dw=ctx.needs_input_grad[2], precision=ctx.precision)
        return (dzh if ctx.needs_input_grad[0] else None), (dzt if ctx.needs_input_grad[1] else None), dw, None, None


def bilinear_gather(z_head, z_tail, w_sym, plan, precision="f32"):
    """``precision``: the step's arithmetic mode.  "f32": exact fp32 matrix cores throughout.  In the 16-bit modes the head stays
    fp32-GRADE: its 128 x 128 products (per (label, drug) pair forward and backward, the weight gradient's outer-product sums) run on
    the split-bf16 matrix cores -- three products of hi / lo halves, fp32 accumulation, ~4e-6 of max."""
    return _BilinearGather.apply(z_head, z_tail, w_sym, plan, precision)


class _BCEWithSigmoid(Function):
    """nn.BCELoss(reduction)(sigmoid(score), target) -> scalar (madrigal/utils.py:616-619, train_ddi_batch.py:285-288)."""

    @staticmethod
    def forward(ctx, score, target, reduction):
        n = score.numel()
        gscale = 1.0 / max(n, 1) if reduction == "mean" else 1.0
        term, ds = ops.bce_logits(score, target, want_term=True, grad_scale=gscale)
        ctx.save_for_backward(ds)
        tot = ops.colsum(term.view(-1, 1))                       # fixed-order sum
        return (tot * gscale if reduction == "mean" else tot).reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss):
        (ds,) = ctx.saved_tensors
        return ops.mul_device_scalar(ds, dloss), None, None


def bce_with_sigmoid(score, target, reduction="mean"):
    if reduction not in ("mean", "sum"):
        raise NotImplementedError(f"loss_readout={reduction!r}")
    return _BCEWithSigmoid.apply(score, target, reduction)


# ------------------------------------------------------------------------------------------- graphs
class _CsrAggregate(Function):
    """out[v] = (add + coef_dev) * x[v] (optional) + sum_{e in row v} w_e x[col_e]  (GIN neighbourhood sum, read-out).
    The backward pass is the same kernel on the reversed edges (``tplan`` from graph_plans.transposed_csr)."""

    @staticmethod
    def forward(ctx, x, rowptr, col, w, tplan, with_self, coef_dev, coef_add, mean):
        x = x if x.is_contiguous() else x.contiguous()
        ctx.tplan, ctx.with_self, ctx.coef_add, ctx.cols = tplan, with_self, coef_add, x.shape[1]
        ctx.save_for_backward(coef_dev)
        return ops.csr_aggregate(x, rowptr, col, edge_weight=w, x_self=x if with_self else None, self_coef_dev=coef_dev,
                                 self_coef_add=coef_add, mean=mean)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (coef_dev,) = ctx.saved_tensors
        tp = ctx.tplan
        dout = dout if dout.is_contiguous() else dout.contiguous()
        dx = ops.csr_aggregate(dout, tp["rowptr"], tp["col"], edge_weight=tp["w"], x_self=dout if ctx.with_self else None,
                               self_coef_dev=coef_dev, self_coef_add=ctx.coef_add)
        return (dx[:, :ctx.cols],) + (None,) * 8


def csr_aggregate(x, rowptr, col, w, tplan, with_self=False, coef_dev=None, coef_add=0.0, mean=False):
    return _CsrAggregate.apply(x, rowptr, col, w, tplan, with_self, coef_dev, coef_add, mean)


class _HgtAttention(Function):
    """Edge softmax + aggregation of EVERY destination node type of one HGTConv in one node of the tape, so that the
    key / value gradient buffer (layout of the flat projection buffer) is allocated and zero-filled once."""

    @staticmethod
    def forward(ctx, kv, heads, plans, *qs):
        from .graph_plans import hgt_reverse_plan
        outs, stats = [], []
        for q, pd in zip(qs, plans):
            o, s = ops.hgt_attention_stats(q, kv, pd, heads)
            outs.append(o)
            stats.append(s)
        ctx.heads, ctx.plans = heads, plans
        ctx.revs = [hgt_reverse_plan(pd) for pd in plans]
        ctx.n = len(qs)
        ctx.save_for_backward(kv, *qs, *outs, *stats)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *douts):
        n = ctx.n
        saved = ctx.saved_tensors
        kv, qs, outs, stats = saved[0], saved[1:1 + n], saved[1 + n:1 + 2 * n], saved[1 + 2 * n:1 + 3 * n]
        dkv = torch.zeros_like(kv)
        dqs = []
        for q, pd, rev, o, s, g in zip(qs, ctx.plans, ctx.revs, outs, stats, douts):
            g = g if g.is_contiguous() else g.contiguous()
            dqs.append(ops.hgt_attention_bwd(q, kv, pd, rev, ctx.heads, g, o, s, dkv))
        return (dkv, None, None) + tuple(dqs)


def hgt_attention_all(kv, heads, plans, qs):
    """-> tuple of pre-activation attention outputs, one per (q, plan) pair."""
    return _HgtAttention.apply(kv, heads, plans, *qs)


class _Activation(Function):
    @staticmethod
    def forward(ctx, x, act):
        x = x if x.is_contiguous() else x.contiguous()
        ctx.act = act
        y = ops.activation_fwd(x, act)
        ctx.save_for_backward(y if act == "relu" else x)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (pre,) = ctx.saved_tensors
        return ops.activation_bwd(dy if dy.is_contiguous() else dy.contiguous(), pre, ctx.act), None


def activation(x, act):
    return x if act in (None, "none") else _Activation.apply(x, act)


class _GatedResidual(Function):
    @staticmethod
    def forward(ctx, o, x, skip):
        o = o if o.is_contiguous() else o.contiguous()
        x = x if x.is_contiguous() else x.contiguous()
        ctx.save_for_backward(o, x, skip)
        return ops.gated_residual(o, x, skip.detach())

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        o, x, skip = ctx.saved_tensors
        d_o, d_x, d_skip = ops.gated_residual_bwd(dout if dout.is_contiguous() else dout.contiguous(), o, x, skip.detach())
        return d_o, d_x, d_skip.view(skip.shape)


def gated_residual(o, x, skip):
    return _GatedResidual.apply(o, x, skip)


_gather_plan = {}


class _GatherRows(Function):
    """table[idx] with repeated indices (embedding lookup).  torch's own backward scatters with atomics (run-to-run
    different low bits); here the gradient rows are summed per table row in index order by mdg_csr_aggregate."""

    @staticmethod
    def forward(ctx, table, idx):
        ctx.save_for_backward(idx)
        ctx.n_rows = table.shape[0]
        return table.index_select(0, idx)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        dout = dout if dout.is_contiguous() else dout.contiguous()
        if ctx.n_rows <= 128 and idx.numel() >= 4096 and bool((idx.numel() % 4) == 0):
            # a small table (the 16 cell lines) under 10^4..10^5 lookups: one workgroup per table row would add its rows one
            # after the other; as onehot(idx)^T dout the sum is the split-reduction weight-gradient kernel (exact fp32
            # matrix cores, the rows of a split in order, the splits in order: deterministic)
            onehot = torch.zeros((idx.numel(), ctx.n_rows), dtype=torch.float32, device=dout.device)
            onehot.scatter_(1, idx.clamp_min(0).unsqueeze(1), (idx >= 0).to(torch.float32).unsqueeze(1))     # (idx < 0: _GatherRowsOr's default rows)
            return ops.grad_weight(onehot, dout), None
        # the index tensor of an embedding lookup is fixed for a run (cell-line ids of the tx stack): its sort is kept per
        # (storage, version, table size); the entry holds the tensor, so the address cannot be recycled under it
        key = (idx.data_ptr(), idx._version, idx.numel(), ctx.n_rows)
        hit = _gather_plan.get(key)
        if hit is None:
            order = torch.argsort(idx, stable=True)
            # row pointers by binary search in the sorted indices: no host synchronisation (torch.bincount reads the maximum
            # back to the host, which would stall the launch queue in the middle of the backward pass)
            rowptr = torch.searchsorted(idx[order], torch.arange(ctx.n_rows + 1, device=idx.device))
            hit = (key, order.contiguous(), rowptr, idx)
            while len(_gather_plan) >= 48:                # a dozen lookup sites per side (cell-line ids, the drug-row glue of encode): oldest out
                _gather_plan.pop(next(iter(_gather_plan)))
            _gather_plan[key] = hit
        order, rowptr = hit[1], hit[2]
        return ops.csr_aggregate(dout, rowptr, order)[:, :dout.shape[1]], None


def gather_rows(table, idx):
    return _GatherRows.apply(table, idx)


class _GatherRowsOr(Function):
    """out[d] = table[idx[d]] where idx[d] >= 0, default[d] (a constant) elsewhere: the KG rows of a drug batch, filler rows for the
    drugs that are not in the KG (madrigal/models/models.py:734-736 does it with an indexed assignment into a filler table and an
    indexed read: torch's backward of that pair sorts and scatters).  Backward: _GatherRows' (the sorted-index plan skips idx < 0)."""

    @staticmethod
    def forward(ctx, table, idx, default):
        ctx.save_for_backward(idx)
        ctx.n_rows = table.shape[0]
        got = table.index_select(0, idx.clamp_min(0))
        return torch.where((idx >= 0).unsqueeze(1), got, default)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        return _GatherRows.backward(ctx, dout) + (None,)


def gather_rows_or(table, idx, default):
    return _GatherRowsOr.apply(table, idx, default)


# ------------------------------------------------------------------------------------------- contrastive pretraining
class _InfoNCE(Function):
    """SimCLR_NovelDDI.contrastive_loss (madrigal/models/simclr.py:74-108) on the already L2-normalised features
    F = normalize(cat(aug1, aug2)): returns (logits, labels, loss); only the loss carries a gradient."""

    @staticmethod
    def forward(ctx, f, hard, temperature, precision):
        f = f if f.is_contiguous() else f.contiguous()
        B = f.shape[0] // 2
        sim = ops.linear(f, f, None, precision=precision, cache_weight=False)
        hard_u8 = None if hard is None else hard.to(device=f.device, dtype=torch.uint8).contiguous()
        logits = torch.empty((2 * B, 2 * B - 1), dtype=torch.float32, device=f.device)
        labels = torch.empty_like(logits)
        row = torch.empty(2 * B, dtype=torch.float32, device=f.device)
        loss = torch.empty(1, dtype=torch.float32, device=f.device)
        ops.check(ops.lib().mdg_infonce_finish(ops._ptr(sim), ops._ptr(hard_u8), ops._ptr(logits), ops._ptr(labels), ops._ptr(row), ops._ptr(loss),
                                               ops._c64(B), ops._f(temperature), ops._stream(f)), "mdg_infonce_finish")
        ctx.save_for_backward(f, sim, hard_u8)
        ctx.temperature, ctx.precision = temperature, precision
        ctx.mark_non_differentiable(logits, labels)
        return logits, labels, loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, _dlogits, _dlabels, dloss):
        f, sim, hard = ctx.saved_tensors
        dsim = ops.info_nce_bwd(sim, hard, dloss, ctx.temperature)
        g = ops.axpby(dsim, ops.transpose(dsim, pad_inner=False))            # sim = F F^T: dF = (dsim + dsim^T) F
        df = ops.linear(g, ops.transpose(f), precision=ctx.precision, cache_weight=False)
        return df[:, :f.shape[1]], None, None, None


def info_nce(aug1, aug2, too_hard_neg, temperature: float, precision="bf16x3"):
    f = l2_normalize(torch.cat([aug1, aug2], dim=0))
    return _InfoNCE.apply(f, too_hard_neg, float(temperature), precision)


class _BilinearAllPairs(Function):
    """The dense head S[l,i,j] = z_h[i]^T W[l] z_t[j] as a differentiable [L,Nh,Nt] tensor: what the reference's own loop
    differentiates (train_ddi_batch.py:285-288: sigmoid(model(...))[labels, heads, tails] -> BCELoss -> backward), so that
    the loop runs unchanged.  The incoming gradient is dense (torch builds it from the index / sigmoid backward); four
    GEMMs through the HIP kernel turn it into dz_h, dz_t, dW.  Memory like the reference: a second [L,Nh,Nt] buffer for the
    transposed gradient.  The gathered head (score_triples) is the fast path; this one is the drop-in path."""

    @staticmethod
    def forward(ctx, z_head, z_tail, w, precision):
        ctx.save_for_backward(z_head, z_tail, w)
        ctx.precision = precision
        return ops.bilinear_allpairs(z_head, z_tail, w, precision=precision)

    @staticmethod
    @once_differentiable
    def backward(ctx, dS):
        zh, zt, w = ctx.saved_tensors
        L, Nh, Nt, D = w.shape[0], zh.shape[0], zt.shape[0], zh.shape[1]
        prec = ctx.precision
        lin = lambda x, wt: ops.linear(x, wt, precision=prec, cache_weight=False)       # noqa: E731
        dS2 = (dS if dS.is_contiguous() else dS.contiguous()).view(L * Nh, Nt)
        dzh = dzt = dw = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:
            U = lin(dS2, ops.transpose(zt)).view(L, Nh, D)                               # U[l,i,:] = sum_j dS[l,i,j] z_t[j,:]
            if ctx.needs_input_grad[0]:                                                  # dz_h[i,k] = sum_{l,m} U[l,i,m] W[l,k,m]
                dzh = lin(U.permute(1, 0, 2).reshape(Nh, L * D), w.permute(1, 0, 2).reshape(D, L * D))
            if ctx.needs_input_grad[2]:                                                  # dW[l,k,m] = sum_i z_h[i,k] U[l,i,m]
                y = lin(U.transpose(1, 2).reshape(L * D, Nh), ops.transpose(zh))         # y[(l,m),k]
                dw = y[:, :D].reshape(L, D, D).transpose(1, 2).contiguous()
        if ctx.needs_input_grad[1]:                                                      # dz_t[j,m] = sum_{l,i} dS[l,i,j] (z_h W_l)[i,m]
            V = lin(zh, w.transpose(1, 2).reshape(L * D, D)).view(Nh, L, D).permute(1, 0, 2).reshape(L * Nh, D)
            dzt = lin(ops.transpose(dS2), ops.transpose(V))
        return (None if dzh is None else dzh[:, :D].contiguous()), (None if dzt is None else dzt[:, :D].contiguous()), dw, None


def bilinear_allpairs(z_head, z_tail, w, precision="bf16x3"):
    return _BilinearAllPairs.apply(z_head, z_tail, w, precision)


# ------------------------------------------------------------------------------------------- HGT on the flat projection buffer
class _HgtProject(Function):
    """All node types' composite projections  x_t -> [ q | k'_0 v'_0 | k'_1 v'_1 ... ]  written straight into the flat
    buffer the edge-attention kernels address (graph_plans.hgt_plan layout): one GEMM per node type, no concatenation.
    args = x_0..x_{n-1}, W_0..W_{n-1}, b_0..b_{n-1} for the n projected node types (``layout`` = their (offset, rows, width))."""

    @staticmethod
    def forward(ctx, layout, total_floats, precision, *args):
        n = len(layout)
        xs, ws, bs = args[:n], args[n:2 * n], args[2 * n:]
        dev = ws[0].device
        flat = torch.zeros(max(total_floats, 128), dtype=torch.float32, device=dev)       # unprojected types stay zero
        xs2 = []
        for (off, rows, width), x, w, b in zip(layout, xs, ws, bs):
            x2 = x if x.is_contiguous() else x.contiguous()
            xs2.append(x2)
            if rows:
                ops.linear(x2, w, b, precision=precision, out=flat[off:off + rows * width].view(rows, width), cache_weight=False)
        ctx.layout, ctx.precision, ctx.n = layout, precision, n
        ctx.save_for_backward(*xs2, *ws)
        return flat.view(-1, 128)

    @staticmethod
    @once_differentiable
    def backward(ctx, dflat):
        n = ctx.n
        xs, ws = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        dflat = (dflat if dflat.is_contiguous() else dflat.contiguous()).view(-1)
        dxs, dws, dbs = [], [], []
        for i, ((off, rows, width), x, w) in enumerate(zip(ctx.layout, xs, ws)):
            g = dflat[off:off + rows * width].view(rows, width)
            dx = None
            if ctx.needs_input_grad[3 + i] and rows:
                dx = ops.linear(g, ops.transpose(w), precision=ctx.precision, cache_weight=False)[:, :x.shape[1]]
            elif ctx.needs_input_grad[3 + i]:
                dx = torch.zeros_like(x)
            dw, db = ops.grad_weight(g, x, ctx.precision, want_bias=True)
            dxs.append(dx)
            dws.append(dw)
            dbs.append(db)
        return (None, None, None, *dxs, *dws, *dbs)


def hgt_project(layout, total_floats, precision, xs, ws, bs):
    return _HgtProject.apply(tuple(layout), total_floats, precision, *xs, *ws, *bs)


class _HgtProjectRows(Function):
    """_HgtProject with the composite weights of all node types stacked in ONE matrix (rows offs[i]..offs[i+1] belong to type
    i; all types share the input width): one weight / bias gradient tensor comes back instead of one pair per type, so the
    parameter-space graph above it stays small.  args = big_w [rows,in], big_b [rows], x_0..x_{n-1}."""

    @staticmethod
    def forward(ctx, layout, total_floats, precision, offs, big_w, big_b, *xs):
        flat = torch.zeros(max(total_floats, 128), dtype=torch.float32, device=big_w.device)      # unprojected types stay zero
        big_w = big_w if big_w.is_contiguous() else big_w.contiguous()
        big_b = big_b if big_b.is_contiguous() else big_b.contiguous()
        xs2 = [x if x.is_contiguous() else x.contiguous() for x in xs]
        lanes = _type_lanes(flat, len(layout), chains=True)            # one GEMM per node type into disjoint rows of ``flat``
        for i, ((off, rows, width), x2) in enumerate(zip(layout, xs2)):
            if offs[i + 1] - offs[i] != width:
                raise ValueError("hgt_project_rows: weight rows disagree with the projection width")
            if rows:
                with lanes.lane(i):
                    ops.linear(x2, big_w[offs[i]:offs[i + 1]], big_b[offs[i]:offs[i + 1]], precision=precision,
                               out=flat[off:off + rows * width].view(rows, width), cache_weight=False)
        lanes.join()
        ctx.layout, ctx.precision, ctx.offs = layout, precision, offs
        ctx.save_for_backward(big_w, *xs2)
        return flat.view(-1, 128)

    @staticmethod
    @once_differentiable
    def backward(ctx, dflat):
        big_w, xs = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        offs = ctx.offs
        dflat = (dflat if dflat.is_contiguous() else dflat.contiguous()).view(-1)
        dw = torch.empty_like(big_w)
        db = torch.empty(big_w.shape[0], dtype=torch.float32, device=big_w.device)
        dxs = []
        lanes = _type_lanes(dflat, len(xs), chains=True)               # per type: dx, and its rows of dw / db (disjoint)
        for i, ((off, rows, width), x) in enumerate(zip(ctx.layout, xs)):
            g = dflat[off:off + rows * width].view(rows, width)
            w = big_w[offs[i]:offs[i + 1]]
            dx = None
            with lanes.lane(i):
                if ctx.needs_input_grad[6 + i]:
                    dx = ops.linear(g, ops.transpose(w), precision=ctx.precision, cache_weight=False)[:, :x.shape[1]] if rows else torch.zeros_like(x)
                ops.grad_weight(g, x, ctx.precision, want_bias=True, out=(dw[offs[i]:offs[i + 1]], db[offs[i]:offs[i + 1]]))
            dxs.append(dx)
        lanes.join(*dxs)
        return (None, None, None, None, dw, db, *dxs)


class _HgtComposite(Function):
    """Composite projection weights of all projected node types of an HGT conv from the live parameters, and the parameter
    gradients from the gradient of those rows: mdg_hgt_composite_fwd / _bwd (csrc/hgt_params.hip).  args: meta (dict of device
    index tables + sizes, built once per conv and relation set), k_rel.weight, v_rel.weight, then the kqv weights of the projected
    types, their biases, and the p_rel parameter of EVERY edge type (in edge-type order).  -> big_w [rows,in], big_b [rows].
    Every parameter's gradient is a fresh view of one flat buffer (no accumulation inside: a conv's parameters enter once)."""

    @staticmethod
    def forward(ctx, meta, k_rel, v_rel, *params):
        nt, R = meta["n_types"], meta["n_edge_types"]
        ws, bs, ps = params[:nt], params[nt:2 * nt], params[2 * nt:]
        assert len(ps) == R
        for t in (k_rel, v_rel, *params):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError("hgt_composite: contiguous fp32 GPU parameters expected")
        ptr_key = tuple(t.data_ptr() for t in params)
        if meta.get("ptr_key") != ptr_key:                  # parameter storage moved (.to(), load_state_dict keeps it): rebuild the pointer tables
            pin = torch.tensor(ptr_key, dtype=torch.int64).pin_memory()
            meta["ptrs"] = pin.to(k_rel.device, non_blocking=True)
            meta["ptr_key"], meta["ptr_pin"] = ptr_key, pin
        ptrs = meta["ptrs"]
        rows, cin = meta["rows"], meta["cin"]
        big_w = torch.empty((rows, cin), dtype=torch.float32, device=k_rel.device)
        big_b = torch.empty(rows, dtype=torch.float32, device=k_rel.device)
        ops.hgt_composite(ptrs, k_rel, v_rel, meta, big_w, big_b)
        ctx.meta = meta
        ctx.save_for_backward(k_rel, v_rel, *params)
        return big_w, big_b

    @staticmethod
    @once_differentiable
    def backward(ctx, dbig_w, dbig_b):
        meta = ctx.meta
        k_rel, v_rel = ctx.saved_tensors[0], ctx.saved_tensors[1]
        params = ctx.saved_tensors[2:]
        nt, R, F, H, cin = meta["n_types"], meta["n_edge_types"], meta["F"], meta["H"], meta["cin"]
        D = F // H
        per_type = 3 * F * cin + 3 * F
        rel = H * R * D * D
        flat = torch.empty(nt * per_type + 2 * rel + R * H, dtype=torch.float32, device=k_rel.device)
        dbig_w = dbig_w if dbig_w.is_contiguous() else dbig_w.contiguous()
        dbig_b = dbig_b if dbig_b.is_contiguous() else dbig_b.contiguous()
        ops.hgt_composite_bwd(meta["ptrs"], k_rel, v_rel, meta, dbig_w, dbig_b, flat)
        gw = [flat[i * per_type: i * per_type + 3 * F * cin].view(3 * F, cin) for i in range(nt)]
        gb = [flat[i * per_type + 3 * F * cin: (i + 1) * per_type] for i in range(nt)]
        base = nt * per_type
        gk = flat[base: base + rel].view(H * R, D, D)
        gv = flat[base + rel: base + 2 * rel].view(H * R, D, D)
        gp = [flat[base + 2 * rel + r * H: base + 2 * rel + (r + 1) * H].view(params[2 * nt + r].shape) for r in range(R)]
        return (None, gk, gv, *gw, *gb, *gp)


def hgt_composite(meta, k_rel, v_rel, ws, bs, ps):
    return _HgtComposite.apply(meta, k_rel, v_rel, *ws, *bs, *ps)


def hgt_project_rows(layout, total_floats, precision, xs, big_w, big_b, offs):
    return _HgtProjectRows.apply(tuple(layout), total_floats, precision, tuple(offs), big_w, big_b, *xs)


_hgt_lanes = {}


class _type_lanes:
    """The destination types of a conv are independent launches (every type reads the shared projection buffer and writes its own
    rows / its own relations' column blocks): they are dealt to a few side streams that fork from the current stream and join it
    again, so that the small types' kernels run beside the large ones instead of in a chain of launch gaps.  One lane (or a
    CPU tensor) keeps the plain loop.
    Inside a stream capture (the KG pass replayed as hipGraphs, NovelDDIEncoder._kg_graphed) the fork / join become edges of the
    graph: parallel branches cost the replay nothing, so there the per-type CHAINS of small launches (``chains=True``: one
    projection GEMM per node type, a type's GELU / out_lin / gated residual, their backward) are dealt out too, on
    two (measured best of 1-4) branches.  Eager launches pay an event round trip per fork and join, which costs the chains more
    than their concurrency returns (finetune step 40.5 -> 41.5 ms): outside a capture ``chains`` keeps the plain loop."""

    def __init__(self, ref: torch.Tensor, n_items: int, chains: bool = False):
        capturing = ref.is_cuda and torch.cuda.is_current_stream_capturing()
        if chains:
            want = 2 if (capturing and _bn_sync["reduce"] is None) else 1        # measured best of 1-4 branches
        else:
            want = 2
        self.cur = self.lanes = None
        if ref.is_cuda and want > 1 and n_items > 1:
            self.cur = torch.cuda.current_stream(ref.device)
            key = (ref.device, self.cur.cuda_stream, want)
            if key not in _hgt_lanes:
                _hgt_lanes[key] = [torch.cuda.Stream(device=ref.device) for _ in range(want)]
            self.lanes = _hgt_lanes[key]
            for st in self.lanes:
                st.wait_stream(self.cur)

    def lane(self, i: int):
        import contextlib
        return contextlib.nullcontext() if self.lanes is None else torch.cuda.stream(self.lanes[i % len(self.lanes)])

    def join(self, *tensors):
        if self.lanes is None:
            return
        for st in self.lanes:
            self.cur.wait_stream(st)
        for t in tensors:
            if isinstance(t, torch.Tensor):
                t.record_stream(self.cur)


class _HgtAttentionFlat(Function):
    """Edge attention of every destination type, queries read from / query gradients written to the flat buffer itself:
    one gradient tensor for the whole projection buffer, no per-type scatter of dq."""

    @staticmethod
    def forward(ctx, flat, heads, plans, qspec, rows16=False):
        from .graph_plans import hgt_reverse_plan
        outs, stats = [], []
        f1 = flat.view(-1)
        # reduced-precision mode: the k' | v' rows are gathered from a bf16 mirror of the projection buffer (half the bytes per edge:
        # the gathers ARE these kernels' time); queries, statistics, sums and gradients stay fp32
        flat16 = ops.f32_to_bf16(flat) if (rows16 and flat.numel() % 8 == 0) else None
        lanes = _type_lanes(flat, len(plans))
        for i, (pd, (off, rows, width)) in enumerate(zip(plans, qspec)):
            q = f1[off:off + rows * width].view(rows, width)[:, 0:128]
            with lanes.lane(i):
                o, s = ops.hgt_attention_stats(q, flat, pd, heads, kv16=flat16)
            outs.append(o)
            stats.append(s)
        lanes.join(*outs, *stats)
        ctx.heads, ctx.plans, ctx.qspec, ctx.n = heads, plans, qspec, len(plans)
        ctx.revs = [hgt_reverse_plan(pd) for pd in plans]
        ctx.flat16 = flat16
        ctx.save_for_backward(flat, *outs, *stats)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *douts):
        n = ctx.n
        flat, outs, stats = ctx.saved_tensors[0], ctx.saved_tensors[1:1 + n], ctx.saved_tensors[1 + n:1 + 2 * n]
        dflat = torch.zeros_like(flat)
        f1, d1 = flat.view(-1), dflat.view(-1)
        gs = [g if g.is_contiguous() else g.contiguous() for g in douts]
        lanes = _type_lanes(flat, n)                  # (after dflat's zero fill and the gradients' copies: the lanes wait for them)
        for i, (pd, rev, (off, rows, width), o, s, g) in enumerate(zip(ctx.plans, ctx.revs, ctx.qspec, outs, stats, gs)):
            q = f1[off:off + rows * width].view(rows, width)[:, 0:128]
            dq = d1[off:off + rows * width].view(rows, width)[:, 0:128]
            with lanes.lane(i):
                ops.hgt_attention_bwd(q, flat, pd, rev, ctx.heads, g, o, s, dflat, dq_out=dq, kv16=ctx.flat16)
        lanes.join()
        return dflat, None, None, None, None


type_lanes = _type_lanes


def hgt_attention_flat(flat, heads, plans, qspec, rows16=False):
    return _HgtAttentionFlat.apply(flat, heads, tuple(plans), tuple(qspec), bool(rows16))
