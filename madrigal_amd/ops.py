"""Python wrappers of the C ABI: validate tensors, pass raw pointers + the current HIP stream.

torch is used for device memory and streams only; all arithmetic is in libmadrigal_hip.so.
Every wrapper raises ``ValueError`` for bad shapes/dtypes/devices (the reference raises
assertion errors in the same situations) and ``MadrigalHipError`` if the library is missing.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from ._lib import check, lib

PREC_F32, PREC_BF16X3, PREC_BF16, PREC_F16 = 0, 1, 2, 3
PRECISIONS = {"f32": PREC_F32, "bf16x3": PREC_BF16X3, "bf16": PREC_BF16}
HEAD_PRECISIONS = dict(PRECISIONS, f16=PREC_F16)          # the all-pairs head also runs on the fp16 matrix cores
EPI_STORE, EPI_STORE_SIGMOID, EPI_ROWSTATS, EPI_TRIKEYS = 0, 1, 2, 3

_c64 = ctypes.c_int64
_vp = ctypes.c_void_p


def _prec(p) -> int:
    if isinstance(p, str):
        if p not in PRECISIONS:
            raise ValueError(f"unknown precision {p!r}; expected one of {sorted(PRECISIONS)}")
        return PRECISIONS[p]
    return int(p)


_raw_stream_of = torch._C._cuda_getCurrentRawStream      # the current stream's handle without building a torch.cuda.Stream (~1 000 lookups per training step)


def _stream_handle(device) -> int:
    idx = device.index
    return _raw_stream_of(torch.cuda.current_device() if idx is None else idx)


def _stream(t: torch.Tensor) -> _vp:
    return _vp(_stream_handle(t.device))


def _ptr(t: Optional[torch.Tensor]) -> _vp:
    return _vp(0 if t is None else t.data_ptr())


def _f32_cuda(t: torch.Tensor, name: str, ndim: Optional[int] = None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise ValueError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise ValueError(f"{name}: must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        raise ValueError(f"{name}: expected float32, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dims, got shape {tuple(t.shape)}")
    return t if t.is_contiguous() else t.contiguous()


_ws_cache = {}


def _workspace(nbytes: int, device) -> Optional[torch.Tensor]:
    """Per-device grow-only scratch buffer (the C ABI never allocates)."""
    if nbytes == 0:
        return None
    if torch.cuda.is_current_stream_capturing():
        # inside a hipGraph capture: scratch from the graph's own pool (a cached buffer would be shared between the replays and
        # whatever eager op later lands on a stream with the same handle)
        return torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=device)
    # one scratch buffer per (device, stream): ops enqueued on different streams may run concurrently
    key = (device.type, device.index, _stream_handle(device))
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# ------------------------------------------------------------------------------- head
def symmetrize(w_original: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """triu(W) + triu(W,1)^T per outcome (madrigal/models/models.py:522-524)."""
    w = _f32_cuda(w_original, "w_original", 3)
    if w.shape[1] != w.shape[2]:
        raise ValueError(f"w_original: expected [L,D,D], got {tuple(w.shape)}")
    out = torch.empty_like(w) if out is None else _f32_cuda(out, "out", 3)
    if out.shape != w.shape:
        raise ValueError("out: shape mismatch")
    check(lib().mdg_symmetrize(_ptr(w), _ptr(out), _c64(w.shape[0]), _c64(w.shape[1]), _stream(w)), "mdg_symmetrize")
    return out


def bilinear_allpairs(z_head: torch.Tensor, z_tail: torch.Tensor, w_sym: torch.Tensor, *, precision="bf16x3",
                      epilogue: int = EPI_STORE, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """All-pairs scores S[l,i,j] = z_head[i]^T W_sym[l] z_tail[j]  -> [L,Nh,Nt] fp32
    (or [L,Nh,2] row statistics with ``EPI_ROWSTATS``).  madrigal/models/models.py:537-547.

    ``EPI_TRIKEYS`` (one drug set against itself, ``z_head is z_tail``): an int32 tensor whose strict lower triangle holds the
    order keys of the scores, for ``rank_normalize`` (which reads nothing else); everything above the diagonal blocks is left
    unwritten -- half the store stream of ``EPI_STORE``."""
    zh, zt, w = _f32_cuda(z_head, "z_head", 2), _f32_cuda(z_tail, "z_tail", 2), _f32_cuda(w_sym, "w_sym", 3)
    D = zh.shape[1]
    if zt.shape[1] != D or w.shape[1] != D or w.shape[2] != D:
        raise ValueError(f"feature dims disagree: z_head {tuple(zh.shape)}, z_tail {tuple(zt.shape)}, w {tuple(w.shape)}")
    if zh.device != zt.device or zh.device != w.device:
        raise ValueError("z_head, z_tail and w_sym must be on the same device")
    L, Nh, Nt = w.shape[0], zh.shape[0], zt.shape[0]
    shape = (L, Nh, 2) if epilogue == EPI_ROWSTATS else (L, Nh, Nt)
    out_dtype = torch.int32 if epilogue == EPI_TRIKEYS else torch.float32
    if out is None:
        out = empty_scores(L, Nh, Nt, zh.device).view(torch.int32) if epilogue == EPI_TRIKEYS else torch.empty(shape, dtype=torch.float32, device=zh.device)
    else:
        if not (isinstance(out, torch.Tensor) and out.is_cuda and out.dtype == out_dtype):
            raise ValueError(f"out: expected a {out_dtype} GPU tensor")
        # contiguous, or a view of row-padded storage (empty_scores): unit inner stride and one pitch for every row
        if tuple(out.shape) != shape or (out.numel() and (out.stride(2) != 1 or out.stride(1) < shape[2] or out.stride(0) != shape[1] * out.stride(1))):
            raise ValueError(f"out: expected {shape}, contiguous or row-pitched (empty_scores), got {tuple(out.shape)} strides {tuple(out.stride())}")
    ldo = out.stride(1) if out.numel() else shape[2]
    if isinstance(precision, str) and precision not in HEAD_PRECISIONS:
        raise ValueError(f"unknown precision {precision!r}; expected one of {sorted(HEAD_PRECISIONS)}")
    prec = HEAD_PRECISIONS[precision] if isinstance(precision, str) else int(precision)
    L_ = lib()
    # the grid's y extent caps one call at 65535 outcomes; chunk above that
    for lo in range(0, max(L, 1), 65535):
        hi = min(L, lo + 65535)
        if hi <= lo:
            break
        nbytes = L_.mdg_bilinear_allpairs_workspace_bytes(_c64(Nh), _c64(Nt), _c64(hi - lo), _c64(D), prec)
        ws = _workspace(nbytes, zh.device)
        check(L_.mdg_bilinear_allpairs_ld(_ptr(zh), _ptr(zt), _vp(w.data_ptr() + lo * D * D * 4),
                                          _vp(out.data_ptr() + lo * out.stride(0) * 4), _c64(ldo), _c64(Nh), _c64(Nt), _c64(hi - lo),
                                          _c64(D), prec, int(epilogue), _ptr(ws), ctypes.c_size_t(nbytes), _stream(zh)),
              "mdg_bilinear_allpairs")
    return out


def empty_scores(L: int, Nh: int, Nt: int, device) -> torch.Tensor:
    """An uninitialised [L,Nh,Nt] fp32 score (or rank) tensor in the layout the head writes fastest: rows padded to a multiple of
    32 floats, so that every row starts on a 128-byte line whatever Nt is (the real drug counts -- 11 607 in
    generate_embeddings.ipynb -- are not multiples of anything).  For Nt % 32 == 0 this is a plain contiguous tensor; otherwise a
    [:, :, :Nt] view of the padded storage: same values and indexing, ``.contiguous()`` compacts it."""
    unit = 32                                                     # floats: one 128-byte line (scripts/head_ragged_bench.py compared 4 .. 64)
    pitch = (Nt + unit - 1) // unit * unit
    return torch.empty((L, Nh, pitch), dtype=torch.float32, device=device)[:, :, :Nt]


# ------------------------------------------------------------------------------- dense blocks
ACTS = {None: 0, "none": 0, "relu": 1, "gelu": 2, "sigmoid": 3, "tanh": 4, "leakyrelu": 5, "softplus": 6, "selu": 7}
_c = ctypes.c_int
_f = ctypes.c_float

_pad_cache = {}


def forward_only(*tensors) -> None:
    """The raw wrappers of this module are forward-only: refuse to run under autograd rather than silently return
    tensors that do not carry gradients.  The differentiable entry points are in madrigal_amd.autograd (each node
    calls these wrappers for its forward and its backward pass with autograd switched off)."""
    if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors):
        raise RuntimeError("madrigal_amd.ops wrappers are forward-only: use madrigal_amd.autograd (or the model classes) "
                           "to record a gradient, or call under torch.no_grad()")


def _pad_last(t: torch.Tensor, mult: int = 4) -> torch.Tensor:
    k = t.shape[-1]
    if k % mult == 0:
        return t
    return torch.nn.functional.pad(t, (0, mult - k % mult))


def _wkey(w: torch.Tensor):
    return (w.data_ptr(), tuple(w.shape), tuple(w.stride()), str(w.device))


def padded_weight(w: torch.Tensor) -> torch.Tensor:
    """nn.Linear weight [N,K] with K zero-padded to a multiple of 4.  Cached per storage address and in-place
    version; the entry keeps the source tensor alive, so the address cannot be recycled under the cache."""
    if w.shape[-1] % 4 == 0 and w.is_contiguous():
        return w.detach()
    k = _wkey(w)
    hit = _pad_cache.get(k)
    if hit is not None and hit[0] == w._version:
        return hit[1]
    p = _pad_last(w.detach()).contiguous()
    _pad_cache[k] = (w._version, p, w)
    return p


_pack_cache = {}


def packed_weight_image(w: torch.Tensor, prec: int):
    """Operand image of a (padded) nn.Linear weight for mdg_linear, built once per (storage, version, precision)."""
    N, K = w.shape
    nbytes = lib().mdg_pack_operand_bytes(_c64(N), _c64(K), _c(prec))
    if nbytes == 0:
        return None
    k = _wkey(w) + (prec,)
    hit = _pack_cache.get(k)
    if hit is not None and hit[0] == w._version:
        return hit[1]
    img = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    check(lib().mdg_pack_operand(_ptr(w), _c64(w.stride(0)), _c64(N), _c64(K), _c(prec), _ptr(img), ctypes.c_size_t(nbytes),
                                 _stream(w)), "mdg_pack_operand")
    _pack_cache[k] = (w._version, img, w)
    return img


def pack_operand(x: torch.Tensor, precision="bf16x3") -> Optional[torch.Tensor]:
    """Operand image of a 2-D fp32 activation for ``linear_packed`` (what mdg_linear's own pre-pass writes); None where the
    arithmetic mode takes the tensor as it is."""
    x = _f32_cuda(x, "x", 2)
    M, K = x.shape
    if K % 4 or x.stride(1) != 1 or x.stride(0) % 4:
        x = _pad_last(x).contiguous()
        K = x.shape[1]
    prec = _prec(precision)
    nbytes = int(lib().mdg_pack_operand_bytes(_c64(M), _c64(K), _c(prec)))
    if nbytes == 0:
        return None
    img = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    check(lib().mdg_pack_operand(_ptr(x), _c64(x.stride(0)), _c64(M), _c64(K), _c(prec), _ptr(img), ctypes.c_size_t(nbytes), _stream(x)),
          "mdg_pack_operand")
    return img


def _rows2d(x: torch.Tensor, name: str):
    """[..., K] fp32 cuda tensor -> (2-D view/copy with unit inner stride and ld % 4 == 0, leading shape)."""
    x = _f32_cuda(x, name)
    lead = tuple(x.shape[:-1])
    return x.reshape(-1, x.shape[-1]), lead


class PackedWeight:
    """A dense block's weight that exists only as its operand image (``shape`` = [rows N, inner K] of the fp32 tensor it stands for):
    accepted by ``linear`` / ``linear_packed`` in place of the tensor.  ``transposed_weight_image`` makes W^T this way in the 16-bit
    modes -- the backward pass never needs W^T itself, only its image."""
    __slots__ = ("shape", "image", "device")

    def __init__(self, shape, image):
        self.shape, self.image, self.device = tuple(int(v) for v in shape), image, image.device


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, scale=None, shift=None,
           act=None, residual: Optional[torch.Tensor] = None, alpha: float = 1.0, beta: float = 1.0,
           precision="bf16x3", out: Optional[torch.Tensor] = None, cache_weight: bool = True,
           weight_image: Optional[torch.Tensor] = None, dropout_p: float = 0.0, dropout_seed: int = 0) -> torch.Tensor:
    """y = alpha * act((x W^T + b) * scale + shift) + beta * residual   (nn.Linear layout W [N,K]).

    ``x`` may be a strided 2-D view (row stride a multiple of 4); ``residual`` may be [N] / [1,N]
    (broadcast over rows) or [M,N].  ``cache_weight``: keep the packed image of ``weight`` (hi/lo bf16 planes, K
    padded) and reuse it while the tensor is unchanged; pass False for one-shot "weights" (e.g. InfoNCE's F F^T).
    ``weight_image``: the (padded) weight's image when the caller keeps one (transposed_weight_image).
    ``dropout_p`` > 0 (training): y = dropout(act(x W^T + b)) + beta * residual, the mask of ``dropout(.., p, seed)`` on the contiguous
    result applied inside the epilogue (mdg_linear_dropout; no scale / shift, alpha = 1)."""
    forward_only(x, weight, bias, residual)
    if x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.shape[1] % 4 == 0 and x.data_ptr() % 16 == 0 \
            and x.is_cuda and x.dtype == torch.float32:
        x2, lead = x, (x.shape[0],)
    else:
        x2, lead = _rows2d(x, "x")
        if x2.shape[1] % 4:
            x2 = _pad_last(x2)
    if isinstance(weight, PackedWeight):
        w, weight_image, w_ld = None, weight.image, 0
        wshape = weight.shape
    else:
        w = padded_weight(_f32_cuda(weight, "weight", 2))
        wshape, w_ld = tuple(w.shape), w.stride(0)
    M, K, N = x2.shape[0], x2.shape[1], wshape[0]
    if wshape[1] != K:
        raise ValueError(f"linear: x has inner dim {x.shape[-1]} but weight is {tuple(weight.shape)}")
    if act not in ACTS:
        raise ValueError(f"unknown activation {act!r}")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x2.device)
    elif out.dim() != 2 or out.shape != (M, N) or out.stride(1) != 1 or out.dtype != torch.float32 or not out.is_cuda:
        raise ValueError(f"out: expected fp32 cuda [{M},{N}] with unit inner stride")
    ldr = 0
    if residual is not None:
        residual = _f32_cuda(residual, "residual") if residual.is_contiguous() else residual
        if residual.numel() == N:
            ldr = 0
        else:
            if residual.dim() != 2:
                residual = residual.reshape(-1, N)
            if residual.shape != (M, N) or residual.stride(1) != 1:
                raise ValueError(f"residual: expected [{M},{N}] or [{N}], got {tuple(residual.shape)}")
            ldr = residual.stride(0)
    for nm, t in (("bias", bias), ("scale", scale), ("shift", shift)):
        if t is not None and (t.numel() != N or not t.is_cuda or t.dtype != torch.float32):
            raise ValueError(f"{nm}: expected fp32 cuda [{N}]")
    prec = _prec(precision)
    wimg = weight_image if weight_image is not None else (packed_weight_image(w, prec) if cache_weight else None)
    if w is None and (wimg is None or prec == PREC_F32):
        raise ValueError("linear: a PackedWeight needs its image and a 16-bit arithmetic mode")
    nbytes = lib().mdg_linear_workspace_bytes(_c64(M), _c64(N), _c64(K), _c(prec), _c(1 if wimg is not None else 0))
    ws = _workspace(nbytes, x2.device)
    if dropout_p > 0.0:
        if scale is not None or shift is not None or alpha != 1.0 or not out.is_contiguous():
            raise ValueError("linear: the dropout epilogue takes no scale / shift / alpha and a contiguous result")
        check(lib().mdg_linear_dropout(_ptr(x2), _c64(x2.stride(0)), _ptr(w), _c64(w_ld), _ptr(wimg), _ptr(out), _c64(out.stride(0)),
                                       _c64(M), _c64(N), _c64(K), _ptr(None if bias is None else bias.detach().contiguous()), _c(ACTS[act]),
                                       _ptr(residual), _c64(ldr), _f(beta), _f(dropout_p), ctypes.c_uint64(dropout_seed & (2 ** 64 - 1)), _c(prec),
                                       _ptr(ws), ctypes.c_size_t(nbytes), _stream(x2)), "mdg_linear_dropout")
        return out.view(*lead, N) if len(lead) != 1 or lead[0] != M else out
    check(lib().mdg_linear(_ptr(x2), _c64(x2.stride(0)), _ptr(w), _c64(w_ld), _ptr(wimg), _ptr(out), _c64(out.stride(0)),
                           _c64(M), _c64(N), _c64(K), _ptr(None if bias is None else bias.detach().contiguous()),
                           _ptr(None if scale is None else scale.contiguous()), _ptr(None if shift is None else shift.contiguous()),
                           _c(ACTS[act]), _ptr(residual), _c64(ldr), _f(alpha), _f(beta), _c(prec), _ptr(ws),
                           ctypes.c_size_t(nbytes), _stream(x2)), "mdg_linear")
    return out.view(*lead, N) if len(lead) != 1 or lead[0] != M else out


def layernorm_packed(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float, precision, want_fp32: bool = True):
    """LayerNorm of a 2-D ``x`` -> (y, image): ``image`` is y as the packed operand of the dense block that consumes it
    (linear_packed), written by the same kernel; None where the arithmetic mode or the width takes no image (then the
    consumer is the ordinary linear).  ``want_fp32=False``: y is None when the image exists (nothing else reads y)."""
    forward_only(x, weight, bias)
    x2 = x if (x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.is_cuda and x.dtype == torch.float32) else _rows2d(x, "x")[0]
    R, d = x2.shape
    prec = _prec(precision)
    if prec == PREC_F32 or d % 64 or R == 0:
        return layernorm(x2, weight, bias, eps), None
    y = torch.empty((R, d), dtype=torch.float32, device=x2.device) if want_fp32 else None      # image only: y itself is never written
    nbytes = int(lib().mdg_pack_operand_bytes(_c64(R), _c64(d), _c(prec)))
    img = torch.empty(nbytes, dtype=torch.uint8, device=x2.device)
    check(lib().mdg_layernorm_packed(_ptr(x2), _c64(x2.stride(0)), _ptr(weight.detach().contiguous()), _ptr(bias.detach().contiguous()), _ptr(y),
                                     _c64(d), _c64(R), _c64(d), _f(eps), _c(prec), _ptr(img), ctypes.c_size_t(nbytes), _stream(x2)),
          "mdg_layernorm_packed")
    return y, img


def linear_packed(x_img: torch.Tensor, M: int, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, act=None,
                  residual: Optional[torch.Tensor] = None, alpha: float = 1.0, beta: float = 1.0, precision="bf16x3",
                  out: Optional[torch.Tensor] = None, weight_image: Optional[torch.Tensor] = None, cache_weight: bool = True) -> torch.Tensor:
    """linear() on an input that already exists as an operand image (layernorm_packed, linear_backward_pack): no pre-pass over x.
    ``weight_image``: the weight's own image if the caller keeps one (transposed_weight_image); ``cache_weight=False``: pack the
    weight inside the call (a one-shot tensor must not enter the per-storage image cache)."""
    if isinstance(weight, PackedWeight):
        w, weight_image, w_ld = None, weight.image, 0
        N, K = weight.shape
    else:
        w = padded_weight(_f32_cuda(weight, "weight", 2))
        N, K = w.shape
        w_ld = w.stride(0)
    if act not in ACTS:
        raise ValueError(f"unknown activation {act!r}")
    prec = _prec(precision)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x_img.device)
    elif out.dim() != 2 or out.shape != (M, N) or out.stride(1) != 1 or out.dtype != torch.float32 or not out.is_cuda:
        raise ValueError(f"out: expected fp32 cuda [{M},{N}] with unit inner stride")
    ldr = 0
    if residual is not None:
        if residual.numel() != N:
            if residual.dim() != 2 or residual.shape != (M, N) or residual.stride(1) != 1:
                raise ValueError(f"residual: expected [{M},{N}] or [{N}]")
            ldr = residual.stride(0)
    wimg = weight_image if weight_image is not None else (packed_weight_image(w, prec) if cache_weight else None)
    nbytes = lib().mdg_linear_packed_x_workspace_bytes(_c64(M), _c64(N), _c64(K), _c(prec), _c(1 if wimg is not None else 0))    # (+ the stream-K slots of the 256-tile kernel)
    ws = _workspace(nbytes, x_img.device)
    check(lib().mdg_linear_packed_x(_ptr(x_img), _c64(M), _c64(K), _ptr(w), _c64(w_ld), _ptr(wimg), _ptr(out), _c64(out.stride(0)), _c64(N),
                                    _ptr(None if bias is None else bias.detach().contiguous()), _c(ACTS[act]), _ptr(residual), _c64(ldr),
                                    _f(alpha), _f(beta), _c(prec), _ptr(ws), ctypes.c_size_t(nbytes), _stream(x_img)), "mdg_linear_packed_x")
    return out


def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Row-wise LayerNorm over the last dim; ``x`` / ``out`` may be strided 2-D views (row stride % 4 == 0)."""
    forward_only(x, weight, bias)
    if x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.is_cuda and x.dtype == torch.float32:
        x2, lead = x, None
    else:
        x2, lead = _rows2d(x, "x")
    R, d = x2.shape
    if out is None:
        out = torch.empty((R, d), dtype=torch.float32, device=x2.device)
    check(lib().mdg_layernorm(_ptr(x2), _c64(x2.stride(0)), _ptr(weight.detach().contiguous()), _ptr(bias.detach().contiguous()),
                              _ptr(out), _c64(out.stride(0)), _c64(R), _c64(d), _f(eps), _stream(x2)), "mdg_layernorm")
    return out.view(*lead, d) if lead is not None and out.is_contiguous() else out


# ------------------------------------------------------------------------------- fusion
def mask_bits(mask: torch.Tensor) -> torch.Tensor:
    """bool [..., S] (True = masked) -> uint32-valued int32 [...] bit field, bit j = mask[..., j]."""
    S = mask.shape[-1]
    if S > 32:
        raise ValueError("at most 32 tokens")
    w = (torch.ones(S, dtype=torch.int64, device=mask.device) << torch.arange(S, device=mask.device))
    return (mask.to(torch.int64) * w).sum(-1).to(torch.int32).contiguous()     # two's complement keeps bit 31


def assemble_tokens(str_emb, kg_emb, cv_emb, tx_emb, *, bottleneck=None, cls=None, pe=None, rows=None,
                    normalize=False, token_index=None) -> torch.Tensor:
    """[n, S, 128] token sequence (see mdg_assemble_tokens).  tx_emb is [16*n_src,128], cell-line major.
    ``token_index`` (int64 [R], values drug*S + s of the live tokens): emit only those rows -> [R,128]."""
    forward_only(str_emb, kg_emb, cv_emb, tx_emb, bottleneck, cls, pe)
    s, k, c, t = (_f32_cuda(a, nm, 2) for a, nm in ((str_emb, "str"), (kg_emb, "kg"), (cv_emb, "cv"), (tx_emb, "tx")))
    n_src = s.shape[0]
    if k.shape != s.shape or c.shape != s.shape or t.shape != (16 * n_src, s.shape[1]):
        raise ValueError("assemble_tokens: modality embeddings disagree in shape")
    n = n_src if rows is None else int(rows.numel())
    nb = 0 if bottleneck is None else int(bottleneck.shape[0])
    pe2 = None if pe is None else _f32_cuda(pe.detach().reshape(-1, pe.shape[-1]), "pe")
    S = (1 if cls is not None else 0) + 3 + nb + 16
    if token_index is None:
        seq = torch.empty((n, S, s.shape[1]), dtype=torch.float32, device=s.device)
        n_tok = 0
    else:
        if token_index.dtype != torch.int64 or not token_index.is_cuda:
            raise ValueError("token_index: int64 cuda tensor")
        n_tok = int(token_index.numel())
        seq = torch.empty((n_tok, s.shape[1]), dtype=torch.float32, device=s.device)
    check(lib().mdg_assemble_tokens(_ptr(s), _ptr(k), _ptr(c), _ptr(t),
                                    _ptr(None if bottleneck is None else bottleneck.detach().contiguous()),
                                    _ptr(None if cls is None else cls.detach().contiguous()), _ptr(pe2),
                                    _ptr(None if rows is None else rows.contiguous()),
                                    _ptr(None if token_index is None else token_index.contiguous()), _c64(n_tok), _ptr(seq), _c64(n),
                                    _c64(n_src), _c(nb), _c(0 if cls is None else 1), _c(0 if pe2 is None else pe2.shape[0]),
                                    _c(1 if normalize else 0), _c64(s.shape[1]), _stream(s)), "mdg_assemble_tokens")
    return seq


def fusion_attention(qkv: torch.Tensor, n: int, S: int, H: int, dh: int, kpm_bits=None, src_bits=None,
                     want_probs: bool = False, row_start: Optional[torch.Tensor] = None,
                     row_bits: Optional[torch.Tensor] = None, p_drop: float = 0.0, seed: int = 0):
    """Self-attention core over q|k|v rows -> (attention output rows, probs [n,H,S,S] | None).
    Dense: qkv [n*S, 3*H*dh].  Compact (``row_start`` [n+1] int64): qkv holds only live token rows."""
    qkv = _f32_cuda(qkv, "qkv", 2)
    d = H * dh
    rows = n * S if row_start is None else qkv.shape[0]
    if qkv.shape != (rows, 3 * d):
        raise ValueError(f"qkv: expected [{rows},{3 * d}], got {tuple(qkv.shape)}")
    if row_start is not None and (row_start.dtype != torch.int64 or row_start.numel() != n + 1 or want_probs):
        raise ValueError("row_start: int64 [n+1]; attention weights need the dense layout")
    out = torch.empty((rows, d), dtype=torch.float32, device=qkv.device)
    probs = torch.empty((n, H, S, S), dtype=torch.float32, device=qkv.device) if want_probs else None
    check(lib().mdg_fusion_attention_dropout(_ptr(qkv), _c64(qkv.stride(0)), _ptr(out), _c64(d), _ptr(kpm_bits), _ptr(src_bits),
                                             _ptr(probs), _ptr(row_start), _ptr(row_bits), _c64(n), _c(S), _c(H), _c(dh), _f(p_drop),
                                             ctypes.c_uint64(seed & (2 ** 64 - 1)), _stream(qkv)), "mdg_fusion_attention")
    return out, probs


def fusion_attention_bwd(qkv: torch.Tensor, dout: torch.Tensor, n: int, S: int, H: int, dh: int, kpm_bits=None, src_bits=None,
                         row_start=None, row_bits=None, p_drop: float = 0.0, seed: int = 0) -> torch.Tensor:
    """Gradient of the q|k|v rows given the gradient of the attention output (weights recomputed, mask replayed)."""
    qkv, dout = _f32_cuda(qkv, "qkv", 2), _f32_cuda(dout, "dout", 2)
    d = H * dh
    if qkv.shape[1] != 3 * d or dout.shape != (qkv.shape[0], d):
        raise ValueError("fusion_attention_bwd: shape mismatch")
    # rows outside every tile (none in practice) would stay unwritten: start from zeros only in that case
    dqkv = torch.empty_like(qkv)
    check(lib().mdg_fusion_attention_bwd(_ptr(qkv), _c64(qkv.stride(0)), _ptr(dout), _c64(d), _ptr(dqkv), _c64(3 * d), _ptr(kpm_bits),
                                         _ptr(src_bits), _ptr(row_start), _ptr(row_bits), _c64(n), _c(S), _c(H), _c(dh), _f(p_drop),
                                         ctypes.c_uint64(seed & (2 ** 64 - 1)), _stream(qkv)), "mdg_fusion_attention_bwd")
    return dqkv


def xattn_pool(q_proj: torch.Tensor, kv_proj: torch.Tensor, n: int, Tk: int, H: int, dh: int, p_drop: float = 0.0,
               seed: int = 0) -> torch.Tensor:
    q = _f32_cuda(q_proj.reshape(-1), "q_proj", 1)
    kv = _f32_cuda(kv_proj, "kv_proj", 2)
    d = H * dh
    if q.numel() != d or kv.shape != (n * Tk, 2 * d):
        raise ValueError(f"xattn_pool: expected q [{d}] and kv [{n * Tk},{2 * d}]")
    out = torch.empty((n, d), dtype=torch.float32, device=kv.device)
    check(lib().mdg_xattn_pool_dropout(_ptr(q), _ptr(kv), _c64(kv.stride(0)), _ptr(out), _c64(d), _c64(n), _c(Tk), _c(H), _c(dh),
                                       _f(p_drop), ctypes.c_uint64(seed & (2 ** 64 - 1)), _stream(kv)), "mdg_xattn_pool")
    return out


def xattn_pool_bwd(q_proj: torch.Tensor, kv_proj: torch.Tensor, dout: torch.Tensor, n: int, Tk: int, H: int, dh: int,
                   p_drop: float = 0.0, seed: int = 0):
    """-> (dq_proj [d], dkv [n*Tk, 2d])."""
    q = _f32_cuda(q_proj.reshape(-1), "q_proj", 1)
    kv, dout = _f32_cuda(kv_proj, "kv_proj", 2), _f32_cuda(dout, "dout", 2)
    d = H * dh
    if q.numel() != d or kv.shape != (n * Tk, 2 * d) or dout.shape != (n, d):
        raise ValueError("xattn_pool_bwd: shape mismatch")
    dkv = torch.empty_like(kv)
    dq_part = torch.empty((n, d), dtype=torch.float32, device=kv.device)
    check(lib().mdg_xattn_pool_bwd(_ptr(q), _ptr(kv), _c64(kv.stride(0)), _ptr(dout), _c64(d), _ptr(dkv), _c64(2 * d), _ptr(dq_part),
                                   _c64(n), _c(Tk), _c(H), _c(dh), _f(p_drop), ctypes.c_uint64(seed & (2 ** 64 - 1)), _stream(kv)),
          "mdg_xattn_pool_bwd")
    return colsum(dq_part), dkv


# ------------------------------------------------------------------------------- graphs
# ------------------------------------------------------------------------------- grouped dense block
GROUP_TILE = 128          # output tile edge of mdg_linear_grouped


def group_tile_table(groups, device) -> torch.Tensor:
    """Tile descriptors of a grouped launch (include/madrigal_hip.h: mdg_linear_grouped) -> int64 [n_tiles, words] on
    ``device``.  ``groups``: dicts with m_base, rows (the group's rows in the stacked x), n_base, n (its rows in the stacked W),
    y_off, ldy (float offset of its output block in y and that block's row stride) and optionally res_off, ldr, alpha, beta."""
    import struct
    words = int(lib().mdg_linear_group_tile_words())
    rows = []
    for g in groups:
        for v in (g["n_base"], g["y_off"], g["ldy"], g.get("res_off", 0) or 0, g.get("ldr", 0)):
            if v % 4:
                raise ValueError("group_tile_table: offsets and row strides must be multiples of 4 floats")
        ab = struct.unpack("<q", struct.pack("<ff", float(g.get("alpha", 1.0)), float(g.get("beta", 1.0))))[0]
        has_res = "res_off" in g and g["res_off"] is not None
        for ty in range((g["rows"] + GROUP_TILE - 1) // GROUP_TILE):
            for tx in range((g["n"] + GROUP_TILE - 1) // GROUP_TILE):
                e = [g["m_base"] + ty * GROUP_TILE, g["m_base"] + g["rows"], g["n_base"] + tx * GROUP_TILE, g["n_base"] + g["n"],
                     g["m_base"], g["n_base"], g["y_off"], g["ldy"], g["res_off"] if has_res else -1, g.get("ldr", 0) if has_res else 0, ab, 0]
                rows.append(e + [0] * (words - len(e)))
    t = torch.tensor(rows, dtype=torch.int64).reshape(-1, words) if rows else torch.zeros((0, words), dtype=torch.int64)
    return t.to(device)


def linear_grouped(x: torch.Tensor, w_all: torch.Tensor, bias_all: Optional[torch.Tensor], tiles: torch.Tensor, y: torch.Tensor,
                   residual: Optional[torch.Tensor] = None, act=None, precision="bf16x3") -> torch.Tensor:
    """G products y_g = alpha_g act(x_g W_g^T + b_g) + beta_g r_g in one launch: ``x`` [rows_total, K] and ``w_all``
    [w_rows_total, K] hold the groups' rows stacked, ``tiles`` = group_tile_table(...), ``y`` / ``residual`` the buffers its
    offsets address.  The packed image of ``w_all`` is kept while the tensor is unchanged (as for linear)."""
    forward_only(x, w_all, bias_all, residual)
    x, w_all = _f32_cuda(x, "x", 2), _f32_cuda(w_all, "w_all", 2)
    if x.stride(1) != 1 or w_all.shape[1] != x.shape[1] or not w_all.is_contiguous() or x.shape[1] % 4 or x.stride(0) % 4:
        raise ValueError("linear_grouped: x [rows,K] (unit inner stride, K and row stride multiples of 4), w_all [w_rows,K] contiguous")
    if not y.is_cuda or y.dtype != torch.float32 or not y.is_contiguous() or (residual is not None and not residual.is_contiguous()):
        raise ValueError("linear_grouped: y / residual must be contiguous fp32 cuda buffers")
    if act not in ACTS:
        raise ValueError(f"unknown activation {act!r}")
    prec = _prec(precision)
    K = x.shape[1]
    wimg = packed_weight_image(w_all, prec)
    nbytes = lib().mdg_linear_grouped_workspace_bytes(_c64(x.shape[0]), _c64(K), _c(prec))
    ws = _workspace(nbytes, x.device)
    check(lib().mdg_linear_grouped(_ptr(x), _c64(x.stride(0)), _c64(x.shape[0]), _c64(K), _ptr(w_all), _c64(w_all.stride(0)), _ptr(wimg),
                                   _c64(w_all.shape[0]), _ptr(None if bias_all is None else bias_all.detach().contiguous()), _ptr(tiles),
                                   _c64(tiles.shape[0]), _ptr(y), _ptr(residual), _c(ACTS[act]), _c(prec), _ptr(ws), ctypes.c_size_t(nbytes),
                                   _stream(x)), "mdg_linear_grouped")
    return y


def csr_aggregate(x: torch.Tensor, rowptr: torch.Tensor, col: Optional[torch.Tensor] = None, *, edge_weight=None,
                  x_self: Optional[torch.Tensor] = None, self_coef_dev: Optional[torch.Tensor] = None,
                  self_coef_add: float = 0.0, mean: bool = False) -> torch.Tensor:
    """out[v] = (self_coef_add + self_coef_dev[0]) * x_self[v] + sum_{e in row v} w[e] * x[col[e]]  (see the C header)."""
    forward_only(x, x_self)
    x = _f32_cuda(x, "x", 2)
    if x.shape[1] % 4:
        x = _pad_last(x)
    if x.shape[0] == 0:                       # no source rows (hence no edges): a valid pointer for the C side all the same
        x = torch.zeros((1, x.shape[1]), dtype=torch.float32, device=x.device)
    F = x.shape[1]
    n_dst = int(rowptr.numel()) - 1
    if rowptr.dtype != torch.int64 or (col is not None and col.dtype != torch.int64):
        raise ValueError("rowptr / col must be int64")
    if x_self is not None:
        x_self = _f32_cuda(x_self, "x_self", 2)
        if x_self.shape[1] % 4:
            x_self = _pad_last(x_self)
        if x_self.shape != (n_dst, F):
            raise ValueError("x_self: shape mismatch")
    out = torch.empty((n_dst, F), dtype=torch.float32, device=x.device)
    check(lib().mdg_csr_aggregate(_ptr(x), _c64(x.stride(0)), _ptr(rowptr.contiguous()), _ptr(None if col is None else col.contiguous()),
                                  _ptr(None if edge_weight is None else edge_weight.contiguous()), _ptr(x_self),
                                  _c64(0 if x_self is None else x_self.stride(0)), _ptr(self_coef_dev), _f(self_coef_add),
                                  _c(1 if mean else 0), _ptr(out), _c64(F), _c64(n_dst), _c64(F), _stream(x)), "mdg_csr_aggregate")
    return out


def hgt_attention(q: torch.Tensor, kv: torch.Tensor, plan: dict, heads: int, apply_gelu: bool = True,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Edge softmax + aggregation for one destination node type.  ``q`` may be a column-slice view
    [n_dst,128] of the k|q|v projection; ``plan`` holds col / item_* / item_ptr (see graph_plans).  ``out``: contiguous
    [n_dst,128] rows to write (a row block of a buffer shared by all destination types)."""
    n_dst = q.shape[0]
    if q.dim() != 2 or q.shape[1] != 128 or q.stride(1) != 1 or not q.is_cuda or q.dtype != torch.float32:
        raise ValueError("q: expected fp32 cuda [n_dst,128] with unit inner stride")
    if out is None:
        out = torch.empty((n_dst, 128), dtype=torch.float32, device=q.device)
    elif tuple(out.shape) != (n_dst, 128) or not out.is_contiguous() or out.dtype != torch.float32 or out.device != q.device:
        raise ValueError("out: expected contiguous fp32 [n_dst,128] on q's device")
    n_items = int(plan["item_dst"].numel())
    nbytes = lib().mdg_hgt_attention_workspace_bytes(_c64(n_items), _c(heads))
    ws = _workspace(nbytes, q.device)
    check(lib().mdg_hgt_attention(_ptr(q), _c64(q.stride(0)), _ptr(kv), _c64(0 if kv is None else kv.stride(0)), _ptr(plan["col"]),
                                  _ptr(plan["item_dst"]), _ptr(plan["item_begin"]), _ptr(plan["item_end"]), _c64(n_items),
                                  _ptr(plan["item_ptr"]), _ptr(out), _c64(128), _c64(n_dst), _c(heads), _c64(128),
                                  _c(1 if apply_gelu else 0), _ptr(ws), ctypes.c_size_t(nbytes), _stream(q)), "mdg_hgt_attention")
    return out


def hgt_attention_rows(buf: torch.Tensor, plan_all: dict, heads: int, out: torch.Tensor, apply_gelu: bool = True) -> torch.Tensor:
    """Edge attention of ALL destination types of a conv in one launch (inference).  ``buf``: the flat projection buffer (queries,
    keys and values); ``plan_all``: the destination types' plans concatenated (models.HGTConv._forward_grouped): q_off [n_dst] float
    offsets of the query rows in ``buf``, col / item_* / item_ptr with destinations numbered across the types; ``out`` [n_dst,128]."""
    n_dst = int(plan_all["q_off"].numel())
    if tuple(out.shape) != (n_dst, 128) or not out.is_contiguous() or out.dtype != torch.float32 or not buf.is_contiguous():
        raise ValueError("hgt_attention_rows: out must be contiguous fp32 [n_dst,128], buf contiguous")
    n_items = int(plan_all["item_dst"].numel())
    nbytes = lib().mdg_hgt_attention_workspace_bytes(_c64(n_items), _c(heads))
    ws = _workspace(nbytes, buf.device)
    kv = buf.view(-1, 128)
    check(lib().mdg_hgt_attention_rows(_ptr(buf), _ptr(plan_all["q_off"]), _ptr(kv), _c64(128), _ptr(plan_all["col"]), _ptr(plan_all["item_dst"]),
                                       _ptr(plan_all["item_begin"]), _ptr(plan_all["item_end"]), _c64(n_items), _ptr(plan_all["item_ptr"]),
                                       _ptr(out), _c64(128), _c64(n_dst), _c(heads), _c(1 if apply_gelu else 0), _ptr(ws),
                                       ctypes.c_size_t(nbytes), _stream(buf)), "mdg_hgt_attention_rows")
    return out


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """F.normalize(x, p=2, dim=-1)."""
    forward_only(x)
    x2, lead = _rows2d(x, "x")
    if x2.shape[1] % 4:
        raise ValueError("l2_normalize: last dim must be a multiple of 4")
    y = torch.empty_like(x2)
    check(lib().mdg_l2_normalize(_ptr(x2), _c64(x2.stride(0)), _ptr(y), _c64(y.stride(0)), _c64(x2.shape[0]), _c64(x2.shape[1]),
                                 _stream(x2)), "mdg_l2_normalize")
    return y.view(*lead, x2.shape[1])


def token_pool(tokens: torch.Tensor, bits: Optional[torch.Tensor], mode: str) -> torch.Tensor:
    """Masked mean / sum / max over the token axis of [n,S,128]."""
    forward_only(tokens)
    t = _f32_cuda(tokens, "tokens", 3)
    n, S, D = t.shape
    out = torch.empty((n, D), dtype=torch.float32, device=t.device)
    check(lib().mdg_token_pool(_ptr(t), _ptr(bits), _ptr(out), _c64(n), _c(S), _c64(D), _c({"mean": 0, "sum": 1, "max": 2}[mode]),
                               _stream(t)), "mdg_token_pool")
    return out


# ------------------------------------------------------------------------------- losses
def info_nce(aug1: torch.Tensor, aug2: torch.Tensor, too_hard_neg: Optional[torch.Tensor], temperature: float,
             precision="bf16x3", want_logits: bool = True):
    """(logits [2B,2B-1], labels [2B,2B-1], loss) of SimCLR_NovelDDI.contrastive_loss (simclr.py:74-108)."""
    forward_only(aug1, aug2)
    a1, a2 = _f32_cuda(aug1, "aug1", 2), _f32_cuda(aug2, "aug2", 2)
    if a1.shape != a2.shape:
        raise ValueError("aug1 / aug2 shapes differ")
    B = a1.shape[0]
    f = l2_normalize(torch.cat([a1, a2], dim=0))
    sim = linear(f, f, None, precision=precision, cache_weight=False)
    hard = None
    if too_hard_neg is not None:
        if too_hard_neg.shape != (B, B):
            raise ValueError("too_hard_neg: expected [B,B]")
        hard = too_hard_neg.to(device=a1.device, dtype=torch.uint8).contiguous()
    logits = torch.empty((2 * B, 2 * B - 1), dtype=torch.float32, device=a1.device) if want_logits else None
    labels = torch.empty_like(logits) if want_logits else None
    row = torch.empty(2 * B, dtype=torch.float32, device=a1.device)
    loss = torch.empty(1, dtype=torch.float32, device=a1.device)
    check(lib().mdg_infonce_finish(_ptr(sim), _ptr(hard), _ptr(logits), _ptr(labels), _ptr(row), _ptr(loss), _c64(B),
                                   _f(temperature), _stream(a1)), "mdg_infonce_finish")
    return logits, labels, loss[0]


def gather_bce(scores: torch.Tensor, labels: torch.Tensor, heads: torch.Tensor, tails: torch.Tensor,
               target: Optional[torch.Tensor] = None, apply_sigmoid: bool = True):
    """pred = sigmoid?(scores)[labels, heads, tails]; loss = BCELoss(pred, target)  (train_ddi_batch.py:285-288)."""
    s = _f32_cuda(scores, "scores", 3)
    n = int(labels.numel())
    for nm, t in (("labels", labels), ("heads", heads), ("tails", tails)):
        if t.dtype != torch.int64 or not t.is_cuda or t.numel() != n:
            raise ValueError(f"{nm}: expected int64 cuda [{n}]")
    if n:
        # torch's advanced indexing (train_ddi_batch.py:286) raises on an index outside the tensor; the kernel would read out of
        # bounds instead, so the ranges are checked here (one small reduction + host read per call; negative indices, which
        # torch would wrap around, never occur in the collator's triples and are refused as well)
        lim = torch.stack([labels.min(), labels.max(), heads.min(), heads.max(), tails.min(), tails.max()]).tolist()
        for (lo, hi), size, nm in zip(((lim[0], lim[1]), (lim[2], lim[3]), (lim[4], lim[5])), s.shape, ("labels", "heads", "tails")):
            if lo < 0 or hi >= size:
                raise IndexError(f"{nm}: index {lo if lo < 0 else hi} is out of bounds for dimension of size {size}")
    pred = torch.empty(n, dtype=torch.float32, device=s.device)
    term = loss = None
    if target is not None:
        target = _f32_cuda(target, "target", 1)
        term = torch.empty(n, dtype=torch.float32, device=s.device)
        loss = torch.zeros(1, dtype=torch.float32, device=s.device)
    check(lib().mdg_gather_bce(_ptr(s), _c64(s.shape[0]), _c64(s.shape[1]), _c64(s.shape[2]), _ptr(labels.contiguous()),
                               _ptr(heads.contiguous()), _ptr(tails.contiguous()), _ptr(target), _ptr(pred), _ptr(term), _ptr(loss),
                               _c64(n), _c(1 if apply_sigmoid else 0), _stream(s)), "mdg_gather_bce")
    return pred, (None if loss is None else loss[0])


# ------------------------------------------------------------------------------- rank normalisation
def _scores3(t: torch.Tensor, name: str) -> torch.Tensor:
    """[L,N,N] fp32 GPU tensor, contiguous or row-pitched (empty_scores); anything else is made contiguous."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 3):
        return _f32_cuda(t, name, 3)
    if t.numel() and (t.stride(2) != 1 or t.stride(1) < t.shape[2] or t.stride(0) != t.shape[1] * t.stride(1)):
        return t.contiguous()
    return t


def rank_normalize(scores: torch.Tensor, out: Optional[torch.Tensor] = None, max_workspace_bytes: int = 8 << 30,
                   fallback_flags: Optional[list] = None) -> torch.Tensor:
    """Normalised ranks per outcome (notebooks/normalize_scores.py:36-74): [L,N,N] fp32 -> [L,N,N] fp32.
    Outcomes are processed in chunks sized to ``max_workspace_bytes`` of sort scratch.  ``scores`` / ``out`` may be row-pitched
    (``empty_scores``); without ``out`` the result has the layout of ``scores``.

    ``scores`` of dtype int32 = the lower-triangle order keys of ``bilinear_allpairs(..., epilogue=EPI_TRIKEYS)``: same ranks,
    and without ``out`` they are written over the keys (the returned fp32 tensor shares the keys' memory).

    ``fallback_flags`` (diagnostics): a list that receives, per chunk that took the MSD fast path, an int32 tensor with one entry
    per outcome -- non-zero where the fast path handed the outcome to the four-pass LSD sort (same ranks either way)."""
    from_keys = isinstance(scores, torch.Tensor) and scores.dtype == torch.int32
    if from_keys:
        if not (scores.is_cuda and scores.dim() == 3 and (scores.numel() == 0 or (scores.stride(2) == 1 and scores.stride(1) >= scores.shape[2]
                                                                                  and scores.stride(0) == scores.shape[1] * scores.stride(1)))):
            raise ValueError("keys: expected the int32 GPU tensor bilinear_allpairs(..., epilogue=EPI_TRIKEYS) returned")
        s = scores.view(torch.float32)
    else:
        s = _scores3(scores, "scores")
    L, N, N2 = s.shape
    if N != N2:
        raise ValueError("scores: expected [L,N,N]")
    if out is None:
        if from_keys:
            out = s
        else:
            out = empty_scores(L, N, N, s.device) if (N and s.stride(1) != N) else torch.empty((L, N, N), dtype=torch.float32, device=s.device)
    else:
        o2 = _scores3(out, "out")
        if o2 is not out or out.shape != s.shape or (out.data_ptr() == s.data_ptr() and not from_keys):
            raise ValueError("out: an fp32 GPU tensor of the shape of scores (contiguous or row-pitched), not aliasing it")
    if L == 0 or N == 0:
        return out
    lb = lib()
    entry = lb.mdg_rank_normalize_keys_ld if from_keys else lb.mdg_rank_normalize_ld
    per = max(lb.mdg_rank_normalize_workspace_bytes(_c64(1), _c64(N)), 1)
    chunk = int(max(1, min(L, 65535, max_workspace_bytes // per)))
    if chunk > 8:
        chunk -= chunk % 8                   # whole launch groups of the MSD path (8 outcomes each): no ragged group at the end of every chunk
    for lo in range(0, L, chunk):
        hi = min(L, lo + chunk)
        nbytes = lb.mdg_rank_normalize_workspace_bytes(_c64(hi - lo), _c64(N))
        ws = _workspace(nbytes, s.device)
        check(entry(_vp(s.data_ptr() + lo * s.stride(0) * 4), _c64(s.stride(1)), _vp(out.data_ptr() + lo * out.stride(0) * 4),
                    _c64(out.stride(1)), _c64(hi - lo), _c64(N), _ptr(ws), ctypes.c_size_t(nbytes), _stream(s)), "mdg_rank_normalize")
        if fallback_flags is not None and lb.mdg_rank_normalize_fast_path(_c64(hi - lo), _c64(N)):
            fallback_flags.append(ws[: 4 * (hi - lo)].view(torch.int32).clone())
    return out


def _hgt_composite_args(ptrs, k_rel, v_rel, meta):
    nt, R = meta["n_types"], meta["n_edge_types"]
    base = ptrs.data_ptr()
    return (_vp(base), _vp(base + 8 * nt), _ptr(k_rel), _ptr(v_rel), _vp(base + 16 * nt), _ptr(meta["rel_r"]), _ptr(meta["rel_src"]), _ptr(meta["rel_row"]),
            _c(meta["n_rel"]), _ptr(meta["type_row"]), _c(nt))


def hgt_composite(ptrs: torch.Tensor, k_rel, v_rel, meta: dict, big_w, big_b) -> None:
    """mdg_hgt_composite_fwd (autograd._HgtComposite): ``ptrs`` = device int64 table [kqv weights | kqv biases | p_rel of every edge type]."""
    check(lib().mdg_hgt_composite_fwd(*_hgt_composite_args(ptrs, k_rel, v_rel, meta), _ptr(big_w), _ptr(big_b), _c(meta["cin"]), _c(meta["H"]),
                                      _c(meta["n_edge_types"]), _c(meta["F"]), _stream(big_w)), "mdg_hgt_composite_fwd")


def hgt_composite_bwd(ptrs: torch.Tensor, k_rel, v_rel, meta: dict, dbig_w, dbig_b, grads) -> None:
    check(lib().mdg_hgt_composite_bwd(*_hgt_composite_args(ptrs, k_rel, v_rel, meta), _ptr(dbig_w), _ptr(dbig_b), _ptr(grads), _c(meta["cin"]), _c(meta["H"]),
                                      _c(meta["n_edge_types"]), _c(meta["F"]), _stream(grads)), "mdg_hgt_composite_bwd")


def gmean(tensors) -> torch.Tensor:
    """Elementwise geometric mean of up to 8 equally shaped fp32 tensors (5-seed rank ensembling).  Row-pitched rank tensors
    (``empty_scores``) of one pitch are averaged in place of their padded storage: the result has the same layout."""
    ts = list(tensors)
    if not 1 <= len(ts) <= 8 or any(t.shape != ts[0].shape for t in ts):
        raise ValueError("gmean: 1..8 tensors of identical shape")
    t0 = ts[0]
    pitched = (t0.dim() == 3 and t0.is_cuda and t0.dtype == torch.float32 and t0.numel() > 0 and t0.stride(2) == 1 and t0.stride(1) > t0.shape[2]
               and all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.stride() == t0.stride() for t in ts)
               and t0.stride(0) == t0.shape[1] * t0.stride(1))
    if pitched:
        L, N, _ = t0.shape
        P = t0.stride(1)
        full = [t.as_strided((L, N, P), (N * P, P, 1)) for t in ts]          # the padded storage itself: contiguous, n % 4 == 0
        out_full = torch.empty((L, N, P), dtype=torch.float32, device=t0.device)
        arr = (ctypes.c_void_p * len(full))(*[t.data_ptr() for t in full])
        check(lib().mdg_gmean(arr, _c(len(full)), _ptr(out_full), _c64(L * N * P), _stream(out_full)), "mdg_gmean")
        return out_full[:, :, :t0.shape[2]]
    ts = [_f32_cuda(t, f"tensors[{i}]") for i, t in enumerate(ts)]
    out = torch.empty_like(ts[0])
    arr = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    check(lib().mdg_gmean(arr, _c(len(ts)), _ptr(out), _c64(ts[0].numel()), _stream(out)), "mdg_gmean")
    return out


def ensemble_ranks(rank_tensors, max_workspace_bytes: int = 8 << 30) -> torch.Tensor:
    """Seed ensembling of the reference (generate_embeddings.ipynb cells 18-20): geometric mean of the seeds'
    normalised-rank tensors, then rank-normalised again per outcome."""
    return rank_normalize(gmean(rank_tensors), max_workspace_bytes=max_workspace_bytes)


# ------------------------------------------------------------------------------- backward building blocks
def _ceil4(n: int) -> int:
    return (n + 3) // 4 * 4


def transpose(x: torch.Tensor, pad_inner: bool = True) -> torch.Tensor:
    """[R, C] (unit inner stride) -> [C, R']; R' = R rounded up to 4 with zero padding when ``pad_inner`` so that the
    result can be an mdg_linear operand as is."""
    if x.dim() != 2 or not x.is_cuda or x.dtype != torch.float32 or x.stride(1) != 1:
        x = _f32_cuda(x, "x", 2)
    R, C = x.shape
    Rp = _ceil4(R) if pad_inner else R
    out = (torch.zeros if Rp != R else torch.empty)((C, Rp), dtype=torch.float32, device=x.device)
    check(lib().mdg_transpose(_ptr(x), _c64(x.stride(0)), _ptr(out), _c64(Rp), _c64(R), _c64(C), _stream(x)), "mdg_transpose")
    return out


_wt_cache = {}


def weight_transposed(w: torch.Tensor) -> torch.Tensor:
    """``transpose(w)`` of a parameter, kept per (storage, in-place version): a step differentiates every layer once per
    side / view, so each weight's transpose is needed two or three times between two optimizer updates (which bump the
    version).  The entry keeps ``w`` alive, so its address cannot be recycled under the cache; a new version of the same
    parameter replaces the old entry (memory: one transposed copy per Linear weight)."""
    if w.is_cuda and torch.cuda.is_current_stream_capturing():
        return transpose(w.detach())            # a captured pass must contain the transpose itself: replays see newer weights
    k = (w.data_ptr(), tuple(w.shape), str(w.device))
    hit = _wt_cache.get(k)
    if hit is not None and hit[0] == w._version:
        return hit[1]
    t = transpose(w.detach())
    _wt_cache[k] = (w._version, t, w)
    return t


_wt_img_cache = {}
_param_img_cache = {}


def parameter_images(w: torch.Tensor, precision):
    """(operand image of W, PackedWeight of W^T) of a parameter [N,K] from ONE pass over it (mdg_linear_backward_pack: the kernel that
    packs a gradient both ways), kept per (storage, in-place version, mode): a training step needs both -- W for the forward block,
    W^T for dx -- once per optimizer update.  None in the modes / shapes that take the tensor as it is (callers fall back to the
    separate packs)."""
    prec = _prec(precision)
    if prec == PREC_F32 or w.dim() != 2 or not w.is_contiguous() or w.shape[1] % 4 or w.data_ptr() % 16 or \
            (w.is_cuda and torch.cuda.is_current_stream_capturing()):
        return None
    k = (w.data_ptr(), tuple(w.shape), str(w.device), prec)
    hit = _param_img_cache.get(k)
    if hit is not None and hit[0] == w._version:
        return hit[1], hit[2]
    N, K = w.shape
    L_ = lib()
    rb = int(L_.mdg_linear_backward_pack_bytes(_c64(N), _c64(K), _c(prec), _c(0)))
    tb = int(L_.mdg_linear_backward_pack_bytes(_c64(N), _c64(K), _c(prec), _c(1)))
    img = torch.empty(rb, dtype=torch.uint8, device=w.device)
    timg = torch.empty(tb, dtype=torch.uint8, device=w.device)
    wd = w.detach()
    check(L_.mdg_linear_backward_pack(_ptr(wd), _c64(K), _c64(N), _c64(K), _c(prec), _ptr(img), _ptr(timg), _ptr(None), _f(0.0), ctypes.c_uint64(0),
                                      _ptr(None), ctypes.c_size_t(0), _stream(wd)), "mdg_linear_backward_pack")
    wt = PackedWeight((K, _ceil4(N)), timg)
    _param_img_cache[k] = (w._version, img, wt, w)
    return img, wt


def transposed_weight_image(w: torch.Tensor, precision):
    """(transpose(w), its operand image) of a parameter, kept per (storage, in-place version, arithmetic mode) like
    ``weight_transposed``: both sides / views of a step multiply their output gradients by the same W.  Image None where the mode
    takes the tensor as it is."""
    prec = _prec(precision)
    if w.is_cuda and torch.cuda.is_current_stream_capturing():
        return transpose(w.detach()), None
    k = (w.data_ptr(), tuple(w.shape), str(w.device), prec)
    hit = _param_img_cache.get(k)
    if hit is not None and hit[0] == w._version:           # made together with W's own image by the forward pass
        return hit[2], hit[2].image
    hit = _wt_img_cache.get(k)
    if hit is not None and hit[0] == w._version:
        return hit[1], hit[2]
    if prec != PREC_F32 and w.is_contiguous():
        # 16-bit modes: W^T is only ever read as an operand image: one transposing pack of W, no fp32 transpose
        N, K = w.shape
        nbytes = int(lib().mdg_pack_operand_bytes(_c64(K), _c64(_ceil4(N)), _c(prec)))
        img = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        wd = w.detach()
        check(lib().mdg_pack_operand_transposed(_ptr(wd), _c64(wd.stride(0)), _c64(N), _c64(K), _c(prec), _ptr(img), ctypes.c_size_t(nbytes), _stream(wd)),
              "mdg_pack_operand_transposed")
        wt = PackedWeight((K, _ceil4(N)), img)
        _wt_img_cache[k] = (w._version, wt, img, w)
        return wt, img
    wt = weight_transposed(w)
    nbytes = int(lib().mdg_pack_operand_bytes(_c64(wt.shape[0]), _c64(wt.shape[1]), _c(prec)))
    img = None
    if nbytes:
        img = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        check(lib().mdg_pack_operand(_ptr(wt), _c64(wt.stride(0)), _c64(wt.shape[0]), _c64(wt.shape[1]), _c(prec), _ptr(img), ctypes.c_size_t(nbytes),
                                     _stream(wt)), "mdg_pack_operand")
    _wt_img_cache[k] = (w._version, wt, img, w)
    return wt, img


def wide_weight_gradient(N: int, K: int) -> bool:
    """dW [N,K] with >= 96 tiles of 128 x 128 fills the chip tile-wise: the 16-bit modes run it as a tile GEMM on transposed images."""
    return ((N + 127) // 128) * ((K + 127) // 128) >= 96


def linear_backward_pack(g: torch.Tensor, precision, want_bias: bool = False, want_row_image: bool = True, dropout_p: float = 0.0,
                         dropout_seed: int = 0):
    """One pass over g = dL/dy [M,N] (contiguous rows, 16-bit operand mode) -> (operand image of g | None, image of g^T, column sums
    of g | None): what the dx GEMM, the dW GEMM and the bias gradient of a wide dense block read (mdg_linear_backward_pack).
    ``dropout_p`` > 0: g is the incoming gradient of a block that ended in dropout(p, seed); the mask is applied while g is read."""
    if g.dim() != 2 or not g.is_cuda or g.dtype != torch.float32 or g.stride(1) != 1:
        raise ValueError("linear_backward_pack: g must be a 2-D fp32 cuda tensor with unit inner stride")
    prec = _prec(precision)
    M, N = g.shape
    L_ = lib()
    rb = int(L_.mdg_linear_backward_pack_bytes(_c64(M), _c64(N), _c(prec), _c(0)))
    tb = int(L_.mdg_linear_backward_pack_bytes(_c64(M), _c64(N), _c(prec), _c(1)))
    if tb == 0:
        raise ValueError("linear_backward_pack: a 16-bit operand mode (bf16 / bf16x3) and a non-empty g")
    row_img = torch.empty(rb, dtype=torch.uint8, device=g.device) if want_row_image else None
    t_img = torch.empty(tb, dtype=torch.uint8, device=g.device)
    db = torch.empty(N, dtype=torch.float32, device=g.device) if want_bias else None
    nbytes = int(L_.mdg_linear_backward_pack_bytes(_c64(M), _c64(N), _c(prec), _c(2))) if want_bias else 0
    ws = _workspace(nbytes, g.device)
    if dropout_p > 0.0 and g.stride(0) != N:
        raise ValueError("linear_backward_pack: the dropout mask is indexed by the contiguous [M,N] position")
    check(L_.mdg_linear_backward_pack(_ptr(g), _c64(g.stride(0)), _c64(M), _c64(N), _c(prec), _ptr(row_img), _ptr(t_img), _ptr(db), _f(dropout_p),
                                      ctypes.c_uint64(dropout_seed & (2 ** 64 - 1)), _ptr(ws), ctypes.c_size_t(nbytes), _stream(g)),
          "mdg_linear_backward_pack")
    return row_img, t_img, db


def linear_tn_packed_g(gt_img: torch.Tensor, x: torch.Tensor, N: int, precision, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW [N,K] = g^T x from the image of g^T (linear_backward_pack) and the layer input x [M,K]."""
    if x.dim() != 2 or not x.is_cuda or x.dtype != torch.float32 or x.stride(1) != 1:
        raise ValueError("linear_tn_packed_g: x must be a 2-D fp32 cuda tensor with unit inner stride")
    M, K = x.shape
    prec = _prec(precision)
    dw = torch.empty((N, K), dtype=torch.float32, device=x.device) if out is None else out
    if tuple(dw.shape) != (N, K) or not dw.is_contiguous() or dw.dtype != torch.float32:
        raise ValueError("linear_tn_packed_g: out must be contiguous fp32 [N,K]")
    nbytes = lib().mdg_linear_tn_packed_g_workspace_bytes(_c64(M), _c64(N), _c64(K), _c(prec))
    ws = _workspace(nbytes, x.device)
    check(lib().mdg_linear_tn_packed_g(_ptr(gt_img), _ptr(x), _c64(x.stride(0)), _ptr(dw), _c64(K), _c64(M), _c64(N), _c64(K), _c(prec), _ptr(ws),
                                       ctypes.c_size_t(nbytes), _stream(x)), "mdg_linear_tn_packed_g")
    return dw


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None, beta: float = 0.0) -> torch.Tensor:
    """Column sums of a 2-D tensor (fixed summation order): out = beta * out + x.sum(0)."""
    if x.dim() != 2 or not x.is_cuda or x.dtype != torch.float32 or x.stride(1) != 1:
        x = _f32_cuda(x, "x", 2)
    R, C = x.shape
    if out is None:
        out, beta = torch.empty(C, dtype=torch.float32, device=x.device), 0.0
    nbytes = lib().mdg_colsum_workspace_bytes(_c64(R), _c64(C))
    ws = _workspace(nbytes, x.device)
    check(lib().mdg_colsum(_ptr(x), _c64(x.stride(0)), _ptr(out), _c64(R), _c64(C), _f(beta), _ptr(ws), ctypes.c_size_t(nbytes),
                           _stream(x)), "mdg_colsum")
    return out


def activation_fwd(pre: torch.Tensor, act) -> torch.Tensor:
    pre = _f32_cuda(pre, "pre")
    y = torch.empty_like(pre)
    check(lib().mdg_activation_fwd(_ptr(pre), _ptr(y), _c64(pre.numel()), _c(ACTS[act]), _stream(pre)), "mdg_activation_fwd")
    return y


def activation_bwd(dy: torch.Tensor, pre: torch.Tensor, act) -> torch.Tensor:
    """dy * act'(pre); for relu ``pre`` may be the activation output itself."""
    dy, pre = _f32_cuda(dy, "dy"), _f32_cuda(pre, "pre")
    if dy.shape != pre.shape:
        raise ValueError("activation_bwd: shape mismatch")
    dx = torch.empty_like(dy)
    check(lib().mdg_activation_bwd(_ptr(dy), _ptr(pre), _ptr(dx), _c64(dy.numel()), _c(ACTS[act]), _stream(dy)), "mdg_activation_bwd")
    return dx


def dropout(x: torch.Tensor, p: float, seed: int) -> torch.Tensor:
    """Inverted dropout with a counter-based mask: the same (seed, p) applied to a gradient is the backward pass."""
    x = _f32_cuda(x, "x")
    y = torch.empty_like(x)
    check(lib().mdg_dropout(_ptr(x), _ptr(y), _c64(x.numel()), _f(p), ctypes.c_uint64(seed & (2 ** 64 - 1)), _stream(x)), "mdg_dropout")
    return y


def activation_dropout_fwd(pre: torch.Tensor, act, p: float, seed: int) -> torch.Tensor:
    """dropout(act(pre), p, seed) in one pass (same mask as ``dropout``)."""
    pre = _f32_cuda(pre, "pre")
    y = torch.empty_like(pre)
    check(lib().mdg_activation_dropout_fwd(_ptr(pre), _ptr(y), _c64(pre.numel()), _c(ACTS[act]), _f(p), ctypes.c_uint64(seed & (2 ** 64 - 1)), _stream(pre)),
          "mdg_activation_dropout_fwd")
    return y


def activation_dropout_bwd(dy: torch.Tensor, pre: torch.Tensor, act, p: float, seed: int) -> torch.Tensor:
    """act'(pre) * dropout-backward(dy) in one pass."""
    dy, pre = _f32_cuda(dy, "dy"), _f32_cuda(pre, "pre")
    if dy.shape != pre.shape:
        raise ValueError("activation_dropout_bwd: shape mismatch")
    dx = torch.empty_like(dy)
    check(lib().mdg_activation_dropout_bwd(_ptr(dy), _ptr(pre), _ptr(dx), _c64(dy.numel()), _c(ACTS[act]), _f(p), ctypes.c_uint64(seed & (2 ** 64 - 1)),
                                           _stream(dy)), "mdg_activation_dropout_bwd")
    return dx


def batchnorm_train_fwd(x: torch.Tensor, gamma, beta, running_mean, running_var, eps: float, momentum: float, act=None):
    """-> (y, stats[5N]); updates the running statistics in place (nn.BatchNorm1d training semantics)."""
    x = _f32_cuda(x, "x", 2)
    R, C = x.shape
    if R < 2:
        raise ValueError("Expected more than 1 value per channel when training")      # torch's message
    y = torch.empty_like(x)
    stats = torch.empty(5 * C, dtype=torch.float32, device=x.device)
    nbytes = lib().mdg_batchnorm_workspace_bytes(_c64(R), _c64(C))
    ws = _workspace(nbytes, x.device)
    check(lib().mdg_batchnorm_train_fwd(_ptr(x), _c64(C), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), _ptr(y),
                                        _c64(C), _ptr(stats), _c64(R), _c64(C), _f(eps), _f(momentum), _c(ACTS[act]), _ptr(ws),
                                        ctypes.c_size_t(nbytes), _stream(x)), "mdg_batchnorm_train_fwd")
    return y, stats


def batchnorm_replay_update(stats: torch.Tensor, running_mean: torch.Tensor, running_var: torch.Tensor, rows: int, eps: float, momentum: float) -> None:
    """A further momentum update of the running statistics from the batch statistics of an earlier training-mode forward."""
    C = running_mean.numel()
    check(lib().mdg_batchnorm_replay_update(_ptr(stats), _ptr(running_mean), _ptr(running_var), _c64(rows), _c64(C), _f(eps), _f(momentum),
                                            _stream(stats)), "mdg_batchnorm_replay_update")


def batchnorm_train_bwd(dy: torch.Tensor, x: torch.Tensor, stats: torch.Tensor):
    """-> (dx, dgamma, dbeta) for the gradient ``dy`` at the BatchNorm output (before any activation)."""
    dy, x = _f32_cuda(dy, "dy", 2), _f32_cuda(x, "x", 2)
    R, C = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty(C, dtype=torch.float32, device=x.device)
    db = torch.empty_like(dg)
    nbytes = lib().mdg_batchnorm_workspace_bytes(_c64(R), _c64(C))
    ws = _workspace(nbytes, x.device)
    check(lib().mdg_batchnorm_train_bwd(_ptr(dy), _ptr(x), _ptr(stats), _ptr(dx), _ptr(dg), _ptr(db), _c64(R), _c64(C), _ptr(ws),
                                        ctypes.c_size_t(nbytes), _stream(x)), "mdg_batchnorm_train_bwd")
    return dx, dg, db


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-5, extra: Optional[torch.Tensor] = None):
    """-> (dx, dweight, dbias); ``x`` is the LayerNorm input (statistics are recomputed).  ``extra``: a gradient of ``x`` that arrives
    through another consumer (the residual connection of a pre-norm block): dx = LayerNorm-backward(dy) + extra in the same pass."""
    d = x.shape[-1]
    x2 = x.reshape(-1, d) if x.stride(-1) == 1 and x.dim() == 2 else _f32_cuda(x, "x").reshape(-1, d)
    dy2 = dy.reshape(-1, d) if dy.stride(-1) == 1 and dy.dim() == 2 else _f32_cuda(dy, "dy").reshape(-1, d)
    R = x2.shape[0]
    dx = torch.empty((R, d), dtype=torch.float32, device=x2.device)
    dg = torch.empty(d, dtype=torch.float32, device=x2.device)
    db = torch.empty_like(dg)
    nbytes = lib().mdg_layernorm_bwd_workspace_bytes(_c64(R), _c64(d))
    ws = _workspace(nbytes, x2.device)
    if extra is not None:
        e2 = extra.reshape(-1, d) if extra.stride(-1) == 1 and extra.dim() == 2 else _f32_cuda(extra, "extra").reshape(-1, d)
        if e2.shape[0] != R:
            raise ValueError("layernorm_bwd: extra must have the shape of x")
        check(lib().mdg_layernorm_bwd_add(_ptr(dy2), _c64(dy2.stride(0)), _ptr(x2), _c64(x2.stride(0)), _ptr(weight.detach().contiguous()), _ptr(e2),
                                          _c64(e2.stride(0)), _ptr(dx), _c64(d), _ptr(dg), _ptr(db), _c64(R), _c64(d), _f(eps), _ptr(ws),
                                          ctypes.c_size_t(nbytes), _stream(x2)), "mdg_layernorm_bwd_add")
        return dx.view(x.shape), dg, db
    check(lib().mdg_layernorm_bwd(_ptr(dy2), _c64(dy2.stride(0)), _ptr(x2), _c64(x2.stride(0)), _ptr(weight.detach().contiguous()), _ptr(dx),
                                  _c64(d), _ptr(dg), _ptr(db), _c64(R), _c64(d), _f(eps), _ptr(ws), ctypes.c_size_t(nbytes), _stream(x2)),
          "mdg_layernorm_bwd")
    return dx.view(x.shape), dg, db


def affine_act(x: torch.Tensor, scale: torch.Tensor, shift: Optional[torch.Tensor] = None, act=None) -> torch.Tensor:
    """act(x * scale + shift) with per-column scale / shift."""
    x = _f32_cuda(x, "x", 2)
    y = torch.empty_like(x)
    check(lib().mdg_affine_act(_ptr(x), _c64(x.stride(0)), _ptr(scale.contiguous()), _ptr(None if shift is None else shift.contiguous()),
                               _ptr(y), _c64(y.stride(0)), _c64(x.shape[0]), _c64(x.shape[1]), _c(ACTS[act]), _stream(x)), "mdg_affine_act")
    return y


def axpby(a: torch.Tensor, b: torch.Tensor, alpha: float = 1.0, beta: float = 1.0) -> torch.Tensor:
    """alpha * a + beta * b; ``b`` may be a trailing-dims broadcast of ``a`` (numel(b) divides numel(a))."""
    a, b = _f32_cuda(a, "a"), _f32_cuda(b, "b")
    if a.numel() % max(b.numel(), 1) or (b.numel() != a.numel() and tuple(a.shape[a.dim() - b.dim():]) != tuple(b.shape)):
        raise ValueError(f"axpby: cannot broadcast {tuple(b.shape)} over {tuple(a.shape)}")
    out = torch.empty_like(a)
    check(lib().mdg_axpby(_ptr(a), _ptr(b), _ptr(out), _c64(a.numel()), _c64(b.numel()), _f(alpha), _f(beta), _stream(a)), "mdg_axpby")
    return out


def assemble_tokens_bwd(dseq: torch.Tensor, str_emb, kg_emb, cv_emb, tx_emb, *, bottleneck=None, cls=None, pe_len: int = 0,
                        normalize: bool = False, token_index=None):
    """Gradients of mdg_assemble_tokens (rows=None) -> dict(str, kg, cv, tx, bottleneck, cls, pe)."""
    s, k, c, t = (_f32_cuda(a, nm, 2) for a, nm in ((str_emb, "str"), (kg_emb, "kg"), (cv_emb, "cv"), (tx_emb, "tx")))
    n, D = s.shape
    nb = 0 if bottleneck is None else int(bottleneck.shape[0])
    has_cls = cls is not None
    S = (1 if has_cls else 0) + 3 + nb + 16
    dseq = _f32_cuda(dseq, "dseq").reshape(-1, D)
    n_tok = 0 if token_index is None else int(token_index.numel())
    if dseq.shape[0] != (n * S if token_index is None else n_tok):
        raise ValueError("assemble_tokens_bwd: dseq rows disagree with the token list")
    dev = s.device
    # non-emitted tokens leave zero gradient; dense scratch for the shared tokens is summed over drugs below
    dstr, dkg, dcv = (torch.zeros((n, D), dtype=torch.float32, device=dev) for _ in range(3))
    dtx = torch.zeros((16 * n, D), dtype=torch.float32, device=dev)
    dlearned = torch.zeros((n, S, D), dtype=torch.float32, device=dev) if (nb or has_cls) else None
    dpe = torch.zeros((n, S, D), dtype=torch.float32, device=dev) if pe_len else None
    check(lib().mdg_assemble_tokens_bwd(_ptr(dseq), _ptr(s), _ptr(k), _ptr(c), _ptr(t),
                                        _ptr(None if bottleneck is None else bottleneck.detach().contiguous()),
                                        _ptr(None if cls is None else cls.detach().contiguous()),
                                        _ptr(None if token_index is None else token_index.contiguous()), _c64(n_tok), _ptr(dstr), _ptr(dkg),
                                        _ptr(dcv), _ptr(dtx), _ptr(dlearned), _ptr(dpe), _c64(n), _c(nb), _c(1 if has_cls else 0),
                                        _c(pe_len), _c(1 if normalize else 0), _c64(D), _stream(s)), "mdg_assemble_tokens_bwd")
    out = {"str": dstr, "kg": dkg, "cv": dcv, "tx": dtx, "bottleneck": None, "cls": None, "pe": None}
    if dlearned is not None:
        tot = colsum(dlearned.view(n, S * D)).view(S, D)
        off = 1 if has_cls else 0
        if has_cls:
            out["cls"] = tot[0]
        if nb:
            out["bottleneck"] = tot[off + 3: off + 3 + nb]
    if dpe is not None:
        out["pe"] = colsum(dpe.view(n, S * D)).view(S, D)[:pe_len]
    return out


def l2_normalize_bwd(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    dy, x = _f32_cuda(dy, "dy", 2), _f32_cuda(x, "x", 2)
    if x.shape[1] % 4:
        raise ValueError("l2_normalize_bwd: feature dim must be a multiple of 4")
    dx = torch.empty_like(x)
    check(lib().mdg_l2_normalize_bwd(_ptr(dy), _c64(dy.stride(0)), _ptr(x), _c64(x.stride(0)), _ptr(dx), _c64(dx.stride(0)),
                                     _c64(x.shape[0]), _c64(x.shape[1]), _stream(x)), "mdg_l2_normalize_bwd")
    return dx


# ------------------------------------------------------------------------------- gathered head (finetune step)
def triple_plan(labels: torch.Tensor, heads: torch.Tensor, tails: torch.Tensor, n_labels: int, n_head: int, n_tail: int) -> dict:
    """Index plumbing for the gathered head: triples sorted by label (and by head drug inside a label), cut into tiles of <= 32 and
    chunks of <= 256 triples of one label; CSR lists of the sorted triples per head drug and per tail drug for the gradient row sums;
    the table of (label, head drug) PAIRS (the batch holds more labelled triples than pairs, and every 128 x 128 product of the head
    depends on the pair only: bilinear_gather_pairs / _bwd).  The reference feeds a new batch of triples to every step
    (train_ddi_batch.py:231-354), so this runs per step: the three sorts are rocprim's (torch.sort), everything between them is the
    mdg_plan_* kernels (csrc/plan.hip) -- ~45 launches and two host reads of a few sizes, where index / scan / repeat_interleave calls
    took ~300 launches and 5 ms of wall time.  tests/helpers.triple_plan_torch is that earlier construction, kept as the checker."""
    T = int(labels.numel())
    dev = labels.device
    for nm, t in (("labels", labels), ("heads", heads), ("tails", tails)):
        if t.dtype != torch.int64 or not t.is_cuda or t.numel() != T or t.dim() != 1:
            raise ValueError(f"{nm}: expected int64 cuda [{T}]")
    lb, st = lib(), _stream(labels)
    labels, heads, tails = labels.contiguous(), heads.contiguous(), tails.contiguous()
    i64 = lambda n: torch.empty(int(n), dtype=torch.int64, device=dev)

    def bounds(vals, n_vals, scale, n_bounds):
        out = i64(n_bounds)
        check(lb.mdg_plan_lower_bounds(_ptr(vals), _c(vals.element_size()), _c64(n_vals), _c64(scale), _c64(n_bounds), _ptr(out), st),
              "mdg_plan_lower_bounds")
        return out

    def by_drug(idx, n):
        """CSR pointer over the drugs + the stable order of the entries by drug (16-bit keys when they fit: half the radix passes)."""
        cast = torch.int16 if n <= 32767 else (torch.int32 if n < 2 ** 31 else torch.int64)
        vals, order = torch.sort(idx.to(cast), stable=True)
        return bounds(vals, idx.numel(), 1, n + 1), order

    def cut_count(ptr, n, size, totals_row):
        first = i64(n + 1)
        check(lb.mdg_plan_cut_count(_ptr(ptr), _c64(n), _c64(size), _ptr(first), _ptr(totals_row), st), "mdg_plan_cut_count")
        return first

    def cut_fill(ptr, first, n, size, total, want_which=True):
        which = i64(total) if want_which else None
        start = i64(total + 1)
        check(lb.mdg_plan_cut_fill(_ptr(ptr), _ptr(first), _c64(n), _c64(size), _c64(total), _ptr(which), _ptr(start), st), "mdg_plan_cut_fill")
        return which, start

    # ONE sort serves the label order and the (label, head drug) pair order: by label, then by head inside a label (any
    # label-sorted order will do for the tiles; a pair's triples must be consecutive for the pair-compressed head).
    # int32 keys when they fit (twice the radix-sort rate).
    big = n_labels * max(n_head, 1) >= 2 ** 31
    key = labels * n_head + heads
    perm = torch.argsort(key if big else key.to(torch.int32), stable=True)
    hs, ts, skey, inv = i64(T), i64(T), i64(T), i64(T)
    sizes = torch.zeros((8, 2), dtype=torch.int64, device=dev)          # rows: tiles | chunks | head pieces | tail pieces | P | status
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lb.mdg_plan_gather(_ptr(perm), _ptr(labels), _ptr(heads), _ptr(tails), _c64(T), _c64(n_labels), _c64(n_head), _c64(n_tail), _ptr(hs),
                             _ptr(ts), _ptr(skey), _ptr(inv), _ptr(status), st), "mdg_plan_gather")
    label_ptr = bounds(skey, T, n_head, n_labels + 1)
    head_ptr, head_rows = by_drug(hs, n_head)
    tail_ptr, tail_rows = by_drug(ts, n_tail)
    tile_first = cut_count(label_ptr, n_labels, 32, sizes[0])
    chunk_first = cut_count(label_ptr, n_labels, 256, sizes[1])
    hp_first = cut_count(head_ptr, n_head, 64, sizes[2])
    tp_first = cut_count(tail_ptr, n_tail, 64, sizes[3])
    pair_of = None
    if T:
        flag = i64(T)
        check(lb.mdg_plan_pair_flags(_ptr(skey), _c64(T), _ptr(flag), st), "mdg_plan_pair_flags")
        pair_of = torch.cumsum(flag, 0)                               # sorted triple -> its pair
        sizes[4, 0].copy_(pair_of[T - 1])
    sizes[5, 0].copy_(status[0])
    host = sizes.cpu()                                                # the first of two host reads
    if int(host[5, 0]) & 1:
        raise ValueError("labels: value outside [0, n_labels)")
    if int(host[5, 0]) & 2:
        raise ValueError("heads / tails: index outside the embedding tables")
    n_tiles, n_chunks = int(host[0, 0]), int(host[1, 0])
    tile_label, tile_start = cut_fill(label_ptr, tile_first, n_labels, 32, n_tiles)
    _, chunk_start = cut_fill(label_ptr, chunk_first, n_labels, 256, n_chunks, want_which=False)

    def pieces(ptr, first, n, total, longest):
        """A drug's list can hold thousands of entries while mdg_csr_aggregate gives a row to one group of lanes: every list cut into
        pieces of <= 64 entries -> (piece_ptr over the entries, row_ptr over the pieces) for a two-level sum (_sum_rows); None when
        no list is long enough to matter."""
        if n == 0 or longest <= 256:
            return None
        return cut_fill(ptr, first, n, 64, total, want_which=False)[1], first
    head_pieces = pieces(head_ptr, hp_first, n_head, int(host[2, 0]), int(host[2, 1]))
    tail_pieces = pieces(tail_ptr, tp_first, n_tail, int(host[3, 0]), int(host[3, 1]))
    # ---- (label, head drug) PAIRS: the plan's triple order IS the pair order; pair p owns the triples pair_ptr[p] .. pair_ptr[p+1]
    pairs = None
    if T:
        P = int(host[4, 0]) + 1
        pair_ptr, pair_drug = i64(P + 1), i64(P)
        check(lb.mdg_plan_pair_table(_ptr(skey), _ptr(pair_of), _c64(T), _c64(n_head), _c64(P), _ptr(pair_ptr), _ptr(pair_drug), st),
              "mdg_plan_pair_table")
        plabel_ptr = i64(n_labels + 1)                                  # first pair of every label (= the pair of its first triple)
        check(lb.mdg_plan_take(_ptr(pair_of), _ptr(label_ptr), _c64(n_labels + 1), _c64(T), _c64(P), _ptr(plabel_ptr), st), "mdg_plan_take")
        psizes = torch.zeros((4, 2), dtype=torch.int64, device=dev)     # rows: pair tiles | pair chunks | drug pieces
        ptile_first = cut_count(plabel_ptr, n_labels, 32, psizes[0])
        pchunk_first = cut_count(plabel_ptr, n_labels, 512, psizes[1])  # (256: 0.82 ms, 512 / 1024: 0.75 ms, 2048: 1.05 ms for the 2.4e6 pairs of the bench step)
        drug_ptr, drug_rows = by_drug(pair_drug, n_head)
        dp_first = cut_count(drug_ptr, n_head, 64, psizes[2])
        of_triple_by_tail = pair_of[tail_rows]
        phost = psizes.cpu()                                            # the second host read
        pn_tiles, pn_chunks = int(phost[0, 0]), int(phost[1, 0])
        ptile_label, ptile_start = cut_fill(plabel_ptr, ptile_first, n_labels, 32, pn_tiles)
        _, pchunk_start = cut_fill(plabel_ptr, pchunk_first, n_labels, 512, pn_chunks, want_which=False)
        pairs = {"P": P, "drug": pair_drug, "ptr": pair_ptr, "tails_by_pair": ts,
                 "of_triple": pair_of, "tile_start": ptile_start, "tile_label": ptile_label, "n_tiles": pn_tiles,
                 "chunk_start": pchunk_start, "n_chunks": pn_chunks, "label_chunk_ptr": pchunk_first, "drug_ptr": drug_ptr, "drug_rows": drug_rows,
                 "of_triple_by_tail": of_triple_by_tail, "drug_pieces": pieces(drug_ptr, dp_first, n_head, int(phost[2, 0]), int(phost[2, 1]))}
    return {"T": T, "L": n_labels, "n_head": n_head, "n_tail": n_tail, "perm": perm, "inv_perm": inv, "heads": hs, "tails": ts, "pairs": pairs,
            "tile_start": tile_start, "tile_label": tile_label, "n_tiles": n_tiles, "chunk_start": chunk_start,
            "n_chunks": n_chunks, "label_chunk_ptr": chunk_first, "head_ptr": head_ptr,
            "head_rows": head_rows, "tail_ptr": tail_ptr, "tail_rows": tail_rows, "head_pieces": head_pieces,
            "tail_pieces": tail_pieces}


def _sum_rows(x, ptr, rows, pieces, edge_weight=None):
    """csr_aggregate(x, ptr, rows) with the long lists cut into the plan's pieces: piece sums first (many short rows), then
    the pieces of a drug in order.  Same fixed order of additions from run to run."""
    if pieces is None:
        return csr_aggregate(x, ptr, rows, edge_weight=edge_weight)
    part = csr_aggregate(x, pieces[0], rows, edge_weight=edge_weight)
    return csr_aggregate(part, pieces[1], None)


def bilinear_gather(z_head: torch.Tensor, z_tail: torch.Tensor, w: torch.Tensor, plan: dict) -> torch.Tensor:
    """score[t] = z_head[h_t]^T w[l_t] z_tail[t_t] for the plan's triples, in the plan's (label-sorted) order."""
    zh, zt, w = _f32_cuda(z_head, "z_head", 2), _f32_cuda(z_tail, "z_tail", 2), _f32_cuda(w, "w", 3)
    if zh.shape != (plan["n_head"], 128) or zt.shape != (plan["n_tail"], 128) or w.shape != (plan["L"], 128, 128):
        raise ValueError("bilinear_gather: operands disagree with the plan (D must be 128)")
    score = torch.empty(plan["T"], dtype=torch.float32, device=zh.device)
    check(lib().mdg_bilinear_gather(_ptr(zh), _ptr(zt), _ptr(w), _ptr(plan["heads"]), _ptr(plan["tails"]), _ptr(plan["tile_start"]),
                                    _ptr(plan["tile_label"]), _c64(plan["n_tiles"]), _ptr(score), _c64(128), _stream(zh)),
          "mdg_bilinear_gather")
    return score


def bilinear_gather_bwd(z_head, z_tail, w, plan: dict, dscore: torch.Tensor, w_t: Optional[torch.Tensor] = None, need_dw: bool = True):
    """-> (dz_head [Nh,128], dz_tail [Nt,128], dw [L,128,128] | None).  ``w_t``: per-label transpose of ``w`` (omit when
    ``w`` is symmetric)."""
    zh, zt, w = _f32_cuda(z_head, "z_head", 2), _f32_cuda(z_tail, "z_tail", 2), _f32_cuda(w, "w", 3)
    ds = _f32_cuda(dscore, "dscore", 1)
    T, L, dev = plan["T"], plan["L"], zh.device
    if ds.numel() != T:
        raise ValueError("dscore: one entry per triple expected")
    wt = w if w_t is None else _f32_cuda(w_t, "w_t", 3)
    gh = torch.empty((T, 128), dtype=torch.float32, device=dev)
    gt = torch.empty((T, 128), dtype=torch.float32, device=dev)
    dw = torch.empty((L, 128, 128), dtype=torch.float32, device=dev) if need_dw else None
    part = torch.empty((max(plan["n_chunks"], 1), 128, 128), dtype=torch.float32, device=dev) if need_dw else None
    check(lib().mdg_bilinear_gather_bwd(_ptr(zh), _ptr(zt), _ptr(w), _ptr(wt), _ptr(plan["heads"]), _ptr(plan["tails"]),
                                        _ptr(plan["tile_start"]), _ptr(plan["tile_label"]), _c64(plan["n_tiles"]), _ptr(plan["chunk_start"]),
                                        _c64(plan["n_chunks"]), _ptr(plan["label_chunk_ptr"]), _c64(L), _ptr(ds), _ptr(gh), _ptr(gt),
                                        _ptr(part), _ptr(dw), _c64(128), _stream(zh)), "mdg_bilinear_gather_bwd")
    dzh = _sum_rows(gh, plan["head_ptr"], plan["head_rows"], plan.get("head_pieces"))
    dzt = _sum_rows(gt, plan["tail_ptr"], plan["tail_rows"], plan.get("tail_pieces"))
    return dzh, dzt, dw


def _matvec_rows(z, w, row_index, pp, out, precision):
    """rows[p] = W[label of p] z[row_index[p]] per (label, drug) pair, in the step's arithmetic mode (exact fp32 / split-bf16 matrix cores)."""
    L_ = lib()
    prec = _prec(precision)
    nbytes = L_.mdg_bilinear_matvec_rows_workspace_bytes(_c64(w.shape[0]), _c(prec))
    ws = _workspace(nbytes, z.device)
    check(L_.mdg_bilinear_matvec_rows_prec(_ptr(z), _ptr(w), _c64(w.shape[0]), _ptr(row_index), _ptr(pp["tile_start"]), _ptr(pp["tile_label"]),
                                           _c64(pp["n_tiles"]), _ptr(out), _c64(128), _c(prec), _ptr(ws), ctypes.c_size_t(nbytes), _stream(z)),
          "mdg_bilinear_matvec_rows")


def bilinear_gather_pairs(z_head: torch.Tensor, z_tail: torch.Tensor, w: torch.Tensor, plan: dict, w_t: Optional[torch.Tensor] = None,
                          precision="f32"):
    """The scores of bilinear_gather through the (label, head drug) pairs: V[p] = W[l_p]^T z_head[i_p] once per pair (the 128 x 128
    product), then score[t] = V[pair(t)] . z_tail[t_t].  -> (score [T] in the plan's order, V [P,128] for the backward pass)."""
    zh, zt, w = _f32_cuda(z_head, "z_head", 2), _f32_cuda(z_tail, "z_tail", 2), _f32_cuda(w, "w", 3)
    if zh.shape != (plan["n_head"], 128) or zt.shape != (plan["n_tail"], 128) or w.shape != (plan["L"], 128, 128):
        raise ValueError("bilinear_gather_pairs: operands disagree with the plan (D must be 128)")
    pp = plan["pairs"]
    score = torch.empty(plan["T"], dtype=torch.float32, device=zh.device)
    if pp is None:
        return score, torch.empty((0, 128), dtype=torch.float32, device=zh.device)
    wt = w if w_t is None else _f32_cuda(w_t, "w_t", 3)
    V = torch.empty((pp["P"], 128), dtype=torch.float32, device=zh.device)
    L_ = lib()
    _matvec_rows(zh, wt, pp["drug"], pp, V, precision)
    check(L_.mdg_gather_rowdot(_ptr(V), _ptr(pp["of_triple"]), _ptr(zt), _ptr(plan["tails"]), _ptr(score), _c64(plan["T"]), _c64(128),
                               _stream(zh)), "mdg_gather_rowdot")
    return score, V


def bilinear_gather_pairs_bwd(z_head, z_tail, w, plan: dict, dscore: torch.Tensor, V: torch.Tensor, need_dw: bool = True, precision="f32"):
    """Backward of bilinear_gather_pairs -> (dz_head, dz_tail, dw | None), one 128 x 128 product per PAIR:
        dz_tail[j]  = sum_{t: tail = j} ds_t V[pair(t)]                      (V = W^T z_head: saved by the forward pass)
        u[p]        = sum_{t in pair p} ds_t z_tail[t_t]
        dz_head[i]  = sum_{p: drug = i} W[l_p] u[p]
        dW[l]       = sum_{p: label = l} z_head[i_p] u[p]^T
    Sums run in fixed (sorted) order, no atomics."""
    zh, zt, w = _f32_cuda(z_head, "z_head", 2), _f32_cuda(z_tail, "z_tail", 2), _f32_cuda(w, "w", 3)
    ds = _f32_cuda(dscore, "dscore", 1)
    T, L, dev = plan["T"], plan["L"], zh.device
    if ds.numel() != T:
        raise ValueError("dscore: one entry per triple expected")
    pp = plan["pairs"]
    if pp is None:
        z = torch.zeros
        return z((plan["n_head"], 128), device=dev), z((plan["n_tail"], 128), device=dev), (z((L, 128, 128), device=dev) if need_dw else None)
    dzt = _sum_rows(V, plan["tail_ptr"], pp["of_triple_by_tail"], plan.get("tail_pieces"), edge_weight=ds.index_select(0, plan["tail_rows"]))
    u = csr_aggregate(zt, pp["ptr"], pp["tails_by_pair"], edge_weight=ds)
    R = torch.empty((pp["P"], 128), dtype=torch.float32, device=dev)
    L_ = lib()
    _matvec_rows(u, w, None, pp, R, precision)
    dzh = _sum_rows(R, pp["drug_ptr"], pp["drug_rows"], pp.get("drug_pieces"))
    dw = None
    if need_dw:
        dw = torch.empty((L, 128, 128), dtype=torch.float32, device=dev)
        part = torch.empty((max(pp["n_chunks"], 1), 128, 128), dtype=torch.float32, device=dev)
        check(L_.mdg_bilinear_gather_bwd_prec(_ptr(zh), _ptr(u), _ptr(w), _ptr(w), _ptr(pp["drug"]), _ptr(None), _ptr(None), _ptr(None), _c64(0),
                                              _ptr(pp["chunk_start"]), _c64(pp["n_chunks"]), _ptr(pp["label_chunk_ptr"]), _c64(L), _ptr(None),
                                              _ptr(None), _ptr(None), _ptr(part), _ptr(dw), _c64(128), _c(_prec(precision)), _stream(zh)),
              "mdg_bilinear_gather_bwd")
    return dzh[:, :128], dzt[:, :128], dw


def bce_logits(score: torch.Tensor, target: torch.Tensor, want_term: bool = True, grad_scale: Optional[float] = None):
    """-> (term | None, dscore | None): nn.BCELoss(sigmoid(score), target) per element and its logit gradient * grad_scale."""
    s, y = _f32_cuda(score, "score", 1), _f32_cuda(target, "target", 1)
    if s.shape != y.shape:
        raise ValueError("bce_logits: shape mismatch")
    term = torch.empty_like(s) if want_term else None
    ds = torch.empty_like(s) if grad_scale is not None else None
    check(lib().mdg_bce_logits(_ptr(s), _ptr(y), _ptr(term), _ptr(ds), _c64(s.numel()), _f(grad_scale or 0.0), _stream(s)), "mdg_bce_logits")
    return term, ds


def symmetrize_bwd(dw_sym: torch.Tensor) -> torch.Tensor:
    dws = _f32_cuda(dw_sym, "dw_sym", 3)
    out = torch.empty_like(dws)
    check(lib().mdg_symmetrize_bwd(_ptr(dws), _ptr(out), _c64(dws.shape[0]), _c64(dws.shape[1]), _stream(dws)), "mdg_symmetrize_bwd")
    return out


def mul_device_scalar(x: torch.Tensor, scalar: torch.Tensor) -> torch.Tensor:
    x = _f32_cuda(x, "x")
    s = _f32_cuda(scalar.reshape(1), "scalar", 1)
    out = torch.empty_like(x)
    check(lib().mdg_mul_device_scalar(_ptr(x), _ptr(s), _ptr(out), _c64(x.numel()), _stream(x)), "mdg_mul_device_scalar")
    return out


# ------------------------------------------------------------------------------- HGT attention, training
def f32_to_bf16(x: torch.Tensor) -> torch.Tensor:
    """bf16 mirror (round to nearest even) of a contiguous fp32 tensor whose element count is a multiple of 8 (mdg_f32_to_bf16)."""
    x = _f32_cuda(x, "x")
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib().mdg_f32_to_bf16(_ptr(x), _ptr(y), _c64(x.numel()), _stream(x)), "mdg_f32_to_bf16")
    return y


def _kv16_ok(kv: torch.Tensor, kv16: Optional[torch.Tensor]) -> None:
    if kv16 is not None and (kv16.dtype != torch.bfloat16 or kv16.shape != kv.shape or not kv16.is_contiguous() or kv16.device != kv.device or
                             kv.stride(0) != 128):
        raise ValueError("kv16: the bf16 mirror of kv (same [rows,128] shape, contiguous, same device)")


def hgt_attention_stats(q: torch.Tensor, kv: torch.Tensor, plan: dict, heads: int, kv16: Optional[torch.Tensor] = None):
    """mdg_hgt_attention without the activation -> (out_pre [n_dst,128], stats [n_dst,heads,2]).  ``kv16``: bf16 mirror of ``kv``
    (f32_to_bf16) to gather the k' | v' rows from -- the reduced-precision mode's half-size rows."""
    n_dst = q.shape[0]
    _kv16_ok(kv, kv16)
    if q.dim() != 2 or q.shape[1] != 128 or q.stride(1) != 1 or not q.is_cuda or q.dtype != torch.float32:
        raise ValueError("q: expected fp32 cuda [n_dst,128] with unit inner stride")
    out = torch.empty((n_dst, 128), dtype=torch.float32, device=q.device)
    stats = torch.empty((n_dst, heads, 2), dtype=torch.float32, device=q.device)
    n_items = int(plan["item_dst"].numel())
    nbytes = lib().mdg_hgt_attention_workspace_bytes(_c64(n_items), _c(heads))
    ws = _workspace(nbytes, q.device)
    check(lib().mdg_hgt_attention_stats(_ptr(q), _c64(q.stride(0)), _ptr(kv), _c64(0 if kv is None else kv.stride(0)), _ptr(plan["col"]),
                                        _ptr(plan["item_dst"]), _ptr(plan["item_begin"]), _ptr(plan["item_end"]), _c64(n_items),
                                        _ptr(plan["item_ptr"]), _ptr(out), _c64(128), _c64(n_dst), _c(heads), _c64(128), _c(0),
                                        _ptr(stats), _ptr(kv16), _ptr(ws), ctypes.c_size_t(nbytes), _stream(q)), "mdg_hgt_attention_stats")
    return out, stats


def hgt_attention_bwd(q: torch.Tensor, kv: torch.Tensor, plan: dict, rev: dict, heads: int, dout: torch.Tensor, out_pre: torch.Tensor,
                      stats: torch.Tensor, dkv: torch.Tensor, dq_out: Optional[torch.Tensor] = None,
                      kv16: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> dq [n_dst,128]; writes this destination type's key / value gradient rows into ``dkv`` (layout of ``kv``).
    ``dq_out``: write dq there instead (a [n_dst,128] view with unit inner stride, e.g. the query slots of ``dkv`` itself).
    ``kv16``: the bf16 mirror the forward pass gathered from (the same rounded rows enter dq and the softmax gradient)."""
    n_dst = q.shape[0]
    _kv16_ok(kv, kv16)
    dout = _f32_cuda(dout, "dout", 2)
    if kv.dim() != 2 or kv.shape[1] != 128 or not kv.is_contiguous() or dkv.shape != kv.shape or not dkv.is_contiguous():
        raise ValueError("hgt_attention_bwd: kv / dkv must be contiguous [rows,128] (value rows follow key rows)")
    if dq_out is None:
        dq = torch.empty((n_dst, 128), dtype=torch.float32, device=q.device)
    else:
        dq = dq_out
        if dq.shape != (n_dst, 128) or dq.stride(1) != 1 or dq.stride(0) % 4 or dq.dtype != torch.float32 or not dq.is_cuda:
            raise ValueError("hgt_attention_bwd: dq_out must be an fp32 cuda [n_dst,128] view with unit inner stride")
    nnz, n_items = int(plan["col"].numel()), int(plan["item_dst"].numel())
    nbytes = lib().mdg_hgt_attention_bwd_workspace_bytes(_c64(nnz), _c64(n_items), _c64(rev["n_items"]), _c(heads))
    ws = _workspace(nbytes, q.device)
    check(lib().mdg_hgt_attention_bwd(_ptr(q), _c64(q.stride(0)), _ptr(kv), _c64(128), _ptr(plan["col"]), _c64(nnz), _ptr(plan["item_dst"]),
                                      _ptr(plan["item_begin"]), _ptr(plan["item_end"]), _c64(n_items), _ptr(plan["item_ptr"]), _c64(n_dst),
                                      _ptr(dout), _c64(dout.stride(0)), _ptr(out_pre), _c64(out_pre.stride(0)), _ptr(stats), _c(heads),
                                      _ptr(rev["t_edge"]), _ptr(rev["t_dst"]), _ptr(rev["item_begin"]), _ptr(rev["item_end"]),
                                      _c64(rev["n_items"]), _ptr(rev["item_ptr"]), _ptr(rev["rows"]), _c64(rev["n_rows"]), _ptr(rev.get("item_row")),
                                      _ptr(dq), _c64(dq.stride(0)), _ptr(dkv), _c64(128), _ptr(kv16), _ptr(ws), ctypes.c_size_t(nbytes),
                                      _stream(q)),
          "mdg_hgt_attention_bwd")
    return dq


def gated_residual(o: torch.Tensor, x: torch.Tensor, skip: torch.Tensor) -> torch.Tensor:
    """sigmoid(skip) * o + (1 - sigmoid(skip)) * x, the gate read on the device."""
    o, x = _f32_cuda(o, "o", 2), _f32_cuda(x, "x", 2)
    if o.shape != x.shape:
        raise ValueError("gated_residual: shape mismatch")
    out = torch.empty_like(o)
    check(lib().mdg_gated_residual(_ptr(o), _ptr(x), _ptr(_f32_cuda(skip.reshape(1), "skip", 1)), _ptr(out), _c64(o.numel()), _stream(o)),
          "mdg_gated_residual")
    return out


def gated_residual_bwd(dout: torch.Tensor, o: torch.Tensor, x: torch.Tensor, skip: torch.Tensor):
    """-> (d_o, d_x, d_skip [1])."""
    dout, o, x = _f32_cuda(dout, "dout", 2), _f32_cuda(o, "o", 2), _f32_cuda(x, "x", 2)
    d_o, d_x = torch.empty_like(o), torch.empty_like(o)
    rowdot = torch.empty(o.shape[0], dtype=torch.float32, device=o.device)
    check(lib().mdg_gated_residual_bwd(_ptr(dout), _ptr(o), _ptr(x), _ptr(_f32_cuda(skip.reshape(1), "skip", 1)), _ptr(d_o), _ptr(d_x),
                                       _ptr(rowdot), _c64(o.shape[0]), _c64(o.shape[1]), _stream(o)), "mdg_gated_residual_bwd")
    return d_o, d_x, colsum(rowdot.view(-1, 1))


def grad_weight(g: torch.Tensor, x: torch.Tensor, precision="f32", want_bias: bool = False, out=None):
    """dW [N,K] = g^T x for g [M,N], x [M,K] (row-major, unit inner stride; rows may be strided); with ``want_bias`` also
    db [N] = column sums of g -> (dW, db).

    Small outputs (the usual case: 128..512-wide layers over 10^4..10^6 rows) run a split-reduction kernel, which produces the bias
    gradient on the side: exact fp32 matrix cores for "f32", operands rounded (bf16) / split (bf16x3) while staged for the 16-bit
    modes (mdg_grad_weight_prec).  Outputs with >= 96 tiles of 128x128 (the 2048-wide fusion transformer)
    already fill the chip tile-wise: in the bf16 modes they go through the forward GEMM kernel on transposed operands,
    ~5x the fp32 matrix-core rate."""
    if out is not None:                         # (dW, db | None): contiguous fp32 tensors to write into (row blocks of a stacked gradient)
        ow, ob = out
        if tuple(ow.shape) != (g.shape[1], x.shape[1]) or not ow.is_contiguous() or ow.dtype != torch.float32 or \
                (want_bias and (ob is None or ob.numel() != g.shape[1] or not ob.is_contiguous())):
            raise ValueError("grad_weight: out must be contiguous fp32 (dW [N,K], db [N])")
    if g.dim() == 2 and x.dim() == 2 and g.shape[0] == 0 and x.shape[0] == 0:          # no rows: exact zeros
        if out is not None:
            out[0].zero_()
            if want_bias:
                out[1].zero_()
            return out if want_bias else out[0]
        dw = torch.zeros((g.shape[1], x.shape[1]), dtype=torch.float32, device=g.device)
        return (dw, torch.zeros(g.shape[1], dtype=torch.float32, device=g.device)) if want_bias else dw
    for nm, v in (("g", g), ("x", x)):
        if v.dim() != 2 or not v.is_cuda or v.dtype != torch.float32 or v.stride(1) != 1:
            raise ValueError(f"grad_weight: {nm} must be a 2-D fp32 cuda tensor with unit inner stride")
    if g.shape[0] != x.shape[0]:
        raise ValueError("grad_weight: g and x disagree in the number of rows")
    M, N, K = g.shape[0], g.shape[1], x.shape[1]
    if _prec(precision) != PREC_F32 and wide_weight_gradient(N, K):
        prec = _prec(precision)
        dw = torch.empty((N, K), dtype=torch.float32, device=g.device) if out is None else out[0]
        nbytes = lib().mdg_linear_tn_workspace_bytes(_c64(M), _c64(N), _c64(K), _c(prec))
        ws = _workspace(nbytes, g.device)
        check(lib().mdg_linear_tn(_ptr(g), _c64(g.stride(0)), _ptr(x), _c64(x.stride(0)), _ptr(dw), _c64(K), _c64(M), _c64(N), _c64(K),
                                  _c(prec), _ptr(ws), ctypes.c_size_t(nbytes), _stream(g)), "mdg_linear_tn")
        if not want_bias:
            return dw
        if out is None:
            return dw, colsum(g)
        out[1].copy_(colsum(g))
        return dw, out[1]
    dw = torch.empty((N, K), dtype=torch.float32, device=g.device) if out is None else out[0]
    db = (torch.empty(N, dtype=torch.float32, device=g.device) if out is None else out[1]) if want_bias else None
    nbytes = lib().mdg_grad_weight_workspace_bytes(_c64(M), _c64(N), _c64(K))
    ws = _workspace(nbytes, g.device)
    check(lib().mdg_grad_weight_prec(_ptr(g), _c64(g.stride(0)), _ptr(x), _c64(x.stride(0)), _ptr(dw), _ptr(db), _c64(M), _c64(N), _c64(K),
                                     _c(_prec(precision)), _ptr(ws), ctypes.c_size_t(nbytes), _stream(g)), "mdg_grad_weight")
    return (dw, db) if want_bias else dw


def info_nce_bwd(sim: torch.Tensor, too_hard_neg: Optional[torch.Tensor], dloss: torch.Tensor, temperature: float) -> torch.Tensor:
    """d loss / d sim for the InfoNCE finish (sim = F F^T [2B,2B]); ``dloss`` is a device scalar."""
    sim = _f32_cuda(sim, "sim", 2)
    B = sim.shape[0] // 2
    hard = None if too_hard_neg is None else too_hard_neg.to(device=sim.device, dtype=torch.uint8).contiguous()
    dsim = torch.empty_like(sim)
    check(lib().mdg_infonce_bwd(_ptr(sim), _ptr(hard), _ptr(_f32_cuda(dloss.reshape(1), "dloss", 1)), _ptr(dsim), _c64(B), _f(temperature),
                                _stream(sim)), "mdg_infonce_bwd")
    return dsim


# ------------------------------------------------------------------------------- SyncBatchNorm (phased BatchNorm)
def _col_reduce(x, y, center, rstd, mode: int) -> torch.Tensor:
    R, C = x.shape
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    nbytes = lib().mdg_batchnorm_workspace_bytes(_c64(max(R, 1)), _c64(C))
    ws = _workspace(nbytes, x.device)
    check(lib().mdg_col_reduce(_ptr(x), _c64(x.stride(0)), _ptr(y), _c64(0 if y is None else y.stride(0)), _ptr(center), _ptr(rstd), _ptr(out),
                               _c64(R), _c64(C), _c(mode), _ptr(ws), ctypes.c_size_t(nbytes), _stream(x)), "mdg_col_reduce")
    return out


def sync_batchnorm_train_fwd(x: torch.Tensor, gamma, beta, running_mean, running_var, eps: float, momentum: float, act, reduce_):
    """BatchNorm1d training forward with statistics over all ranks: ``reduce_(t)`` sums a small device tensor over ranks
    in place.  The total row count stays on the device (no host read inside the step).  -> (y, stats[5C], count_dev[1])."""
    x = _f32_cuda(x, "x", 2)
    R, C = x.shape
    cnt = torch.tensor([float(R)], dtype=torch.float64).to(x.device, non_blocking=True)
    s = _col_reduce(x, None, None, None, 0)
    reduce_(s)
    reduce_(cnt)
    stats = torch.empty(5 * C, dtype=torch.float32, device=x.device)
    fin = lib().mdg_batchnorm_finalize
    check(fin(_ptr(s), _ptr(None), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), _ptr(stats), ctypes.c_double(0.0), _ptr(cnt),
              _c64(C), _f(eps), _f(momentum), _c(0), _stream(x)), "mdg_batchnorm_finalize")
    q = _col_reduce(x, None, stats, None, 1)
    reduce_(q)
    check(fin(_ptr(s), _ptr(q), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), _ptr(stats), ctypes.c_double(0.0), _ptr(cnt),
              _c64(C), _f(eps), _f(momentum), _c(1), _stream(x)), "mdg_batchnorm_finalize")
    y = affine_act(x, stats[2 * C:3 * C], stats[3 * C:4 * C], act)
    return y, stats, cnt


def sync_batchnorm_train_bwd(dy: torch.Tensor, x: torch.Tensor, stats: torch.Tensor, count: torch.Tensor, reduce_):
    """-> (dx, dgamma_local, dbeta_local): dx uses the sums over ALL ranks, the parameter gradients stay local partial sums
    (they are summed over ranks with every other parameter gradient)."""
    dy, x = _f32_cuda(dy, "dy", 2), _f32_cuda(x, "x", 2)
    R, C = x.shape
    db = _col_reduce(dy, None, None, None, 0)
    dg = _col_reduce(dy, x, stats[0:C], stats[C:2 * C], 2)
    both = torch.cat([db, dg])
    reduce_(both)
    dx = torch.empty_like(x)
    check(lib().mdg_batchnorm_bwd_apply(_ptr(dy), _ptr(x), _ptr(stats), _ptr(both[:C].contiguous()), _ptr(both[C:].contiguous()), _ptr(dx),
                                        _c64(R), _c64(C), ctypes.c_double(0.0), _ptr(count), _stream(x)), "mdg_batchnorm_bwd_apply")
    return dx, dg, db
