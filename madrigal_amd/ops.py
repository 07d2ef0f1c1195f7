"""Python wrappers of the C ABI: validate tensors, pass raw pointers + the current HIP stream.

torch is used for device memory and streams only; all arithmetic is in libmadrigal_hip.so.
Every wrapper raises ``ValueError`` for bad shapes/dtypes/devices (the reference raises
assertion errors in the same situations) and ``MadrigalHipError`` if the library is missing.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from ._lib import check, lib

PREC_F32, PREC_BF16X3, PREC_BF16 = 0, 1, 2
PRECISIONS = {"f32": PREC_F32, "bf16x3": PREC_BF16X3, "bf16": PREC_BF16}
EPI_STORE, EPI_STORE_SIGMOID, EPI_ROWSTATS = 0, 1, 2

_c64 = ctypes.c_int64
_vp = ctypes.c_void_p


def _prec(p) -> int:
    if isinstance(p, str):
        if p not in PRECISIONS:
            raise ValueError(f"unknown precision {p!r}; expected one of {sorted(PRECISIONS)}")
        return PRECISIONS[p]
    return int(p)


def _stream(t: torch.Tensor) -> _vp:
    return _vp(torch.cuda.current_stream(t.device).cuda_stream)


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
    key = (device.type, device.index)
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
    (or [L,Nh,2] row statistics with ``EPI_ROWSTATS``).  madrigal/models/models.py:537-547."""
    zh, zt, w = _f32_cuda(z_head, "z_head", 2), _f32_cuda(z_tail, "z_tail", 2), _f32_cuda(w_sym, "w_sym", 3)
    D = zh.shape[1]
    if zt.shape[1] != D or w.shape[1] != D or w.shape[2] != D:
        raise ValueError(f"feature dims disagree: z_head {tuple(zh.shape)}, z_tail {tuple(zt.shape)}, w {tuple(w.shape)}")
    if zh.device != zt.device or zh.device != w.device:
        raise ValueError("z_head, z_tail and w_sym must be on the same device")
    L, Nh, Nt = w.shape[0], zh.shape[0], zt.shape[0]
    shape = (L, Nh, 2) if epilogue == EPI_ROWSTATS else (L, Nh, Nt)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=zh.device)
    else:
        out = _f32_cuda(out, "out")
        if tuple(out.shape) != shape or not out.is_contiguous():
            raise ValueError(f"out: expected contiguous {shape}, got {tuple(out.shape)}")
    prec = _prec(precision)
    L_ = lib()
    # the grid's y extent caps one call at 65535 outcomes; chunk above that
    for lo in range(0, max(L, 1), 65535):
        hi = min(L, lo + 65535)
        if hi <= lo:
            break
        nbytes = L_.mdg_bilinear_allpairs_workspace_bytes(_c64(Nh), _c64(Nt), _c64(hi - lo), _c64(D), prec)
        ws = _workspace(nbytes, zh.device)
        check(L_.mdg_bilinear_allpairs(_ptr(zh), _ptr(zt), _vp(w.data_ptr() + lo * D * D * 4),
                                       _vp(out.data_ptr() + lo * out.stride(0) * 4), _c64(Nh), _c64(Nt), _c64(hi - lo),
                                       _c64(D), prec, int(epilogue), _ptr(ws), ctypes.c_size_t(nbytes), _stream(zh)),
              "mdg_bilinear_allpairs")
    return out
