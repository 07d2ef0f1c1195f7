"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every declared symbol."""
import ctypes
import os

import pytest


def test_library_exports_every_declared_symbol():
    from madrigal_amd import _lib
    syms = _lib.declared_symbols()
    assert "mdg_bilinear_allpairs" in syms and "mdg_last_error" in syms
    L = _lib.lib()
    for s in syms:
        assert hasattr(L, s), s
    assert L.mdg_build_arch() == b"gfx950"
    assert L.mdg_abi_version() >= 1


def test_workspace_queries_need_no_gpu():
    from madrigal_amd import _lib
    L = _lib.lib()
    c = ctypes.c_int64
    assert L.mdg_bilinear_allpairs_workspace_bytes(c(4096), c(4096), c(896), c(128), 0) == 0
    b3 = L.mdg_bilinear_allpairs_workspace_bytes(c(4096), c(4096), c(896), c(128), 1)
    b1 = L.mdg_bilinear_allpairs_workspace_bytes(c(4096), c(4096), c(896), c(128), 2)
    assert b3 == 2 * b1 and b1 >= 4096 * 128 * 2 + 896 * 128 * 128 * 2


def test_ops_refuse_cpu_tensors():
    import torch
    from madrigal_amd import ops
    z = torch.zeros(4, 128)
    w = torch.zeros(1, 128, 128)
    with pytest.raises(ValueError, match="GPU"):
        ops.bilinear_allpairs(z, z, w)


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from madrigal_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(tmp_path, "nope.so"))
    with pytest.raises(_lib.MadrigalHipError):
        _lib.lib()
