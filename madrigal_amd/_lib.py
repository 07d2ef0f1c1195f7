"""ctypes binding of libmadrigal_hip.so (the C ABI declared in include/madrigal_hip.h).

The product path has NO fallback: if the shared object is missing or a call fails, an
exception is raised.  ``declared_symbols()`` parses the public header so tests can check that
every declared entry point is exported.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libmadrigal_hip.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "madrigal_hip.h")

_lib = None


class MadrigalHipError(RuntimeError):
    pass


def declared_symbols() -> List[str]:
    """Names of all functions declared in include/madrigal_hip.h."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdg_[a-z0-9_]+)\s*\(", text)))


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MadrigalHipError(
                f"{LIB_PATH} not found: build it with `python -m madrigal_amd.build` "
                "(there is no CPU or PyTorch fallback for the HIP path)")
        # torch first: its wheel bundles its own libamdhip64; loading ours afterwards makes the dynamic linker bind
        # libmadrigal_hip.so to that same, already initialised HIP runtime (two runtimes in one process do not
        # share devices, streams or allocations).
        import torch  # noqa: F401
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.mdg_last_error.restype = ctypes.c_char_p
        _lib.mdg_build_arch.restype = ctypes.c_char_p
        for name in declared_symbols():
            fn = getattr(_lib, name)            # AttributeError here = header/library mismatch
            if name.endswith("_workspace_bytes"):
                fn.restype = ctypes.c_size_t
        _lib.mdg_abi_version.restype = ctypes.c_int
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().mdg_last_error().decode(errors="replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise MadrigalHipError(f"{what} failed (code {rc}): {msg}")
