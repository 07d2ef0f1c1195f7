"""Deterministic, platform-independent parameter fill shared by the golden generator
and the tests.  TEST INFRASTRUCTURE ONLY (see oracle/madrigal_oracle.py header).

Golden fixtures store inputs and expected outputs only; weights are regenerated from
(seed, parameter name, shape) with numpy's PCG64, which is bit-identical across
machines.  The same fill is loaded into the reference module (generation, this
container) and into the madrigal_amd module / the oracle (tests)."""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(key.encode())])


def det_tensor(seed: int, key: str, shape: Tuple[int, ...], dtype=torch.float32) -> torch.Tensor:
    """Value for one state_dict entry, chosen by the entry's name and rank."""
    rng = _rng(seed, key)
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_var":
        a = rng.uniform(0.5, 1.5, size=shape)
    elif leaf == "running_mean":
        a = rng.standard_normal(shape) * 0.2
    elif leaf == "eps":
        a = rng.uniform(0.0, 0.2, size=shape)
    elif leaf in ("x_attn_query", "tx_bottleneck_tokens", "cls", "pe") or leaf.startswith("skip") \
            or ".skip." in key or ".p_rel." in key:
        a = rng.standard_normal(shape) * 0.5
    elif len(shape) <= 1 and leaf == "weight":          # norm scales
        a = 1.0 + 0.1 * rng.standard_normal(shape)
    elif len(shape) <= 1:                               # biases
        a = 0.1 * rng.standard_normal(shape)
    else:                                               # matrices: fan-in scaling on the last dim
        a = rng.standard_normal(shape) / np.sqrt(shape[-1])
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(dtype)


def det_state_dict(seed: int, shapes: Dict[str, Tuple[int, ...]], skip: Iterable[str] = ()) -> Dict[str, torch.Tensor]:
    skip = set(skip)
    return {k: det_tensor(seed, k, s) for k, s in shapes.items() if k not in skip}


def fill_module(module: torch.nn.Module, seed: int, skip: Iterable[str] = ()) -> Dict[str, torch.Tensor]:
    """Overwrite every state_dict entry of ``module`` (except ``skip``) in place and
    return the tensors that were loaded."""
    sd = module.state_dict()
    new = det_state_dict(seed, {k: tuple(v.shape) for k, v in sd.items()}, skip)
    module.load_state_dict({**sd, **new}, strict=True)
    return new


def det_input(seed: int, key: str, shape, scale: float = 1.0) -> torch.Tensor:
    a = _rng(seed, "input:" + key).standard_normal(tuple(shape)) * scale
    return torch.from_numpy(a.astype(np.float32))
