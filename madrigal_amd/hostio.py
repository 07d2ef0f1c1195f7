"""Host -> device transfers that do not stall the launch queue.

A pageable ``tensor.to(device)`` waits until everything queued on the stream has run: issued in the middle of a training loop
it ends the host's run-ahead and the GPU idles while the next step is being queued.  The reference's loops hand the model
per-iteration host tensors (drug indices, modality masks drawn per iteration: pretrain.py:59-71); the step functions here
accept those tensors on the CPU and move them through a small ring of pinned staging buffers with asynchronous copies."""
from __future__ import annotations

import torch

_rings = {}


def upload(host: torch.Tensor, device, tag: str, depth: int = 4) -> torch.Tensor:
    """CPU tensor -> device tensor on the current stream, asynchronously.  ``tag`` names the ring (one per logical input); a
    staging buffer is reused only after the copy issued from it has completed (its event), which ``depth`` iterations later
    it long has."""
    if host.device.type != "cpu":
        return host.to(device)
    device = torch.device(device)
    key = (tag, str(device), host.dtype)
    ring = _rings.setdefault(key, {"slot": 0, "bufs": [None] * depth})
    i = ring["slot"]
    ring["slot"] = (i + 1) % depth
    ent = ring["bufs"][i]
    n = host.numel()
    if ent is None or ent[0].numel() < n:
        ent = ring["bufs"][i] = [torch.empty(max(n, 1), dtype=host.dtype).pin_memory(), None]
    if ent[1] is not None:
        ent[1].synchronize()
    stage = ent[0][:n].view(host.shape)
    stage.numpy()[...] = host.numpy()            # plain memcpy (torch's CPU copy may fan out over an oversubscribed thread pool)
    out = stage.to(device, non_blocking=True)
    ent[1] = torch.cuda.Event()
    ent[1].record(torch.cuda.current_stream(device))
    return out
