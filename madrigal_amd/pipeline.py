"""All-pairs inference: encode every drug once -> (all-gather) -> score every (head, tail, outcome).

MI355X-native counterpart of ``notebooks/generate_embeddings.ipynb`` cells 9-10 and
``get_*_scores_for_all_pairs_among_drugs`` (madrigal/evaluate/predict.py:381-463, 502-579): the
reference encodes once (``model.encoder(...)``), then loops over chunks of 10-30 outcomes calling
``model.decoder(z, z, (start, end)).cpu().numpy()`` into an ``np.memmap``.  Here the head writes the
[L,N,N] tensor straight into HBM (288 GB holds the reference's whole 80 GB tensor) or, when the caller
wants it on the host / in a memmap, streams outcome chunks through two pinned buffers on a copy stream
so that the D2H copy of chunk k overlaps the kernel of chunk k+1.

Multi-GPU (one process per GPU): encode+fuse is independent per drug, so rank r encodes a contiguous
block of drugs (the KG encoder, which is per-graph rather than per-drug, runs destination-partitioned:
every rank computes its block of every node type and one all-gather per conv reassembles them); one
all-gather of the [N/G,128] blocks over RCCL gives every rank z[N,128]; the head is sharded by
outcome (``label_range``: rank-local rank normalisation afterwards; what bench.py and the rank pipeline use)
or by head row (``head_rows``: BASELINE configs[3]'s "row-sharded" -- rank r owns S[:, rows_r, :]; the rank
normalisation of a row-sharded tensor would need a global sort, so it is for callers that want raw scores).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import ops
from .data import MoleculeBatch
from .parallel import all_gather_rows, shard_range


def slice_molecules(mols: MoleculeBatch, lo: int, hi: int) -> MoleculeBatch:
    """Graphs [lo,hi) of a packed molecule batch as a new packed batch (atoms / bonds re-indexed)."""
    n2g = mols.node2graph
    a0 = int(torch.searchsorted(n2g, torch.tensor(lo, device=n2g.device)))
    a1 = int(torch.searchsorted(n2g, torch.tensor(hi, device=n2g.device)))
    keep = (mols.edge_list[:, 1] >= a0) & (mols.edge_list[:, 1] < a1)
    el = mols.edge_list[keep].clone()
    el[:, :2] -= a0
    return MoleculeBatch(mols.node_feature[a0:a1], el, mols.edge_feature[keep], n2g[a0:a1] - lo, hi - lo,
                         mols.edge_weight[keep])


def slice_batch(batch: dict, lo: int, hi: int) -> dict:
    """Drugs [lo,hi) of a reference-format batch dict."""
    tx = {c: {"sigs": v["sigs"][lo:hi], "drugs": v["drugs"][lo:hi], "dosages": v["dosages"][lo:hi],
              "cell_lines": v["cell_lines"][lo:hi]} for c, v in batch["tx"].items()}
    return {"drugs": batch["drugs"][lo:hi], "strs": slice_molecules(batch["strs"], lo, hi), "cv": batch["cv"][lo:hi],
            "tx": tx, "masks": batch["masks"][lo:hi]}


@torch.no_grad()
def generate_embeddings(model, batch: dict, batch_kg: dict, masks: Optional[torch.Tensor] = None, rank: int = 0,
                        world: int = 1, kg_filler: Optional[torch.Tensor] = None, on_encoded=None) -> torch.Tensor:
    """z[N,128] for every drug of ``batch`` (``model.encoder`` on all drugs, generate_embeddings.ipynb raw
    line 226).  With ``world > 1`` each rank encodes its block and the blocks are all-gathered.  ``on_encoded``: called
    (no arguments) between the rank's own encode+fuse and the exchange step -- bench.py records a HIP event there."""
    masks = batch["masks"] if masks is None else masks
    n = int(batch["drugs"].shape[0])
    if world == 1:
        z = model.encoder(batch["drugs"], masks, batch["strs"], batch_kg, batch["cv"], batch["tx"], kg_filler=kg_filler)
        if on_encoded is not None:
            on_encoded()
        return z
    lo, hi = shard_range(n, rank, world)
    local = batch.get("_shards", {}).get((lo, hi))
    if local is None:
        local = slice_batch(batch, lo, hi)
        batch.setdefault("_shards", {})[(lo, hi)] = local
    # the KG encoder is per graph, not per drug: its convs run destination-partitioned over the ranks (HGTConv.forward) instead
    # of replicated (MDG_SHARD_KG=0 keeps it replicated)
    import os
    kg_shard = (rank, world, None) if os.environ.get("MDG_SHARD_KG", "1") != "0" else None
    z_local = model.encoder(local["drugs"], masks[lo:hi], local["strs"], batch_kg, local["cv"], local["tx"], kg_filler=kg_filler,
                            kg_shard=kg_shard)
    if on_encoded is not None:
        on_encoded()
    return all_gather_rows(z_local.contiguous(), n, rank, world)


@torch.no_grad()
def score_all_pairs(model, z: torch.Tensor, label_range: Optional[Tuple[int, int]] = None, out=None,
                    host_chunk: int = 16, epilogue: int = ops.EPI_STORE, head_rows: Optional[Tuple[int, int]] = None):
    """Scores of every ordered drug pair for outcomes ``label_range`` (default: all) -> [L',N,N] fp32.

    ``head_rows`` = (r0, r1): the row-sharded head -- only head drugs [r0, r1) against ALL tail drugs -> [L', r1 - r0, N]
    (``decoder(z[r0:r1], z, ...)``, the general sweep; HBM destination only).  ``shard_range(N, rank, world)`` gives a rank's rows;
    the row blocks of the ranks concatenate along dim 1 to the full tensor (no collective touches the scores).

    ``out`` None / a CUDA tensor: one head launch writes the whole tensor into HBM.
    ``out`` a numpy array or ``np.memmap`` (the reference's destination, predict.py:410-429): outcome chunks
    of ``host_chunk`` are produced on the GPU and copied out through two pinned staging buffers."""
    dec = model.decoder
    L_all = dec.parametrizations.weight.original.shape[0] if hasattr(dec, "parametrizations") else dec.weight.shape[0]
    lo, hi = (0, L_all) if label_range is None else label_range
    N = z.shape[0]
    if head_rows is not None:
        r0, r1 = head_rows
        if not (0 <= r0 <= r1 <= N):
            raise ValueError(f"head_rows {head_rows} outside [0, {N}]")
        if out is None:
            out = ops.empty_scores(hi - lo, r1 - r0, N, z.device)
        if not isinstance(out, torch.Tensor):
            raise ValueError("head_rows: HBM destination only")
        if r1 == r0:
            return out
        return dec(z[r0:r1], z, (lo, hi), epilogue=epilogue, out=out)
    if out is None:
        out = ops.empty_scores(hi - lo, N, N, z.device)      # rows on 128-byte lines whatever N is (a [:, :, :N] view when N % 32 != 0)
    if isinstance(out, torch.Tensor):
        return dec(z, z, (lo, hi), epilogue=epilogue, out=out)
    if tuple(out.shape) != (hi - lo, N, N) or out.dtype != np.float32:
        raise ValueError(f"out: expected float32 array of shape {(hi - lo, N, N)}")
    copy_stream = torch.cuda.Stream(device=z.device)
    dev_buf = [ops.empty_scores(host_chunk, N, N, z.device) for _ in range(2)]     # row-pitched on the device, compacted by the copy
    pin_buf = [torch.empty((host_chunk, N, N), dtype=torch.float32, pin_memory=True) for _ in range(2)]
    done = [None, None]
    pending = [None, None]
    k = 0
    for s in range(lo, hi, host_chunk):
        e = min(hi, s + host_chunk)
        b = k & 1
        if done[b] is not None:                        # staging pair b is free once its last copy has landed
            done[b].synchronize()
            ps, pe = pending[b]
            out[ps - lo:pe - lo] = pin_buf[b][: pe - ps].numpy()
        dec(z, z, (s, e), epilogue=epilogue, out=dev_buf[b][: e - s])
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(z.device))
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(ready)
            pin_buf[b][: e - s].copy_(dev_buf[b][: e - s], non_blocking=True)
            done[b] = torch.cuda.Event()
            done[b].record(copy_stream)
        pending[b] = (s, e)
        k += 1
    for b in range(2):
        if done[b] is not None:
            done[b].synchronize()
            ps, pe = pending[b]
            out[ps - lo:pe - lo] = pin_buf[b][: pe - ps].numpy()
    return out


@torch.no_grad()
def rank_all_pairs(model, z: torch.Tensor, label_range: Optional[Tuple[int, int]] = None, out: Optional[torch.Tensor] = None,
                   max_workspace_bytes: int = 8 << 30) -> torch.Tensor:
    """Normalised ranks of every drug pair per outcome -> [L',N,N] fp32, for callers whose product is the ranks (the reference
    materialises the raw scores into a memmap, predict.py:410-429, and ranks them afterwards on the CPU, notebooks/normalize_scores.py:
    36-85).  The head writes only the order keys of the strict lower triangle -- all the rank normalisation reads -- and the sort puts
    the ranks over them: half the head's store stream and no score tensor next to the rank tensor.  Bit-identical to
    ``ops.rank_normalize(score_all_pairs(...))``."""
    dec = model.decoder
    L_all = dec.parametrizations.weight.original.shape[0] if hasattr(dec, "parametrizations") else dec.weight.shape[0]
    lo, hi = (0, L_all) if label_range is None else label_range
    N = z.shape[0]
    if out is None:
        out = ops.empty_scores(hi - lo, N, N, z.device)
    elif not (isinstance(out, torch.Tensor) and out.is_cuda and out.dtype == torch.float32 and tuple(out.shape) == (hi - lo, N, N)):
        raise ValueError(f"out: expected a float32 GPU tensor of shape {(hi - lo, N, N)} (ops.empty_scores)")
    pitch = out.stride(1) if out.numel() else N
    if N * pitch * 4 >= 2 ** 32 or pitch % 4:
        # beyond the symmetric sweep's 32-bit slab offsets (N > 32 767) or on an unpadded tensor: scores first, ranks over them in place of
        # a second tensor is not possible (rank_normalize reads what it overwrites) -- one outcome chunk of scores at a time
        chunk = max(1, min(hi - lo, (8 << 30) // max(N * pitch * 4, 1)))
        tmp = ops.empty_scores(chunk, N, N, z.device)
        for s in range(lo, hi, chunk):
            e = min(hi, s + chunk)
            dec(z, z, (s, e), out=tmp[: e - s])
            ops.rank_normalize(tmp[: e - s], out=out[s - lo: e - lo], max_workspace_bytes=max_workspace_bytes)
        return out
    keys = dec(z, z, (lo, hi), epilogue=ops.EPI_TRIKEYS, out=out.view(torch.int32))
    return ops.rank_normalize(keys, max_workspace_bytes=max_workspace_bytes)
