"""Destination-sorted (CSR) plans for the two GNN aggregations.

Host-side glue (torch ops on the device, no arithmetic of the model): the reference hands the
encoders COO edge lists (torchdrug ``edge_list`` [E,3], PyG ``edge_index_dict``); the HIP
aggregation kernels want edges grouped by destination so that one lane group owns one output row
(no atomics, fixed summation order).  Plans are built once per graph object and cached on it: the
KG is static across steps and the molecule batch is the single full batch of the run
(train_ddi_batch.py:116).
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import torch

import os

# edges per work item: bounds the serial walk of one wave over a heavy-tailed KG destination (one wave walks its item 4 edges per
# step, each step a dependent col -> row gather of ~2 us: at 512 edges the longest items ARE the kernel's duration)
HGT_CHUNK = 128     # 512 -> 128: encode 12.4 -> 11.8 ms, finetune step 68.7 -> 66.0 ms


def _rowptr(sorted_dst: torch.Tensor, n: int) -> torch.Tensor:
    deg = torch.bincount(sorted_dst, minlength=n)
    rp = torch.zeros(n + 1, dtype=torch.int64, device=sorted_dst.device)
    torch.cumsum(deg, 0, out=rp[1:])
    return rp


def molecule_plan(graph) -> dict:
    """CSR-by-destination view of a packed molecule batch (duck-typed torchdrug PackedMolecule)."""
    dev = graph.node_feature.device
    cached = getattr(graph, "_mdg_plan", None)
    if cached is not None and cached["device"] == dev:
        return cached
    el = graph.edge_list
    src, dst = el[:, 0].long(), el[:, 1].long()
    A = int(graph.node_feature.shape[0])
    order = torch.argsort(dst, stable=True)
    ew = getattr(graph, "edge_weight", None)
    w_sorted = None
    if ew is not None:
        ew = ew.float()
        if not bool(torch.all(ew == 1.0)):
            w_sorted = ew[order].contiguous()
    ef = graph.edge_feature.float()[order]
    E = ef.shape[0]
    # bond features + a ones column (weighted in-degree) + zero pad to a multiple of 4
    width = ef.shape[1] + 1
    pad = (-width) % 4
    ef_aug = torch.cat([ef, torch.ones(E, 1, device=dev), torch.zeros(E, pad, device=dev)], dim=1).contiguous()
    n2g = graph.node2graph.long()
    n_graphs = int(getattr(graph, "batch_size", int(n2g.max()) + 1 if n2g.numel() else 0))
    if n2g.numel() > 1 and not bool(torch.all(n2g[1:] >= n2g[:-1])):
        raise ValueError("node2graph must be non-decreasing (packed graphs)")
    plan = {"device": dev, "rowptr": _rowptr(dst[order], A), "col": src[order].contiguous(), "w": w_sorted,
            "edge_feat_aug": ef_aug, "edge_feat_dim": int(ef.shape[1]), "graph_rowptr": _rowptr(n2g, n_graphs),
            "n_graphs": n_graphs}
    try:
        graph._mdg_plan = plan
    except Exception:       # objects that refuse new attributes: rebuild every call
        pass
    return plan


def hgt_reverse_plan(pd: dict) -> dict:
    """Reversed edge lists of one destination type's HGT plan (backward pass of mdg_hgt_attention): the edges stably
    sorted by key row, cut into work items of <= HGT_CHUNK edges.  Cached inside the plan."""
    if "rev" in pd:
        return pd["rev"]
    col, rowptr = pd["col"], pd["rowptr"]
    dev = col.device
    n_dst = int(rowptr.numel()) - 1
    deg = rowptr[1:] - rowptr[:-1]
    dst = torch.repeat_interleave(torch.arange(n_dst, device=dev), deg)
    order = torch.argsort(col, stable=True)
    cs = col[order]
    rows, cnt = torch.unique_consecutive(cs, return_counts=True)
    n_rows = int(rows.numel())
    rptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    torch.cumsum(cnt, 0, out=rptr[1:])
    chunks = (cnt + HGT_CHUNK - 1) // HGT_CHUNK
    item_ptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    torch.cumsum(chunks, 0, out=item_ptr[1:])
    item_row = torch.repeat_interleave(torch.arange(n_rows, device=dev), chunks)
    k = torch.arange(item_row.numel(), device=dev) - item_ptr[item_row]
    begin = rptr[item_row] + k * HGT_CHUNK
    end = torch.minimum(begin + HGT_CHUNK, rptr[item_row + 1])
    pd["rev"] = {"t_edge": order.contiguous(), "t_dst": dst[order].contiguous(), "item_begin": begin.contiguous(),
                 "item_end": end.contiguous(), "item_ptr": item_ptr, "rows": rows.contiguous(), "n_rows": n_rows,
                 "n_items": int(item_row.numel()), "item_row": item_row.contiguous()}
    return pd["rev"]


def transposed_csr(rowptr: torch.Tensor, col, w, n_src: int, mean: bool = False) -> dict:
    """CSR of the reversed edges of (rowptr, col, w): the plan of the backward pass of mdg_csr_aggregate
    (d x[u] = sum over edges u -> v of w_e * d out[v]).  ``col`` None = row v owns source rows rowptr[v]..rowptr[v+1];
    ``mean`` folds the 1/len(row) factor of the forward mean into the reversed weights."""
    dev = rowptr.device
    n_dst = int(rowptr.numel()) - 1
    counts = rowptr[1:] - rowptr[:-1]
    E = int(rowptr[-1].item()) if n_dst > 0 else 0
    row_of_edge = torch.repeat_interleave(torch.arange(n_dst, device=dev), counts)
    src = torch.arange(E, device=dev) if col is None else col
    order = torch.argsort(src, stable=True)
    wt = None
    if w is not None or mean:
        wt = torch.ones(E, device=dev) if w is None else w.float()
        if mean:
            wt = wt / counts.clamp_min(1).to(torch.float32)[row_of_edge]
        wt = wt[order].contiguous()
    return {"rowptr": _rowptr(src[order], n_src), "col": row_of_edge[order].contiguous(), "w": wt}


def hgt_plan(edge_index_dict, edge_types: Sequence[Tuple[str, str, str]], sizes: Dict[str, int], device, used=None, dst_range=None) -> dict:
    """Layout + CSR plan of one HGTConv call.

    Projection layout: node type t owns rows of ``width[t] = 128 + 256 * R_t`` floats in one flat buffer starting at
    float offset ``base[t]``: [ q (128) | k'_0 v'_0 | k'_1 v'_1 | ... ] where slot i belongs to the i-th USED edge
    type whose source is t (one GEMM per node type writes the whole row).  Viewed as [*,128] the relation-transformed
    key of source node j under slot i is row (base[t] + j*width[t] + 128 + 256*i) / 128 and its value the next row.

    Per destination type: the incoming edges of every used edge type concatenated, stably sorted by destination,
    ``col[e]`` = that key row; rows longer than HGT_CHUNK are split into work items.

    ``dst_range`` {type: (lo, hi)} (destination-partitioned conv, one rank of a multi-GPU encode): only the edges whose
    destination lies in [lo, hi) are kept and destinations are renumbered from lo, so the per-destination lists describe
    hi - lo rows; the projection layout (sources: every node) is unchanged.  A row's edge order is the unpartitioned one
    (stable sorts), so its sums come out bit-identical."""
    present = [et for et in edge_types if et in edge_index_dict]
    used = present if used is None else [et for et in present if et in set(used)]
    slot, nrel = {}, {t: 0 for t in sizes}
    for et in used:
        slot[et] = nrel[et[0]]
        nrel[et[0]] += 1
    width = {t: 128 + 256 * nrel[t] for t in sizes}
    base, total = {}, 0
    for t in sizes:
        base[t] = total
        total += sizes[t] * width[t]
    per_dst = {}
    for t, n_t in sizes.items():
        cols, dsts = [], []
        for et in used:
            if et[2] != t:
                continue
            ei = edge_index_dict[et].to(device).long()
            if dst_range is not None:
                lo, hi = dst_range[t]
                keep = (ei[1] >= lo) & (ei[1] < hi)
                ei = torch.stack([ei[0][keep], ei[1][keep] - lo])
            if ei.shape[1] == 0:
                continue
            s = et[0]
            cols.append((base[s] + 128 + 256 * slot[et]) // 128 + ei[0] * (width[s] // 128))
            dsts.append(ei[1])
        if cols:
            col, dst = torch.cat(cols), torch.cat(dsts)
            order = torch.argsort(dst, stable=True)
            col, dst = col[order].contiguous(), dst[order]
        else:
            col = torch.zeros(0, dtype=torch.int64, device=device)
            dst = col
        if dst_range is not None:
            n_t = dst_range[t][1] - dst_range[t][0]
        rowptr = _rowptr(dst, n_t)
        deg = rowptr[1:] - rowptr[:-1]
        chunks = (deg + HGT_CHUNK - 1) // HGT_CHUNK
        item_ptr = torch.zeros(n_t + 1, dtype=torch.int64, device=device)
        torch.cumsum(chunks, 0, out=item_ptr[1:])
        item_dst = torch.repeat_interleave(torch.arange(n_t, device=device), chunks)
        k = torch.arange(item_dst.numel(), device=device) - item_ptr[item_dst]
        item_begin = rowptr[item_dst] + k * HGT_CHUNK
        item_end = torch.minimum(item_begin + HGT_CHUNK, rowptr[item_dst + 1])
        per_dst[t] = {"col": col, "rowptr": rowptr, "item_ptr": item_ptr, "item_dst": item_dst.contiguous(),
                      "item_begin": item_begin.contiguous(), "item_end": item_end.contiguous()}
    return {"used": used, "slot": slot, "nrel": nrel, "width": width, "base": base, "total_floats": total, "per_dst": per_dst}
