"""Shared helpers for the parity tests (CPU and GPU)."""
import numpy as np
import torch

from oracle.params import det_state_dict

# name, heads, head_dim, ffn, layers, norm_first, agg, nb, actn   (same list as oracle/gen_golden.py)
FUSION_CASES = [
    ("drugbank163", 8, 64, 256, 2, True, "x-attn", 4, "gelu"),
    ("twosides105", 2, 256, 512, 2, True, "x-attn", 2, "gelu"),
    ("twosides321", 8, 256, 1024, 2, True, "x-attn", 2, "gelu"),
    ("cl_default", 4, 128, 512, 3, False, "x-attn", 0, "gelu"),
    ("cls_small", 4, 32, 64, 1, True, "cls", 2, "relu"),
    ("mean_small", 4, 32, 64, 2, False, "mean", 0, "gelu"),
    ("max_small", 2, 64, 128, 1, True, "max", 2, "gelu"),
]

# name, fusion, nb, pos, heads, head_dim, ffn, layers, norm_first, agg, normalize, adapt
ENCODE_CASES = [
    ("drugbank163", "transformer", 4, "sinusoidal", 8, 64, 256, 2, True, "x-attn", False, False),
    ("uniproj", "transformer_uni_proj", 2, "learnable", 2, 64, 128, 2, True, "x-attn", True, False),
    ("cls_adapt", "transformer", 2, "learnable", 4, 32, 64, 1, False, "cls", False, True),
    ("meanfuse", "mean", 0, "learnable", 4, 32, 64, 1, False, "x-attn", True, False),
]


def t(a):
    return torch.from_numpy(np.asarray(a))


def fusion_param_shapes(H, dh, ffn, nl, agg):
    d = H * dh
    s = {"embed2latent.weight": (d, 128), "embed2latent.bias": (d,), "latent2embed.weight": (128, d),
         "latent2embed.bias": (128,)}
    for i in range(nl):
        p = f"transformer_encoder.layers.{i}."
        s.update({p + "self_attn.in_proj_weight": (3 * d, d), p + "self_attn.in_proj_bias": (3 * d,),
                  p + "self_attn.out_proj.weight": (d, d), p + "self_attn.out_proj.bias": (d,),
                  p + "linear1.weight": (ffn, d), p + "linear1.bias": (ffn,), p + "linear2.weight": (d, ffn),
                  p + "linear2.bias": (d,), p + "norm1.weight": (d,), p + "norm1.bias": (d,),
                  p + "norm2.weight": (d,), p + "norm2.bias": (d,)})
    if agg == "x-attn":
        s.update({"x_attn_kv_norm.weight": (d,), "x_attn_kv_norm.bias": (d,), "x_attn_query_norm.weight": (d,),
                  "x_attn_query_norm.bias": (d,), "x_attn_mha_layer.in_proj_weight": (3 * d, d),
                  "x_attn_mha_layer.in_proj_bias": (3 * d,), "x_attn_mha_layer.out_proj.weight": (d, d),
                  "x_attn_mha_layer.out_proj.bias": (d,), "x_attn_query": (1, d)})
    return s


def fusion_params(seed, H, dh, ffn, nl, agg):
    return det_state_dict(seed, fusion_param_shapes(H, dh, ffn, nl, agg))


def mlp_shapes(in_dim, hidden, out, p, norm, order="nd"):
    """state_dict shapes of MLPEncoder / MLPAdaptor (madrigal/models/models.py:121-180)."""
    s, idx = {}, 0
    s[f"fc.{idx}.weight"], s[f"fc.{idx}.bias"] = (hidden[0], in_dim), (hidden[0],)
    idx += 2
    for i in range(len(hidden) - 1):
        slots = (["n"] if norm not in (None, "None") else []) + (["d"] if p != 0 else [])
        if order == "dn":
            slots = slots[::-1]
        for sl in slots:
            if sl == "n":
                s[f"fc.{idx}.weight"], s[f"fc.{idx}.bias"] = (hidden[i],), (hidden[i],)
                if norm == "bn":
                    s[f"fc.{idx}.running_mean"], s[f"fc.{idx}.running_var"] = (hidden[i],), (hidden[i],)
                    s[f"fc.{idx}.num_batches_tracked"] = ()
            idx += 1
        s[f"fc.{idx}.weight"], s[f"fc.{idx}.bias"] = (hidden[i + 1], hidden[i]), (hidden[i + 1],)
        idx += 2
    s[f"fc.{idx}.weight"], s[f"fc.{idx}.bias"] = (out, hidden[-1]), (out,)
    return s


def chemcpa_shapes(num_genes=978, width=512, depth=2, dim=128, n_cov=16):
    s = {}
    for name, sizes in (("encoder", [num_genes] + [width] * depth + [dim]),
                        ("decoder", [dim] + [width] * depth + [2 * num_genes])):
        for k in range(len(sizes) - 1):
            s[f"{name}.network.{3 * k}.weight"], s[f"{name}.network.{3 * k}.bias"] = (sizes[k + 1], sizes[k]), (sizes[k + 1],)
            if k < len(sizes) - 2:
                for leaf, shp in (("weight", (sizes[k + 1],)), ("bias", (sizes[k + 1],)), ("running_mean", (sizes[k + 1],)),
                                  ("running_var", (sizes[k + 1],)), ("num_batches_tracked", ())):
                    s[f"{name}.network.{3 * k + 1}.{leaf}"] = shp
    s["covariates_embeddings.0.weight"] = (n_cov, dim)
    return s


def rel_err(a, b):
    """Norm-wise relative error max|a-b| / max|b| (the 1e-4 bar of BASELINE.json is read this way:
    relative to the score scale, since individual logits pass through zero)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def gin_shapes(input_dim=67, hidden=(128, 128, 128, 128), edge_dim=18, n_mlp=3, batch_norm=True):
    s, dims = {}, [input_dim] + list(hidden)
    for k in range(len(dims) - 1):
        p = f"layers.{k}."
        s[p + "eps"] = (1,)
        if batch_norm:
            for leaf, shp in (("weight", (dims[k + 1],)), ("bias", (dims[k + 1],)), ("running_mean", (dims[k + 1],)),
                              ("running_var", (dims[k + 1],)), ("num_batches_tracked", ())):
                s[p + "batch_norm." + leaf] = shp
        md = [dims[k]] + [dims[k + 1]] * n_mlp
        for j in range(n_mlp):
            s[p + f"mlp.layers.{j}.weight"], s[p + f"mlp.layers.{j}.bias"] = (md[j + 1], md[j]), (md[j + 1],)
        s[p + "edge_linear.weight"], s[p + "edge_linear.bias"] = (dims[k], edge_dim), (dims[k],)
    return s


def hgt_shapes(kg, in_dim=128, hidden=128, out=128, heads=4, n_layers=2):
    s = {}
    node_types, edge_types = kg.metadata()
    for i in range(n_layers):
        p, cin = f"convs.{i}.", in_dim if i == 0 else hidden
        for tname in node_types:
            s[p + f"kqv_lin.lins.{tname}.weight"], s[p + f"kqv_lin.lins.{tname}.bias"] = (3 * hidden, cin), (3 * hidden,)
            s[p + f"out_lin.lins.{tname}.weight"], s[p + f"out_lin.lins.{tname}.bias"] = (hidden, hidden), (hidden,)
            s[p + f"skip.{tname}"] = (1,)
        dh = hidden // heads
        s[p + "k_rel.weight"] = s[p + "v_rel.weight"] = (heads * len(edge_types), dh, dh)
        for e in edge_types:
            s[p + "p_rel." + "__".join(e)] = (1, heads)
    for tname in node_types:
        s[f"lin_dict.{tname}.weight"], s[f"lin_dict.{tname}.bias"] = (out, hidden), (out,)
    return s


def model_shapes_for_case(case, kg, L):
    """state_dict shapes of NovelDDIMultilabel(NovelDDIEncoder(...)) for one ENCODE_CASES row, as the
    reference lays it out (SURVEY.md 8b state-dict contract)."""
    name, fusion, nb, pos, H, dh, ffn, nl, nf, agg, normalize, adapt = case
    s = {"decoder.bias": (L,), "decoder.parametrizations.weight.original": (L, 128, 128)}

    def add(prefix, d):
        s.update({prefix + k: v for k, v in d.items()})
    add("encoder.str_encoder.", gin_shapes())
    add("encoder.kg_encoder.", hgt_shapes(kg))
    add("encoder.cv_encoder.", mlp_shapes(559, [512, 256], 128, 0.2, None))
    add("encoder.tx_encoder.", chemcpa_shapes())
    add("encoder.transformer.", fusion_param_shapes(H, dh, ffn, nl, agg))
    add("encoder.uni_projector.", mlp_shapes(128, [512, 512], 128, 0.2, "ln"))
    if fusion == "transformer_uni_proj":
        add("encoder.uni_fuser.", mlp_shapes(128, [512, 512], 128, 0.2, "ln"))
    if nb > 0:
        s["encoder.tx_bottleneck_tokens"] = (nb, 128)
    if agg == "cls":
        s["encoder.cls"] = (1, 128)
    max_len = (19 if nb == 0 else 3) + (1 if agg == "cls" else 0)
    if pos == "sinusoidal":
        seq = 19 + nb + (1 if agg == "cls" else 0)
        s["encoder.pos_encoder.pe"] = (1, seq if nb > 0 else max_len, 128)
    else:
        s["encoder.pos_encoder.pe"] = (1, max_len, 128)
    return s


from oracle.pipeline import oracle_pipeline  # noqa: E402,F401  (whole-path CPU oracle)


def set_switch(monkeypatch, name: str, value: str) -> None:
    """Flip one of the library's MDG_* tuning switches for the following launches: the library reads a switch once, so the
    environment change is followed by mdg_tuning_reload() (conftest's autouse fixture reloads again after the test)."""
    from madrigal_amd._lib import lib
    monkeypatch.setenv(name, value)
    lib().mdg_tuning_reload()


def gin_bias_case():
    """One molecule that separates the two readings of torchdrug's GIN edge-linear bias (oracle.gin_forward's docstring):
    atom 0 has three incoming bonds, atoms 1-3 one each, atom 4 none (isolated), atom 5 one (from atom 3).  Positive one-hot
    features, a one-layer conv whose MLP is the identity and whose edge_linear has a small positive weight and the bias
    ``b``: every pre-activation is positive (ReLU = identity), so the node outputs of the per-edge reading minus those of the
    per-atom reading are EXACTLY (deg_v - 1) b.  -> (MoleculeBatch, params, b, in-degree)"""
    import torch
    from madrigal_amd import data as D
    dim, fe = 8, 18
    src = torch.tensor([1, 2, 3, 0, 0, 0, 3])
    dst = torch.tensor([0, 0, 0, 1, 2, 3, 5])
    g = torch.Generator().manual_seed(77)
    x = torch.zeros(6, dim)
    x[torch.arange(6), torch.randint(0, dim, (6,), generator=g)] = 1.0
    ef = torch.zeros(7, fe)
    ef[torch.arange(7), torch.randint(0, fe, (7,), generator=g)] = 1.0
    mols = D.MoleculeBatch(x, torch.stack([src, dst, torch.zeros(7, dtype=torch.int64)], 1), ef, torch.zeros(6, dtype=torch.int64), 1)
    b = torch.rand(dim, generator=g) + 0.25
    p = {"layers.0.eps": torch.tensor([0.0]), "layers.0.mlp.layers.0.weight": torch.eye(dim), "layers.0.mlp.layers.0.bias": torch.zeros(dim),
         "layers.0.edge_linear.weight": torch.rand(dim, fe, generator=g) * 0.125, "layers.0.edge_linear.bias": b}
    return mols, p, b, torch.bincount(dst, minlength=6).float()


def smooth_relu(model, name="gelu"):
    """The model with every ReLU replaced by a smooth activation (nn.ReLU modules; the GIN layers' ``activation`` string): the
    counterpart of ``oracle.relu_as`` for the whole-model gradient comparisons."""
    import torch
    from madrigal_amd.models import _make_act
    for mod in list(model.modules()):
        for cname, child in list(mod.named_children()):
            if isinstance(child, torch.nn.ReLU):
                setattr(mod, cname, _make_act(name))
        if getattr(mod, "activation", None) == "relu":
            mod.activation = name
    return model


def assert_tensors_agree(errs, strict, loose, max_outliers=3, what=""):
    """The acceptance rule of the whole-model gradient comparisons between two fp32 implementations of a ReLU network (the HIP path
    against the CPU oracle's autograd; a multi-rank step against the single-process one).  ``errs`` = one (relative error, name) per
    parameter tensor.  Such a comparison has a failure mode that is not an error of either side: a pre-activation within fp32 rounding
    of zero takes derivative 0 on one side and 1 on the other, and ONE row of the weight gradient it feeds (and that layer's bias entry)
    moves by that sample's whole contribution -- measured 3e-3 ... 2e-2 of the tensor's scale in a handful of entries of one or two
    tensors (scripts/pretrain_flip_probe.py), where an arithmetic defect (a wrong term, a wrong scale, a few percent lost in a
    reduction) moves every tensor downstream of it.  So, on EVERY seed -- no search, no seed is allowed to fail, the seeds are a fixed
    ascending range -- all tensors but at most ``max_outliers`` sit under ``strict`` and every tensor sits under ``loose``."""
    errs = sorted(errs, reverse=True)
    over = [e for e in errs if e[0] >= strict]
    assert len(over) <= max_outliers, (what, f"{len(over)} of {len(errs)} tensors beyond {strict}", over[:8])
    assert errs[0][0] < loose, (what, errs[:4])
    return errs[0], len(over)


def triple_plan_torch(labels: torch.Tensor, heads: torch.Tensor, tails: torch.Tensor, n_labels: int, n_head: int, n_tail: int) -> dict:
    """CHECKER for madrigal_amd.ops.triple_plan (its construction until round 5: torch sorts / index / scan calls on the device): triples sorted by label, cut into tiles of <= 32 and chunks of <= 256 triples of one label;
    CSR lists of the sorted triples per head drug and per tail drug for the gradient row sums."""
    T = int(labels.numel())
    dev = labels.device
    for nm, t in (("labels", labels), ("heads", heads), ("tails", tails)):
        if t.dtype != torch.int64 or not t.is_cuda or t.numel() != T or t.dim() != 1:
            raise ValueError(f"{nm}: expected int64 cuda [{T}]")
    # ONE sort serves the label order and the (label, head drug) pair order: by label, then by head inside a label (any
    # label-sorted order will do for the tiles; a pair's triples must be consecutive for the pair-compressed head).
    # int32 keys when they fit (twice the radix-sort rate).
    big = n_labels * max(n_head, 1) >= 2 ** 31
    key = labels * n_head + heads if big else (labels * n_head + heads).to(torch.int32)
    perm = torch.argsort(key, stable=True)
    hs, ts = heads[perm].contiguous(), tails[perm].contiguous()
    try:
        counts = torch.bincount(labels, minlength=n_labels)
    except RuntimeError as e:
        raise ValueError(f"labels: value outside [0, n_labels) ({e})") from None
    if counts.numel() != n_labels:
        raise ValueError("labels: value outside [0, n_labels)")
    zero = torch.zeros(1, dtype=torch.int64, device=dev)
    label_ptr = torch.cat([zero, torch.cumsum(counts, 0)])
    lab = torch.arange(n_labels, device=dev)

    def cut(size):
        per = (counts + size - 1) // size
        first = torch.cumsum(per, 0) - per
        which = torch.repeat_interleave(lab, per)
        start = label_ptr[which] + size * (torch.arange(which.numel(), device=dev) - first[which])
        return per, which.contiguous(), torch.cat([start, torch.tensor([T], dtype=torch.int64, device=dev)]).contiguous()
    _, tile_label, tile_start = cut(32)
    chunks_per, _, chunk_start = cut(256)
    label_chunk_ptr = torch.cat([zero, torch.cumsum(chunks_per, 0)]).contiguous()

    def by_drug(idx, n):
        # the range check of the drug indices rides on the histogram: bincount raises on a negative entry and returns more than n
        # bins when one is >= n (no separate max / min reductions over the triples)
        order = torch.argsort(idx if n >= 2 ** 31 else idx.to(torch.int32), stable=True)
        try:
            cnt = torch.bincount(idx, minlength=n)
        except RuntimeError as e:
            raise ValueError(f"heads / tails: index outside the embedding tables ({e})") from None
        if cnt.numel() != n:
            raise ValueError("heads / tails: index outside the embedding tables")
        return torch.cat([zero, torch.cumsum(cnt, 0)]).contiguous(), order.contiguous()

    def pieces(ptr):
        """A drug's list can hold thousands of entries while mdg_csr_aggregate gives a row to one group of lanes: cut every
        list into pieces of <= 64 entries -> (piece_ptr over the entries, row_ptr over the pieces) for a two-level sum
        (_sum_rows); None when no list is long enough to matter."""
        counts_ = ptr[1:] - ptr[:-1]
        if counts_.numel() == 0 or int(counts_.max()) <= 256:
            return None
        per = (counts_ + 63) // 64
        first = torch.cumsum(per, 0) - per
        which = torch.repeat_interleave(torch.arange(counts_.numel(), device=dev), per)
        start = ptr[which] + 64 * (torch.arange(which.numel(), device=dev) - first[which])
        return (torch.cat([start, ptr[-1:]]).contiguous(), torch.cat([zero, torch.cumsum(per, 0)]).contiguous())
    head_ptr, head_rows = by_drug(hs, n_head)
    tail_ptr, tail_rows = by_drug(ts, n_tail)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(T, device=dev)
    # ---- (label, head drug) PAIRS: the batch holds more labelled triples than pairs, and every 128 x 128 product of the head
    # depends on the pair only (bilinear_gather_pairs / _bwd).  Triples re-sorted by (label, head): pair p owns the triples
    # pair_ptr[p] .. pair_ptr[p+1] of that order.
    pairs = None
    if T:
        pkey, pcnt = torch.unique_consecutive(key[perm], return_counts=True)     # the plan's triple order IS the pair order
        P = int(pkey.numel())
        pkey = pkey.to(torch.int64)
        pair_label, pair_drug = pkey // n_head, (pkey % n_head).contiguous()
        pair_ptr = torch.cat([zero, torch.cumsum(pcnt, 0)]).contiguous()
        pair_of_triple = torch.repeat_interleave(torch.arange(P, device=dev), pcnt)  # sorted triple -> its pair
        pcounts = torch.bincount(pair_label, minlength=n_labels)
        plabel_ptr = torch.cat([zero, torch.cumsum(pcounts, 0)])

        def pcut(size):
            per = (pcounts + size - 1) // size
            first = torch.cumsum(per, 0) - per
            which = torch.repeat_interleave(lab, per)
            start = plabel_ptr[which] + size * (torch.arange(which.numel(), device=dev) - first[which])
            return per, which.contiguous(), torch.cat([start, torch.tensor([P], dtype=torch.int64, device=dev)]).contiguous()
        _, ptile_label, ptile_start = pcut(32)
        pchunks_per, _, pchunk_start = pcut(512)          # (256: 0.82 ms, 512 / 1024: 0.75 ms, 2048: 1.05 ms for the 2.4e6 pairs of the bench step)
        drug_ptr, drug_rows = by_drug(pair_drug, n_head)
        pairs = {"P": P, "drug": pair_drug, "ptr": pair_ptr, "tails_by_pair": ts,
                 "of_triple": pair_of_triple, "tile_start": ptile_start, "tile_label": ptile_label, "n_tiles": int(ptile_label.numel()),
                 "chunk_start": pchunk_start, "n_chunks": int(pchunk_start.numel()) - 1,
                 "label_chunk_ptr": torch.cat([zero, torch.cumsum(pchunks_per, 0)]).contiguous(), "drug_ptr": drug_ptr, "drug_rows": drug_rows,
                 "of_triple_by_tail": pair_of_triple[tail_rows].contiguous(), "drug_pieces": pieces(drug_ptr)}
    return {"T": T, "L": n_labels, "n_head": n_head, "n_tail": n_tail, "perm": perm, "inv_perm": inv, "heads": hs, "tails": ts, "pairs": pairs,
            "tile_start": tile_start, "tile_label": tile_label, "n_tiles": int(tile_label.numel()), "chunk_start": chunk_start,
            "n_chunks": int(chunk_start.numel()) - 1, "label_chunk_ptr": label_chunk_ptr, "head_ptr": head_ptr,
            "head_rows": head_rows, "tail_ptr": tail_ptr, "tail_rows": tail_rows, "head_pieces": pieces(head_ptr),
            "tail_pieces": pieces(tail_ptr)}
