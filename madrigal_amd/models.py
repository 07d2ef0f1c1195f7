"""MI355X-native drop-in for ``madrigal.models.models`` (+ the torchdrug GIN, PyG HGTConv and
chemCPA ``TxAdaptingComPert.predict`` pieces it calls): same class names, constructor and
``forward`` signatures, attribute names and ``state_dict`` keys as the reference, so that
``madrigal/utils.py:get_model`` / ``create_optimizer`` and checkpoints keep working; the arithmetic
runs in hand-written HIP kernels behind the C ABI of ``include/madrigal_hip.h`` (no PyTorch
arithmetic on the hot path, no CPU fallback: the ops raise if the library is missing).

Inference (``eval()`` under ``torch.no_grad()``) runs the fused forward kernels with cached derived weights.  Training
mode / autograd (dropout, BatchNorm batch statistics, gradients) runs the same modules through the tape nodes of
``madrigal_amd/autograd.py``: forward and backward arithmetic in HIP kernels, torch only as the tape.

Reference lines cited per class (paths relative to the reference checkout).
"""
from __future__ import annotations

import math
import os
from contextlib import contextmanager
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import autograd as ag
from . import ops
from .data import CELL_LINES, MOL_DIM, NUM_MODALITIES, NUM_NON_TX_MODALITIES
from .graph_plans import hgt_plan, molecule_plan, transposed_csr

TX_INPUT_DIM = 978
CELL_LINES_CAPITALIZED = [c.upper() for c in CELL_LINES]

_state = {"precision": "bf16x3"}


def set_precision(p: str) -> None:
    """Arithmetic of every matrix product on the path: 'f32' (exact fp32 MFMA), 'bf16x3'
    (split-bf16, fp32-grade; default) or 'bf16'."""
    if p not in ops.PRECISIONS:
        raise ValueError(f"unknown precision {p!r}")
    _state["precision"] = p


def get_precision() -> str:
    return _state["precision"]


@contextmanager
def precision(p: str):
    old = get_precision()
    set_precision(p)
    try:
        yield
    finally:
        set_precision(old)


def _lin(x, w, b=None, **kw):
    return ops.linear(x, w, b, precision=_state["precision"], **kw)


def _train_path(m: nn.Module) -> bool:
    """True when the call must go through the autograd nodes (madrigal_amd/autograd.py): the module is in
    training mode (dropout / BatchNorm batch statistics) or a gradient is being recorded for its parameters."""
    return m.training or (torch.is_grad_enabled() and any(p.requires_grad for p in m.parameters()))


def _linT(x, w, b=None, act=None, x_image=None):
    return ag.linear(x, w, b, act, _state["precision"], x_image)


def _linT_drop(x, w, b, act, drop: nn.Dropout, residual=None, x_image=None):
    """residual + drop(act(x W^T + b)) as one autograd node (dropout inside the dense block: autograd._Linear)."""
    return ag.linear_dropout(x, w, b, act, _state["precision"], drop.p, drop.training, residual, x_image=x_image)


def _require_eval(m: nn.Module) -> None:
    """Guard of the few modules that have no differentiated / training-mode path (stand-alone position encoders, the
    SimCLR projection head): refuse rather than return tensors without a graph."""
    if m.training:
        raise RuntimeError(f"{type(m).__name__}: this module has no training-mode path on the HIP side; call .eval() first")
    if torch.is_grad_enabled() and any(p.requires_grad for p in m.parameters()):
        raise RuntimeError(f"{type(m).__name__}: this module is forward-only on the HIP side: call it under torch.no_grad()")


# madrigal/models/models.py:31.  Fresh instances per use; only the type matters for the fused epilogue.
def _make_act(name):
    table = {'relu': nn.ReLU, 'leakyrelu': nn.LeakyReLU, 'tanh': nn.Tanh, 'sigmoid': nn.Sigmoid, 'selu': nn.SELU,
             'softplus': nn.Softplus, 'gelu': nn.GELU, None: nn.Identity}
    if name not in table:
        raise NotImplementedError(name)
    return table[name]()


_ACT_OF = {nn.ReLU: "relu", nn.LeakyReLU: "leakyrelu", nn.Tanh: "tanh", nn.Sigmoid: "sigmoid", nn.SELU: "selu",
           nn.Softplus: "softplus", nn.GELU: "gelu", nn.Identity: None}


def _cached(owner, tag, tensors, build):
    """Weights derived from parameters (folded BatchNorm, block-diagonal relation matrices, padded / augmented
    matrices) are rebuilt only when one of the source tensors changed (storage address or in-place version).
    The cache lives ON the owning module (it dies with it: Python recycles ids, the allocator recycles addresses)
    and each entry pins its source tensors so that their addresses cannot be reused while the entry is alive."""
    srcs = tuple(t for t in tensors if t is not None)
    key = tuple((t.data_ptr(), t._version, str(t.device)) for t in srcs)
    slot = owner.__dict__.setdefault("_mdg_derived", {})
    hit = slot.get(tag)
    if hit is None or hit[0] != key:
        hit = (key, build(), srcs)
        slot[tag] = hit
    return hit[1]


def _put_diag(H: int, blocks: torch.Tensor) -> torch.Tensor:
    """blocks [H,R,D,D] -> [R,H,D,H,D] with out[r, h, :, h, :] = blocks[h, r] and zeros elsewhere (block-diagonal placement
    as one broadcast product with an identity mask: a single differentiable op in parameter space)."""
    eye = torch.eye(H, dtype=blocks.dtype, device=blocks.device)
    # [R,H,D,1,D] * [1,H,1,H,1]: the block of head h lands in column-block h only
    return blocks.permute(1, 0, 2, 3).unsqueeze(3) * eye.view(1, H, 1, H, 1)


def _bn_scale_shift(bn: nn.BatchNorm1d):
    """Eval-mode BatchNorm as y = x * scale + shift."""
    def build():
        scale = torch.rsqrt(bn.running_var + bn.eps)
        if bn.weight is not None:
            scale = scale * bn.weight.detach()
        shift = -bn.running_mean * scale
        if bn.bias is not None:
            shift = shift + bn.bias.detach()
        return scale.contiguous(), shift.contiguous()
    return _cached(bn, "affine", (bn.running_var, bn.running_mean, bn.weight, bn.bias), build)


def _run_sequential(seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """Execute an nn.Sequential of Linear / activation / LayerNorm / BatchNorm1d / Dropout (eval) with
    the fused kernels: every Linear absorbs the BatchNorm and activation that FOLLOW it; a BatchNorm
    that PRECEDES a Linear (MLPEncoder's 'nd' order) is folded into that Linear's weights."""
    mods = list(seq)
    i = 0
    pre_affine = None          # (scale, shift) of an eval BatchNorm waiting for the next Linear
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Dropout) or isinstance(m, nn.Identity):
            i += 1
        elif isinstance(m, nn.LayerNorm):
            x = ops.layernorm(x, m.weight, m.bias, m.eps)
            i += 1
        elif isinstance(m, nn.BatchNorm1d):
            pre_affine = _bn_scale_shift(m)
            i += 1
        elif isinstance(m, nn.Linear):
            w, b = m.weight.detach(), None if m.bias is None else m.bias.detach()
            if pre_affine is not None:          # W (s*x + t) + b = (W*s) x + (W t + b)
                sc, sh = pre_affine

                def fold(w=w, b=b, sc=sc, sh=sh):
                    return (w * sc.unsqueeze(0)).contiguous(), ((w @ sh) if b is None else b + w @ sh).contiguous()
                w, b = _cached(m, "prefold", (m.weight, m.bias, sc, sh), fold)
                pre_affine = None
            j, scale, shift, act = i + 1, None, None, None
            if j < len(mods) and isinstance(mods[j], nn.BatchNorm1d):
                scale, shift = _bn_scale_shift(mods[j])
                j += 1
            if j < len(mods) and type(mods[j]) in _ACT_OF and not isinstance(mods[j], nn.Identity):
                act = _ACT_OF[type(mods[j])]
                j += 1
            x = _lin(x, w, b, scale=scale, shift=shift, act=act)
            i = j
        elif type(m) in _ACT_OF:
            raise NotImplementedError("activation without a preceding Linear")
        else:
            raise NotImplementedError(type(m).__name__)
    if pre_affine is not None:
        raise NotImplementedError("trailing BatchNorm without a Linear")
    return x


def _bn_or_affine(x, bn: nn.BatchNorm1d, act):
    """BatchNorm1d inside the differentiated graph: batch statistics when bn.training, else the folded running
    statistics as a constant per-column affine map."""
    if bn.training or not bn.track_running_stats:
        return ag.batchnorm_act(x, bn, act)
    if (bn.weight is not None and bn.weight.requires_grad) and torch.is_grad_enabled():
        raise NotImplementedError("gradients of an eval-mode BatchNorm's affine parameters (freeze them or use train())")
    scale, shift = _bn_scale_shift(bn)
    return ag.affine_act(x, scale, shift, act)


def _run_sequential_train(seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """The same nn.Sequential walked through the autograd nodes: Linear absorbs the activation (and the
    BatchNorm + activation) that follows it; Dropout / BatchNorm honour each sub-module's own ``training`` flag."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Identity):
            i += 1
        elif isinstance(m, nn.Dropout):
            x = ag.dropout(x, m.p, m.training)
            i += 1
        elif isinstance(m, nn.LayerNorm):
            x = ag.layernorm(x, m.weight, m.bias, m.eps)
            i += 1
        elif isinstance(m, nn.BatchNorm1d):
            x = _bn_or_affine(x, m, None)
            i += 1
        elif isinstance(m, nn.Linear):
            j, bn, act = i + 1, None, None
            if j < len(mods) and isinstance(mods[j], nn.BatchNorm1d):
                bn = mods[j]
                j += 1
            if j < len(mods) and type(mods[j]) in _ACT_OF and not isinstance(mods[j], nn.Identity):
                act = _ACT_OF[type(mods[j])]
                j += 1
            if bn is None:
                x = _linT(x, m.weight, m.bias, act)
            else:
                x = _bn_or_affine(_linT(x, m.weight, m.bias, None), bn, act)
            i = j
        elif type(m) in _ACT_OF:
            raise NotImplementedError("activation without a preceding Linear")
        else:
            raise NotImplementedError(type(m).__name__)
    return x


# ------------------------------------------------------------------------------------- MLPs
class _MLPStack(nn.Module):
    """Shared body of MLPEncoder and MLPAdaptor.  The two are SIBLINGS, as in the reference (models.py:121 / :459 are
    unrelated classes): its create_optimizer (madrigal/utils.py:467-479) sorts parameters into learning-rate groups with
    ``isinstance(child, MLPEncoder)`` / ``isinstance(child, MLPAdaptor)``, and an adaptor that IS an encoder would drop
    uni_projector / uni_fuser out of the fusion group (oracle/check_dropin.py)."""

    def __init__(self, in_dim: int, hidden_dims: list, output_dim: int, p: float, norm: str, actn: str, order: str = 'nd'):
        super().__init__()
        self.n_layer = len(hidden_dims) - 1
        self.in_dim = in_dim
        layers = [nn.Linear(in_dim, hidden_dims[0]), _make_act(actn)]
        for i in range(self.n_layer):
            layers += self.compose_layer(hidden_dims[i], hidden_dims[i + 1], norm, actn, p, order)
        layers.append(nn.Linear(hidden_dims[-1], output_dim))
        self.fc = nn.Sequential(*layers)

    @staticmethod
    def compose_layer(in_dim, out_dim, norm, actn, p=0.0, order='nd'):
        if norm in (None, 'None'):
            nl = None
        elif norm == 'bn':
            nl = nn.BatchNorm1d(in_dim)
        elif norm == 'ln':
            nl = nn.LayerNorm(in_dim)
        else:
            raise NotImplementedError(norm)
        if order == 'nd':
            layers = ([nl] if nl is not None else []) + ([nn.Dropout(p)] if p != 0 else [])
        elif order == 'dn':
            layers = ([nn.Dropout(p)] if p != 0 else []) + ([nl] if nl is not None else [])
        else:
            raise NotImplementedError(order)
        layers.append(nn.Linear(in_dim, out_dim))
        if actn is not None:
            layers.append(_make_act(actn))
        return layers

    def forward(self, x):
        lead = x.shape[:-1]
        run = _run_sequential_train if _train_path(self) or ag.needs_grad(x) else _run_sequential
        return run(self.fc, x.reshape(-1, x.shape[-1])).reshape(*lead, -1)


class MLPEncoder(_MLPStack):
    """madrigal/models/models.py:121-180 (cv encoder; also tx_encoder='mlp')."""


class MLPAdaptor(_MLPStack):
    """madrigal/models/models.py:459-518 (uni_projector / uni_fuser); same structure as MLPEncoder, not a subclass of it."""


class VAE(nn.Module):
    """madrigal/models/models.py:183-208 is not on the encode -> fuse -> score path."""

    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("VAE is outside the encode -> fuse -> score path")


# ------------------------------------------------------------------------------------- structure encoder
class GraphIsomorphismNetwork(nn.Module):
    """torchdrug==0.2.1 ``models.GraphIsomorphismNetwork`` as used at madrigal/models/models.py:217,720
    (parameter layout = modality_pretraining/str/GIN_256x4_muv.pt).  PARITY UNPINNED against the real
    wheel (absent from the image): restated from the GIN paper + torchdrug's layer semantics.

    Per layer (the path torchdrug 0.2.1 executes: ``MessagePassingBase.forward`` -> ``GraphIsomorphismConv.
    message_and_aggregate``): update_v = sum_{u->v} w_uv h_u + edge_linear(sum_{u->v} w_uv e_uv) -- the bond features are
    summed per destination atom first and ``edge_linear`` (weight AND bias) is applied ONCE per atom, atoms without
    bonds included; h' = ReLU(BN(MLP((1+eps) h_v + update_v))).  The per-edge work is a pure row gather
    (mdg_csr_aggregate); the MLP runs through mdg_linear with the eval BatchNorm and ReLU fused into the last layer's
    epilogue; read-out = segment mean / sum.  ``edge_bias_per_edge=True`` (not a torchdrug argument) selects the
    un-fused ``message()`` reading instead: sum_{u->v} w_uv (h_u + edge_linear(e_uv)), i.e. the bias times the weighted
    in-degree (rounds 1-3 of this build; differs by (1 - deg_v) b_e per atom)."""

    def __init__(self, input_dim=None, hidden_dims=None, edge_input_dim=None, num_mlp_layer=2, eps=0, learn_eps=False,
                 short_cut=False, batch_norm=False, activation="relu", concat_hidden=False, readout="sum",
                 edge_bias_per_edge=False):
        super().__init__()
        self.edge_bias_per_edge = bool(edge_bias_per_edge)
        if short_cut or concat_hidden:
            raise NotImplementedError("short_cut / concat_hidden are not used by Madrigal")
        if readout not in ("sum", "mean"):
            raise NotImplementedError(readout)
        self.input_dim, self.output_dim = input_dim, hidden_dims[-1]
        self.dims = [input_dim] + list(hidden_dims)
        self.num_mlp_layer, self.readout_kind, self.activation = num_mlp_layer, readout, activation
        self.layers = nn.ModuleList()
        for i in range(len(self.dims) - 1):
            layer = nn.Module()
            e = torch.tensor([float(eps)], dtype=torch.float32)
            if learn_eps:
                layer.eps = nn.Parameter(e)
            else:
                layer.register_buffer("eps", e)
            if batch_norm:
                layer.batch_norm = nn.BatchNorm1d(self.dims[i + 1])
            layer.mlp = nn.Module()
            md = [self.dims[i]] + [self.dims[i + 1]] * num_mlp_layer
            layer.mlp.layers = nn.ModuleList([nn.Linear(md[j], md[j + 1]) for j in range(num_mlp_layer)])
            layer.edge_linear = nn.Linear(edge_input_dim, self.dims[i]) if edge_input_dim else None
            self.layers.append(layer)

    def _bond_sums(self, plan):
        """[A, pad4(fe + 1)]: per destination atom the (weighted) sum of its bonds' features, then the column that meets
        ``edge_linear.bias``: ones for EVERY atom (torchdrug applies edge_linear once per atom to the summed features),
        or, under ``edge_bias_per_edge``, the weighted in-degree the aggregation of the edges' ones column leaves there."""
        esum = ops.csr_aggregate(plan["edge_feat_aug"], plan["rowptr"], None, edge_weight=plan["w"])
        if not self.edge_bias_per_edge:
            esum[:, plan["edge_feat_dim"]] = 1.0
        return esum

    def _forward_train(self, graph, input):
        """Training-mode / differentiated pass: the same kernels through the autograd nodes; BatchNorm uses batch
        statistics over the atoms; the backward of every aggregation is mdg_csr_aggregate on the reversed edges."""
        plan = molecule_plan(graph)
        A = int(input.shape[0])
        if "t_edges" not in plan:
            plan["t_edges"] = transposed_csr(plan["rowptr"], plan["col"], plan["w"], A)
            plan["t_readout"] = transposed_csr(plan["graph_rowptr"], None, None, A, mean=(self.readout_kind == "mean"))
        h = ops._pad_last(input.float()).contiguous()
        esum = None
        for layer in self.layers:
            if isinstance(layer.eps, nn.Parameter) and layer.eps.requires_grad and torch.is_grad_enabled():
                raise NotImplementedError("learn_eps=True is not trained on the HIP path (Madrigal's GIN uses a fixed eps)")
            k_in = layer.mlp.layers[0].weight.shape[1]
            agg = ag.csr_aggregate(h, plan["rowptr"], plan["col"], plan["w"], plan["t_edges"], with_self=True,
                                   coef_dev=layer.eps.detach(), coef_add=1.0)
            if agg.shape[1] != k_in:
                agg = agg[:, :k_in]
            if layer.edge_linear is not None:
                if esum is None:
                    esum = self._bond_sums(plan)
                fe = plan["edge_feat_dim"]
                el = layer.edge_linear             # [W_e | b_e | 0] against [sum_e e_uv | 1 (per_edge: weighted degree) | 0]
                we = torch.cat([el.weight, el.bias.unsqueeze(1), el.weight.new_zeros(k_in, esum.shape[1] - fe - 1)], dim=1)
                agg = ag.add(agg, _linT(esum, we, None))
            u = agg
            n_mlp = len(layer.mlp.layers)
            for j, lin in enumerate(layer.mlp.layers):
                if j == n_mlp - 1 and hasattr(layer, "batch_norm"):
                    u = _bn_or_affine(_linT(u, lin.weight, lin.bias, None), layer.batch_norm, self.activation)
                else:
                    u = _linT(u, lin.weight, lin.bias, self.activation)
            h = u
        g = ag.csr_aggregate(h, plan["graph_rowptr"], None, None, plan["t_readout"], mean=(self.readout_kind == "mean"))
        return {"graph_feature": g[:, : self.output_dim], "node_feature": h}

    def forward(self, graph, input, all_loss=None, metric=None):
        if _train_path(self) or ag.needs_grad(input):
            return self._forward_train(graph, input)
        plan = molecule_plan(graph)
        h = ops._pad_last(input.float()).contiguous()          # 67 atom features -> 68 (zero column)
        esum = None
        for layer in self.layers:
            k_in = layer.mlp.layers[0].weight.shape[1]
            agg = ops.csr_aggregate(h, plan["rowptr"], plan["col"], edge_weight=plan["w"], x_self=h,
                                    self_coef_dev=layer.eps.detach(), self_coef_add=1.0)        # [A, pad4(k_in)]
            if layer.edge_linear is not None:
                if esum is None:       # per-atom sum of (weighted) bond features + the bias column: layer independent
                    esum = self._bond_sums(plan)
                fe = plan["edge_feat_dim"]

                def build(layer=layer, rows=agg.shape[1], cols=esum.shape[1], fe=fe, k_in=k_in, dev=h.device):
                    we = torch.zeros(rows, cols, device=dev)
                    we[:k_in, :fe] = layer.edge_linear.weight.detach()
                    we[:k_in, fe] = layer.edge_linear.bias.detach()
                    return we
                we = _cached(layer.edge_linear, ("aug", agg.shape[1], esum.shape[1]), (layer.edge_linear.weight, layer.edge_linear.bias), build)
                agg = _lin(esum, we, None, residual=agg)                 # + W_e sum_e(e_uv) + b_e
            u = agg
            n_mlp = len(layer.mlp.layers)
            for j, lin in enumerate(layer.mlp.layers):
                scale = shift = None
                if j == n_mlp - 1 and hasattr(layer, "batch_norm"):      # BatchNorm(eval) + the conv's activation
                    scale, shift = _bn_scale_shift(layer.batch_norm)
                u = _lin(u, lin.weight, lin.bias, scale=scale, shift=shift, act=self.activation)
            h = u
        g = ops.csr_aggregate(h, plan["graph_rowptr"], None, mean=(self.readout_kind == "mean"))
        return {"graph_feature": g[:, : self.output_dim], "node_feature": h}


# ------------------------------------------------------------------------------------- KG encoder
class _FlatRows(dict):
    """{node type: rows} whose tensors are consecutive row blocks of ONE buffer (``flat`` = (buffer, type order)): the next
    grouped conv takes the buffer as its stacked input instead of concatenating the blocks again."""
    flat = None


class HGTConv(nn.Module):
    """torch-geometric==2.3.1 ``HGTConv`` as used at madrigal/models/models.py:76-79,90-94 (parameter
    layout of PyG 2.3: kqv_lin.lins.<type>, out_lin.lins.<type>, k_rel/v_rel.weight [H*R,D,D] indexed
    h*R + r, skip.<type>, p_rel.<src>__<rel>__<dst>).  PARITY UNPINNED against the real wheel.

    One mdg_linear per node type produces q and, for every edge type leaving that node type, the
    relation-transformed key/value (the per-head relation matrices, as a block-diagonal 128x128 matrix with
    p_rel/sqrt(D) folded in, are composed with the K / V projection weights); the edge softmax over ALL incoming
    edges of a node fused with the weighted value sum is mdg_hgt_attention; output projection, GELU and the
    sigmoid(skip)-gated residual are fused epilogues."""

    batched_weights = True         # composite projection weights of all node types built at once (False, tests only: one node type at a time)

    def __init__(self, in_channels, out_channels, metadata, heads=1, group="sum", **kwargs):
        super().__init__()
        if out_channels % heads != 0:
            raise ValueError("out_channels must be divisible by heads")
        if out_channels != 128:
            raise NotImplementedError("the HIP HGT kernels are built for hidden size 128 (all shipped configs)")
        self.node_types = list(metadata[0])
        self.edge_types = [tuple(e) for e in metadata[1]]
        if not isinstance(in_channels, dict):
            in_channels = {t: in_channels for t in self.node_types}
        self.in_channels, self.out_channels, self.heads, self.group = in_channels, out_channels, heads, group
        self.dst_node_types = {e[2] for e in self.edge_types}
        D = out_channels // heads
        self.kqv_lin = nn.Module()
        self.kqv_lin.lins = nn.ModuleDict({t: nn.Linear(in_channels[t], 3 * out_channels) for t in self.node_types})
        self.out_lin = nn.Module()
        self.out_lin.lins = nn.ModuleDict({t: nn.Linear(out_channels, out_channels) for t in self.node_types})
        R = len(self.edge_types)
        self.k_rel = nn.Module()
        self.k_rel.weight = nn.Parameter(torch.empty(heads * R, D, D))
        self.v_rel = nn.Module()
        self.v_rel.weight = nn.Parameter(torch.empty(heads * R, D, D))
        self.skip = nn.ParameterDict({t: nn.Parameter(torch.ones(1)) for t in self.node_types})
        self.p_rel = nn.ParameterDict({"__".join(e): nn.Parameter(torch.ones(1, heads)) for e in self.edge_types})
        for w in (self.k_rel.weight, self.v_rel.weight):
            nn.init.xavier_uniform_(w.view(heads * R * D, D))
        self._plan_cache = {}
        self._host_cache = {}

    def _plan(self, edge_index_dict, sizes, device, want, dst_range=None):
        used = [et for et in self.edge_types if et in edge_index_dict and et[2] in want]
        key = (id(edge_index_dict), str(device), tuple(sorted(sizes.items())), tuple(used),
               None if dst_range is None else tuple(sorted(dst_range.items())))
        hit = self._plan_cache.get(key)
        if hit is None:
            hit = hgt_plan(edge_index_dict, self.edge_types, sizes, device, used, dst_range)
            if len(self._plan_cache) > 8:
                self._plan_cache.clear()
            self._plan_cache[key] = hit
        return hit

    def _skip_alpha(self, t: str) -> float:
        p = self.skip[t]
        key = (p.data_ptr(), p._version)
        hit = self._host_cache.get(t)
        if hit is None or hit[0] != key:
            hit = (key, float(torch.sigmoid(p.detach()).item()))       # one sync per parameter version
            self._host_cache[t] = hit
        return hit[1]

    def _relation_matrices(self, r: int, et) -> Tuple[torch.Tensor, torch.Tensor]:
        """[128,128] matrices (nn.Linear layout) applying the per-head [D,D] relation matrices of edge type r
        to K (scaled by p_rel[h]/sqrt(D)) and to V:  k' = k @ blockdiag(A_h)  ==  linear(k, blockdiag(A_h)^T)."""
        H, R = self.heads, len(self.edge_types)
        D = self.out_channels // H
        idx = torch.arange(H, device=self.k_rel.weight.device) * R + r
        pr = self.p_rel["__".join(et)].detach().view(H, 1, 1) / math.sqrt(D)
        wk = torch.block_diag(*(self.k_rel.weight.detach()[idx] * pr).unbind(0)).t()
        wv = torch.block_diag(*self.v_rel.weight.detach()[idx].unbind(0)).t()
        return wk, wv

    def _composite_projection(self, t: str, plan: dict):
        """One weight per node type producing  q | k'_0 v'_0 | k'_1 v'_1 ...  in a single GEMM: the relation
        transform of every used edge type leaving t is composed with the K / V projection
        (k' = (x W_k^T + b_k) B_r  =  x (B_r^T W_k)^T + B_r^T b_k).  Rebuilt only when a parameter changes."""
        F = self.out_channels
        lin = self.kqv_lin.lins[t]
        rels = [et for et in plan["used"] if et[0] == t]
        srcs = [lin.weight, lin.bias, self.k_rel.weight, self.v_rel.weight] + [self.p_rel["__".join(et)] for et in rels]

        def build():
            W, b = lin.weight.detach(), lin.bias.detach()
            Wk, Wq, Wv = W[0:F], W[F:2 * F], W[2 * F:3 * F]
            bk, bq, bv = b[0:F], b[F:2 * F], b[2 * F:3 * F]
            ws, bs = [Wq], [bq]
            for et in rels:
                mk, mv = self._relation_matrices(self.edge_types.index(et), et)
                ws += [mk @ Wk, mv @ Wv]
                bs += [mk @ bk, mv @ bv]
            return torch.cat(ws, 0).contiguous(), torch.cat(bs, 0).contiguous()
        return _cached(self, ("proj", t, tuple(rels)), srcs, build)

    def _relation_blocks_train(self):
        """[R,128,128] block-diagonal relation matrices (nn.Linear layout, k' = linear(k, MK[r])) for the keys (p_rel/sqrt(D)
        folded in) and the values, assembled from the live parameters in a handful of batched torch ops (parameter space)."""
        H, R, D = self.heads, len(self.edge_types), self.out_channels // self.heads
        dev = self.k_rel.weight.device
        pr = torch.cat([self.p_rel["__".join(et)] for et in self.edge_types], dim=0) / math.sqrt(D)        # [R,H]
        kr = self.k_rel.weight.view(H, R, D, D) * pr.t().reshape(H, R, 1, 1)
        vr = self.v_rel.weight.view(H, R, D, D)
        out = []
        for blocks in (kr, vr):
            # m[r, h, b, h, a] = blocks[h, r, a, b]  (transposed block on the diagonal: the nn.Linear layout of x @ blockdiag)
            m = _put_diag(H, blocks.transpose(-1, -2))
            out.append(m.reshape(R, H * D, H * D))
        return out[0], out[1]

    def _composite_projection_train(self, t: str, plan: dict, mk_all, mv_all):
        """The composite projection of ``_composite_projection`` from the LIVE parameters: two batched [R_t,128,128] x
        [128,in] products and one concatenation per node type (parameter space, O(parameters) — not O(nodes)); gradients
        reach kqv_lin, k_rel, v_rel and p_rel through torch's autograd."""
        F = self.out_channels
        lin = self.kqv_lin.lins[t]
        W, b = lin.weight, lin.bias
        Wk, Wq, Wv = W[0:F], W[F:2 * F], W[2 * F:3 * F]
        bk, bq, bv = b[0:F], b[F:2 * F], b[2 * F:3 * F]
        rels = [self.edge_types.index(e) for e in plan["used"] if e[0] == t]
        if not rels:
            return Wq, bq
        idx = self.__dict__.setdefault("_rel_idx", {}).get((t, tuple(rels)))
        if idx is None or idx.device != W.device:
            idx = torch.tensor(rels, dtype=torch.int64, device=W.device)
            self.__dict__["_rel_idx"][(t, tuple(rels))] = idx
        mk, mv = mk_all.index_select(0, idx), mv_all.index_select(0, idx)                       # [R_t,128,128]
        wkv = torch.stack([torch.matmul(mk, Wk), torch.matmul(mv, Wv)], dim=1).reshape(-1, W.shape[1])      # rows: (r, k|v, 128)
        bkv = torch.stack([torch.matmul(mk, bk), torch.matmul(mv, bv)], dim=1).reshape(-1)
        return torch.cat([Wq, wkv], 0), torch.cat([bq, bkv], 0)

    def _composite_all_train(self, types, plan: dict):
        """Every projected node type's composite weight at once (all types share the input width): the rows
        q | k'_r v'_r ... of all types as ONE matrix [rows, in] in the order of the flat projection buffer, built from the live
        parameters with two batched products over the used relations (bmm of the [Ru,128,128] relation blocks with the
        source types' K / V weights) instead of four products, six slices and two concatenations per node type -- and,
        above all, with a backward graph of ~25 nodes instead of ~25 per type.  -> (W [rows,in], b [rows], first row per type)."""
        F = self.out_channels
        mk_all, mv_all = self._relation_blocks_train()
        key = ("all", tuple(types), tuple(plan["used"]))
        meta = self.__dict__.setdefault("_rel_idx", {}).get(key)
        dev = mk_all.device
        if meta is None or meta[0].device != dev:
            rel_idx, src_slot, perm, offs, row = [], [], [], [], 0
            for i, t in enumerate(types):
                offs.append(row)
                perm.extend(range(i * F, (i + 1) * F))
                row += F
                for e in plan["used"]:
                    if e[0] == t:
                        g = len(rel_idx)
                        rel_idx.append(self.edge_types.index(e))
                        src_slot.append(i)
                        perm.extend(range(len(types) * F + g * 2 * F, len(types) * F + (g + 1) * 2 * F))
                        row += 2 * F
            offs.append(row)
            # source type of each used relation as a 0/1 selection matrix: several relations leave the same type, and the
            # backward of an index_select with repeated indices adds with atomics (run-to-run different low bits); as a
            # product with a constant matrix the gradient is a plain GEMM
            sel = torch.zeros(len(rel_idx), len(types), device=dev)
            if rel_idx:
                sel[torch.arange(len(rel_idx)), torch.tensor(src_slot)] = 1.0
            meta = (torch.tensor(rel_idx, dtype=torch.int64, device=dev), sel, torch.tensor(perm, dtype=torch.int64, device=dev), offs)
            self.__dict__["_rel_idx"][key] = meta
        rel_idx, sel, perm, offs = meta
        lins = [self.kqv_lin.lins[t] for t in types]
        NT, cin = len(types), lins[0].weight.shape[1]
        Wk, Wq, Wv = torch.stack([l.weight for l in lins]).view(NT, 3, F, cin).unbind(1)              # [NT,F,in] each
        bk, bq, bv = torch.stack([l.bias for l in lins]).view(NT, 3, F).unbind(1)
        if rel_idx.numel() == 0:
            return Wq.reshape(NT * F, cin).index_select(0, perm), bq.reshape(-1).index_select(0, perm), offs
        mk, mv = mk_all.index_select(0, rel_idx), mv_all.index_select(0, rel_idx)                       # [Ru,128,128]
        Ru = int(rel_idx.numel())

        def of_source(v):                                              # [NT, ...] -> [Ru, ...]: row of each relation's source type
            return torch.matmul(sel, v.reshape(NT, -1)).view(Ru, *v.shape[1:])
        K = torch.bmm(mk, of_source(Wk))
        V = torch.bmm(mv, of_source(Wv))
        bK = torch.bmm(mk, of_source(bk).unsqueeze(-1)).squeeze(-1)
        bV = torch.bmm(mv, of_source(bv).unsqueeze(-1)).squeeze(-1)
        big_w = torch.cat([Wq.reshape(NT * F, cin), torch.stack([K, V], dim=1).reshape(-1, cin)], 0).index_select(0, perm)
        big_b = torch.cat([bq.reshape(-1), torch.stack([bK, bV], dim=1).reshape(-1)], 0).index_select(0, perm)
        return big_w, big_b, offs

    def _composite_all_hip(self, types, plan: dict):
        """``_composite_all_train`` as ONE autograd node over two kernels (autograd._HgtComposite, csrc/hgt_params.hip): the composite
        rows q | k'_r v'_r ... of all projected types from the live kqv_lin / k_rel / v_rel / p_rel, their gradients from the rows'
        gradient.  -> (W [rows,in], b [rows], first row per type)."""
        F = self.out_channels
        lins = [self.kqv_lin.lins[t] for t in types]
        cin = lins[0].weight.shape[1]
        dev = self.k_rel.weight.device
        key = ("hip", tuple(types), tuple(plan["used"]))
        meta = self.__dict__.setdefault("_rel_idx", {}).get(key)
        if meta is None or meta["rel_r"].device != dev:
            rel_r, rel_src, rel_row, type_row, offs, row = [], [], [], [], [], 0
            for i, t in enumerate(types):
                offs.append(row)
                type_row.append(row)
                row += F
                for e in plan["used"]:
                    if e[0] == t:
                        rel_r.append(self.edge_types.index(e))
                        rel_src.append(i)
                        rel_row.append(row)
                        row += 2 * F
            offs.append(row)
            i32 = lambda v: torch.tensor(v if v else [0], dtype=torch.int32, device=dev)
            meta = dict(rel_r=i32(rel_r), rel_src=i32(rel_src), rel_row=i32(rel_row), type_row=i32(type_row), n_rel=len(rel_r), n_types=len(types),
                        n_edge_types=len(self.edge_types), rows=row, cin=int(cin), F=F, H=self.heads, offs=offs)
            self.__dict__["_rel_idx"][key] = meta
        ps = [self.p_rel["__".join(et)] for et in self.edge_types]
        big_w, big_b = ag.hgt_composite(meta, self.k_rel.weight, self.v_rel.weight, [l.weight for l in lins], [l.bias for l in lins], ps)
        return big_w, big_b, meta["offs"]

    def _forward_train(self, x_dict, edge_index_dict, needed_types=None, shard=None):
        """Differentiated pass on the same flat projection layout as inference: one composite GEMM per node type writes
        q | k'_r v'_r ... into the flat buffer (ag.hgt_project), edge attention of all destination types reads queries, keys
        and values from it and returns ONE gradient buffer (ag.hgt_attention_flat), then GELU, output projection and the
        sigmoid(skip) gate.  The composite weights are rebuilt from the live parameters each call.

        ``shard`` = (rank, world, group): the destination-partitioned conv of the data-parallel steps.  The projections (one
        GEMM per node type over all nodes) run on every rank; the edge attention -- every KG edge -- and the output stage only
        for this rank's block of each destination type; the blocks of all ranks and types travel in ONE all-gather whose
        backward is the reverse exchange (all-reduce of the gathered rows' gradients, own block kept: parallel.py).  A rank's
        backward pass then holds the gradient contributions of the edges INTO its blocks only; they are partial sums of the
        parameter gradients, which the step's gradient all-reduce completes like every other data-parallel partial."""
        F, H = self.out_channels, self.heads
        dev = next(iter(x_dict.values())).device
        sizes = {t: int(x.shape[0]) for t, x in x_dict.items()}
        want = set(self.dst_node_types if needed_types is None else needed_types)
        dst_range = None
        if shard is not None and shard[1] > 1:
            from .parallel import shard_range
            dst_range = {t: shard_range(sizes[t], shard[0], shard[1]) for t in sizes}
        plan = self._plan(edge_index_dict, sizes, dev, want, dst_range)
        types = [t for t in x_dict if not ((plan["nrel"][t] == 0 and t not in want) or sizes[t] == 0)]
        layout = [(plan["base"][t], sizes[t], plan["width"][t]) for t in types]
        spec = dict(zip(types, layout))
        xs = [x_dict[t].float() for t in types]
        if types and len({x.shape[1] for x in xs}) == 1 and self.batched_weights:
            # (_composite_all_train assembles the same rows with ~30 torch ops and torch's autograd, rounds 2-3: the tests' second reference)
            big_w, big_b, offs = self._composite_all_hip(types, plan)
            flat = ag.hgt_project_rows(layout, plan["total_floats"], _state["precision"], xs, big_w, big_b, offs)
        else:                                                         # node types of different input width: one weight each
            mk_all, mv_all = self._relation_blocks_train()
            ws, bs = zip(*(self._composite_projection_train(t, plan, mk_all, mv_all) for t in types)) if types else ((), ())
            flat = ag.hgt_project(layout, plan["total_floats"], _state["precision"], xs, list(ws), list(bs))
        rng = {t: ((0, sizes[t]) if dst_range is None else dst_range[t]) for t in sizes}
        dst_all = [t for t in self.node_types if t in self.dst_node_types and t in x_dict and t in want and sizes[t] > 0]
        dst_types = [t for t in dst_all if rng[t][1] > rng[t][0]]        # (a rank's block of a small type may be empty)

        def block(t):                                                  # this rank's rows of type t inside the flat buffer
            off, _, width = spec[t]
            return (off + rng[t][0] * width, rng[t][1] - rng[t][0], width)
        rows16 = _state["precision"] == "bf16"
        pres = ag.hgt_attention_flat(flat, H, [plan["per_dst"][t] for t in dst_types], [block(t) for t in dst_types], rows16=rows16) if dst_types else ()
        out = {}
        # the destination types' output stages (GELU, out_lin, gated residual: a chain of small launches per type, and as many again
        # in the backward pass, which autograd runs on the stream of the forward): parallel branches when the pass is being captured
        # as a hipGraph (ag.type_lanes), a plain loop otherwise
        lanes = ag.type_lanes(flat, len(dst_types) if dst_range is None else 1, chains=True)
        for i, (t, pre) in enumerate(zip(dst_types, pres)):
            lin = self.out_lin.lins[t]
            with lanes.lane(i):
                o = _linT(ag.activation(pre, "gelu"), lin.weight, lin.bias)
                xr = x_dict[t].float()
                out[t] = ag.gated_residual(o, xr[rng[t][0]:rng[t][1]] if dst_range is not None else xr, self.skip[t]) if x_dict[t].shape[-1] == F else o
        lanes.join(*out.values())
        if dst_range is None:
            return out
        # one exchange step for every destination type (as the inference conv): each rank's blocks back to back, gathered with a
        # gradient, re-cut per type
        from .parallel import all_gather_rows_grad, shard_range
        rank, world, group = shard
        per_rank = [[shard_range(sizes[t], r, world)[1] - shard_range(sizes[t], r, world)[0] for t in dst_all] for r in range(world)]
        parts = [out[t] if t in out else torch.zeros((0, F), dtype=torch.float32, device=dev) for t in dst_all]
        local = torch.cat(parts, dim=0) if parts else torch.zeros(0, F, device=dev)
        totals = [sum(v) for v in per_rank]
        full = all_gather_rows_grad(local, sum(totals), rank, world, group, sizes=totals)
        res, start = {}, [sum(totals[:r]) for r in range(world)]
        for ti, t in enumerate(dst_all):
            cut = []
            for r in range(world):
                off = start[r] + sum(per_rank[r][:ti])
                cut.append(full[off: off + per_rank[r][ti]])
            res[t] = torch.cat(cut, dim=0)
        return res

    def _forward_grouped(self, x_dict, plan, sizes, want, dev):
        """Inference conv with the per-node-type layers grouped: the composite projections of all node types are ONE launch
        (mdg_linear_grouped over the stacked inputs and the stacked composite weights) and so are the output projections with
        their sigmoid(skip)-gated residuals -- 4 launches instead of ~40 per conv, the same tiles and arithmetic (bit-identical
        rows).  The outputs of all destination types live in one buffer, handed to the next conv as it stands.  None when
        the node types differ in input width."""
        F = self.out_channels
        types_p = [t for t in x_dict if not (plan["nrel"][t] == 0 and t not in want) and sizes[t] > 0]
        if not types_p or len({x_dict[t].shape[1] for t in types_p}) != 1:
            return None
        cin = x_dict[types_p[0]].shape[1]
        if cin % 4:
            return None
        # the stacked input rows: the previous grouped conv's output buffer as it stands, else one concatenation (kept for
        # constant inputs: the KG's node features)
        flat = getattr(x_dict, "flat", None)
        if flat is not None and flat[1] == tuple(types_p) and flat[0].shape[1] == cin:
            X_all = flat[0]
        else:
            xs = [x_dict[t] for t in types_p]
            key = tuple((x.data_ptr(), x._version, tuple(x.shape)) for x in xs)
            hit = self.__dict__.get("_xcat")
            if hit is None or hit[0] != key:
                hit = self.__dict__["_xcat"] = (key, torch.cat([x.float() for x in xs], 0).contiguous(), xs)
            X_all = hit[1]
        row0, r = {}, 0
        for t in types_p:
            row0[t] = r
            r += sizes[t]
        # stacked composite weights (rebuilt only when a parameter changes) and the tile tables (per plan)
        srcs = [p for t in types_p for p in (self.kqv_lin.lins[t].weight, self.kqv_lin.lins[t].bias)] + [self.k_rel.weight, self.v_rel.weight] + \
               [self.p_rel["__".join(et)] for et in plan["used"]]

        def build_w():
            wb = [self._composite_projection(t, plan) for t in types_p]
            return torch.cat([w for w, _ in wb], 0).contiguous(), torch.cat([b for _, b in wb], 0).contiguous()
        W_all, b_all = _cached(self, ("proj_all", tuple(types_p), tuple(plan["used"])), srcs, build_w)
        dst_types = [t for t in self.node_types if t in self.dst_node_types and t in x_dict and t in want and sizes[t] > 0]
        gated = cin == F
        tkey = ("tables", tuple(types_p), tuple(dst_types), tuple(sizes[t] for t in types_p), cin)
        tabs = plan.get(tkey)
        if tabs is None:
            n0, groups = 0, []
            for t in types_p:
                groups.append(dict(m_base=row0[t], rows=sizes[t], n_base=n0, n=plan["width"][t], y_off=plan["base"][t], ldy=plan["width"][t]))
                n0 += plan["width"][t]
            d0, drow = 0, {}
            for t in dst_types:
                drow[t] = d0
                d0 += sizes[t]
            # the destination types' attention plans concatenated: destinations, items and edges numbered across the types
            pieces = {k: [] for k in ("q_off", "col", "item_dst", "item_begin", "item_end", "item_ptr")}
            n_it = n_e = 0
            for t in dst_types:
                pd = plan["per_dst"][t]
                pieces["q_off"].append(plan["base"][t] + torch.arange(sizes[t], device=dev, dtype=torch.int64) * plan["width"][t])
                pieces["col"].append(pd["col"])
                pieces["item_dst"].append(pd["item_dst"] + drow[t])
                pieces["item_begin"].append(pd["item_begin"] + n_e)
                pieces["item_end"].append(pd["item_end"] + n_e)
                pieces["item_ptr"].append(pd["item_ptr"][:-1] + n_it)
                n_it += int(pd["item_dst"].numel())
                n_e += int(pd["col"].numel())
            pieces["item_ptr"].append(torch.tensor([n_it], dtype=torch.int64, device=dev))
            att = {k: torch.cat(v).contiguous() for k, v in pieces.items()} if dst_types else None
            tabs = plan[tkey] = {"proj": ops.group_tile_table(groups, dev), "drow": drow, "rows_dst": d0, "out": None, "alphas": None, "att": att}
        buf = torch.empty(max(plan["total_floats"], 128), dtype=torch.float32, device=dev)
        ops.linear_grouped(X_all, W_all, b_all, tabs["proj"], buf, precision=_state["precision"])
        drow, rows_dst = tabs["drow"], tabs["rows_dst"]
        agg_all = torch.empty((rows_dst, F), dtype=torch.float32, device=dev)
        if tabs["att"] is not None:
            probe = self.__dict__.get("_attention_probe")            # bench.py: HIP events around this launch + its edge count
            if probe is not None:
                probe["start"].record(torch.cuda.current_stream(dev))
            ops.hgt_attention_rows(buf, tabs["att"], self.heads, agg_all, apply_gelu=True)
            if probe is not None:
                probe["end"].record(torch.cuda.current_stream(dev))
                probe["edges"], probe["dst_rows"], probe["buffer_floats"] = int(tabs["att"]["col"].numel()), int(rows_dst), int(plan["total_floats"])
        alphas = tuple(self._skip_alpha(t) if gated else 1.0 for t in dst_types)
        if tabs["out"] is None or tabs["alphas"] != alphas:             # the gates sit in the table: rebuilt when a skip parameter changes
            groups = [dict(m_base=drow[t], rows=sizes[t], n_base=F * i, n=F, y_off=drow[t] * F, ldy=F, alpha=a, beta=1.0 - a,
                           **({"res_off": row0[t] * cin, "ldr": cin} if gated else {}))
                      for i, (t, a) in enumerate(zip(dst_types, alphas))]
            tabs["out"], tabs["alphas"] = ops.group_tile_table(groups, dev), alphas
        osrc = [p for t in dst_types for p in (self.out_lin.lins[t].weight, self.out_lin.lins[t].bias)]
        Wo, bo = _cached(self, ("out_all", tuple(dst_types)), osrc,
                         lambda: (torch.cat([self.out_lin.lins[t].weight.detach() for t in dst_types], 0).contiguous(),
                                  torch.cat([self.out_lin.lins[t].bias.detach() for t in dst_types], 0).contiguous()))
        out_all = torch.empty((rows_dst, F), dtype=torch.float32, device=dev)
        ops.linear_grouped(agg_all, Wo, bo, tabs["out"], out_all, residual=X_all if gated else None, precision=_state["precision"])
        out = _FlatRows({t: out_all[drow[t]: drow[t] + sizes[t]] for t in dst_types})
        for t in self.node_types:                                       # destination types without nodes: empty rows, as the per-type path returns
            if t in self.dst_node_types and t in x_dict and t in want and sizes[t] == 0:
                out[t] = torch.zeros((0, F), dtype=torch.float32, device=dev)
        out.flat = (out_all, tuple(dst_types))
        return out

    def forward(self, x_dict, edge_index_dict, needed_types=None, shard=None):
        """``needed_types`` (extension): compute only these destination node types (the encoder reads
        ['drug'] of the LAST conv only, models.py:729); default = every destination type, as PyG does.
        ``shard`` = (rank, world, group) (extension, inference): destination-partitioned conv for the multi-GPU encode -- the
        K/Q/V projections (one GEMM per node type over all nodes) run on every rank, the edge attention (the bulk: every KG edge)
        and the output projection only for this rank's block of each destination type, then the blocks of all ranks and types
        travel in ONE all-gather.  Rows come out bit-identical to the unpartitioned conv."""
        if _train_path(self) or ag.needs_grad(*x_dict.values()):
            return self._forward_train(x_dict, edge_index_dict, needed_types, shard)
        F = self.out_channels
        dev = next(iter(x_dict.values())).device
        sizes = {t: int(x.shape[0]) for t, x in x_dict.items()}
        want = set(self.dst_node_types if needed_types is None else needed_types)
        dst_range = None
        if shard is not None and shard[1] > 1:
            from .parallel import shard_range
            dst_range = {t: shard_range(sizes[t], shard[0], shard[1]) for t in sizes}
        plan = self._plan(edge_index_dict, sizes, dev, want, dst_range)
        if dst_range is None:
            out = self._forward_grouped(x_dict, plan, sizes, want, dev)
            if out is not None:
                return out
        buf = torch.empty(max(plan["total_floats"], 128), dtype=torch.float32, device=dev)
        proj = {}
        for t, x in x_dict.items():
            if plan["nrel"][t] == 0 and t not in want:
                continue
            w, b = self._composite_projection(t, plan)
            n_t, wd = sizes[t], plan["width"][t]
            proj[t] = buf[plan["base"][t]: plan["base"][t] + n_t * wd].view(n_t, wd)
            if n_t:
                _lin(x.float(), w, b, out=proj[t])
        kv = buf.view(-1, 128)                             # key rows / value rows, addressed by the plan's col
        out = {}
        for t in self.node_types:
            if t not in self.dst_node_types or t not in x_dict or t not in want:
                continue
            lo, hi = (0, sizes[t]) if dst_range is None else dst_range[t]
            agg = ops.hgt_attention(proj[t][lo:hi, 0:F], kv, plan["per_dst"][t], self.heads, apply_gelu=True)
            lin = self.out_lin.lins[t]
            if x_dict[t].shape[-1] == F:
                a = self._skip_alpha(t)
                out[t] = _lin(agg, lin.weight, lin.bias, alpha=a, residual=x_dict[t][lo:hi].float(), beta=1.0 - a)
            else:
                out[t] = _lin(agg, lin.weight, lin.bias)
        if dst_range is None:
            return out
        # one exchange step for every destination type: each rank's blocks back to back, gathered, re-cut per type
        from .parallel import all_gather_rows, shard_range
        rank, world, group = shard
        types = list(out.keys())
        per_rank = [[shard_range(sizes[t], r, world)[1] - shard_range(sizes[t], r, world)[0] for t in types] for r in range(world)]
        local = torch.cat([out[t] for t in types], dim=0) if types else torch.zeros(0, F, device=dev)
        totals = [sum(v) for v in per_rank]
        full = all_gather_rows(local.contiguous(), sum(totals), rank, world, group, sizes=totals)
        res, start = {}, [sum(totals[:r]) for r in range(world)]
        for ti, t in enumerate(types):
            parts = []
            for r in range(world):
                off = start[r] + sum(per_rank[r][:ti])
                parts.append(full[off: off + per_rank[r][ti]])
            res[t] = torch.cat(parts, dim=0)
        return res


class HGT(nn.Module):
    """madrigal/models/models.py:71-96: HGTConv stack (ReLU only between convs i >= 1 and the last) + per-type Linear."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, num_heads, metadata, group='sum'):
        super().__init__()
        self.convs = nn.ModuleList([HGTConv(in_channels, hidden_channels, metadata, num_heads, group=group)])
        for _ in range(num_layers - 1):
            self.convs.append(HGTConv(hidden_channels, hidden_channels, metadata, num_heads, group=group))
        self.lin_dict = nn.ModuleDict({t: nn.Linear(hidden_channels, out_channels) for t in metadata[0]})

    def forward(self, x_dict, edge_index_dict, only_types=None, shard=None):
        """``only_types`` (extension): node types whose output the caller reads; the last conv and the final
        Linear are then restricted to them (same values for those types).  ``shard`` = (rank, world, group) (extension,
        inference): every conv runs destination-partitioned over the ranks (HGTConv.forward)."""
        last = len(self.convs) - 1
        kw = {} if shard is None else {"shard": shard}
        out = self.convs[0](x_dict, edge_index_dict, needed_types=only_types if last == 0 else None, **kw)
        for i in range(1, len(self.convs)):
            out = self.convs[i](out, edge_index_dict, needed_types=only_types if i == last else None, **kw)
            if i < last:
                out = {t: (ag.activation(x, "relu") if x.requires_grad else torch.relu_(x)) for t, x in out.items()}
        if _train_path(self) or ag.needs_grad(*out.values()):
            return {t: _linT(x, self.lin_dict[t].weight, self.lin_dict[t].bias) for t, x in out.items()}
        return {t: _lin(x, self.lin_dict[t].weight, self.lin_dict[t].bias) for t, x in out.items()}


class HAN(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("HAN (madrigal/models/models.py:41-67) is not used by any shipped config")


class RGCN(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("RGCN (madrigal/models/models.py:99-117) is not used by any shipped config")


# ------------------------------------------------------------------------------------- chemCPA tx encoder
class ChemCPAMLP(nn.Module):
    """chemCPA ``MLP`` (madrigal/chemcpa/chemCPA/model.py:161-231): (Linear, BatchNorm1d, ReLU) x (n-1) + Linear."""

    def __init__(self, sizes, batch_norm=True, last_layer_act="linear", append_layer_width=None, append_layer_position=None):
        super().__init__()
        if append_layer_width:
            raise NotImplementedError("append_layer_width is not used by Madrigal (models.py:285,324)")
        if last_layer_act != "linear":
            raise NotImplementedError(last_layer_act)
        layers = []
        for s in range(len(sizes) - 1):
            layers.append(nn.Linear(sizes[s], sizes[s + 1]))
            if batch_norm and s < len(sizes) - 2:
                layers.append(nn.BatchNorm1d(sizes[s + 1]))
            layers.append(nn.ReLU())
        layers = layers[:-1]
        # indices must match the reference's Sequential (Linear 3k, BN 3k+1, ReLU 3k+2)
        self.network = nn.Sequential(*layers)
        self.activation = last_layer_act

    def forward(self, x, residual=None):
        mods = list(self.network)
        if _train_path(self) or ag.needs_grad(x, residual):
            y = _run_sequential_train(self.network, x)
            return y if residual is None else ag.add(y, residual)
        if residual is None:
            return _run_sequential(self.network, x)
        # fuse `+ residual` into the last Linear
        x = _run_sequential(nn.Sequential(*mods[:-1]), x) if len(mods) > 1 else x
        return _lin(x, mods[-1].weight, mods[-1].bias, residual=residual)


class TxAdaptingComPert(nn.Module):
    """Forward (``predict``) part of chemCPA's ComPert autoencoder as Madrigal uses it
    (madrigal/chemcpa/chemCPA/model.py:290-519 constructor, :655-712 predict).  ``use_drugs=False`` only
    (configs/chemcpa/chemcpa_finetune_configs.yaml:18); training (``update``), adversaries, dosers and
    optimisers are outside the path."""

    def __init__(self, num_genes: int, num_drugs: int, covariate_names_unique: Dict[str, List[str]], seed=0, patience=5,
                 doser_type="logsigm", decoder_activation="linear", hparams="", drug_embeddings=None,
                 append_layer_width=None, use_drugs=True, disable_adv=False, **kwargs):
        super().__init__()
        if use_drugs:
            raise NotImplementedError("use_drugs=True (drug-embedding term of chemCPA) is not used by Madrigal's shipped configs")
        if not isinstance(hparams, dict):
            raise ValueError("hparams must be a dict (autoencoder_width, autoencoder_depth, dim)")
        self.num_genes, self.num_drugs, self.covariate_names_unique = num_genes, num_drugs, covariate_names_unique
        self.num_covariates = [len(v) for v in covariate_names_unique.values()]
        assert 0 not in self.num_covariates
        self.hparams, self.use_drugs, self.disable_adv = hparams, use_drugs, disable_adv
        w, dpt, dim = hparams["autoencoder_width"], hparams["autoencoder_depth"], hparams["dim"]
        self.encoder = ChemCPAMLP([num_genes] + [w] * dpt + [dim])
        self.decoder = ChemCPAMLP([dim] + [w] * dpt + [num_genes * 2], last_layer_act=decoder_activation)
        self.adversary_drugs = self.drug_embeddings = self.drug_embedding_encoder = self.dosers = None
        self.covariates_embeddings = nn.ModuleList([nn.Embedding(n, dim) for n in self.num_covariates])

    def predict(self, genes, drugs=None, drugs_idx=None, dosages=None, covariates=None, return_latent_basal=False,
                return_latent_treated=False, compute_reconstruction=True, covariate_indices=None):
        """Same outputs as the reference.  ``compute_reconstruction=False`` skips the decoder (62 % of
        this block's flops) whose output Madrigal discards (models.py:761) and returns None in its slot;
        ``covariate_indices`` may replace the one-hot ``covariates`` (the reference only takes argmax)."""
        assert (drugs is not None) or (drugs_idx is not None and dosages is not None)
        if covariate_indices is None:
            covariate_indices = [c.argmax(1) for c in covariates]
        train = _train_path(self) or ag.needs_grad(genes)
        if train and compute_reconstruction:
            raise NotImplementedError("the reconstruction head is not differentiated on the HIP path (Madrigal discards it, "
                                      "models.py:761): call predict(compute_reconstruction=False)")
        emb_sum = None
        for emb, idx in zip(self.covariates_embeddings, covariate_indices):
            idx = idx.to(emb.weight.device)
            e = ag.gather_rows(emb.weight, idx) if train else emb.weight.detach().index_select(0, idx)
            emb_sum = e if emb_sum is None else (ag.add(emb_sum, e) if train else emb_sum + e)
        last_emb = e
        need_basal = return_latent_basal or emb_sum is None
        if need_basal:
            latent_basal = self.encoder(genes)
            latent_treated = latent_basal if emb_sum is None else (ag.add(latent_basal, emb_sum) if train else latent_basal + emb_sum)
        else:
            latent_basal = None
            latent_treated = self.encoder(genes, residual=emb_sum)          # + cov embedding fused in the epilogue
        recon = None
        if compute_reconstruction:
            g = self.decoder(latent_treated)
            dim = g.size(1) // 2
            recon = torch.cat([g[:, :dim], torch.nn.functional.softplus(g[:, dim:])], dim=1)
        out = (recon, last_emb)
        if return_latent_basal:
            out += (latent_basal,)
        if return_latent_treated:
            out += (latent_treated,)
        return out


# ------------------------------------------------------------------------------------- encoder factories
def _ckpt_dir():
    return os.getenv("ENCODER_CKPT_DIR", "")


def get_str_encoder(str_encoder_name, str_encoder_hparams, embed_dim, atom_dim, use_modality_pretrain=True):
    """madrigal/models/models.py:213-232."""
    if str_encoder_name != 'gin':
        raise NotImplementedError(f"str encoder {str_encoder_name!r}: only 'gin' is on the path (GAT is unused by shipped configs)")
    hp = str_encoder_hparams
    enc = GraphIsomorphismNetwork(input_dim=atom_dim, hidden_dims=hp['gin_hidden_dims'] + [embed_dim],
                                  edge_input_dim=hp['gin_edge_input_dim'], num_mlp_layer=hp['gin_num_mlp_layer'],
                                  eps=hp['gin_eps'], batch_norm=hp['gin_batch_norm'], activation=hp['gin_actn'],
                                  readout=hp['gin_readout'])
    if use_modality_pretrain:
        sd = torch.load(_ckpt_dir() + 'str/GIN_256x4_muv.pt', map_location='cpu')
        clean = {}
        for k, v in sd.items():
            if k.startswith('model.'):
                clean[k[len('model.'):]] = v
            elif k.startswith('layer'):
                clean[k] = v
        enc.load_state_dict(clean)
    return enc


def get_kg_encoder(kg_encoder_name, kg_encoder_hparams, embed_dim, all_kg_data, use_modality_pretrain=True):
    """madrigal/models/models.py:235-247."""
    if 'hgt' not in kg_encoder_name:
        raise NotImplementedError(f"kg encoder {kg_encoder_name!r}: only HGT is on the path")
    hp = kg_encoder_hparams
    enc = HGT(in_channels=all_kg_data.x_dict['drug'].shape[1], hidden_channels=hp['hgt_hidden_dim'], out_channels=embed_dim,
              num_layers=hp['hgt_num_layers'], num_heads=hp['hgt_att_heads'], metadata=all_kg_data.metadata(),
              group=hp['hgt_group'])
    if use_modality_pretrain:
        enc.load_state_dict(torch.load(_ckpt_dir() + 'kg/hgt_best.pt', map_location='cpu'))
    return enc


def get_tabular_mod_encoder(mod_encoder_name, mod_encoder_hparams, embed_dim, use_modality_pretrain=True, mod="cv"):
    """madrigal/models/models.py:250-259."""
    assert mod_encoder_name == 'mlp'
    hp = mod_encoder_hparams
    enc = MLPEncoder(hp['cv_input_dim'], hp['cv_mlp_hidden_dims'], embed_dim, hp['cv_mlp_dropout'], hp['cv_mlp_norm'],
                     hp['cv_mlp_actn'], hp['cv_mlp_order'])
    if use_modality_pretrain:
        enc.load_state_dict(torch.load(_ckpt_dir() + f'{mod}/{mod}_model_ae.pt', map_location='cpu'))
    return enc


def get_tx_encoder(tx_encoder_name, tx_encoder_hparams, embed_dim, use_modality_pretrain=True):
    """madrigal/models/models.py:262-348.  With ``use_drugs=False`` the frozen drug-embedding table the
    reference builds from two external data files (:271-275) is never used, so it is not read here."""
    if tx_encoder_name == 'mlp':
        hp = tx_encoder_hparams
        return MLPEncoder(hp['tx_input_dim'], hp['tx_mlp_hidden_dims'], embed_dim, hp['tx_mlp_dropout'], hp['tx_mlp_norm'],
                          hp['tx_mlp_actn'], hp['tx_mlp_order'])
    if tx_encoder_name != 'chemcpa':
        raise NotImplementedError(tx_encoder_name)
    m = tx_encoder_hparams["model"]
    hparams, additional, use_drugs = m["hparams"], dict(m.get("additional_params", {})), m["use_drugs"]
    num_drugs = int(m.get("num_drugs", 0))
    if not use_modality_pretrain:
        enc = TxAdaptingComPert(num_genes=TX_INPUT_DIM, num_drugs=num_drugs, covariate_names_unique={"cell_iname": CELL_LINES_CAPITALIZED},
                                **additional, hparams=hparams, drug_embeddings=None, append_layer_width=None,
                                use_drugs=use_drugs, disable_adv=True)
        cell_lines = np.array([c.lower() for c in CELL_LINES_CAPITALIZED])
        return enc, cell_lines
    state_dict, _, cov_sds, model_config, _history = torch.load(_ckpt_dir() + f"tx/{m['pretrained_model_ckpt']}",
                                                                map_location='cpu', weights_only=False)
    assert model_config['use_drugs'] == use_drugs
    assert len(cov_sds) == 1
    for key in list(state_dict.keys()):
        if key.startswith("adversary_") or key == "drug_embeddings.weight":
            state_dict.pop(key)
    for k in list(additional):
        if k in model_config:
            additional[k] = model_config[k]
    enc = TxAdaptingComPert(num_genes=TX_INPUT_DIM, num_drugs=num_drugs, covariate_names_unique=model_config["covariate_names_unique"],
                            **additional, hparams=model_config["hparams"], drug_embeddings=None, append_layer_width=None,
                            use_drugs=use_drugs, disable_adv=True)
    enc.load_state_dict(state_dict, strict=False)
    for emb, sd in zip(enc.covariates_embeddings, cov_sds):
        emb.load_state_dict(sd)
    cell_lines = np.array([c.lower() for c in model_config["covariate_names_unique"]["cell_iname"]])
    return enc, cell_lines


# ------------------------------------------------------------------------------------- fusion
class TransformerFusion(nn.Module):
    """madrigal/models/models.py:352-455.  Holds a stock ``nn.TransformerEncoder`` / ``nn.MultiheadAttention``
    ONLY for their parameters (identical state_dict keys; ``transformer_encoder.layers[-1].self_attn`` stays
    a hook target, predict.py:643); the forward pass is mdg_linear / mdg_layernorm / mdg_fusion_attention /
    mdg_xattn_pool.  Works batch-major internally; per-drug results do not depend on ``batch_first``."""

    def __init__(self, embed_dim, num_tx_bottlenecks, transformer_num_layers, transformer_att_heads, transformer_head_dim,
                 transformer_ffn_dim, transformer_dropout=0.1, transformer_actn='relu', transformer_norm_first=False,
                 transformer_batch_first=True, transformer_agg='mean'):
        super().__init__()
        self.batch_first = transformer_batch_first
        self.norm_first = transformer_norm_first
        self.num_heads, self.head_dim = transformer_att_heads, transformer_head_dim
        self.latent_dim = transformer_head_dim * transformer_att_heads
        self.actn = transformer_actn
        self.num_tx_bottlenecks = num_tx_bottlenecks
        self.embed2latent = nn.Linear(embed_dim, self.latent_dim)
        layer = nn.TransformerEncoderLayer(d_model=self.latent_dim, nhead=transformer_att_heads, dim_feedforward=transformer_ffn_dim,
                                           dropout=transformer_dropout, activation=transformer_actn,
                                           norm_first=transformer_norm_first, batch_first=transformer_batch_first)
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=transformer_num_layers, enable_nested_tensor=False)
        self.latent2embed = nn.Linear(self.latent_dim, embed_dim)
        self.transformer_agg = transformer_agg
        if transformer_agg == 'x-attn':
            self.x_attn_kv_norm = nn.LayerNorm(self.latent_dim)
            self.x_attn_query_norm = nn.LayerNorm(self.latent_dim)
            self.x_attn_mha_layer = nn.MultiheadAttention(embed_dim=self.latent_dim, num_heads=transformer_att_heads,
                                                          dropout=transformer_dropout, batch_first=transformer_batch_first)
            self.x_attn_dropout = nn.Dropout(transformer_dropout)
            self.x_attn_query = nn.Parameter(torch.randn(1, self.latent_dim))
            kpm = torch.zeros(1, NUM_MODALITIES + num_tx_bottlenecks, dtype=torch.bool)
            if num_tx_bottlenecks > 0:           # only the bottleneck tokens are keys (models.py:382-385)
                kpm[:, :NUM_NON_TX_MODALITIES] = True
                kpm[:, -len(CELL_LINES):] = True
            self.x_attn_key_padding_mask = kpm
        self.last_attention_weights = None

    # ---- one transformer layer on a set of token rows (dense or compact) --------------------
    def _layer(self, L, h, attend, keep_rows=None):
        """``attend(qkv) -> attention output rows``; ``keep_rows`` (int64 index) prunes the rows that
        continue after the attention (last layer: only the pooled key tokens are ever read again)."""
        sa = L.self_attn

        def norm(x, ln, want_fp32=True):
            """LayerNorm whose kernel also writes the output as the packed operand of the dense block that consumes it
            (image None: fp32 mode / odd width -> the block packs its input itself).  ``want_fp32=False``:
            the fp32 rows are skipped when the image exists (pre-norm: only linear1 reads norm2's output)."""
            return ops.layernorm_packed(x, ln.weight, ln.bias, ln.eps, _state["precision"], want_fp32)

        def lin_of(x, img, w, b, rows=None, **kw):
            return _lin(x, w, b, **kw) if img is None else ops.linear_packed(img, x.shape[0] if rows is None else rows, w, b,
                                                                             precision=_state["precision"], **kw)

        def in_proj(a, img=None):
            """q | k | v rows.  When only ``keep_rows`` continue past the attention, only THEIR queries are ever used (a query
            row decides its own output row and nothing else): keys and values for all rows, queries for the kept rows only --
            the other rows' query slots stay unwritten and so do the outputs computed from them, which are dropped."""
            w, b = sa.in_proj_weight.detach(), sa.in_proj_bias.detach()
            if keep_rows is None or a.shape[0] < 4 * keep_rows.numel() // 3:
                return lin_of(a, img, w, b)
            d = w.shape[1]
            qkv = torch.empty((a.shape[0], 3 * d), dtype=torch.float32, device=a.device)
            lin_of(a, img, w[d:], b[d:], out=qkv[:, d:])
            qkv[:, :d].index_copy_(0, keep_rows, _lin(a.index_select(0, keep_rows), w[:d], b[:d]))
            return qkv
        if self.norm_first:
            a, a_img = norm(h, L.norm1)
            att = attend(in_proj(a, a_img), a)
            if keep_rows is not None:
                att, h = att.index_select(0, keep_rows), h.index_select(0, keep_rows)
            h = _lin(att, sa.out_proj.weight, sa.out_proj.bias, residual=h)
            f, f_img = norm(h, L.norm2, want_fp32=False)
            u = lin_of(f, f_img, L.linear1.weight, L.linear1.bias, rows=h.shape[0], act=self.actn)
            return _lin(u, L.linear2.weight, L.linear2.bias, residual=h)
        att = attend(in_proj(h), h)
        if keep_rows is not None:
            att, h = att.index_select(0, keep_rows), h.index_select(0, keep_rows)
        t = _lin(att, sa.out_proj.weight, sa.out_proj.bias, residual=h)
        h, h_img = norm(t, L.norm1)
        u = lin_of(h, h_img, L.linear1.weight, L.linear1.bias, act=self.actn)
        t = _lin(u, L.linear2.weight, L.linear2.bias, residual=h)
        return ops.layernorm(t, L.norm2.weight, L.norm2.bias, L.norm2.eps)

    def _layer_train(self, L, h, attend, keep_rows=None):
        """The same layer through the autograd nodes, with the layer's three dropouts and the attention-weight
        dropout active when the sub-modules are in training mode (nn.TransformerEncoderLayer._sa_block/_ff_block)."""
        sa = L.self_attn
        p_att = sa.dropout if sa.training else 0.0

        def sa_block(x, x_image=None):
            att = attend(_linT(x, sa.in_proj_weight, sa.in_proj_bias, x_image=x_image), p_att)
            return att

        # the three dropouts live inside their dense blocks (GEMM epilogue / activation pass; backward inside the gradient's packing
        # pass) and the residual adds inside the epilogues: x + dropout(linear(...)) is one node and one launch
        def ff_block(x, residual, x_image=None):
            u = _linT_drop(x, L.linear1.weight, L.linear1.bias, self.actn, L.dropout, x_image=x_image)
            return _linT_drop(u, L.linear2.weight, L.linear2.bias, None, L.dropout2, residual)
        if self.norm_first:
            # the norms write the operand image of the block they feed on the side (no packing pass over their output)
            # ... and hand x on to the residual connection, so that the two gradients of x meet inside the norm's backward kernel
            prec = _state["precision"]
            y1, img1, h = ag.layernorm_fork(h, L.norm1.weight, L.norm1.bias, L.norm1.eps, image_precision=prec)
            att = sa_block(y1, img1)
            if keep_rows is not None:
                att, h = att.index_select(0, keep_rows), h.index_select(0, keep_rows)
            h = _linT_drop(att, sa.out_proj.weight, sa.out_proj.bias, None, L.dropout1, h)
            y2, img2, h = ag.layernorm_fork(h, L.norm2.weight, L.norm2.bias, L.norm2.eps, image_precision=prec)
            return ff_block(y2, h, img2)
        att = sa_block(h)
        if keep_rows is not None:
            att, h = att.index_select(0, keep_rows), h.index_select(0, keep_rows)
        h = ag.layernorm(_linT_drop(att, sa.out_proj.weight, sa.out_proj.bias, None, L.dropout1, h), L.norm1.weight, L.norm1.bias, L.norm1.eps)
        return ag.layernorm(ff_block(h, h), L.norm2.weight, L.norm2.bias, L.norm2.eps)

    def _x_attn_pool_train(self, h_keys, n, Tk):
        d, H, dh = self.latent_dim, self.num_heads, self.head_dim
        mha = self.x_attn_mha_layer
        kvn = ag.layernorm(h_keys, self.x_attn_kv_norm.weight, self.x_attn_kv_norm.bias, self.x_attn_kv_norm.eps)
        w, b = mha.in_proj_weight, mha.in_proj_bias             # row slices of a parameter: torch views carry the gradient
        kvp = _linT(kvn, w[d:], b[d:])
        q = self.x_attn_query
        if self.norm_first:
            q = ag.layernorm(q, self.x_attn_query_norm.weight, self.x_attn_query_norm.bias, self.x_attn_query_norm.eps)
        qp = _linT(q, w[:d], b[:d])
        pooled = ag.xattn_pool(qp, kvp, n, Tk, H, dh, mha.dropout if mha.training else 0.0)
        o = _linT(pooled, mha.out_proj.weight, mha.out_proj.bias)
        o = ag.dropout(o, self.x_attn_dropout.p, self.x_attn_dropout.training)
        o = ag.add(o, q.reshape(-1))
        if not self.norm_first:
            o = ag.layernorm(o, self.x_attn_query_norm.weight, self.x_attn_query_norm.bias, self.x_attn_query_norm.eps)
        return _linT(o, self.latent2embed.weight, self.latent2embed.bias)

    def _forward_train(self, h_in, n, S, attend, keep_rows, dense: bool, Tk=None):
        """Shared training-mode body of ``forward`` (dense) and ``forward_tokens`` (live tokens)."""
        d = self.latent_dim
        h = _linT(h_in, self.embed2latent.weight, self.embed2latent.bias)
        layers = self.transformer_encoder.layers
        agg = self.transformer_agg
        for li, L in enumerate(layers):
            last = li == len(layers) - 1
            h = self._layer_train(L, h, attend, keep_rows if (last and agg in ('x-attn', 'cls')) else None)
        self.last_attention_weights = None
        if agg == 'x-attn':
            return self._x_attn_pool_train(h, n, Tk)
        if agg == 'cls':
            return _linT(h, self.latent2embed.weight, self.latent2embed.bias)
        raise NotImplementedError(f"transformer_agg={agg!r} in training mode (the shipped configs train with 'x-attn')")

    def _x_attn_pool(self, h_keys, n, Tk):
        """Cross-attention pooling over the (already selected) key tokens h_keys [n*Tk, d] (models.py:422-443)."""
        d, H, dh = self.latent_dim, self.num_heads, self.head_dim
        mha = self.x_attn_mha_layer
        w, b = mha.in_proj_weight.detach(), mha.in_proj_bias.detach()
        ln = self.x_attn_kv_norm
        _, img = ops.layernorm_packed(h_keys, ln.weight, ln.bias, ln.eps, _state["precision"], want_fp32=False)      # the norm writes the K|V projection's packed operand itself
        if img is not None:
            kvp = ops.linear_packed(img, h_keys.shape[0], w[d:], b[d:], precision=_state["precision"])
        else:
            kvp = _lin(ops.layernorm(h_keys, ln.weight, ln.bias, ln.eps), w[d:], b[d:])                 # [n*Tk, 2d] = K|V
        # the projected query depends on parameters only: computed once per parameter version (a 1-row product through the
        # tile kernel costs ~70 us of launch + operand packing per step otherwise)
        def build_q():
            q_ = self.x_attn_query.detach()
            if self.norm_first:
                q_ = ops.layernorm(q_, self.x_attn_query_norm.weight, self.x_attn_query_norm.bias, self.x_attn_query_norm.eps)
            return q_, _lin(q_, w[:d], b[:d])
        q, qp = _cached(self, ("x_attn_q", _state["precision"]),
                        [self.x_attn_query, self.x_attn_query_norm.weight, self.x_attn_query_norm.bias, mha.in_proj_weight, mha.in_proj_bias], build_q)
        pooled = ops.xattn_pool(qp, kvp, n, Tk, H, dh)
        o = _lin(pooled, mha.out_proj.weight, mha.out_proj.bias, residual=q.reshape(-1))        # + query (broadcast)
        if not self.norm_first:
            o = ops.layernorm(o, self.x_attn_query_norm.weight, self.x_attn_query_norm.bias, self.x_attn_query_norm.eps)
        return _lin(o, self.latent2embed.weight, self.latent2embed.bias)

    def _key_positions(self):
        return (~self.x_attn_key_padding_mask[0]).nonzero().flatten().tolist()

    def forward(self, fusion_sequence, fusion_mask, src_mask=None):
        """Dense path, the reference's signature: fusion_sequence [n,S,D] (batch-major, as the encoder passes
        it), fusion_mask bool [n,S] (True = padding), src_mask bool [S,S] (True = not allowed) -> [n,D].
        Also produces the last layer's attention weights for forward hooks on ``layers[-1].self_attn``."""
        n, S, D = fusion_sequence.shape
        d, H, dh = self.latent_dim, self.num_heads, self.head_dim
        kbits = None if fusion_mask is None else ops.mask_bits(fusion_mask)
        sbits = None if src_mask is None else ops.mask_bits(src_mask)
        if _train_path(self) or ag.needs_grad(fusion_sequence):
            def attend_t(qkv, p_att):
                return ag.fusion_attention(qkv, n, S, H, dh, kbits, sbits, p_drop=p_att)
            dev = fusion_sequence.device
            if self.transformer_agg == 'x-attn':
                keys = torch.tensor(self._key_positions(), device=dev)
                keep = (torch.arange(n, device=dev).unsqueeze(1) * S + keys.unsqueeze(0)).flatten()
                Tk = int(keys.numel())
            else:
                keep, Tk = torch.arange(n, device=dev) * S, None
            return self._forward_train(fusion_sequence.reshape(n * S, D), n, S, attend_t, keep, True, Tk)
        h = _lin(fusion_sequence.reshape(n * S, D), self.embed2latent.weight, self.embed2latent.bias)
        layers = self.transformer_encoder.layers
        for li, L in enumerate(layers):
            want = li == len(layers) - 1
            seen = {}

            def attend(qkv, x_in, want=want, seen=seen):
                att, pr = ops.fusion_attention(qkv, n, S, H, dh, kbits, sbits, want_probs=want)
                seen["att"], seen["pr"], seen["in"] = att, pr, x_in
                return att
            h = self._layer(L, h, attend)
            if want:
                self.last_attention_weights = seen["pr"]
                for hook in list(L.self_attn._forward_hooks.values()):      # analysis hooks expect (attn_out, weights)
                    hook(L.self_attn, (seen["in"],), (seen["att"], seen["pr"]))
        agg = self.transformer_agg
        if agg == 'x-attn':
            keys = self._key_positions()
            h3 = h.view(n, S, d)
            h_keys = h3[:, keys, :].reshape(n * len(keys), d) if len(keys) != S else h
            return self._x_attn_pool(h_keys, n, len(keys))
        if agg == 'cls':
            return _lin(h.view(n, S * d)[:, :d], self.latent2embed.weight, self.latent2embed.bias)
        if agg in ('mean', 'max'):
            e = _lin(h, self.latent2embed.weight, self.latent2embed.bias).view(n, S, -1)
            return ops.token_pool(e, kbits, agg)
        raise NotImplementedError(agg)

    # ---- compact path: padded tokens are never materialised ----------------------------------
    def supports_live_tokens(self) -> bool:
        """Padded tokens cannot influence the result (they are masked as keys and never pooled) except when the
        cross-attention pooling has no bottleneck: then its fixed key mask lets every token in, padding included
        (models.py:382-385).  Forward hooks on the last attention need the dense [n,H,S,S] weights."""
        if self.transformer_agg == 'x-attn' and self.num_tx_bottlenecks == 0:
            return False
        if self.transformer_agg == 'max':
            return False
        return len(self.transformer_encoder.layers[-1].self_attn._forward_hooks) == 0

    def live_token_plan(self, fusion_mask: torch.Tensor, src_mask: Optional[torch.Tensor]) -> dict:
        """Index plumbing for the compact path (torch ops + one small host round trip per distinct mask).

        Live tokens are stored drug after drug.  Attention tiles hold up to 32 consecutive live rows covering WHOLE
        drugs (on average ~5 live tokens per drug, so several drugs share one 32x32 MFMA tile); ``row_bits[r]`` bit j
        says that row r may not attend the j-th row of its tile: rows of other drugs, and the reference's [S,S]
        source mask between rows of the same drug."""
        n, S = fusion_mask.shape
        live = ~fusion_mask
        dev = live.device
        counts = live.sum(1)
        row_start = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=row_start[1:])
        token_index = live.flatten().nonzero().flatten()
        R = int(token_index.numel())
        cidx = torch.cumsum(live.to(torch.int64), 1) - 1                                   # compact index inside the drug
        # greedy packing of consecutive drugs into tiles of <= 32 rows (host: n small integers)
        cnt = counts.cpu().numpy()
        tile_of_drug = np.empty(n, dtype=np.int64)
        starts, fill, t = [0], 0, 0
        for i in range(n):
            c = int(cnt[i])
            if fill + c > 32:
                t += 1
                starts.append(starts[-1] + fill)
                fill = 0
            tile_of_drug[i] = t
            fill += c
        starts.append(starts[-1] + fill)
        tile_start = torch.tensor(starts, dtype=torch.int64, device=dev)                   # [n_tiles+1] row offsets
        tile_of_drug = torch.from_numpy(tile_of_drug).to(dev)
        row_drug = token_index // S
        row_pos = token_index % S
        off_in_tile = (row_start[:-1] - tile_start[tile_of_drug])                          # first row of the drug inside its tile
        # bit j allowed <=> j in [off, off+count) of the same drug and not blocked by the source mask
        allowed = torch.zeros(n, S, dtype=torch.int64, device=dev)
        if src_mask is not None:
            ok = live.unsqueeze(1) & ~src_mask.unsqueeze(0)                                # [n, query pos, key pos]
        else:
            ok = live.unsqueeze(1).expand(n, S, S)
        allowed = (ok.to(torch.int64) << (cidx.clamp_min(0) + off_in_tile.unsqueeze(1)).unsqueeze(1)).sum(-1)   # [n,S]
        row_bits = (~allowed[row_drug, row_pos]).to(torch.int32).contiguous()              # set bit = NOT allowed
        plan = {"n": n, "S": S, "R": R, "row_start": row_start, "token_index": token_index, "row_bits": row_bits,
                "tile_start": tile_start, "n_tiles": int(tile_start.numel()) - 1}
        if self.transformer_agg == 'x-attn':
            off = S - (NUM_MODALITIES + self.num_tx_bottlenecks)                           # 1 if a cls token leads
            keys = torch.tensor([k + off for k in self._key_positions()], device=dev)
            plan["key_rows"] = (row_start[:-1].unsqueeze(1) + cidx[:, keys]).flatten().contiguous()
            plan["Tk"] = int(keys.numel())
        elif self.transformer_agg == 'cls':
            plan["key_rows"] = row_start[:-1].contiguous()
        return plan

    def merged_live_plan(self, a: dict, b: dict) -> dict:
        """The live-token plan of two batches laid one after the other (``a``'s drugs and token rows first): one transformer pass
        over both (the finetune step's head side and tail side).  Row / tile / key tables are offset by ``a``'s row count; the
        per-row bit masks are tile-local and carry over.  Kept for the last pair of plans seen."""
        key = (id(a), id(b))
        hit = self.__dict__.get("_merged_plan")
        if hit is not None and hit[0] == key:
            return hit[1]
        assert a["S"] == b["S"] and a.get("Tk") == b.get("Tk")
        Ra = a["R"]
        m = {"n": a["n"] + b["n"], "S": a["S"], "R": Ra + b["R"], "n_tiles": a["n_tiles"] + b["n_tiles"],
             "row_start": torch.cat([a["row_start"][:-1], b["row_start"] + Ra]).contiguous(),
             "tile_start": torch.cat([a["tile_start"][:-1], b["tile_start"] + Ra]).contiguous(),
             "row_bits": torch.cat([a["row_bits"], b["row_bits"]]).contiguous()}
        if "key_rows" in a:
            m["key_rows"] = torch.cat([a["key_rows"], b["key_rows"] + Ra]).contiguous()
        if "Tk" in a:
            m["Tk"] = a["Tk"]
        self.__dict__["_merged_plan"] = (key, m, a, b)          # (holds a and b: their ids cannot be recycled meanwhile)
        return m

    def forward_tokens(self, tokens: torch.Tensor, plan: dict) -> torch.Tensor:
        """Same result as ``forward`` on the dense sequence, computed on the live token rows only
        (tokens [R,D] in plan['token_index'] order)."""
        n, S, H, dh = plan["n"], plan["S"], self.num_heads, self.head_dim
        if _train_path(self) or ag.needs_grad(tokens):
            def attend_t(qkv, p_att):
                return ag.fusion_attention(qkv, plan["n_tiles"], S, H, dh, row_start=plan["tile_start"], row_bits=plan["row_bits"],
                                           p_drop=p_att)
            return self._forward_train(tokens, n, S, attend_t, plan.get("key_rows"), False, plan.get("Tk"))
        h = _lin(tokens, self.embed2latent.weight, self.embed2latent.bias)
        layers = self.transformer_encoder.layers
        agg = self.transformer_agg

        def attend(qkv, _x):
            return ops.fusion_attention(qkv, plan["n_tiles"], S, H, dh, row_start=plan["tile_start"], row_bits=plan["row_bits"])[0]
        for li, L in enumerate(layers):
            last = li == len(layers) - 1
            keep = plan.get("key_rows") if (last and agg in ('x-attn', 'cls')) else None
            h = self._layer(L, h, attend, keep_rows=keep)
        self.last_attention_weights = None
        if agg == 'x-attn':
            return self._x_attn_pool(h, n, plan["Tk"])
        if agg == 'cls':
            return _lin(h, self.latent2embed.weight, self.latent2embed.bias)
        if agg == 'mean':
            e = _lin(h, self.latent2embed.weight, self.latent2embed.bias)
            return ops.csr_aggregate(e, plan["row_start"], None, mean=True)
        raise NotImplementedError(agg)


# ------------------------------------------------------------------------------------- decoder
class Symmetric(nn.Module):
    """madrigal/models/models.py:522-524 (parametrisation registered at :922)."""

    def forward(self, W):
        if W.is_cuda and W.dim() == 3 and ag.needs_grad(W):
            return ag.symmetrize(W)
        if W.is_cuda:
            return ops.symmetrize(W.detach() if W.dim() == 3 else W.detach().unsqueeze(0)).view_as(W)
        return W.triu() + W.triu(1).transpose(-1, -2)      # parameter materialisation off the GPU (state_dict tools)


class BilinearDDIScorer(nn.Bilinear):
    """madrigal/models/models.py:526-547: S[l,i,j] = input1[i]^T W[l] input2[j] -> [L', n1, n2] raw logits,
    optional ``label_range`` slice of the outcomes.  ``bias`` exists in the state_dict and is unused, as in
    the reference."""

    def __init__(self, input_dim1: int, input_dim2: int, output_dim: int):
        super().__init__(in1_features=input_dim1, in2_features=input_dim2, out_features=output_dim)
        self._sym_cache = None

    def symmetric_weight(self) -> torch.Tensor:
        """W_sym on the device, recomputed only when the underlying parameter changes (the reference
        re-runs triu/transpose on every forward)."""
        if hasattr(self, "parametrizations") and "weight" in self.parametrizations:
            w = self.parametrizations.weight.original
            key = (w.data_ptr(), w._version, str(w.device))
            if self._sym_cache is None or self._sym_cache[0] != key:
                self._sym_cache = (key, ops.symmetrize(w.detach()))
            return self._sym_cache[1]
        return self.weight.detach()

    def bilinear(self, input1, input2, weight, epilogue=ops.EPI_STORE, out=None):
        ops.forward_only(input1, input2)
        return ops.bilinear_allpairs(input1, input2, weight, precision=_state["precision"], epilogue=epilogue, out=out)

    def score_triples(self, input1, input2, plan: dict) -> torch.Tensor:
        """Extension (finetune step): scores of the plan's (label, head, tail) triples only, in the plan's
        label-sorted order, differentiable w.r.t. both embeddings and the weight (train_ddi_batch.py:285-286
        reads exactly these entries of the dense result)."""
        return ag.bilinear_gather(input1, input2, self.weight, plan, _state["precision"])

    def forward(self, input1, input2, label_range: tuple = None, epilogue=ops.EPI_STORE, out=None):
        if ag.needs_grad(input1, input2) or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            # drop-in path of the reference's own training loop: a differentiable dense [L,Nh,Nt] result (memory and
            # backward cost proportional to L*Nh*Nt, as in the reference; score_triples() is the fast path)
            if epilogue != ops.EPI_STORE or out is not None:
                raise ValueError("epilogue / out are inference-only arguments")
            w = self.weight
            if label_range is not None:
                assert len(label_range) == 2
                w = w[label_range[0]:label_range[1]]
            return ag.bilinear_allpairs(input1, input2, w.contiguous(), _state["precision"])
        w = self.symmetric_weight()
        if label_range is not None:
            assert len(label_range) == 2
            w = w[label_range[0]:label_range[1]]
        return self.bilinear(input1, input2, w, epilogue, out)


# ------------------------------------------------------------------------------------- position encodings
class PositionEncodingSinusoidal(nn.Module):
    """madrigal/models/models.py:551-587.  Inside NovelDDIEncoder the table is added by mdg_assemble_tokens."""

    def __init__(self, d_model: int, dropout: float = 0.1, max_len: int = 19, num_tx_bottlenecks: int = 0, transformer_agg: str = 'cls'):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(1, max_len, d_model)
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        if num_tx_bottlenecks > 0:
            seq = NUM_MODALITIES + num_tx_bottlenecks + (1 if transformer_agg == 'cls' else 0)
            full = torch.zeros(1, seq, d_model)
            full[:, :max_len] = pe
            pe = full
        self.register_buffer('pe', pe)

    def table(self, train: bool = False):
        return self.pe[0]

    def forward(self, x):
        _require_eval(self)
        return x + self.pe


class PositionEncodingLearnable(nn.Module):
    """madrigal/models/models.py:590-603."""

    def __init__(self, d_model: int, dropout: float = 0.1, max_len: int = 19, num_tx_bottlenecks: int = 0, transformer_agg: str = 'cls'):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        self.max_len = max_len
        self.pe = nn.Parameter(torch.randn(1, max_len, d_model))

    def table(self, train: bool = False):
        return self.pe[0] if train else self.pe.detach()[0]

    def forward(self, x):
        _require_eval(self)
        x = x.clone()
        x[:, :self.max_len, :] += self.pe.detach()
        return x


# ------------------------------------------------------------------------------------- the encoder
class NovelDDIEncoder(nn.Module):
    """madrigal/models/models.py:607-899: four modality encoders -> token assembly (bottleneck tokens,
    masks, position encoding) -> transformer fusion (+ the uni-modal projector branch)."""

    def __init__(self, all_kg_data, feat_dim, str_encoder_name, str_encoder_hparams, kg_encoder_name, kg_encoder_hparams,
                 cv_encoder_name, cv_encoder_hparams, tx_encoder_name, tx_encoder_hparams, num_tx_bottlenecks, pos_emb_dropout,
                 transformer_fusion_hparams, proj_hparams, fusion='transformer_uni_proj', str_node_feat_dim=MOL_DIM,
                 use_modality_pretrain=True, normalize=False, pos_emb_type='learnable', adapt_before_fusion=False, **kwargs):
        super().__init__()
        if NUM_NON_TX_MODALITIES != 3:
            raise NotImplementedError("extra tabular modalities (NON_TX_MODALITIES env override) are outside the path")
        self.embed_dim, self.fusion, self.normalize = feat_dim, fusion, normalize
        self.adapt_before_fusion = adapt_before_fusion
        self.use_tx_basal = kwargs.get('use_tx_basal', False)
        self.str_encoder = get_str_encoder(str_encoder_name, str_encoder_hparams, feat_dim, str_node_feat_dim, use_modality_pretrain)
        self.kg_encoder_name = kg_encoder_name
        self.kg_encoder = get_kg_encoder(kg_encoder_name, kg_encoder_hparams, feat_dim, all_kg_data, use_modality_pretrain)
        self.cv_encoder = get_tabular_mod_encoder(cv_encoder_name, cv_encoder_hparams, feat_dim, use_modality_pretrain, mod="cv")
        self.tabular_mod_encoders = nn.ModuleDict()
        if tx_encoder_name == 'mlp':
            self.tx_encoder_dict = nn.ModuleDict({c: get_tx_encoder(tx_encoder_name, tx_encoder_hparams, feat_dim, False) for c in CELL_LINES})
        elif tx_encoder_name == 'chemcpa':
            self.tx_encoder_dict = None
            self.tx_encoder, cell_lines = get_tx_encoder(tx_encoder_name, tx_encoder_hparams, feat_dim, use_modality_pretrain)
            self.set_cell_line_categories(cell_lines)
        else:
            raise NotImplementedError(tx_encoder_name)
        self.num_tx_bottlenecks = num_tx_bottlenecks
        self.transformer_agg = transformer_fusion_hparams['transformer_agg']
        self.transformer_att_heads = transformer_fusion_hparams['transformer_att_heads']
        pos_emb_max_len = NUM_MODALITIES if num_tx_bottlenecks == 0 else NUM_NON_TX_MODALITIES
        if self.transformer_agg == 'cls':
            pos_emb_max_len += 1
        if num_tx_bottlenecks > 0:
            self.tx_bottleneck_tokens = nn.Parameter(torch.randn(num_tx_bottlenecks, feat_dim))
        cls_ = {'learnable': PositionEncodingLearnable, 'sinusoidal': PositionEncodingSinusoidal}.get(pos_emb_type)
        if cls_ is None:
            raise NotImplementedError(pos_emb_type)
        self.pos_encoder = cls_(d_model=feat_dim, dropout=pos_emb_dropout, max_len=pos_emb_max_len,
                                num_tx_bottlenecks=num_tx_bottlenecks, transformer_agg=self.transformer_agg)
        self.transformer = TransformerFusion(feat_dim, num_tx_bottlenecks, **transformer_fusion_hparams)
        if self.transformer_agg == 'cls':
            self.cls = nn.Parameter(torch.randn(1, feat_dim))
        ph = proj_hparams
        self.uni_projector = MLPAdaptor(feat_dim, ph['proj_hidden_dims'], feat_dim, ph['proj_dropout'], ph['proj_norm'], ph['proj_actn'], ph['proj_order'])
        if fusion == 'transformer_uni_proj':
            self.uni_fuser = MLPAdaptor(feat_dim, ph['proj_hidden_dims'], feat_dim, ph['proj_dropout'], ph['proj_norm'], ph['proj_actn'], ph['proj_order'])
        # run the fusion transformer on live (unmasked) tokens only: identical z, a fraction of the rows
        self.live_tokens_only = True
        # run the KG encoder on a second HIP stream beside the structure / cv / tx encoders
        self.overlap_kg = True
        self._plan_cache = None
        self._kg_stream = None

    def set_cell_line_categories(self, cell_lines_lowercase) -> None:
        """Category order of the reference's sklearn OneHotEncoder (sorted unique names, models.py:648-649)."""
        cats = sorted(set(str(c) for c in cell_lines_lowercase))
        self._cell_line_index = {c: i for i, c in enumerate(cats)}

    # -- encoders ---------------------------------------------------------------------------
    def _mask_plan(self, batch_masks: torch.Tensor, dev, compact: bool) -> dict:
        """Everything that depends only on the modality masks (row partitions, token masks, live-token plan, tx rows),
        computed once per distinct mask tensor: steady-state encodes then run without a host synchronisation."""
        key = (batch_masks.data_ptr(), batch_masks._version, tuple(batch_masks.shape), str(dev), compact)
        if self._plan_cache is None:
            self._plan_cache = {}
        hit = self._plan_cache.get(key)
        if hit is not None and hit[1] is batch_masks:
            return hit[0]
        n = batch_masks.shape[0]
        rows = uni_rows = uni_col = None
        masks_f, n_uni = batch_masks, 0
        if self.fusion == 'transformer_uni_proj':                     # models.py:781-790
            avail = (~batch_masks).sum(dim=1)
            assert bool(torch.all(avail > 0))
            multi = avail > 1
            rows = multi.nonzero().flatten()
            uni_rows = (~multi).nonzero().flatten()
            n_uni = int(uni_rows.numel())
            uni_col = (~batch_masks[uni_rows]).to(torch.int64).argmax(dim=1)          # the single available modality
            masks_f = batch_masks[rows]
        nf = n if rows is None else int(rows.numel())
        nb, has_cls = self.num_tx_bottlenecks, self.transformer_agg == 'cls'
        parts = []
        if has_cls:
            parts.append(torch.zeros(nf, 1, dtype=torch.bool, device=dev))
        parts.append(masks_f[:, :NUM_NON_TX_MODALITIES])
        if nb > 0:                                                    # bottleneck tokens are never padding (:805)
            parts.append(torch.zeros(nf, nb, dtype=torch.bool, device=dev))
        parts.append(masks_f[:, NUM_NON_TX_MODALITIES:])
        kpm = torch.cat(parts, dim=1)
        src = None
        if nb > 0:                                                    # non-TX and TX tokens only meet through the bottleneck (:813-816)
            S0 = NUM_MODALITIES + nb
            src = torch.zeros(S0, S0, dtype=torch.bool, device=dev)
            src[:NUM_NON_TX_MODALITIES, -len(CELL_LINES):] = True
            src[-len(CELL_LINES):, :NUM_NON_TX_MODALITIES] = True
            if has_cls:
                full = torch.zeros(S0 + 1, S0 + 1, dtype=torch.bool, device=dev)
                full[1:, 1:] = src
                src = full
        mp = {"rows": rows, "uni_rows": uni_rows, "uni_col": uni_col, "n_uni": n_uni, "nf": nf, "kpm": kpm, "src": src,
              "live": self.transformer.live_token_plan(kpm, src) if (compact and nf > 0) else None,
              # rows of the cell-line-major tx stack whose (drug, cell line) signature exists
              "tx_rows": (~batch_masks[:, NUM_NON_TX_MODALITIES:]).t().reshape(-1).nonzero().flatten()}
        if rows is not None:                                          # index lists of the training path's own gather nodes (encode)
            ar = torch.arange(len(CELL_LINES), device=dev).unsqueeze(1) * n
            mp["tx_rows_of_multi"] = (ar + rows.unsqueeze(0)).reshape(-1)             # rows of the [16 n, D] tx stack: (cell line, multi-modal drug)
            mp["uni_flat"] = uni_col * n + uni_rows                                    # rows of [str | kg | cv | tx stack]: the single modality of a uni-modal drug
            perm = torch.empty(n, dtype=torch.int64, device=dev)
            perm[rows] = torch.arange(nf, device=dev)
            perm[uni_rows] = nf + torch.arange(n_uni, device=dev)
            mp["merge_perm"] = perm                                                    # drug d's row in [fused rows | uni-modal rows]
        # a few entries, oldest out: the head and the tail side of a step carry different masks
        while len(self._plan_cache) >= 6:
            self._plan_cache.pop(next(iter(self._plan_cache)))
        self._plan_cache[key] = (mp, batch_masks)          # (holds the mask tensor: its address cannot be recycled meanwhile)
        return mp

    def _kg_graphed(self, kg_data, dev):
        """Training: the KG encoder's forward and backward as two captured hipGraphs (torch.cuda.make_graphed_callables).  The KG,
        hence every shape and every index plan of the pass, is the same in every step, and the pass is ~290 + ~290 small launches
        over ten node types: replayed as graphs they cost the host two launches and the GPU no launch gaps.  Parameters are read at
        replay time (the optimizer updates them in place), gradients come back in static buffers.  Off (None) under data-parallel
        SyncBatchNorm reductions or for CPU tensors; a capture failure falls back to eager launches.
        Per step kind (``encoder.kg_graph``; MDG_KG_GRAPH=0/1 overrides): measured on one MI355X (scripts/kg_graph_probe.py) the
        replay costs the host 2.7 ms instead of 8.5 ms per forward+backward but the GPU 11.3 ms instead of ~9 ms (hipGraph nodes are
        dispatched no faster than eager launches here), so it helps the host-bound contrastive step (17.7-19.8 -> 16.0 ms: PretrainStep
        switches it on) and hurts the GPU-bound finetune step (51.3 -> 52.4 ms: FinetuneStep leaves it off)."""
        want = os.environ.get("MDG_KG_GRAPH")                 # "0" / "1" overrides the step's own preference (kg_graph attribute)
        want = (want == "1") if want in ("0", "1") else bool(getattr(self, "kg_graph", False))
        if dev.type != 'cuda' or not want or ag._bn_sync["reduce"] is not None:
            return None
        params = list(self.kg_encoder.parameters())
        key = (id(kg_data), _state["precision"], tuple(p.requires_grad for p in params), torch.is_grad_enabled())
        hit = self.__dict__.get("_kg_graph_cache")
        if hit is not None and hit[0] == key:
            return hit[1]
        if not torch.is_grad_enabled() or not any(p.requires_grad for p in params):
            return None
        enc = self.kg_encoder

        class DrugRows(nn.Module):
            def __init__(self):
                super().__init__()
                self.enc = enc

            def forward(self, x_drug):                     # the argument only anchors the callable's signature
                return self.enc(kg_data.x_dict, kg_data.edge_index_dict, only_types=('drug',))['drug']
        try:
            runner = torch.cuda.make_graphed_callables(DrugRows(), (kg_data.x_dict['drug'],), allow_unused_input=True)
        except Exception as e:                              # capture is an optimisation: keep training on eager launches
            import warnings
            warnings.warn(f"KG encoder graph capture failed ({type(e).__name__}: {e}); running it eagerly")
            runner = None
        self.__dict__["_kg_graph_cache"] = (key, runner, kg_data)
        return runner

    def _encode_tx(self, batch_tx_dict, n: int, device, present_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[16*n, D] cell-line-major tx embeddings (models.py:753-769).  ``present_rows`` (int64 rows of the stack):
        encode only the (drug, cell line) rows that exist; the others are left zero (callers that pass it never
        read them)."""
        if self.tx_encoder_dict is not None:
            return torch.cat([self.tx_encoder_dict[c](batch_tx_dict[c]['sigs']) for c in CELL_LINES], dim=0)
        sigs = torch.cat([batch_tx_dict[c]['sigs'] for c in CELL_LINES], dim=0)
        # covariate (cell line) index per row of the stack; host work + H2D copy only when the name arrays change
        names = [np.asarray(batch_tx_dict[c]['cell_lines']) for c in CELL_LINES]
        ckey = (str(device),) + tuple((a.size, str(a[0]) if a.size else "", bool(a.size and np.all(a == a[0]))) for a in names)
        hit = self.__dict__.get("_cov_cache")
        if hit is None or hit[0] != ckey or not ckey[1:] or not all(k[2] for k in ckey[1:]):
            idx = []
            for c, a in zip(CELL_LINES, names):
                if a.size and not np.all(a == a[0]):
                    idx.append(torch.tensor([self._cell_line_index[str(x)] for x in a], dtype=torch.int64))
                else:
                    idx.append(torch.full((a.size,), self._cell_line_index[str(a[0]) if a.size else c], dtype=torch.int64))
            hit = (ckey, torch.cat(idx).to(device))
            self.__dict__["_cov_cache"] = hit
        cov = hit[1]
        sel = present_rows
        if sel is not None:
            sigs, cov = sigs.index_select(0, sel), cov.index_select(0, sel)
        zeros = torch.zeros(sigs.shape[0], dtype=torch.int64, device=device)
        if sigs.shape[0] == 0:
            lat = torch.zeros(0, self.embed_dim, device=device)
        else:
            lat = self.tx_encoder.predict(genes=sigs, drugs_idx=zeros, dosages=zeros, covariates=None, covariate_indices=[cov],
                                          return_latent_basal=self.use_tx_basal, return_latent_treated=(not self.use_tx_basal),
                                          compute_reconstruction=False)[2]
        if sel is None:
            return lat
        full = torch.zeros(16 * n, self.embed_dim, device=device)
        full[sel] = lat
        return full

    def encode(self, batch_drugs, batch_masks, batch_mols, batch_kg, batch_cv, batch_tx_dict, raw_encoder_output=False, **kwargs):
        train = _train_path(self)
        norm = ag.l2_normalize if train else ops.l2_normalize
        dev = batch_cv.device
        n, Dm = batch_drugs.shape[0], self.embed_dim
        compact = (not raw_encoder_output and self.live_tokens_only and self.fusion in ('transformer', 'transformer_uni_proj')
                   and self.transformer.supports_live_tokens())
        kg_data, kg_map = batch_kg['data'], batch_kg['drug_index_map']
        # drugs absent from the KG get filler rows that are always masked (models.py:734-736); the size of the
        # table needs the largest drug id (the reference's .item() syncs here as well)
        # Per-iteration inputs may arrive on the CPU (what a loader yields; the contrastive loop draws fresh masks every
        # iteration): the index work that needs their VALUES is then done on the host and the tensors travel through pinned
        # buffers, so nothing here waits for the device (hostio.py).  Device inputs behave as in the reference (.item() syncs).
        host_masks = host_drugs = None
        if dev.type == 'cuda':
            from . import hostio
            if batch_masks.device.type == 'cpu':
                host_masks, batch_masks = batch_masks, hostio.upload(batch_masks, dev, "masks")
            if batch_drugs.device.type == 'cpu':
                host_drugs, batch_drugs = batch_drugs, hostio.upload(batch_drugs, dev, "drugs")
        filler = kwargs.get('kg_filler')
        if filler is None:
            mkey = (kg_map.data_ptr(), kg_map._version, kg_map.numel())
            hit = self.__dict__.get("_kg_map_max")
            if hit is None or hit[0] != mkey:
                hit = self.__dict__["_kg_map_max"] = (mkey, int(kg_map.max().item()) if kg_map.numel() else -1, kg_map)
            dmax = int(host_drugs.numpy().max()) if host_drugs is not None and n else (int(batch_drugs.max().item()) if n else -1)
            filler = torch.randn((max(dmax, hit[1]) + 1, Dm), device=dev)

        # ``kg_share`` (extension, a dict owned by the caller): the KG encoder sees the same graph on the head and on the
        # tail side of one step and has neither dropout nor batch statistics, so its two passes are identical; the
        # second side reuses the first side's node embeddings (the tape then adds both sides' gradients into one
        # backward pass — the same sum the reference forms from its two identical passes).
        share = kwargs.get('kg_share')

        def run_kg():
            key = (id(kg_data), "drug")
            kg_valid = None if share is None else share.get(key)
            if kg_valid is None:
                kg_kw = {} if kwargs.get('kg_shard') is None else {"shard": kwargs['kg_shard']}
                graphed = self._kg_graphed(kg_data, dev) if (train and not kg_kw) else None
                if graphed is not None:
                    kg_valid = graphed(kg_data.x_dict['drug'])
                else:
                    kg_valid = self.kg_encoder(kg_data.x_dict, kg_data.edge_index_dict, only_types=('drug',), **kg_kw)['drug']
                if share is not None:
                    share[key] = kg_valid
            if train:
                # own gather node instead of torch's indexed assignment + indexed read (their backward sorts and scatters: rocprim
                # launches in every step): src[d] = row of drug d in the KG encoder's output, -1 for drugs outside the KG
                gkey = (kg_map.data_ptr(), kg_map._version, kg_map.numel(), batch_drugs.data_ptr(), batch_drugs._version, tuple(batch_drugs.shape),
                        batch_drugs.stride(), int(filler.shape[0]))
                cache = self.__dict__.setdefault("_kg_gather", {})
                hit = cache.get(gkey)
                if hit is None or hit[2] is not kg_map or hit[3] is not batch_drugs:
                    pos = torch.full((int(filler.shape[0]),), -1, dtype=torch.int64, device=dev)
                    pos[kg_map] = torch.arange(kg_map.numel(), device=dev)
                    while len(cache) >= 4:                             # head and tail batches of a couple of steps
                        cache.pop(next(iter(cache)))
                    hit = cache[gkey] = (gkey, pos[batch_drugs].contiguous(), kg_map, batch_drugs)
                return ag.gather_rows_or(kg_valid, hit[1], filler.to(dev)[batch_drugs])
            table = filler.to(dev).clone()
            table[kg_map] = kg_valid
            return table[batch_drugs]
        main = torch.cuda.current_stream(dev)
        # under autograd the backward of every node runs on the stream its forward ran on (torch semantics), so the KG
        # encoder's backward overlaps the other encoders' backward the same way its forward overlaps their forward
        # (measured: finetune step 115 -> 110 ms with the KG stream, -> 107 ms with the structure encoder on a third stream;
        # the tx encoder on a fourth made it slower).  Data-parallel steps keep one stream: their collectives stay in one order.
        overlap = self.overlap_kg and (not train or ag._bn_sync["reduce"] is None)
        if overlap:
            if self._kg_stream is None or self._kg_stream.device != dev:
                self._kg_stream = torch.cuda.Stream(device=dev)
            self._kg_stream.wait_stream(main)
            with torch.cuda.stream(self._kg_stream):
                kg_out = run_kg()
        # Training: the reference encodes the head side and the tail side (or the two contrastive views) of one step separately
        # (models.py:945-946, simclr.py:134-135).  When both passes are given the same molecule batch / tx dict, the encoders
        # WITHOUT dropout -- GIN and the chemCPA encoder: Linear / BatchNorm(batch statistics) / ReLU only, all rows whatever the
        # masks -- return the same tensor twice; the second pass reuses the first one's output (autograd adds both consumers'
        # gradients into one backward pass, the sum the reference forms from two identical passes) and its BatchNorm layers
        # replay their running-statistics update, so the module state after the step is what two passes leave behind.
        share_enc = (train and share is not None and ag._bn_sync["reduce"] is None and os.environ.get("MDG_SHARE_SIDES", "1") != "0")
        skey, tkey = ("str", id(batch_mols)), ("tx", id(batch_tx_dict))
        s_hit = share.get(skey) if share_enc else None
        # training: the structure encoder (a chain of small launches over the atoms, forward and backward) on a third stream
        str_side = overlap and train and s_hit is None
        if s_hit is not None:
            str_out = s_hit[0]
            ag.replay_batchnorm(s_hit[1])
        else:
            with ag.record_batchnorm() as s_log:
                if str_side:
                    if self.__dict__.get("_str_stream") is None or self._str_stream.device != dev:
                        self.__dict__["_str_stream"] = torch.cuda.Stream(device=dev)
                    self._str_stream.wait_stream(main)
                    with torch.cuda.stream(self._str_stream):
                        str_out = self.str_encoder(batch_mols, batch_mols.node_feature.float())["graph_feature"]
                else:
                    str_out = self.str_encoder(batch_mols, batch_mols.node_feature.float())["graph_feature"]
            if share_enc:
                share[skey] = (str_out, s_log, batch_mols)
        cv_out = kwargs['cv_out'] if kwargs.get('cv_out') is not None else self.cv_encoder(batch_cv)     # (embed() encodes both sides' rows in one pass)
        # tx embeddings of absent cell lines are masked tokens: the live-token path never reads them.  In training mode
        # every row goes through the tx encoder, as in the reference: its BatchNorm batch statistics include them.
        skip_absent = compact and not train
        tx_shareable = share_enc and self.tx_encoder_dict is None and not any(isinstance(m, nn.Dropout) and m.p > 0 for m in self.tx_encoder.modules())
        t_hit = share.get(tkey) if tx_shareable else None
        if t_hit is not None:
            tx_out = t_hit[0]
            ag.replay_batchnorm(t_hit[1])
        else:
            with ag.record_batchnorm() as t_log:
                tx_out = self._encode_tx(batch_tx_dict, n, dev, present_rows=self._mask_plan(batch_masks, dev, compact)["tx_rows"] if skip_absent else None)
            if tx_shareable:
                share[tkey] = (tx_out, t_log, batch_tx_dict)
        if str_side:
            main.wait_stream(self._str_stream)
            str_out.record_stream(main)
        if overlap:
            main.wait_stream(self._kg_stream)
            kg_out.record_stream(main)
        else:
            kg_out = run_kg()
        if raw_encoder_output:
            all_embeds = torch.stack([str_out, kg_out, cv_out] + list(tx_out.split(n)), dim=1)
            if host_masks is not None:               # the available (drug, modality) pairs, row-major as boolean indexing lists them
                from . import hostio
                pairs = hostio.upload(torch.from_numpy(np.flatnonzero(~host_masks.numpy().reshape(-1))), dev, "pairs")
                uni = all_embeds.reshape(n * all_embeds.shape[1], Dm).index_select(0, pairs)
            else:
                uni = all_embeds[~batch_masks]
            if self.normalize:
                uni = norm(uni)
            if kwargs.get('defer_projector'):        # the caller projects both views' rows in one pass (SimCLR_NovelDDI.forward)
                return uni
            return self.uni_projector(uni)
        if self.adapt_before_fusion:
            str_out, kg_out, cv_out, tx_out = (self.uni_projector(t) for t in (str_out, kg_out, cv_out, tx_out))
        if self.fusion in ('mean', 'add'):
            if train:
                raise NotImplementedError(f"fusion={self.fusion!r} is not differentiated on the HIP path (shipped configs fuse with the transformer)")
            all_embeds = torch.stack([str_out, kg_out, cv_out] + list(tx_out.split(n)), dim=1)
            if self.normalize:
                all_embeds = ops.l2_normalize(all_embeds)
            return ops.token_pool(all_embeds, ops.mask_bits(batch_masks), 'mean' if self.fusion == 'mean' else 'sum')
        if self.fusion not in ('transformer', 'transformer_uni_proj'):
            raise NotImplementedError(self.fusion)
        mp = self._mask_plan(batch_masks, dev, compact)
        rows, uni_rows, nf = mp["rows"], mp["uni_rows"], mp["nf"]
        nb = self.num_tx_bottlenecks
        has_cls = self.transformer_agg == 'cls'
        tok_args = dict(bottleneck=self.tx_bottleneck_tokens if nb > 0 else None, cls=self.cls if has_cls else None,
                        pe=self.pos_encoder.table(), rows=rows, normalize=self.normalize)
        if nf == 0:
            z_f = torch.zeros(0, Dm, device=dev)
        elif train:
            s_, k_, c_, t_ = str_out, kg_out, cv_out, tx_out
            if rows is not None:                         # only multi-modal drugs enter the transformer (models.py:781-790)
                s_, k_, c_ = (ag.gather_rows(v, rows) for v in (s_, k_, c_))
                t_ = ag.gather_rows(t_, mp["tx_rows_of_multi"])      # rows c * n + rows[.] of the [16 n, D] stack, cell line by cell line
            plan = mp["live"] if compact else None
            tokens = ag.assemble_tokens(s_, k_, c_, t_, bottleneck=self.tx_bottleneck_tokens if nb > 0 else None,
                                        cls=self.cls if has_cls else None, pe=self.pos_encoder.table(train=True),
                                        normalize=self.normalize, token_index=None if plan is None else plan["token_index"])
            tokens = ag.dropout(tokens, self.pos_encoder.dropout.p, self.pos_encoder.dropout.training)
            if compact and kwargs.get('defer_fusion'):
                # the caller runs the transformer itself -- over this side's and the other side's tokens in ONE pass
                # (NovelDDIMultilabel.embed) -- and hands the fused rows to ``finish``
                return PendingFusion(tokens, plan, self.transformer, lambda z_f_: self._finish_fused(z_f_, mp, n, str_out, kg_out, cv_out, tx_out, norm))
            z_f = self.transformer.forward_tokens(tokens, plan) if compact else self.transformer(tokens, fusion_mask=mp["kpm"], src_mask=mp["src"])
        elif compact:
            plan = mp["live"]
            tokens = ops.assemble_tokens(str_out, kg_out, cv_out, tx_out, token_index=plan["token_index"], **tok_args)
            z_f = self.transformer.forward_tokens(tokens, plan)
        else:
            seq = ops.assemble_tokens(str_out, kg_out, cv_out, tx_out, **tok_args)
            z_f = self.transformer(seq, fusion_mask=mp["kpm"], src_mask=mp["src"])
        if self.fusion != 'transformer_uni_proj':
            return z_f
        if train:
            return self._finish_fused(z_f, mp, n, str_out, kg_out, cv_out, tx_out, norm)
        z = torch.empty((n, Dm), dtype=torch.float32, device=dev)
        z[rows] = z_f
        if mp["n_uni"] > 0:
            all_embeds = torch.stack([str_out, kg_out, cv_out] + list(tx_out.split(n)), dim=1)
            uni = all_embeds[uni_rows, mp["uni_col"]]
            if self.normalize:
                uni = norm(uni)
            z[uni_rows] = self.uni_fuser(uni)
        return z

    def _finish_fused(self, z_f, mp, n, str_out, kg_out, cv_out, tx_out, norm):
        """Training path, after the transformer: the fused rows of the multi-modal drugs and, for 'transformer_uni_proj', the
        projected single modality of the uni-modal ones, in drug order (models.py:781-812)."""
        if self.fusion != 'transformer_uni_proj':
            return z_f
        # the merge through own gather nodes (no indexed assignments: their backward sorts): the single available modality of a
        # uni-modal drug is row uni_col * n + drug of [str | kg | cv | tx (cell line by cell line)]; the fused and the uni-modal
        # rows, one after the other, are read back in drug order
        parts = [z_f]
        if mp["n_uni"] > 0:
            uni = ag.gather_rows(torch.cat([str_out, kg_out, cv_out, tx_out], dim=0), mp["uni_flat"])
            if self.normalize:
                uni = norm(uni)
            parts.append(self.uni_fuser(uni))
        return ag.gather_rows(torch.cat(parts, dim=0) if len(parts) > 1 else z_f, mp["merge_perm"])

    def forward(self, batch_drugs, batch_masks, batch_mols, batch_kg, batch_cv, batch_tx_dict, raw_encoder_output=False, **kwargs):
        return self.encode(batch_drugs, batch_masks, batch_mols, batch_kg, batch_cv, batch_tx_dict, raw_encoder_output, **kwargs)


class PendingFusion:
    """One side of a training step stopped in front of the fusion transformer (NovelDDIEncoder.encode(defer_fusion=True)): its
    live tokens, their plan, and what remains to be done with the fused rows."""

    def __init__(self, tokens, plan, transformer, finish):
        self.tokens, self.plan, self.transformer, self.finish = tokens, plan, transformer, finish

    def run(self):
        return self.finish(self.transformer.forward_tokens(self.tokens, self.plan))

    @staticmethod
    def run_pair(a: "PendingFusion", b: "PendingFusion"):
        """Both sides through the transformer in ONE pass (every op of it is per token row or per drug: the rows come out as the
        two passes would give them, with one set of launches, and every parameter receives one gradient instead of two to add)."""
        tr = a.transformer
        plan = tr.merged_live_plan(a.plan, b.plan)
        z = tr.forward_tokens(torch.cat([a.tokens, b.tokens], dim=0), plan)
        na = a.plan["n"]
        return a.finish(z[:na]), b.finish(z[na:])


class NovelDDIMultilabel(nn.Module):
    """madrigal/models/models.py:914-953: encoder on both sides + symmetric bilinear head -> raw logits [L,Nh,Nt]."""

    def __init__(self, encoder, feat_dim, prediction_dim, prediction_dim_single_drug=None, normalize=False, use_single_drug=False):
        super().__init__()
        self.encoder = encoder
        self.embed_dim = feat_dim
        self.normalize = normalize
        self.use_single_drug = use_single_drug
        self.decoder = BilinearDDIScorer(feat_dim, feat_dim, prediction_dim)
        nn.utils.parametrize.register_parametrization(self.decoder, 'weight', Symmetric())
        # the reference encodes head and tail separately even when they are the same object (full-batch mode,
        # train_ddi_batch.py:285); in eval mode the two results are identical, so one pass is reused.
        self.reuse_identical_sides = True
        # the KG encoder is deterministic and stateless (no dropout, no BatchNorm): head and tail side share one pass
        self.share_kg_between_sides = True

    def forward(self, batch_head, batch_tail, batch_head_mod_masks, batch_tail_mod_masks, batch_kg, label_range=None,
                single_drug=False, **kwargs):
        z_head, z_tail = self.embed(batch_head, batch_tail, batch_head_mod_masks, batch_tail_mod_masks, batch_kg, **kwargs)
        return self.decoder(z_head, z_tail, label_range)

    def embed(self, batch_head, batch_tail, batch_head_mod_masks, batch_tail_mod_masks, batch_kg, **kwargs):
        """Head- and tail-side embeddings exactly as ``forward`` feeds them to the decoder (models.py:940-951).  In
        training mode the two sides are encoded separately even when they are the same batch (independent dropout
        masks / one BatchNorm update per side, as the reference does)."""
        if self.share_kg_between_sides and 'kg_share' not in kwargs:
            kwargs = dict(kwargs, kg_share={})

        # training: both sides' tokens go through the fusion transformer in one pass (PendingFusion.run_pair; MDG_FUSE_SIDES=0: two)
        defer = (self.training and torch.is_grad_enabled() and batch_head_mod_masks.is_cuda and os.environ.get("MDG_FUSE_SIDES", "1") != "0"
                 and 'defer_fusion' not in kwargs)

        cv_pair = (None, None)
        if defer and batch_head['cv'].shape[1:] == batch_tail['cv'].shape[1:] and \
                not any(isinstance(m_, nn.modules.batchnorm._BatchNorm) for m_ in self.encoder.cv_encoder.modules()):
            # the cell-viability encoder (an MLP with dropout, no batch statistics: per row) over both sides' rows at once
            nh = batch_head['cv'].shape[0]
            cv_both = self.encoder.cv_encoder(torch.cat([batch_head['cv'], batch_tail['cv']], dim=0))
            cv_pair = (cv_both[:nh], cv_both[nh:])

        def enc(b, m, cv=None):
            return self.encoder(b['drugs'], m, b['strs'], batch_kg, b['cv'], b['tx'], **(dict(kwargs, defer_fusion=True, cv_out=cv) if defer else kwargs))
        z_head = enc(batch_head, batch_head_mod_masks, cv_pair[0])
        same = (self.reuse_identical_sides and not self.training and batch_head is batch_tail and
                (batch_head_mod_masks is batch_tail_mod_masks or torch.equal(batch_head_mod_masks, batch_tail_mod_masks)))
        z_tail = z_head if same else enc(batch_tail, batch_tail_mod_masks, cv_pair[1])
        if isinstance(z_head, PendingFusion) and isinstance(z_tail, PendingFusion):
            z_head, z_tail = PendingFusion.run_pair(z_head, z_tail)
        else:
            z_head = z_head.run() if isinstance(z_head, PendingFusion) else z_head
            z_tail = z_tail.run() if isinstance(z_tail, PendingFusion) else z_tail
        if self.normalize:
            norm = ag.l2_normalize if ag.needs_grad(z_head, z_tail) else ops.l2_normalize
            z_head = norm(z_head)
            z_tail = z_head if same else norm(z_tail)
        return z_head, z_tail

    def score_triples(self, batch_head, batch_tail, batch_head_mod_masks, batch_tail_mod_masks, batch_kg, plan: dict, **kwargs):
        """Extension for the finetune step: raw logits of the plan's (label, head, tail) triples only, in the ORIGINAL
        order of the triples the plan was built from (``ops.triple_plan``) — the entries train_ddi_batch.py:285-286
        gathers from the dense [L,N,N] result — differentiable end to end on the HIP path."""
        z_head, z_tail = self.embed(batch_head, batch_tail, batch_head_mod_masks, batch_tail_mod_masks, batch_kg, **kwargs)
        return self.decoder.score_triples(z_head, z_tail, plan).index_select(0, plan["inv_perm"])
