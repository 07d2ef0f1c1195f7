"""AdamW on the HIP path + the reference's parameter grouping (madrigal/utils.py:446-613).

``AdamW`` is a ``torch.optim.Optimizer`` subclass (param_groups, ``state_dict`` layout of ``torch.optim.AdamW``:
``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter, so LR schedulers and checkpoints interoperate); ``step()`` updates
every parameter in ONE kernel launch (mdg_adamw_multi) instead of torch's per-tensor / foreach arithmetic.
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, List

import numpy as np
import torch
import torch.nn as nn

from ._lib import check, lib


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by the reference")
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._chunk = int(lib().mdg_adamw_chunk_elems())
        self._layouts = {}
        self._count = {}                # parameter -> number of updates so far (mirrored into state[p]["step"] on demand)

    def _sync_steps(self) -> None:
        for p, k in self._count.items():
            st = self.state.get(p)
            if st:
                st["step"].fill_(float(k))

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._count = {}                # re-read from the loaded ``step`` tensors at the next update

    @staticmethod
    def _upload(lay: dict, name: str, host: np.ndarray, dev) -> torch.Tensor:
        """host array -> device tensor without waiting for the stream: a ring of pinned staging buffers per table (a buffer is
        reused only after the copy issued from it has completed, which its event says)."""
        ring = lay.setdefault("ring_" + name, {"slot": 0, "bufs": [None] * 3})
        i = ring["slot"]
        ring["slot"] = (i + 1) % len(ring["bufs"])
        ent = ring["bufs"][i]
        if ent is None or ent[0].shape != host.shape or ent[0].numpy().dtype != host.dtype:
            ent = ring["bufs"][i] = [torch.from_numpy(np.empty_like(host)).pin_memory(), None]
        if ent[1] is not None:
            ent[1].synchronize()                        # issued three steps ago: long complete
        ent[0].numpy()[...] = host
        out = ent[0].to(dev, non_blocking=True)
        ent[1] = torch.cuda.Event()
        ent[1].record(torch.cuda.current_stream(dev))
        return out

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        by_dev: Dict[torch.device, list] = {}
        count, hyper_of = self._count, {}
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                st = self.state[p]
                k = count.get(p)
                if k is None:                                   # first update of this parameter (or the first after a load)
                    if g.is_sparse or p.dtype != torch.float32 or not p.is_cuda:
                        raise RuntimeError("madrigal_amd.optim.AdamW: dense fp32 parameters on the GPU only")
                    if not p.is_contiguous():
                        raise RuntimeError("madrigal_amd.optim.AdamW: parameters must be contiguous")
                    if not st:
                        st["step"] = torch.tensor(0.0)
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    k = int(float(st["step"]))
                # the step counters live in a host-side table and reach the state's ``step`` tensors when the state is read
                # (state_dict): a tensor increment + read-back per parameter and step was a third of this loop
                k += 1
                count[p] = k
                if not g.is_contiguous():
                    g = g.contiguous()
                hyper = hyper_of.get((gi, k))
                if hyper is None:
                    hyper = hyper_of[(gi, k)] = (group["lr"], b1, b2, group["eps"], group["weight_decay"], 1.0 / (1.0 - b1 ** k),
                                                 1.0 / math.sqrt(1.0 - b2 ** k), 0.0)
                by_dev.setdefault(p.device, []).append((p, g, st["exp_avg"], st["exp_avg_sq"], hyper))
        for dev, items in by_dev.items():
            # chunk layout (offsets / lengths / owning tensor) depends only on the tensor sizes: built once, vectorised
            sizes = tuple(p.numel() for p, *_ in items)
            lay = self._layouts.get((dev, sizes))
            if lay is None:
                offs, lens, owner = [], [], []
                for ti, n in enumerate(sizes):
                    o = np.arange(0, n, self._chunk, dtype=np.int64)
                    offs.append(o * 4)
                    lens.append(np.minimum(self._chunk, n - o).astype(np.int32))
                    owner.append(np.full(o.shape, ti, dtype=np.int32))
                offs, lens, owner = (np.concatenate(a) if a else np.zeros(0, dtype=np.int64) for a in (offs, lens, owner))
                lay = {"offs": offs, "owner": owner, "n": int(offs.shape[0]), "t_len": torch.from_numpy(lens).to(dev),
                       "t_own": torch.from_numpy(owner).to(dev), "base": None, "t_ptr": None}
                self._layouts[(dev, sizes)] = lay
            if lay["n"] == 0:
                continue
            # per-step tables: the chunk pointers (gradients are re-allocated by every backward pass, so their addresses move)
            # and the per-tensor hyper-parameters.  Both go through rotating PINNED host buffers and asynchronous copies: a
            # pageable .to(device) waits for everything queued on the stream (the whole backward pass), which would stop the
            # host from queueing the next step while this one still runs.
            base = np.asarray([(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()) for p, g, m, v, _ in items], dtype=np.int64)
            if lay["base"] is None or not np.array_equal(base, lay["base"]):
                lay["base"] = base
                lay["t_ptr"] = self._upload(lay, "ptr", base[lay["owner"]] + lay["offs"][:, None], dev)
            t_hyp = self._upload(lay, "hyp", np.asarray([h for *_, h in items], dtype=np.float32), dev)
            with torch.cuda.device(dev):
                check(lib().mdg_adamw_multi(ctypes.c_void_p(lay["t_ptr"].data_ptr()), ctypes.c_void_p(lay["t_len"].data_ptr()),
                                            ctypes.c_void_p(lay["t_own"].data_ptr()), ctypes.c_void_p(t_hyp.data_ptr()),
                                            ctypes.c_int64(lay["n"]), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                      "mdg_adamw_multi")
            t_hyp.record_stream(torch.cuda.current_stream(dev))
            lay["t_ptr"].record_stream(torch.cuda.current_stream(dev))
            # the kernel wrote the parameters and the moments behind torch's back: bump their in-place version counters so
            # that everything keyed on them (packed / derived weight caches of the inference path, autograd's saved-tensor
            # checks) sees the update
            torch.autograd.graph.increment_version([t for p, _, m, v, _ in items for t in (p, m, v)])
        return loss


class LinearWarmupCosineDecaySchedule(torch.optim.lr_scheduler._LRScheduler):
    """madrigal/utils.py:665-680 (train_ddi_batch.py:101-102): linear warm-up from 0 over ``warmup_epochs`` steps, then
    ``num_cycles`` half-cosines down to 0 at ``total_epochs``; applied to every parameter group's own base rate."""

    def __init__(self, optimizer, warmup_epochs, total_epochs, num_cycles=1.0, last_epoch=-1):
        self.warmup_epochs, self.total_epochs, self.num_cycles = warmup_epochs, total_epochs, num_cycles
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        e = self.last_epoch
        if e < self.warmup_epochs:
            f = e / self.warmup_epochs
        else:
            f = (1.0 + math.cos(math.pi * self.num_cycles * (e - self.warmup_epochs) / (self.total_epochs - self.warmup_epochs))) / 2.0
        return [b * f for b in self.base_lrs]


def adjust_learning_rate(optimizer, cur_epoch: float, lr: float, warmup_epochs: float, num_epochs: float) -> float:
    """madrigal/utils.py:680-692: the contrastive loop's schedule -- linear warm-up to ``lr`` over ``warmup_epochs``, then one
    half-cosine down to 0 at ``num_epochs``; ONE rate for every parameter group (the reference's TODO stands), written into the
    groups and returned.  ``cur_epoch`` is fractional: pretrain.py:65 passes epoch + i / iters_per_epoch, every iteration."""
    if cur_epoch < warmup_epochs:
        lr = lr * cur_epoch / warmup_epochs
    else:
        lr = lr * 0.5 * (1. + math.cos(math.pi * (cur_epoch - warmup_epochs) / (num_epochs - warmup_epochs)))
    for group in optimizer.param_groups:
        group['lr'] = lr
    return lr


class PretrainSchedule:
    """The per-iteration rate of pretrain.py:59-66 as a hook for ``train.PretrainStep(scheduler=...)``: called with the optimizer
    before every step, it counts the iterations itself (``iters_per_epoch`` of them to an epoch) and applies
    ``adjust_learning_rate`` with hparams['pretrain_lr'], ['warmup_epochs'], ['pretrain_num_epochs']."""

    def __init__(self, lr: float, warmup_epochs: float, num_epochs: float, iters_per_epoch: int, start_epoch: int = 0):
        self.lr, self.warmup_epochs, self.num_epochs, self.iters_per_epoch = lr, warmup_epochs, num_epochs, int(iters_per_epoch)
        self.iteration = int(start_epoch) * self.iters_per_epoch
        self.last_lr = None

    def __call__(self, optimizer) -> float:
        epoch, i = divmod(self.iteration, self.iters_per_epoch)
        self.last_lr = adjust_learning_rate(optimizer, epoch + i / self.iters_per_epoch, self.lr, self.warmup_epochs, self.num_epochs)
        self.iteration += 1
        return self.last_lr


def parameter_names_outside(model: nn.Module, forbidden: tuple, prefix: str = "") -> List[str]:
    """Names of the parameters that do not live inside a module of a ``forbidden`` type (the reference's
    get_parameter_names, madrigal/utils.py:446-460, including its exclusion of the encoder's own cls / bottleneck
    tokens from the module-level parameters)."""
    out = []
    for name, child in model.named_children():
        if isinstance(child, forbidden):
            continue
        out += parameter_names_outside(child, forbidden, f"{prefix}{name}.")
    out += [prefix + n for n in model._parameters.keys() if "cls" not in n and "bottleneck_tokens" not in n]
    return out


def parameter_groups(model: nn.Module, hparams: dict, include_learned_tokens: bool = False) -> List[dict]:
    """madrigal/utils.py:463-613: one (no-decay, decay) pair of parameter groups per model part with its own learning
    rate — structure encoder, KG encoder, cv encoder(s), tx encoder(s), fusion (+ position encoding and projectors),
    decoder.  A parameter decays when it is outside every LayerNorm and is not a bias; the decoder always decays.

    The reference's grouping leaves the encoder's learned ``cls`` / ``tx_bottleneck_tokens`` out of every group (its
    get_parameter_names drops them and the line meant to add them back reads the wrong module, utils.py:459,478), so they
    are never updated; ``include_learned_tokens=False`` reproduces that (pinned by tests/golden/param_groups.npz),
    ``True`` trains them with the fusion group."""
    from . import models as M
    decay = {n for n in parameter_names_outside(model, (nn.LayerNorm,)) if "bias" not in n}
    named = dict(model.named_parameters())

    def owner(name: str) -> str:
        parts = name.split(".")
        mod = model
        kind = "fusion"                                   # module-level parameters ([CLS], bottleneck tokens) and glue
        for part in parts[:-1]:
            mod = getattr(mod, part) if not part.isdigit() else mod[int(part)]
            if isinstance(mod, M.GraphIsomorphismNetwork):
                return "str"
            if isinstance(mod, (M.HGT, M.HAN, M.RGCN)):
                return "kg"
            if isinstance(mod, M.TxAdaptingComPert):
                return "tx"
            if isinstance(mod, M.MLPAdaptor):
                return "fusion"
            if isinstance(mod, M.MLPEncoder):
                return "tx" if "tx_encoder" in name else "tab"
            if isinstance(mod, (M.TransformerFusion, M.PositionEncodingSinusoidal, M.PositionEncodingLearnable)):
                return "fusion"
            if isinstance(mod, M.BilinearDDIScorer):
                return "decoder"
        return kind
    lrs = {"str": hparams["structure_encoder_lr"], "kg": hparams["kg_encoder_lr"], "tab": hparams["perturb_encoders_lr"],
           "tx": hparams["perturb_encoders_lr"], "fusion": hparams["fusion_lr"], "decoder": hparams["decoder_lr"]}
    buckets: Dict[tuple, list] = {}
    for name, p in named.items():
        leaf = name.split(".")[-1]
        token = "cls" in leaf or "bottleneck_tokens" in leaf
        if token and not include_learned_tokens:
            continue
        kind = owner(name)
        wd = hparams["wd"] if (kind == "decoder" or name in decay or token) else 0.0
        buckets.setdefault((kind, wd), []).append(p)
    return [{"params": ps, "weight_decay": wd, "lr": lrs[kind]} for (kind, wd), ps in buckets.items() if ps]


def create_optimizer(model: nn.Module, hparams: dict, include_learned_tokens: bool = False) -> AdamW:
    """AdamW on the HIP path over ``parameter_groups`` (the reference's create_optimizer, madrigal/utils.py:463-613)."""
    if hparams.get("optimizer", "adamw") != "adamw":
        raise NotImplementedError("only AdamW runs on the HIP path (the reference's default, parse_args.py:135)")
    return AdamW(parameter_groups(model, hparams, include_learned_tokens), betas=(hparams["beta1"], hparams["beta2"]), eps=hparams["eps"])
