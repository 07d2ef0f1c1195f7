#!/usr/bin/env python3
"""Headline benchmark: drug-pair x outcome scores/sec for all-pairs scoring (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--drugs 4096] [--outcomes 896]
                    [--config twosides321] [--precision bf16x3] [--head-only]

One "step" = one pass of the hot path over one synthetic drug set, inputs resident in HBM:
    encode + fuse every drug (GIN structure encoder, HGT over the whole KG, cv MLP, chemCPA tx encoder,
    token assembly, fusion transformer)  ->  [all-gather of the embedding shards]  ->  symmetrise W  ->
    all-pairs bilinear head, scores materialised as the reference does ([L, N, N] fp32 raw logits).
This is BASELINE configs[1]/[3]: 4-modality model (TWOSIDES hyper-parameters, hardy_sweep_321), ~4k drugs,
~900 outcomes, KG of 1.3e5 nodes / 8e6 directed edges.  value = scores produced per second.

N GPUs: one process per GPU (torch.distributed over RCCL).  Drugs are sharded by rank for encode+fuse (the KG
encoder is per-graph and replicated), the [N/G,128] embedding blocks are all-gathered over xGMI (the path's one
exchange step), then the head is sharded by outcome:
  --scaling strong (default) = BASELINE configs[3] as named: the FIXED job `--drugs`^2 x `--outcomes` split over the
      ranks (896 / 8 = 112 outcomes per GPU), every rank writing its own slab of the score tensor;
  --scaling weak: every rank scores its own `--outcomes` outcomes (global outcomes = outcomes x n_gpus).
At N > 1 the line of the default mode also carries the other mode's numbers under "weak_scaling".  Rank 0 prints ONE
JSON line; it states the world size RCCL saw and a checksum of the all-gathered embeddings.

Also in the line (N = 1 unless noted): "roofline_cfg5" -- BASELINE configs[4], 100 352 drugs x 1 024 outcomes = 1.03e13
scores through the fp16 head with the row-statistics epilogue (nothing materialised), priced against the dense 16-bit
MFMA peak (outcome-sharded over the ranks at N > 1); "finetune" -- DDI-finetune steps/s; "ranks" -- the rank normalisation
of the score tensor the headline just produced (notebooks/normalize_scores.py) and the 5-seed ensembling on a subset;
"pretrain" -- contrastive-pretraining steps/s.

`python bench.py --gpus N` WITHOUT a launcher (no WORLD_SIZE in the environment) starts its N ranks itself: N fresh child
processes of this script, created before this process has made any GPU call, each with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR=127.0.0.1 / MASTER_PORT set; rank 0's line is this process's output and its exit code the worst child's.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3   # exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TFLOPS = 2500.0
FLOP_PER_SCORE = 256.0         # 2*D at D=128
BYTES_PER_SCORE = 4.0          # fp32 score stored
# one drug set on both sides (the all-pairs job): the symmetric sweep
HEAD_KERNEL = {"f32": "bilinear_allpairs_sym_kernel<0, 0, 8, 0>", "bf16x3": "bilinear_allpairs_sym_kernel<1, 0, 8, 1>",
               "bf16": "bilinear_allpairs_sym_kernel<2, 0, 8, 0>", "f16": "bilinear_allpairs_sym_kernel<3, 0, 8, 0>"}


def stress_leg(args, rank, world, dev, backend):
    """BASELINE configs[4]: 100 352 drugs x 1 024 outcomes, 16-bit bilinear head, row-statistics epilogue (the 20 TB score
    tensor cannot exist: per (outcome, head drug) the sum and the maximum over all tail drugs are kept).  Outcomes are
    sharded over the ranks (no collective: z is generated from the same seed on every rank).  The kernel's launch is
    bracketed by HIP events on its own stream; SURVEY 8(d): achieved = scores/s x 256 flop / 2.5e15."""
    import torch
    import torch.distributed as dist
    from madrigal_amd import ops
    N, L_all, prec = args.stress_drugs, args.stress_outcomes, args.stress_precision
    lo, hi = (rank * L_all) // world, ((rank + 1) * L_all) // world
    g = torch.Generator(device=dev).manual_seed(7)
    z = torch.randn(N, 128, device=dev, generator=g)
    w = ops.symmetrize((torch.randn(L_all, 128, 128, device=dev, generator=g) / 128 ** 0.5)[lo:hi].contiguous())
    out = torch.empty(hi - lo, N, 2, device=dev)
    ops.bilinear_allpairs(z[:8192], z[:8192], w[:4], precision=prec, epilogue=ops.EPI_ROWSTATS)          # warm-up (code load)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    ops.bilinear_allpairs(z, z, w, precision=prec, epilogue=ops.EPI_ROWSTATS, out=out)
    e1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev if backend != "gloo" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    kernel_ms = e0.elapsed_time(e1)
    # size-independent check inside the run: sum_j S[l,i,j] = z_i^T W_l (sum_j z_j) on a block of rows (fp64 on the device)
    ref = torch.einsum("id,lde,e->li", z[:256].double(), w[:2].double(), z.sum(0).double())
    err = float((out[:2, :256, 0].double() - ref).abs().max() / ref.abs().max())
    scores = float(L_all) * N * N
    ach = scores / dt * FLOP_PER_SCORE / 1e12
    return {"bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS,
            "traffic": None, "kernel": f"bilinear_allpairs_kernel<{ {'bf16': 2, 'f16': 3}[prec] }, 2, 8, 2> (row statistics, 64 head rows per wave)",
            "kernel_ms_rank0": kernel_ms, "wall_ms_max_over_ranks": dt * 1e3, "scores": scores, "scores_per_s": scores / dt, "n_gpus": world,
            "dtype": prec, "row_sum_identity_rel_err": err,
            "workload": f"BASELINE configs[4]: {N} drugs x {L_all} outcomes = {scores:.3e} scores, {prec} operands / fp32 accumulate, "
                        f"row statistics only (nothing materialised), outcomes {lo}..{hi} on rank 0" + ("" if world == 1 else f" of {world}"),
            "formula": "achieved = scores/s x 256 flop per score (2 x D, SURVEY 8d) / 1e12; peak = dense 16-bit MFMA 2.5 PFLOP/s"}


def pmc_traffic(n_drugs: int, n_outcomes: int, precision: str):
    """HBM bytes per launch of the head kernel from the committed rocprofv3 PMC passes (WRITE_SIZE + 2 x
    FETCH_SIZE, gfx950 correction), if a profile of exactly this workload exists under profiles/."""
    import glob
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "*pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        w = d.get("workload", {})
        if w.get("drugs") == n_drugs and w.get("outcomes") == n_outcomes and w.get("precision") == precision:
            for name, k in d.get("kernels", {}).items():
                if "bilinear_allpairs" in name:
                    # (bytes past the L2: for this kernel's write-once store stream that is the HBM traffic; older files call the field hbm_*)
                    return k.get("past_l2_bytes_per_launch_corrected", k.get("hbm_bytes_per_launch_corrected")), os.path.basename(f)
    return None, None


def _rank_source_sha16() -> str:
    """Hash of the CODE of csrc/ranks.hip: `//` comments and blank lines do not count (scripts/rank_pmc.sh records the same hash)."""
    import hashlib, re
    src = open(os.path.join(REPO, "madrigal_amd", "csrc", "ranks.hip"), encoding="utf-8").read()
    code = "\n".join(l for l in (re.sub(r"\s*//.*$", "", ln).rstrip() for ln in src.splitlines()) if l.strip())
    return hashlib.sha256(code.encode()).hexdigest()[:16]


def rank_pmc_traffic(n_drugs: int, msd: bool):
    """Bytes past the L2 per OUTCOME of the rank normalisation from the committed PMC passes (scripts/rank_pmc.sh: sum over its kernels of
    WRITE_SIZE + 2 x FETCH_SIZE per launch, divided by the outcomes per launch) -- a profile of THIS build only: the file records the
    path it measured (MSD / LSD) and the hash of csrc/ranks.hip; anything else gives (None, why)."""
    import glob
    want = _rank_source_sha16()
    why = "no rank PMC profile under profiles/ for this drug count"
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "*rank*pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        w = d.get("workload", {})
        if w.get("drugs") != n_drugs or d.get("hbm_bytes_per_outcome_corrected") is None:
            continue
        if w.get("ranks_hip_sha16") != want or bool(w.get("msd_path")) != msd:
            why = f"{os.path.basename(f)} measured another build / path of the sort (ranks.hip {w.get('ranks_hip_sha16')}, this build {want}): not quoted"
            continue
        return d["hbm_bytes_per_outcome_corrected"], f"committed profile {os.path.basename(f)} (same ranks.hip, same path; separate rocprofv3 --pmc passes, not this run)"
    return None, why


def host_threads() -> int:
    """Threads for the CPU baselines: the cores this process may actually use (cgroup CPU quota and affinity mask; os.cpu_count()
    reports every core of the host, and torch's index / scatter ops collapse when oversubscribed 16x)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


ORACLE_CASE = {   # name, fusion, nb, pos, heads, head_dim, ffn, layers, norm_first, agg, normalize, adapt  (configs.SHIPPED restated)
    "twosides321": ("twosides321", "transformer_uni_proj", 2, "sinusoidal", 8, 256, 1024, 2, True, "x-attn", False, False),
    "twosides105": ("twosides105", "transformer", 2, "learnable", 2, 256, 512, 2, True, "x-attn", False, False),
    "drugbank163": ("drugbank163", "transformer", 4, "sinusoidal", 8, 64, 256, 2, True, "x-attn", False, False),
}


def cpu_encode_fuse(params, batch, bkg, config: str, sample: int = 64, kg_edge_keep: int = 16):
    """The oracle's encode+fuse (eval mode; same stages as NovelDDIEncoder.encode, madrigal/models/models.py:717-896) timed on
    the host on a bounded sample: the KG encoder (per graph, not per drug) on the graph thinned to every ``kg_edge_keep``-th
    edge and scaled by that factor (its cost is edge-proportional: 209 s for the whole 8M-edge graph on 256 threads), the
    per-drug stages (GIN, cv MLP, chemCPA tx encoder, token assembly + fusion transformer) on the first ``sample`` drugs
    and scaled linearly to N."""
    import torch
    from madrigal_amd import data as D
    from madrigal_amd.pipeline import slice_batch
    from oracle import madrigal_oracle as O
    case = ORACLE_CASE[config]
    torch.set_num_threads(host_threads())
    name, fusion, nb, pos, H, dh, ffn, nl, nf, agg, normalize, adapt = case
    enc = O._sub(params, "encoder.")
    kg = bkg["data"]
    N = int(batch["drugs"].shape[0])
    thin = {k: v[:, ::kg_edge_keep].contiguous() for k, v in kg.edge_index_dict.items()}
    with torch.no_grad():
        t0 = time.perf_counter()
        O.hgt_forward(O._sub(enc, "kg_encoder."), kg.x_dict, thin, kg.node_types, kg.edge_types, num_layers=2, heads=4, hidden=128)
        t_kg = (time.perf_counter() - t0) * kg_edge_keep
        n = min(sample, N)
        b = slice_batch(batch, 0, n)
        mols = b["strs"]
        t0 = time.perf_counter()
        str_out = O.gin_forward(O._sub(enc, "str_encoder."), mols.node_feature, mols.edge_list, mols.edge_feature, mols.node2graph,
                                mols.batch_size, num_layers=4, num_mlp_layer=3)["graph_feature"]
        cv_out = O.mlp_encoder_forward(O._sub(enc, "cv_encoder."), b["cv"], 2, None, "relu", 0.2)
        sigs = torch.cat([b["tx"][c]["sigs"] for c in D.CELL_LINES])
        _, _, _, treated = O.chemcpa_predict(O._sub(enc, "tx_encoder."), sigs, torch.arange(16).repeat_interleave(n), 3, 3, with_decoder=False)
        all_embeds = torch.stack([str_out, torch.zeros_like(str_out), cv_out] + list(treated.split(n)), dim=1)
        if pos == "sinusoidal":
            enc["pos_encoder.pe"] = O.sinusoidal_pe_table(128, (D.NUM_MODALITIES if nb == 0 else D.NUM_NON_TX_MODALITIES), nb, agg)
        cfg = dict(fusion=fusion, normalize=normalize, adapt_before_fusion=adapt, pos_emb_type=pos, num_tx_bottlenecks=nb, agg=agg,
                   num_layers=nl, num_heads=H, norm_first=nf, actn="gelu", proj=dict(n_hidden=2, norm="ln", actn="relu", dropout=0.2, order="nd"))
        O.fuse_modalities(enc, all_embeds, b["masks"], cfg)
        t_drug = time.perf_counter() - t0
    return {"kg_encoder_s": t_kg, "kg_edges_kept": f"1/{kg_edge_keep}", "per_drug_stages_s_on_sample": t_drug, "sample_drugs": n,
            "encode_fuse_s_extrapolated": t_kg + t_drug * N / n}


def cpu_baseline(n_drugs: int, n_outcomes: int, seconds: float = 12.0, encode=None):
    """The oracle (CPU restatement of the reference path, pinned to the reference by tests/golden) timed on the host cores on a
    bounded sample of the same workload: the bilinear head (same torch ops as madrigal/models/models.py:539) on a block of
    head rows against all tail drugs and outcomes, and -- ``encode`` = (params, batch, bkg, config) -- encode+fuse."""
    import torch
    from oracle import madrigal_oracle as O
    cores = host_threads()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    z = torch.randn(n_drugs, 128, generator=g)
    w = torch.randn(n_outcomes, 128, 128, generator=g) / 128 ** 0.5
    rows = 64
    t0 = time.perf_counter()
    O.bilinear_scores(z[:rows], z, w)
    dt = time.perf_counter() - t0
    rows = int(min(n_drugs, max(64, rows * (seconds / 3.0) / max(dt, 1e-3))))
    rows = max(64, rows // 64 * 64)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        O.bilinear_scores(z[:rows], z, w)
        times.append(time.perf_counter() - t0)
    med = sorted(times)[1]
    head_rate = rows * n_drugs * n_outcomes / med
    out = {"value": head_rate, "unit": "scores/s", "cores": cores, "kind": "port", "head_only_scores_per_s": head_rate,
           "sample": f"oracle bilinear_scores on {rows} head rows x {n_drugs} tail drugs x {n_outcomes} outcomes, fp32 torch CPU, "
                     f"{cores} threads, median of 3, extrapolated linearly in rows"}
    if encode is not None:
        try:
            e = cpu_encode_fuse(*encode)
            total = float(n_drugs) * n_drugs * n_outcomes
            out.update(encode_fuse=e, value=total / (total / head_rate + e["encode_fuse_s_extrapolated"]))
            out["sample"] += (f"; + oracle encode+fuse: HGT on the KG thinned to {e['kg_edges_kept']} of its edges, scaled back ({e['kg_encoder_s']:.1f} s), "
                              f"GIN / cv / chemCPA / fusion on {e['sample_drugs']} drugs scaled to {n_drugs}; value = whole job (encode+fuse + all scores) per second")
        except Exception as ex:
            out["encode_fuse"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    return out


def cpu_finetune_step(params, batch, bkg, config: str, n_outcomes: int, triples, sample: int = 128, kg_edge_keep: int = 8):
    """One finetune step of the oracle on the host (torch CPU autograd over the restated forward with training-mode BatchNorm
    statistics, BCE on the gathered logits, backward; the reference's step is exactly torch autograd over these ops,
    train_ddi_batch.py:275-350), on a bounded sample: the first ``sample`` drugs on both sides with their share of the labelled
    triples, and a KG thinned to every ``kg_edge_keep``-th edge.  Scaled to the full step: per-drug stages x N / sample, KG
    encoder x kg_edge_keep (edge-proportional), head x triples / sampled triples."""
    import torch
    from madrigal_amd import data as D
    from madrigal_amd.pipeline import slice_batch
    from oracle import madrigal_oracle as O
    from oracle.pipeline import oracle_encoders
    case = ORACLE_CASE[config]
    torch.set_num_threads(host_threads())
    name, fusion, nb, pos, H, dh, ffn, nl, nf, agg, normalize, adapt = case
    N = int(batch["drugs"].shape[0])
    n = min(sample, N)
    b = slice_batch(batch, 0, n)
    kg = bkg["data"]
    thin = D.KGData(kg.x_dict, {k: v[:, ::kg_edge_keep].contiguous() for k, v in kg.edge_index_dict.items()}, list(kg.node_types), list(kg.edge_types))
    lab, hd, tl, y = triples
    keep = ((hd < n) & (tl < n)).nonzero().flatten()[: 20_000]          # (w[lab] materialises 64 KB per sampled triple on the host)
    lab, hd, tl, y = lab[keep], hd[keep], tl[keep], y[keep]
    pr = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in params.items()}
    enc = O._sub(pr, "encoder.")
    if pos == "sinusoidal":
        enc["pos_encoder.pe"] = O.sinusoidal_pe_table(128, (D.NUM_MODALITIES if nb == 0 else D.NUM_NON_TX_MODALITIES), nb, agg)
    cfg = dict(fusion=fusion, normalize=normalize, adapt_before_fusion=adapt, pos_emb_type=pos, num_tx_bottlenecks=nb, agg=agg,
               num_layers=nl, num_heads=H, norm_first=nf, actn="gelu", proj=dict(n_hidden=2, norm="ln", actn="relu", dropout=0.2, order="nd"))
    filler = torch.zeros(max(int(b["drugs"].max()) + 1, int(bkg["drug_index_map"].max()) + 1), 128)
    t = {}
    with O.batch_statistics():
        t0 = time.perf_counter()
        kg_valid = O.hgt_forward(O._sub(enc, "kg_encoder."), thin.x_dict, thin.edge_index_dict, thin.node_types, thin.edge_types,
                                 num_layers=2, heads=4, hidden=128)["drug"]
        t["kg_fwd"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        zs = []
        for _side in range(2):                         # head side and tail side, as the reference encodes them
            mols = b["strs"]
            str_out = O.gin_forward(O._sub(enc, "str_encoder."), mols.node_feature, mols.edge_list, mols.edge_feature, mols.node2graph,
                                    mols.batch_size, num_layers=4, num_mlp_layer=3)["graph_feature"]
            kg_out = O.place_kg_rows(kg_valid, bkg["drug_index_map"], b["drugs"], filler)
            cv_out = O.mlp_encoder_forward(O._sub(enc, "cv_encoder."), b["cv"], 2, None, "relu", 0.2)
            sigs = torch.cat([b["tx"][c]["sigs"] for c in D.CELL_LINES])
            _, _, _, treated = O.chemcpa_predict(O._sub(enc, "tx_encoder."), sigs, torch.arange(16).repeat_interleave(n), 3, 3, with_decoder=False)
            all_embeds = torch.stack([str_out, kg_out, cv_out] + list(treated.split(n)), dim=1)
            zs.append(O.fuse_modalities(enc, all_embeds, b["masks"], cfg))
        t["encode_fwd"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        w = O.symmetric(pr["decoder.parametrizations.weight.original"])
        s = torch.einsum("td,tde,te->t", zs[0][hd], w[lab], zs[1][tl])
        loss = torch.nn.BCELoss()(torch.sigmoid(s), y)
        t["head_fwd"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    loss.backward()
    t["backward"] = time.perf_counter() - t0
    fwd = t["kg_fwd"] + t["encode_fwd"] + t["head_fwd"]
    bwd_over_fwd = t["backward"] / max(fwd, 1e-9)
    T_full = int(triples[0].numel())
    est = ((t["kg_fwd"] * kg_edge_keep + t["encode_fwd"] * N / n + t["head_fwd"] * T_full / max(int(keep.numel()), 1)) * (1.0 + bwd_over_fwd))
    return {"step_s_extrapolated": est, "steps_per_s": 1.0 / est, "measured_s": t, "sample_drugs": n, "sample_triples": int(keep.numel()),
            "kg_edges_kept": f"1/{kg_edge_keep}", "loss": float(loss.detach())}


def f32_exact_leg(model, z, scores, enc_ms, args):
    """The exact-fp32 head (v_mfma_f32_32x32x2_f32, no operand splitting) beside the headline's mode, on the embeddings the headline
    just encoded: kernel time by HIP events on the launch stream, the whole-job rate it would give (the headline's encode+fuse time +
    this head), its share of the 157 TFLOP/s fp32 matrix peak -- nominal (256 flop per score) and as executed (the symmetric sweep
    computes the tiles on / right of the block diagonal: (nb + 1) / (2 nb) of them) -- and the per-entry picture of the headline's
    mode against it on a seeded sample of entries (both tensors live in HBM at once: 2 x 60 GB at the default shape)."""
    import torch
    from madrigal_amd import models as M, ops
    L, N, _ = scores.shape
    out = ops.empty_scores(L, N, N, scores.device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    times = []
    with torch.no_grad(), M.precision("f32"):
        model.decoder(z, z, (0, L), out=out)                      # warm-up
        torch.cuda.synchronize()
        for _ in range(3):
            e0.record()
            model.decoder(z, z, (0, L), out=out)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
    ms = sum(times) / len(times)
    per_launch = float(L) * N * N
    nb = -(-N // 256)
    executed = (nb + 1) / (2.0 * nb)
    tf = per_launch * FLOP_PER_SCORE / (ms * 1e-3) / 1e12
    g = torch.Generator(device=scores.device).manual_seed(17)
    n_s = 4_000_000
    li = torch.randint(0, L, (n_s,), device=scores.device, generator=g)
    ii = torch.randint(0, N, (n_s,), device=scores.device, generator=g)
    jj = torch.randint(0, N, (n_s,), device=scores.device, generator=g)
    ref, got = out[li, ii, jj].double(), scores[li, ii, jj].double()
    d, rms = (got - ref).abs(), float(ref.pow(2).mean().sqrt())
    res = {"kernel": HEAD_KERNEL["f32"], "kernel_ms": ms, "kernel_ms_runs": times, "whole_job_scores_per_s": per_launch / ((enc_ms + ms) * 1e-3),
           "head_only_scores_per_s": per_launch / (ms * 1e-3), "tflops_nominal": tf, "tflops_executed": tf * executed,
           "frac_of_fp32_mfma_peak_nominal": tf / MFMA_F32_PEAK_TFLOPS, "frac_of_fp32_mfma_peak_executed": tf * executed / MFMA_F32_PEAK_TFLOPS,
           "executed_share_of_tiles": executed, "peak_tflops": MFMA_F32_PEAK_TFLOPS,
           "headline_mode_vs_exact": {"mode": args.precision, "sampled_entries": n_s, "rms_score": rms,
                                      "max_abs_diff_over_max_abs": float(d.max() / ref.abs().max()),
                                      "share_within_1e-4_rel_plus_1e-4_rms": float((d <= 1e-4 * ref.abs() + 1e-4 * rms).double().mean()),
                                      "share_within_1e-4_rel": float((d <= 1e-4 * ref.abs()).double().mean()),
                                      "median_rel_diff": float((d / ref.abs().clamp_min(1e-30)).median())},
           "what": "same embeddings, same W_sym, exact fp32 products; encode+fuse time taken from the headline's timed steps"}
    del out, ref, got, d
    torch.cuda.empty_cache()
    return res


def ranks_leg(scores, args, model=None, z=None):
    """The product of the reference's scoring job is the normalised-rank tensor (notebooks/normalize_scores.py:36-85; README.md:43):
    per outcome, the strict lower triangle of the [N,N] score slice ranked (1-based, ascending), divided by N(N-1)/2, mirrored.
    Here: ``ops.rank_normalize`` over the outcomes the headline just scored (HIP events on the launch stream; the sort scratch is
    bounded, so the call walks the outcomes in chunks), then the 5-seed ensembling of generate_embeddings.ipynb cells 18-20
    (geometric mean of five rank tensors, ranked again) on a stated subset.  Algorithmic bytes per outcome: the M = N(N-1)/2 kept
    scores read once (4 B each) and the N^2 normalised ranks written once (4 B each); the radix sort between them moves more
    (DESIGN.md 4), which is exactly what the roofline fraction is there to show."""
    import torch
    from madrigal_amd import ops
    L_all, N, _ = scores.shape
    L = L_all if args.rank_outcomes < 0 else min(L_all, args.rank_outcomes)
    s = scores[:L]
    out = ops.empty_scores(L, N, N, s.device)
    ops.rank_normalize(s[:2], out=out[:2])                               # warm-up (code load, workspace)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    passes, walls = [], []
    for _ in range(2):            # two passes over all outcomes, the faster one reported: the first touches 60 GB of fresh rank tensor
        t0 = time.perf_counter()
        e0.record()
        flags = []
        ops.rank_normalize(s, out=out, fallback_flags=flags)
        e1.record()
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
        passes.append(e0.elapsed_time(e1))
    ms, wall = min(passes), min(walls)
    handed = int(sum(int((f != 0).sum()) for f in flags))
    rank_traffic, rank_traffic_src = rank_pmc_traffic(N, bool(flags))
    M = N * (N - 1) // 2
    # size-independent checks inside the run: a permutation of 1..M per outcome (sum of ranks), symmetric, zero diagonal
    denom = N * (N - 1) / 2
    chk = out[: min(L, 4)].double()
    want = M * (M + 1) / 2
    perm_err = float(((chk.sum(dim=(1, 2)) / 2 * denom - want).abs() / want).max())
    sym = bool(torch.equal(out[0], out[0].T)) and float(out[0].diagonal().abs().max()) == 0.0
    alg = (M * 4.0 + N * N * 4.0) * L
    res = {"metric": "rank-normalised scores/sec (normalize_scores)", "value": float(L) * N * N / (ms * 1e-3), "unit": "scores/s", "outcomes": L, "drugs": N,
           "ms_total": ms, "passes_ms": passes, "ms_per_outcome": ms / L, "wall_ms": wall * 1e3, "dtype": "u32 order-preserving keys of fp32 scores, u32 ranks, f64 divide -> f32",
           "checks": {"rank_sum_rel_err": perm_err, "symmetric_zero_diagonal": sym},
           "roofline": {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic": None if rank_traffic is None else rank_traffic * L, "traffic_source": rank_traffic_src,
                        "kernel": "mdg_rank_normalize (extract + 4 x 8-bit stable LSD radix passes on LDS-sorted 16384-key tiles + blocked rank store)" if not flags else
                                  "mdg_rank_normalize: exact-layout MSD path (sampled bucket table -> bucket counts per 128 x 128 output block + scan -> one partition -> in-LDS "
                                  "counting sort per 8192-key bucket, one word per key in place -> output blocks gathered through the layout tables); outcomes it hands back "
                                  "(point masses) take the LSD passes",
                        "outcomes_handed_to_lsd": handed if flags else None,
                        "algorithmic_bytes_per_outcome": alg / L, "formula": "(M x 4 B keys read + N^2 x 4 B ranks written) x outcomes / launch time"}}
    del chk
    # seed ensembling on a subset: 5 rank tensors of `sub` outcomes each (disjoint outcome slices of this run stand in for the seeds)
    sub = max(1, min(8, L // 5))
    seeds = [out[k * sub:(k + 1) * sub] for k in range(5)] if L >= 5 else [out[:1]] * 5
    ops.ensemble_ranks([t_[:1] for t_ in seeds])
    torch.cuda.synchronize()
    e0.record()
    ens = ops.ensemble_ranks(seeds)
    e1.record()
    torch.cuda.synchronize()
    res["ensemble"] = {"seeds": 5, "outcomes": int(ens.shape[0]), "ms": e0.elapsed_time(e1), "ms_per_outcome": e0.elapsed_time(e1) / int(ens.shape[0]),
                       "what": "gmean of 5 normalised-rank tensors then rank_normalize (generate_embeddings.ipynb cells 18-20)"}
    del ens
    if model is not None and z is not None and L == L_all:
        # the same product without the score tensor: the head writes the order keys of the strict lower triangle only (EPI_TRIKEYS: half
        # the store stream) into the rank tensor's own memory and the sort puts the ranks over them (pipeline.rank_all_pairs)
        from madrigal_amd import models as M
        from madrigal_amd.pipeline import rank_all_pairs
        ref = out[:2].clone()
        with torch.no_grad(), M.precision(args.precision):
            dec = model.decoder
            keys = out.view(torch.int32)
            dec(z, z, (0, 2), epilogue=ops.EPI_TRIKEYS, out=keys[:2])          # warm-up
            torch.cuda.synchronize()
            ek = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ek[0].record()
            dec(z, z, (0, L), epilogue=ops.EPI_TRIKEYS, out=keys)
            ek[1].record()
            got = ops.rank_normalize(keys)
            ek[2].record()
            torch.cuda.synchronize()
        res["from_head_keys"] = {"head_keys_ms": ek[0].elapsed_time(ek[1]), "ranks_ms": ek[1].elapsed_time(ek[2]),
                                 "ms_per_outcome_head_plus_ranks": ek[0].elapsed_time(ek[2]) / L,
                                 "equal_to_ranks_of_materialised_scores": bool(torch.equal(got[:2], ref)),
                                 "what": "head with the lower-triangle-keys epilogue (half the store stream, no score tensor) + rank_normalize over the keys in place"}
        del got, keys, ref
    del out
    if not args.no_cpu_baseline:
        try:
            import multiprocessing as mp
            from oracle.bench_workers import rank_one
            cores = host_threads()
            n_proc = max(1, min(cores, 16))
            ctx = mp.get_context("spawn")
            t0 = time.perf_counter()
            with ctx.Pool(n_proc) as pool:
                dts = pool.map(rank_one, [(100 + i, N) for i in range(n_proc)])
            wall_pool = time.perf_counter() - t0
            per = sorted(d for d, _ in dts)[len(dts) // 2]
            res["cpu_baseline"] = {"value": n_proc * float(N) * N / max(d for d, _ in dts), "unit": "scores/s", "cores": n_proc, "kind": "port",
                                   "seconds_per_outcome_one_core": per, "pool_wall_s_incl_spawn": wall_pool,
                                   "sample": f"the reference's run_slice (interval 1: mask, one default argsort + inverse permutation over all N^2 entries, divide, mirror) "
                                             f"restated in numpy, {n_proc} outcomes of {N} x {N} on {n_proc} processes (one outcome per process, as its Pool().map); "
                                             f"value = {n_proc} x N^2 / slowest worker's time"}
        except Exception as e:
            res["cpu_baseline"] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
    return res


def finetune_leg(model, batch, bkg, filler, N, L, args, rank=0, world=1, backend="nccl", precision="bf16"):
    """DDI-finetune steps/s on the same model and batch (train_ddi_batch.py:275-350, 'full_full' mode): zero_grad ->
    encode head and tail side (training mode: dropout, BatchNorm batch statistics) -> scores of the labelled triples
    (gathered head) -> BCE -> backward through every encoder -> AdamW over the reference's parameter groups.
    ``precision``: BASELINE configs[1] names bf16 for this step -- GEMM operands rounded to bf16, fp32 accumulation, fp32 master
    weights and optimizer state, the gathered head fp32-grade on the split-bf16 matrix cores; "bf16x3" is the fp32-grade arithmetic of the
    inference headline.
    world > 1: the SAME step data-parallel (strong scaling): drug-sharded encoders with SyncBatchNorm, all-gather of the
    embeddings, triples dealt to the ranks, flat all-reduce of the gradients (madrigal_amd/train.py)."""
    import torch
    import torch.distributed as dist
    from madrigal_amd import data as D, models as M, ops
    from madrigal_amd.optim import create_optimizer
    from madrigal_amd.train import FinetuneStep
    dev = batch["cv"].device
    if world > 1:                                      # the headline gave every rank its own outcomes: one model again
        with torch.no_grad():
            model.decoder.parametrizations.weight.original.copy_(
                torch.randn(L, 128, 128, generator=torch.Generator().manual_seed(1000)) / 128 ** 0.5)
    # What changes from step to step in the reference's loop (train_ddi_batch.py:231-354) changes here: a DIFFERENT set of labelled
    # triples (the triple plan -- sorts by label / pair / drug -- is rebuilt inside every timed step), a DIFFERENT head batch and tail
    # batch (distinct drug sets: other molecules, signatures, KG row orders) and freshly drawn modality masks on each side, so that no
    # plan keyed on a batch or mask object survives from one step to the next (with one batch object on both sides every
    # such cache always hit).  `fresh_sides = False` (the "cached_plans" figure beside the value) is round 4's step: the fixed batch.
    sets = [tuple(t.to(dev) for t in D.make_labelled_triples(N, L, args.finetune_triples, s_)) for s_ in range(3)]
    lab, hd, tl, y = sets[0]
    T = int(lab.numel())
    hp = dict(optimizer="adamw", structure_encoder_lr=1e-5, kg_encoder_lr=1e-5, perturb_encoders_lr=1e-5, fusion_lr=1e-6, decoder_lr=1e-4,
              wd=1e-2, beta1=0.9, beta2=0.999, eps=1e-8)
    avail = batch["masks"].cpu()
    n_var = 2 if world == 1 else 0                     # (the data-parallel leg keeps the fixed batch: its shards are cut once)
    sides = []
    for v_ in range(n_var):
        pair = []
        for side_seed in (200 + v_, 300 + v_):
            b_, _ = D.make_batch(N, side_seed, kg=bkg["data"], masks=avail)
            pair.append(D.batch_to(b_, dev))
        sides.append(pair)
    g_m = torch.Generator(device=dev).manual_seed(99)
    avail_dev = avail.to(dev)

    def draw_masks():
        """A step's modality masks, drawn on the device: every available modality but the structure dropped with probability 0.3 (the
        'random_sample' family of train_ddi_batch.py; True = absent)."""
        drop = torch.rand(avail_dev.shape, generator=g_m, device=dev) < 0.3
        drop[:, 0] = False
        return avail_dev | drop
    fs = FinetuneStep(model, create_optimizer(model, hp), rank=rank, world=world)
    torch.manual_seed(4321 + rank)
    def one_step(i_, fresh):
        if fresh and sides:
            # (preparing the next step's mask and triple plans on a side stream while this step runs was tried twice in round 5 -- with
            # the torch-built triple plan: 42.3 against 41.8 ms; masks alone, after the plan moved to csrc/plan.hip: 42.1 against 42.4 --
            # and removed: the fresh step is bound by its kernels (41 ms of them; two different molecule batches and mask patterns per
            # step are more work than one shared batch), not by the host reads of its plans)
            hb, tb = sides[i_ % n_var]
            return fs.step(hb, tb, draw_masks(), draw_masks(), bkg, *sets[(i_ + 1) % 3], kg_filler=filler)
        return fs.step(batch, batch, batch["masks"], batch["masks"], bkg, *sets[(i_ + 1) % 3], kg_filler=filler)

    def timed(fresh):
        # warm-up: one step per triple set -- each set has its own tensor sizes (pair counts, tile tables), whose first allocation from the
        # device is slow (a 45-ms step then takes 60-100 ms); afterwards the caching allocator holds them
        ls = [float(one_step(k_, fresh)) for k_ in (0, 1, 2)][-1:]
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        mk = [torch.cuda.Event(enable_timing=True) for _ in range(args.finetune_steps + 1)]
        t_ = time.perf_counter()
        mk[0].record()
        for i_ in range(args.finetune_steps):
            ls.append(one_step(i_, fresh))
            mk[i_ + 1].record()                      # (no host sync: the steps stay back to back, the host queues ahead)
        torch.cuda.synchronize()
        return ls, mk, t_
    with M.precision(precision):
        cached = None
        if sides:
            ls_c, mk_c, _ = timed(False)
            ms_c = sorted(mk_c[i_].elapsed_time(mk_c[i_ + 1]) for i_ in range(args.finetune_steps))
            cached = {"ms_per_step": ms_c[len(ms_c) // 2], "what": "the same step with ONE fixed batch object on both sides and fixed masks (round 4's figure): "
                                                                   "mask / gather / molecule plans hit their caches"}
        losses, marks, t0 = timed(bool(sides))
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    step_ms = [marks[i_].elapsed_time(marks[i_ + 1]) for i_ in range(args.finetune_steps)]
    if world > 1:
        tmax = torch.tensor([dt], device=dev if backend != "gloo" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    dt /= args.finetune_steps
    # One stalled step (seen once in ten runs on the shared GPU boxes: 2.4 s for a 54-ms step, nothing in the step's own kernels)
    # would set the mean of so few steps; the figure of merit is the median of the individually stamped steps, the mean of the timed
    # region and every step's own time are reported beside it.
    med = sorted(step_ms)[len(step_ms) // 2] * 1e-3
    mean_dt, dt = dt, (med if world == 1 else dt)
    out = {"metric": "DDI-finetune steps/sec", "value": 1.0 / dt, "unit": "steps/s", "ms_per_step": dt * 1e3, "n_gpus": world,
           "scaling": "strong", "steps": args.finetune_steps, "warmup": 3, "triples_per_step": T, "drugs": N, "outcomes": L,
           "timing": "median of the steps' own times (HIP events between back-to-back steps)" if world == 1 else "timed region / steps, max over ranks",
           "step_ms": step_ms, "ms_per_step_mean_of_timed_region": mean_dt * 1e3,
           "dtype": {"bf16": "bf16 GEMM operands, fp32 accumulate / master weights / optimizer; gathered head fp32-grade (split-bf16 products)", "bf16x3": "f32 via split-bf16 (bf16x3) MFMA",
                     "f32": "f32"}[precision],
           "loss_first_last": [float(losses[0]), float(losses[-1])],
           "parallelism": "single GPU" if world == 1 else
           f"drug-sharded encoders (SyncBatchNorm), all-gather(z) / reduce-scatter(dz), triples dealt to {world} ranks, flat gradient all-reduce",
           "work": "optimizer.zero_grad, triple plan of a fresh set of labelled triples, encode+fuse a fresh head batch and a fresh tail batch under "
                   "freshly drawn modality masks (training mode), gathered bilinear head on the labelled triples, BCE, backward through all "
                   "encoders, AdamW step" if sides else
                   "optimizer.zero_grad, triple plan of a fresh set of labelled triples, encode+fuse head side and tail side of the fixed batch (training "
                   "mode), gathered bilinear head on the labelled triples, BCE, backward through all encoders, AdamW step",
           "fresh_batches_and_masks_every_step": bool(sides), "cached_plans": cached}
    if rank == 0 and world == 1:
        # roofline of the step's dominant kernel, the wide fusion-transformer GEMM (linear_kernel<bf16, 256x256 tile>; 14 of the
        # 69 ms of kernel time, profiles/): one FFN-sized launch at the step's own row count, HIP events on its stream
        # live fusion tokens of BOTH sides: the step runs them through the transformer in one pass (NovelDDIMultilabel.embed)
        live = 2 * (int((~batch["masks"]).sum()) + int(model.encoder.num_tx_bottlenecks) * N)
        d = int(model.encoder.transformer.embed2latent.weight.shape[0])
        x = torch.randn(live, d, device=dev)
        w = torch.randn(d, d, device=dev) * d ** -0.5
        with torch.no_grad():
            def timed(fn, reps=7):
                for _ in range(2):
                    fn()
                ts = []
                for _ in range(reps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                return sorted(ts)[len(ts) // 2]
            img = ops.pack_operand(x, precision)
            if img is not None and d % 64 == 0:
                # the kernel alone: in the step its x arrives packed by the producer (LayerNorm / the previous block's epilogue)
                ms = timed(lambda: ops.linear_packed(img, live, w, precision=precision))
            else:
                ms = timed(lambda: ops.linear(x, w, precision=precision))
            ms_call = timed(lambda: ops.linear(x, w, precision=precision, cache_weight=False))       # + the operand pre-pass of x and w
        ach = 2.0 * live * d * d / (ms * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS,
                           "traffic": None, "kernel": "linear_pp_kernel (256 x 256 ping-pong dense block, wide fusion-transformer GEMM)", "kernel_ms": ms,
                           "with_operand_prepass_ms": ms_call, "with_operand_prepass_tflops": 2.0 * live * d * d / (ms_call * 1e-3) / 1e12,
                           "shape": [live, d, d], "formula": "2 M N K flop / kernel launch time (median of 7, HIP events); peak = dense bf16 MFMA"
                           + ("; the split-bf16 mode issues 3 products per element" if precision == "bf16x3" else "")}
    return out


def pretrain_leg(model, bkg, masks_all, args, precision="bf16x3"):
    """BASELINE configs[2] as the reference ships it (configs/cl_pretrain/*.yaml: raw_encoder_output, 'str_center_uni' views,
    separate predictors, T = 0.1, mlp_dim 512): one iteration of pretrain.py:59-93 = fresh view draw on the host, both views
    through the encoders -> uni_projector -> predictors, InfoNCE, backward, AdamW, on a 2048-drug batch over the bench's KG."""
    import numpy as np
    import torch
    from madrigal_amd import data as D, masks as MK, models as M
    from madrigal_amd.optim import AdamW
    from madrigal_amd.simclr import SimCLR_NovelDDI
    from madrigal_amd.train import PretrainStep
    B = args.pretrain_batch
    dev = bkg["drug_index_map"].device
    # the first B drugs of the bench's drug table: their KG membership is a property of the bench's KG (drug_index_map)
    avail = masks_all[:B].clone().cpu()
    avail[:, 2] = torch.where(avail[:, 1:].all(dim=1), torch.zeros(B, dtype=torch.bool), avail[:, 2])       # every drug owns a second modality
    bkg_cpu = {"data": bkg["data"].to("cpu"), "drug_index_map": bkg["drug_index_map"].cpu()}
    batch, _ = D.make_batch(B, 0, kg=bkg_cpu["data"], masks=avail)
    np.random.seed(0)
    sim = SimCLR_NovelDDI(model.encoder, dim=128, mlp_dim=512, T=0.1, raw_encoder_output=True).to(dev).train()
    b = D.batch_to(batch, dev)
    kgc = bkg
    draw = MK.StrCenterUniSampler(MK.get_pretrain_masks(list(range(B)), avail.numpy().astype(np.int64), "str_center_uni", False, 0.2))
    from madrigal_amd.optim import PretrainSchedule
    # pretrain.py:65: the rate is set per iteration (madrigal/utils.py:680-692); a 100-iteration epoch, one warm-up epoch of ten
    sched = PretrainSchedule(lr=1e-5, warmup_epochs=1, num_epochs=10, iters_per_epoch=100, start_epoch=1)
    step = PretrainStep(sim, AdamW(sim.parameters(), lr=1e-5, weight_decay=1e-2), scheduler=sched)
    data = (b["strs"], kgc, b["cv"], b["tx"])
    losses = []
    import gc
    with M.precision(precision):
        for i in range(4 + args.pretrain_steps):
            if i == 4:
                torch.cuda.synchronize()
                gc.collect()
                gc.freeze()                                  # the earlier legs' objects out of the collector's way: this loop is host-bound
                t0 = time.perf_counter()
            m1, m2 = draw(range(B))                          # host tensors: the step uploads them (madrigal_amd/hostio.py)
            losses.append(step.step(batch["drugs"], m1, m2, None, data))
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.pretrain_steps
    gc.unfreeze()
    out = {"metric": "contrastive-pretraining steps/sec", "value": 1.0 / dt, "unit": "steps/s", "ms_per_step": dt * 1e3, "drugs_per_s": B / dt,
           "n_gpus": 1, "steps": args.pretrain_steps, "warmup": 4, "batch": B, "dtype": "f32 via split-bf16 (bf16x3) MFMA",
           "loss_first_last": [float(losses[0]), float(losses[-1])], "lr_last": sched.last_lr,
           "config": {"workload": "BASELINE configs[2] as shipped: SimCLR_NovelDDI(raw_encoder_output=True), 'str_center_uni' views drawn per "
                                  "iteration on the host, separate predictors, T=0.1, mlp_dim=512, AdamW with the per-iteration "
                                  "warm-up / cosine rate of pretrain.py:65; the bench's KG"}}
    # roofline of the step's dominant kernel family, the KG edge attention (forward + two backward kernels: a quarter of the step's
    # kernel time, profiles/): the forward launch over every destination type of the first conv, HIP events on its own stream.
    # per edge one 512-byte key row and one 512-byte value row are gathered plus the 8-byte source index; per
    # destination node one 512-byte query row is read and one 512-byte row written.
    try:
        conv = model.encoder.kg_encoder.convs[0]
        probe = {"start": torch.cuda.Event(enable_timing=True), "end": torch.cuda.Event(enable_timing=True)}
        ms_all = []
        sim.eval()
        with torch.no_grad(), M.precision(precision):
            for _ in range(4):
                conv.__dict__["_attention_probe"] = probe
                conv(kgc["data"].x_dict, kgc["data"].edge_index_dict)
                torch.cuda.synchronize()
                ms_all.append(probe["start"].elapsed_time(probe["end"]))
        conv.__dict__.pop("_attention_probe", None)
        ms = sorted(ms_all[1:])[1]
        # bytes that must cross HBM once: the projection buffer (queries, keys', values' of every node), the edge list, the output
        # rows; the rows GATHERED per edge (two 512-byte rows + the 8-byte index) are mostly re-reads served by L2 / the
        # Infinity Cache -- both rates are reported
        unique = probe["buffer_floats"] * 4.0 + probe["edges"] * 8.0 + probe["dst_rows"] * 512.0
        gathered = probe["edges"] * (1024.0 + 8.0) + probe["dst_rows"] * 1024.0
        out["roofline"] = {"bound": "hbm", "achieved": unique / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": unique / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "traffic": None, "kernel": "hgt_attention_kernel (edge softmax + weighted value sum, all destination types of one conv)",
                           "kernel_ms": ms, "edges": probe["edges"], "destination_rows": probe["dst_rows"], "gathered_row_rate_GBps": gathered / (ms * 1e-3) / 1e9,
                           "algorithmic_bytes": unique,
                           "formula": "(projection buffer once + edges x 8 B + destination rows x 512 B) / launch time; gathered_row_rate = "
                                      "(edges x (2 x 512 B + 8 B) + destination rows x 1024 B) / launch time (cache-served re-reads included)"}
    except Exception as e:
        out["roofline"] = {"bound": "hbm", "achieved": None, "error": f"{type(e).__name__}: {e}"[:300]}
    finally:
        sim.train()
    if not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_pretrain_step({k: v.detach().cpu() for k, v in sim.state_dict().items()}, batch, bkg_cpu, avail, B)
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
    return out


def cpu_pretrain_step(params, batch, bkg, avail, B: int, sample: int = 96, kg_edge_keep: int = 16):
    """One contrastive-pretraining iteration of the oracle on the host (oracle.pipeline.oracle_simclr under training-mode batch
    statistics + torch CPU autograd: pretrain.py:59-93 is exactly that over the reference's modules) on a bounded sample: the first
    ``sample`` drugs, one 'str_center_uni' view pair, the KG thinned to every ``kg_edge_keep``-th edge.  Scaled to the full step:
    the per-drug stages x B / sample, the KG encoder (run once per view, as the reference does) x kg_edge_keep."""
    import numpy as np
    import torch
    from madrigal_amd import data as D, masks as MK
    from madrigal_amd.pipeline import slice_batch
    from oracle import madrigal_oracle as O
    from oracle.pipeline import oracle_simclr
    torch.set_num_threads(host_threads())
    n = min(sample, B)
    b = slice_batch(batch, 0, n)
    kg = bkg["data"]
    thin = {"data": D.KGData(kg.x_dict, {k: v[:, ::kg_edge_keep].contiguous() for k, v in kg.edge_index_dict.items()}, list(kg.node_types), list(kg.edge_types)),
            "drug_index_map": bkg["drug_index_map"]}
    np.random.seed(1)
    draw = MK.StrCenterUniSampler(MK.get_pretrain_masks(list(range(n)), avail[:n].numpy().astype(np.int64), "str_center_uni", False, 0.2))
    m1, m2 = draw(range(n))
    pr = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in params.items()}
    filler = torch.zeros(max(int(b["drugs"].max()) + 1, int(bkg["drug_index_map"].max()) + 1), 128)
    # the KG encoder alone (twice per step in the reference: once per view) to split the time
    enc = O._sub({"encoder." + k: v for k, v in O._sub(pr, "base_encoder.").items()}, "encoder.")
    with torch.no_grad():
        t0 = time.perf_counter()
        O.hgt_forward(O._sub(enc, "kg_encoder."), thin["data"].x_dict, thin["data"].edge_index_dict, thin["data"].node_types, thin["data"].edge_types,
                      num_layers=2, heads=4, hidden=128)
        t_kg = time.perf_counter() - t0
    t0 = time.perf_counter()
    with O.batch_statistics():
        ref = oracle_simclr(pr, b, thin, m1, m2, None, 0.1, filler)
    t_fwd = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref["loss"].backward()
    t_bwd = time.perf_counter() - t0
    ratio = t_bwd / max(t_fwd, 1e-9)
    kg_fwd = 2 * t_kg
    est = ((t_fwd - kg_fwd) * B / n + kg_fwd * kg_edge_keep) * (1.0 + ratio)
    return {"value": 1.0 / est, "unit": "steps/s", "cores": host_threads(), "kind": "port",
            "sample": f"oracle SimCLR step (torch CPU autograd, fp32) on {n} drugs, one view pair, KG thinned to 1/{kg_edge_keep} of its edges; "
                      f"scaled to {B} drugs and the full KG (per-drug stages x {B / n:.1f}, KG encoder x {kg_edge_keep})",
            "detail": {"forward_s": t_fwd, "backward_s": t_bwd, "kg_forward_once_s": t_kg, "step_s_extrapolated": est, "loss": float(ref["loss"].detach())}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--drugs", type=int, default=4096)
    ap.add_argument("--outcomes", type=int, default=896)
    ap.add_argument("--config", default="twosides321", choices=["twosides321", "twosides105", "drugbank163"])
    ap.add_argument("--precision", default="bf16x3", choices=["f32", "bf16x3", "bf16"])
    ap.add_argument("--kg-nodes", type=int, default=130000)
    ap.add_argument("--kg-edges", type=int, default=8000000)
    ap.add_argument("--head-only", action="store_true", help="time the scoring stage alone (embeddings given)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32-exact", action="store_true", help="skip the exact-fp32 head beside the headline's mode (N = 1)")
    ap.add_argument("--pretrain-steps", type=int, default=20, help="contrastive-pretraining leg (BASELINE configs[2]); 0 disables; single GPU only")
    ap.add_argument("--pretrain-batch", type=int, default=2048)
    ap.add_argument("--ddp-legs", action="store_true", help="N > 1: after the line, also run the data-parallel finetune step and the sharded cfg5 "
                    "run (untested over RCCL; results to stderr and gpurun_out/bench_ddp_legs_n<N>.json)")
    ap.add_argument("--finetune-steps", type=int, default=5, help="second half of BASELINE's metric: DDI-finetune steps/s "
                    "(encode both sides + gathered head + BCE + backward + AdamW), timed at N=1 after the headline; 0 = skip")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="N > 1: strong = the fixed drugs^2 x outcomes job "
                    "split by outcome over the ranks (BASELINE configs[3]); weak = --outcomes per rank")
    ap.add_argument("--stress-drugs", type=int, default=100_352, help="BASELINE configs[4] row-statistics run; 0 = skip")
    ap.add_argument("--stress-outcomes", type=int, default=1024)
    ap.add_argument("--stress-precision", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--finetune-precision", default="bf16", choices=["bf16", "bf16x3", "f32"], help="BASELINE configs[1] names bf16 for the "
                    "finetune step; the line also carries the bf16x3 (fp32-grade) step under finetune.fp32_grade")
    ap.add_argument("--rank-outcomes", type=int, default=-1, help="ranks leg (N = 1): rank-normalise this many outcomes of the score tensor the "
                    "headline produced; -1 = all of them, 0 = skip")
    ap.add_argument("--finetune-triples", type=int, default=1_000_000, help="positive triples; with 2 negatives each and both "
                    "directions (the reference's collation) 6x as many labelled triples per step")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: be the launcher.  This process has not touched the GPU (torch is not even imported yet) and never will:
        # it starts one child per rank -- fresh processes, nothing is exec'ed from a process that initialised HIP -- and waits.
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        procs = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=None if r == 0 else subprocess.DEVNULL))
        codes = [p_.wait() for p_ in procs]
        raise SystemExit(max(abs(c) for c in codes))

    import torch
    import torch.distributed as dist
    from madrigal_amd import configs, data as D, models as M, ops
    from madrigal_amd.parallel import all_gather_rows, shard_range
    from madrigal_amd.pipeline import generate_embeddings

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # MDG_BENCH_BACKEND=gloo: functional rehearsal with several ranks on ONE card (no RCCL); never used for numbers
    backend = os.environ.get("MDG_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    N, L = args.drugs, args.outcomes
    M.set_precision(args.precision)
    model = batch = bkg = filler = cpu_inputs = None
    if not args.head_only:
        # same synthetic batch on every rank (seeded); each rank encodes only its drug block
        batch, bkg = D.make_batch(N, 0, kg_nodes=args.kg_nodes, kg_edges=args.kg_edges)
        torch.manual_seed(1234)                       # identical encoder weights on every rank
        model = configs.build_model(args.config, bkg["data"], L).to(dev).eval()
        cpu_inputs = ({k: v.detach().cpu() for k, v in model.state_dict().items()}, batch, bkg, args.config) if (world == 1 and not args.no_cpu_baseline) else None
        batch = D.batch_to(batch, dev)
        bkg = {"data": bkg["data"].to(dev), "drug_index_map": bkg["drug_index_map"].to(dev)}
        filler = torch.randn(N, 128, device=dev, generator=torch.Generator(device=dev).manual_seed(5))   # drugs absent from the KG (always masked)

    def headline(mode: str, keep_scores: bool = False) -> dict:
        """K timed steps of the whole job in one sharding mode -> timings of this rank (+ max over ranks)."""
        if mode == "strong":                          # ONE model of L outcomes, rank r scores outcomes [lo_l, hi_l)
            lo_l, hi_l = (rank * L) // world, ((rank + 1) * L) // world
            w_seed = 1000
        else:                                         # every rank its own L outcomes
            lo_l, hi_l, w_seed = 0, L, 1000 + rank
        w_orig = (torch.randn(L, 128, 128, generator=torch.Generator().manual_seed(w_seed)) / 128 ** 0.5).to(dev)
        Lr = hi_l - lo_l
        out = ops.empty_scores(Lr, N, N, dev)          # rows on 128-byte lines: a plain contiguous tensor when N % 32 == 0
        if args.head_only:
            lo, hi = shard_range(N, rank, world)
            z_shard = torch.randn(N, 128, generator=torch.Generator().manual_seed(0))[lo:hi].to(dev)
            w_sym = torch.empty_like(w_orig)
        else:
            with torch.no_grad():
                model.decoder.parametrizations.weight.original.copy_(w_orig)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]       # start | encoded | gathered | scored
        keep = {}

        @torch.no_grad()
        def step(i=None):
            if i is not None:
                ev[i][0].record()
            if args.head_only:
                if i is not None:
                    ev[i][1].record()
                z = all_gather_rows(z_shard, N, rank, world) if world > 1 else z_shard
                ops.symmetrize(w_orig, out=w_sym)
                if i is not None:
                    ev[i][2].record()
                ops.bilinear_allpairs(z, z, w_sym[lo_l:hi_l], precision=args.precision, out=out)
            else:
                z = generate_embeddings(model, batch, bkg, rank=rank, world=world, kg_filler=filler,
                                        on_encoded=None if i is None else ev[i][1].record)
                model.decoder.symmetric_weight()          # W_sym (cached until the parameter changes)
                if i is not None:
                    ev[i][2].record()
                model.decoder(z, z, (lo_l, hi_l), out=out)
            if i is not None:
                ev[i][3].record()
            keep["z"] = z

        for _ in range(args.warmup):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], device=dev if backend != "gloo" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        # what the exchange step delivered: every rank must hold the same z[N,128], made of every rank's block
        z = keep["z"]
        zsum = z.double().sum().reshape(1)
        coll = {"world_size_seen": world, "backend": "none" if world == 1 else backend, "z_rows": int(z.shape[0]), "z_checksum": float(zsum)}
        if world > 1:
            both = torch.cat([zsum, -zsum]).to(dev if backend != "gloo" else "cpu")
            dist.all_reduce(both, op=dist.ReduceOp.MAX)          # max(sum) and -min(sum) over ranks
            coll.update(world_size_seen=dist.get_world_size(), z_checksum_spread_over_ranks=float(both[0] + both[1]),
                        z_block_checksums=[float(z[slice(*shard_range(N, r, world))].double().sum()) for r in range(world)],
                        op="all_gather_into_tensor of [N/G,128] fp32 blocks")
        mean = lambda a, b: sum(e[a].elapsed_time(e[b]) for e in ev) / args.steps
        phases = [mean(0, 1), mean(1, 2), mean(2, 3)]        # this rank: encode+fuse of its drug block | all-gather(z) (+ W_sym) | head
        if world > 1:                                        # every rank's phases in the line: the curve explains itself
            allp = torch.tensor(phases, dtype=torch.float64, device=dev if backend != "gloo" else "cpu").repeat(world, 1) * 0
            allp[rank] = torch.tensor(phases, dtype=torch.float64)
            dist.all_reduce(allp)
            per_rank = allp.cpu().tolist()
        else:
            per_rank = [phases]
        res = {"mode": mode, "dt": dt, "Lr": Lr, "L_total": L if mode == "strong" else L * world, "collective": coll,
               "enc_ms": mean(0, 2),                          # encode+fuse incl. the exchange step
               "head_ms": mean(2, 3),                         # head launch (+ its two operand-split pre-passes)
               "phases_ms_per_rank": {"columns": ["encode_fuse_own_block", "all_gather_z", "head_own_outcomes"], "rows": per_rank}}
        if keep_scores:
            res["scores"] = out
            res["z"] = z
        else:
            del out
        torch.cuda.empty_cache()
        return res

    main_mode = args.scaling if world > 1 else "strong"
    want_ranks = world == 1 and args.rank_outcomes != 0
    want_exact = world == 1 and not args.head_only and args.precision != "f32" and not args.no_f32_exact
    h = headline(main_mode, keep_scores=want_ranks or want_exact)
    f32_exact = None
    if want_exact:
        try:
            f32_exact = f32_exact_leg(model, h["z"], h["scores"], h["enc_ms"], args)
        except Exception as e:
            f32_exact = {"kernel_ms": None, "error": f"{type(e).__name__}: {e}"[:400]}
        torch.cuda.empty_cache()
    ranks = None
    if want_ranks:
        try:
            ranks = ranks_leg(h.pop("scores"), args, None if args.head_only else model, h.pop("z", None))
        except Exception as e:
            ranks = {"metric": "rank-normalised scores/sec", "value": None, "error": f"{type(e).__name__}: {e}"[:400]}
        h.pop("scores", None)
        h.pop("z", None)
        torch.cuda.empty_cache()
    other = headline("weak" if main_mode == "strong" else "strong") if world > 1 else None

    def secondary_legs():
        """Finetune / pretraining / cfg5 legs -> (finetune, pretrain, stress) blocks (each survives its own failure)."""
        nonlocal model, batch, bkg
        finetune = None
        if args.finetune_steps > 0 and not args.head_only:
            try:
                finetune = finetune_leg(model, batch, bkg, filler, N, L, args, rank, world, backend, precision=args.finetune_precision)
                if args.finetune_precision != "bf16x3":
                    second = finetune_leg(model, batch, bkg, filler, N, L, args, rank, world, backend, precision="bf16x3")
                    finetune["fp32_grade"] = {k: second[k] for k in ("value", "unit", "ms_per_step", "dtype", "loss_first_last")}
                if world == 1 and not args.no_cpu_baseline and cpu_inputs is not None:
                    try:
                        trip = tuple(t.cpu() for t in D.make_labelled_triples(N, L, args.finetune_triples, 0))
                        c = cpu_finetune_step(cpu_inputs[0], cpu_inputs[1], cpu_inputs[2], args.config, L, trip)
                        finetune["cpu_baseline"] = {"value": c["steps_per_s"], "unit": "steps/s", "cores": host_threads(), "kind": "port",
                                                    "sample": f"oracle training step (torch CPU autograd, fp32) on {c['sample_drugs']} drugs per side, "
                                                              f"{c['sample_triples']} labelled triples, KG thinned to {c['kg_edges_kept']} of its edges; scaled to "
                                                              f"{N} drugs / {finetune['triples_per_step']} triples / the full KG", "detail": c}
                    except Exception as e:
                        finetune["cpu_baseline"] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
            except Exception as e:          # the headline line must survive a failure of the secondary leg
                finetune = {"metric": "DDI-finetune steps/sec", "value": None, "error": f"{type(e).__name__}: {e}"[:400]}
        pretrain = None
        if args.pretrain_steps > 0 and world == 1 and not args.head_only:
            try:
                pretrain = pretrain_leg(model, bkg, batch["masks"], args)
            except Exception as e:
                pretrain = {"metric": "contrastive-pretraining steps/sec", "value": None, "error": f"{type(e).__name__}: {e}"[:400]}
        stress = None
        if args.stress_drugs > 0:
            model = batch = bkg = None
            torch.cuda.empty_cache()
            try:
                stress = stress_leg(args, rank, world, dev, backend)
            except Exception as e:
                stress = {"bound": "mfma", "achieved": None, "error": f"{type(e).__name__}: {e}"[:400]}

        return finetune, pretrain, stress

    def scores_per_s(r):
        return float(r["L_total"]) * N * N * args.steps / r["dt"]

    def emit(finetune, pretrain, stress):
        """Rank 0 prints THE line of the run."""
        if rank != 0:
            return
        per_launch = float(h["Lr"]) * N * N
        head_ms, enc_ms = h["head_ms"], h["enc_ms"]
        traffic, traffic_src = pmc_traffic(N, h["Lr"], args.precision)
        if args.precision == "f32":
            achieved = per_launch * FLOP_PER_SCORE / (head_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / MFMA_F32_PEAK_TFLOPS, "traffic": traffic}
        else:
            achieved = per_launch * BYTES_PER_SCORE / (head_ms * 1e-3) / 1e9
            nprod = 3 if args.precision == "bf16x3" else 1
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    # scores/s x 256 flop / dense 16-bit peak (what the materialising head reaches of the MFMA roof)
                    "mfma_frac": per_launch * FLOP_PER_SCORE / (head_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                    # matrix-core ISSUE rate: the split-bf16 mode issues 3 products per score element
                    "mfma_issue_frac_16bit": per_launch * FLOP_PER_SCORE * nprod / (head_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS}
        roof.update({"kernel": HEAD_KERNEL[args.precision], "kernel_ms": head_ms, "traffic_source": traffic_src,
                     "traffic_from_profile": traffic is not None,       # PMC passes cannot run inside this process: read from profiles/, not measured in this run
                     "algorithmic_bytes_per_launch": per_launch * BYTES_PER_SCORE,
                     "head_only_scores_per_s": per_launch / (head_ms * 1e-3), "encode_fuse_ms": enc_ms})
        per_gpu = f"{h['Lr']} of {L} outcomes per GPU (fixed job)" if main_mode == "strong" else f"{L} outcomes per GPU"
        wl = (f"all-pairs bilinear head only, {N} x {N} drugs x {per_gpu}" if args.head_only else
              f"all-pairs inference, whole job per step: encode+fuse {N} drugs (4 modalities, {args.config}: GIN + HGT over a "
              f"{args.kg_nodes}-node / {args.kg_edges}-edge KG + cv MLP + chemCPA tx, fusion transformer) then score {N} x {N} "
              f"pairs x {per_gpu}, [L,N,N] fp32 logits materialised in HBM; BASELINE configs[1]/[3]")
        line = {"metric": "drug-pair x outcome scores/sec (all-pairs)", "value": scores_per_s(h), "unit": "scores/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": h["dt"] / args.steps * 1e3, "higher_is_better": True,
                "scaling": main_mode, "vs_baseline": None,
                "dtype": {"f32": "f32", "bf16x3": "f32 via split-bf16 (bf16x3) MFMA", "bf16": "bf16"}[args.precision],
                "data": "synthetic",
                "config": {"workload": wl, "drugs": N, "outcomes_per_gpu": h["Lr"], "outcomes_total": h["L_total"], "feature_dim": 128,
                           "model": None if args.head_only else args.config, "precision": args.precision,
                           "parallelism": "single GPU" if world == 1 else
                           f"drug-sharded encode+fuse, all-gather(z) over {'RCCL' if backend == 'nccl' else backend}, outcome-sharded head x{world} ({main_mode} scaling)"},
                "collective": h["collective"],
                "roofline": roof}
        if other is not None:
            line[f"{other['mode']}_scaling"] = {"value": scores_per_s(other), "unit": "scores/s", "ms_per_step": other["dt"] / args.steps * 1e3,
                                                "outcomes_per_gpu": other["Lr"], "outcomes_total": other["L_total"], "scaling": other["mode"],
                                                "head_ms": other["head_ms"], "encode_fuse_ms": other["enc_ms"]}
        if f32_exact is not None:
            line["f32_exact"] = f32_exact
        if stress is not None:
            line["roofline_cfg5"] = stress
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N, L, encode=cpu_inputs)
        line["phases_ms_per_rank"] = h["phases_ms_per_rank"]
        if ranks is not None:
            line["ranks"] = ranks
        if finetune is not None:
            line["finetune"] = finetune
        if pretrain is not None:
            line["pretrain"] = pretrain

        def pick(d, *path):
            for k_ in path:
                if not isinstance(d, dict) or d.get(k_) is None:
                    return None
                d = d[k_]
            return d
        # LAST key: the driver keeps the tail of the line -- both halves of BASELINE's metric and the figures quoted beside the headline
        line["summary"] = {"scores_per_s": line["value"], "head_frac_of_8TBs": roof.get("frac"), "encode_fuse_ms": enc_ms,
                           "finetune_steps_per_s": pick(finetune, "value"), "finetune_ms": pick(finetune, "ms_per_step"),
                           "finetune_ms_cached_plans": pick(finetune, "cached_plans", "ms_per_step"),
                           "finetune_fp32_grade_ms": pick(finetune, "fp32_grade", "ms_per_step"), "dense_block_frac": pick(finetune, "roofline", "frac"),
                           "ranks_ms_per_outcome": pick(ranks, "ms_per_outcome"), "ranks_frac": pick(ranks, "roofline", "frac"),
                           "ranks_traffic_over_algorithmic": None if pick(ranks, "roofline", "traffic") is None else
                           pick(ranks, "roofline", "traffic") / (pick(ranks, "roofline", "algorithmic_bytes_per_outcome") * pick(ranks, "outcomes")),
                           "ranks_outcomes_handed_to_lsd": pick(ranks, "roofline", "outcomes_handed_to_lsd"),
                           "f32_exact_scores_per_s": pick(f32_exact, "whole_job_scores_per_s") or pick(f32_exact, "value"),
                           "share_within_1e-4_rel": pick(f32_exact, "headline_mode_vs_exact", "share_within_1e-4_rel"),
                           "cfg5_frac": pick(stress, "roofline", "frac") or pick(stress, "frac"), "pretrain_ms": pick(pretrain, "ms_per_step"),
                           "cpu_scores_per_s": pick(line.get("cpu_baseline"), "value")}
        print(json.dumps(line), flush=True)
    if world == 1:
        emit(*secondary_legs())
    else:
        # N > 1: the line (the scaling curve's point) leaves FIRST.  The data-parallel finetune step and the sharded cfg5 stress run
        # have never executed over RCCL (one GPU in the build box): they are OPT-IN there (--ddp-legs / MDG_BENCH_DDP_LEGS=1), run after
        # the line under a watchdog, and report to stderr and gpurun_out/bench_ddp_legs_n<N>.json -- a hang or a crash in them cannot
        # take the headline with it (it is printed by then); a default driver run does not execute them at all.  A hang in an
        # opted-in run ends the rank with exit code 3: it must not read as success.
        emit(None, None, None)
        import threading
        limit = float(os.environ.get("MDG_BENCH_DDP_LEGS_SECONDS", "420"))
        if limit > 0 and (args.ddp_legs or os.environ.get("MDG_BENCH_DDP_LEGS", "0") == "1"):
            dog = threading.Timer(limit, lambda: (sys.stderr.write(f"[bench] secondary legs exceeded {limit:.0f} s on rank {rank}: leaving\n"), sys.stderr.flush(), os._exit(3)))
            dog.daemon = True
            dog.start()
            ft, pt, st_ = secondary_legs()
            dog.cancel()
            if rank == 0:
                side = {"n_gpus": world, "finetune": ft, "pretrain": pt, "roofline_cfg5": st_}
                sys.stderr.write("[bench] secondary legs: " + json.dumps(side) + "\n")
                try:
                    os.makedirs("gpurun_out", exist_ok=True)
                    with open(os.path.join("gpurun_out", f"bench_ddp_legs_n{world}.json"), "w") as fh:
                        json.dump(side, fh)
                except OSError:
                    pass
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
