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
exchange step), then every rank scores ITS OWN `--outcomes` outcomes against all N x N pairs (outcome-sharded
head).  Per-GPU work is fixed => "scaling": "weak"; global outcomes = outcomes x n_gpus.
Rank 0 prints ONE JSON line.
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
HEAD_KERNEL = {"f32": "bilinear_allpairs_kernel<0, 0, 8>", "bf16x3": "bilinear_allpairs_kernel<1, 0, 8>",
               "bf16": "bilinear_allpairs_kernel<2, 0, 8>"}


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
                if "bilinear_allpairs_kernel" in name:
                    return k.get("hbm_bytes_per_launch_corrected"), os.path.basename(f)
    return None, None


def cpu_baseline(n_drugs: int, n_outcomes: int, seconds: float = 12.0):
    """The oracle's bilinear head (same torch ops as the reference's CPU path,
    madrigal/models/models.py:539) on a bounded row sample of the same workload."""
    import torch
    from oracle import madrigal_oracle as O
    cores = os.cpu_count() or 1
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
    return {"value": rows * n_drugs * n_outcomes / med, "unit": "scores/s", "cores": cores, "kind": "port",
            "sample": f"oracle bilinear_scores (the scoring stage only: encode is amortised over N^2 L scores) on {rows} head "
                      f"rows x {n_drugs} tail drugs x {n_outcomes} outcomes, fp32 torch CPU, {cores} threads, median of 3, "
                      f"extrapolated linearly in rows"}


def finetune_leg(model, batch, bkg, filler, N, L, args, rank=0, world=1, backend="nccl"):
    """DDI-finetune steps/s on the same model and batch (train_ddi_batch.py:275-350, 'full_full' mode): zero_grad ->
    encode head and tail side (training mode: dropout, BatchNorm batch statistics) -> scores of the labelled triples
    (gathered head) -> BCE -> backward through every encoder -> AdamW over the reference's parameter groups.
    world > 1: the SAME step data-parallel (strong scaling): drug-sharded encoders with SyncBatchNorm, all-gather of the
    embeddings, triples dealt to the ranks, flat all-reduce of the gradients (madrigal_amd/train.py)."""
    import torch
    import torch.distributed as dist
    from madrigal_amd import data as D
    from madrigal_amd.optim import create_optimizer
    from madrigal_amd.train import FinetuneStep
    dev = batch["cv"].device
    if world > 1:                                      # the headline gave every rank its own outcomes: one model again
        with torch.no_grad():
            model.decoder.parametrizations.weight.original.copy_(
                torch.randn(L, 128, 128, generator=torch.Generator().manual_seed(1000)) / 128 ** 0.5)
    lab, hd, tl, y = (t.to(dev) for t in D.make_labelled_triples(N, L, args.finetune_triples, 0))
    T = int(lab.numel())
    hp = dict(optimizer="adamw", structure_encoder_lr=1e-5, kg_encoder_lr=1e-5, perturb_encoders_lr=1e-5, fusion_lr=1e-6, decoder_lr=1e-4,
              wd=1e-2, beta1=0.9, beta2=0.999, eps=1e-8)
    fs = FinetuneStep(model, create_optimizer(model, hp), rank=rank, world=world)
    torch.manual_seed(4321 + rank)
    losses = [float(fs.step(batch, batch, batch["masks"], batch["masks"], bkg, lab, hd, tl, y, kg_filler=filler))]     # warm-up
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.finetune_steps):
        losses.append(fs.step(batch, batch, batch["masks"], batch["masks"], bkg, lab, hd, tl, y, kg_filler=filler))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev if backend != "gloo" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    dt /= args.finetune_steps
    return {"metric": "DDI-finetune steps/sec", "value": 1.0 / dt, "unit": "steps/s", "ms_per_step": dt * 1e3, "n_gpus": world,
            "scaling": "strong", "steps": args.finetune_steps, "warmup": 1, "triples_per_step": T, "drugs": N, "outcomes": L,
            "loss_first_last": [float(losses[0]), float(losses[-1])],
            "parallelism": "single GPU" if world == 1 else
            f"drug-sharded encoders (SyncBatchNorm), all-gather(z) / reduce-scatter(dz), triples dealt to {world} ranks, flat gradient all-reduce",
            "work": "optimizer.zero_grad, encode+fuse head side and tail side (training mode), gathered bilinear head on the "
                    "labelled triples, BCE, backward through all encoders, AdamW step"}


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
    ap.add_argument("--finetune-steps", type=int, default=3, help="second half of BASELINE's metric: DDI-finetune steps/s "
                    "(encode both sides + gathered head + BCE + backward + AdamW), timed at N=1 after the headline; 0 = skip")
    ap.add_argument("--finetune-triples", type=int, default=1_000_000, help="positive triples; with 2 negatives each and both "
                    "directions (the reference's collation) 6x as many labelled triples per step")
    args = ap.parse_args()

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
    out = torch.empty(L, N, N, dtype=torch.float32, device=dev)
    if args.head_only:
        g = torch.Generator().manual_seed(0)
        z_all = torch.randn(N, 128, generator=g)
        lo, hi = shard_range(N, rank, world)
        z_shard = z_all[lo:hi].to(dev)
        w_orig = (torch.randn(L, 128, 128, generator=torch.Generator().manual_seed(1000 + rank)) / 128 ** 0.5).to(dev)
        w_sym = torch.empty_like(w_orig)
        model = None
    else:
        # same synthetic batch on every rank (seeded); each rank encodes only its drug block
        batch, bkg = D.make_batch(N, 0, kg_nodes=args.kg_nodes, kg_edges=args.kg_edges)
        torch.manual_seed(1234)                       # identical encoder weights on every rank
        model = configs.build_model(args.config, bkg["data"], L)
        with torch.no_grad():                         # this rank's own outcomes
            model.decoder.parametrizations.weight.original.copy_(
                torch.randn(L, 128, 128, generator=torch.Generator().manual_seed(1000 + rank)) / 128 ** 0.5)
        model = model.to(dev).eval()
        batch = D.batch_to(batch, dev)
        bkg = {"data": bkg["data"].to(dev), "drug_index_map": bkg["drug_index_map"].to(dev)}
        filler = torch.randn(N, 128, device=dev)       # rows of drugs absent from the KG (always masked)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]

    @torch.no_grad()
    def step(i=None):
        if i is not None:
            ev[i][0].record()
        if args.head_only:
            z = all_gather_rows(z_shard, N, rank, world) if world > 1 else z_shard
            ops.symmetrize(w_orig, out=w_sym)
            if i is not None:
                ev[i][1].record()
            ops.bilinear_allpairs(z, z, w_sym, precision=args.precision, out=out)
        else:
            z = generate_embeddings(model, batch, bkg, rank=rank, world=world, kg_filler=filler)
            model.decoder.symmetric_weight()          # W_sym (cached until the parameter changes)
            if i is not None:
                ev[i][1].record()
            model.decoder(z, z, out=out)
        if i is not None:
            ev[i][2].record()

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

    finetune = None
    if args.finetune_steps > 0 and not args.head_only:
        del out
        torch.cuda.empty_cache()
        try:
            finetune = finetune_leg(model, batch, bkg, filler, N, L, args, rank, world, backend)
        except Exception as e:          # the headline line must survive a failure of the secondary leg
            finetune = {"metric": "DDI-finetune steps/sec", "value": None, "error": f"{type(e).__name__}: {e}"[:400]}
    enc_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps
    head_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps     # head launch (+ its two operand-split pre-passes)
    value = float(L) * N * N * world * args.steps / dt
    if rank == 0:
        per_launch = float(L) * N * N
        traffic, traffic_src = pmc_traffic(N, L, args.precision)
        if args.precision == "f32":
            achieved = per_launch * FLOP_PER_SCORE / (head_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / MFMA_F32_PEAK_TFLOPS, "traffic": traffic}
        else:
            achieved = per_launch * BYTES_PER_SCORE / (head_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "mfma_frac_bf16": per_launch * FLOP_PER_SCORE * (3 if args.precision == "bf16x3" else 1)
                    / (head_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS}
        roof.update({"kernel": HEAD_KERNEL[args.precision], "kernel_ms": head_ms, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": per_launch * BYTES_PER_SCORE,
                     "head_only_scores_per_s": per_launch / (head_ms * 1e-3), "encode_fuse_ms": enc_ms})
        wl = (f"all-pairs bilinear head only, {N} x {N} drugs x {L} outcomes per GPU" if args.head_only else
              f"all-pairs inference, whole job per step: encode+fuse {N} drugs (4 modalities, {args.config}: GIN + HGT over a "
              f"{args.kg_nodes}-node / {args.kg_edges}-edge KG + cv MLP + chemCPA tx, fusion transformer) then score {N} x {N} "
              f"pairs x {L} outcomes per GPU, [L,N,N] fp32 logits materialised in HBM; BASELINE configs[1]/[3]")
        line = {"metric": "drug-pair x outcome scores/sec (all-pairs)", "value": value, "unit": "scores/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None,
                "dtype": {"f32": "f32", "bf16x3": "f32 via split-bf16 (bf16x3) MFMA", "bf16": "bf16"}[args.precision],
                "data": "synthetic",
                "config": {"workload": wl, "drugs": N, "outcomes_per_gpu": L, "outcomes_total": L * world, "feature_dim": 128,
                           "model": None if args.head_only else args.config, "precision": args.precision,
                           "parallelism": "single GPU" if world == 1 else
                           f"drug-sharded encode+fuse, all-gather(z) over RCCL, outcome-sharded head x{world}"},
                "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N, L)
        if finetune is not None:
            line["finetune"] = finetune
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
