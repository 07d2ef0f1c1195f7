#!/usr/bin/env python3
"""Headline benchmark: drug-pair x outcome scores/sec for all-pairs scoring (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--drugs 4096] [--outcomes 896] [--precision bf16x3]

One "step" = one pass of the all-pairs bilinear head over every (head, tail, outcome) triple of the
synthetic drug set, scores materialised as the reference does ([L, N, N] fp32, raw logits), with the
inputs (z, W) already resident in HBM.  N GPUs: one process per GPU (torch.distributed over RCCL);
the drug embeddings are sharded by rank and all-gathered over xGMI (the path's one exchange step),
then every rank scores ITS OWN `--outcomes` outcomes against all N x N pairs (outcome-sharded head,
SURVEY.md 8e): per-GPU work is fixed, so scaling is "weak" and the global outcome count is
outcomes x n_gpus.  Rank 0 prints ONE JSON line.
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
            "sample": f"oracle bilinear_scores on {rows} head rows x {n_drugs} tail drugs x {n_outcomes} outcomes "
                      f"(fp32 torch CPU, {cores} threads), median of 3, extrapolated linearly in rows"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--drugs", type=int, default=4096)
    ap.add_argument("--outcomes", type=int, default=896)
    ap.add_argument("--precision", default="bf16x3", choices=["f32", "bf16x3", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from madrigal_amd import ops
    from madrigal_amd.parallel import shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    N, L = args.drugs, args.outcomes
    g = torch.Generator().manual_seed(0)
    z_all = torch.randn(N, 128, generator=g)
    lo, hi = shard_range(N, rank, world)
    z_shard = z_all[lo:hi].to(dev)                                     # this rank's drug embeddings
    gw = torch.Generator().manual_seed(1000 + rank)
    w_orig = (torch.randn(L, 128, 128, generator=gw) / 128 ** 0.5).to(dev)   # this rank's outcomes
    out = torch.empty(L, N, N, dtype=torch.float32, device=dev)
    w_sym = torch.empty_like(w_orig)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        if world > 1:
            from madrigal_amd.parallel import all_gather_rows
            z = all_gather_rows(z_shard, N, rank, world)
        else:
            z = z_shard
        ops.symmetrize(w_orig, out=w_sym)
        if i is not None:
            ev[i][0].record()
        ops.bilinear_allpairs(z, z, w_sym, precision=args.precision, out=out)
        if i is not None:
            ev[i][1].record()

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
        tmax = torch.tensor([dt], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    scores_per_step = float(L) * N * N * world
    value = scores_per_step * args.steps / dt
    if rank == 0:
        per_launch_scores = float(L) * N * N
        if args.precision == "f32":
            achieved = per_launch_scores * FLOP_PER_SCORE / (kern_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / MFMA_F32_PEAK_TFLOPS, "traffic": None,
                    "kernel": "bilinear_allpairs_kernel<f32,store>", "kernel_ms": kern_ms}
        else:
            achieved = per_launch_scores * BYTES_PER_SCORE / (kern_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                    "kernel": f"bilinear_allpairs_kernel<{args.precision},store>", "kernel_ms": kern_ms,
                    "mfma_frac_bf16": per_launch_scores * FLOP_PER_SCORE * (3 if args.precision == 'bf16x3' else 1)
                    / (kern_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS}
        line = {"metric": "drug-pair x outcome scores/sec (all-pairs)", "value": value, "unit": "scores/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "f32" else
                ("bf16x3(f32-grade)" if args.precision == "bf16x3" else "bf16"), "data": "synthetic",
                "config": {"workload": f"all-pairs bilinear head, {N} drugs x {N} drugs x {L} outcomes per GPU "
                                       f"([L,N,N] fp32 logits materialised in HBM); BASELINE configs[1]/[3] shape",
                           "drugs": N, "outcomes_per_gpu": L, "outcomes_total": L * world, "feature_dim": 128,
                           "precision": args.precision,
                           "parallelism": "single GPU" if world == 1 else f"drug-sharded encode + all-gather(z) over RCCL, outcome-sharded head x{world}"},
                "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N, L)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
