"""Rank-normalisation throughput (notebooks/normalize_scores.py:36-74): [L,N,N] fp32 scores -> normalised ranks.

    python scripts/rank_bench.py [N] [L] [--no-oracle]      MDG_RANKS_MSD / MDG_RANKS_GROUP select the path (ranks.hip)

HIP events around the call; the ranks of the first outcome against the CPU oracle (numpy double argsort) unless --no-oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from madrigal_amd import ops, _lib
if os.environ.get("MDG_AB_LIB"):                 # A/B builds (scripts/build_variant.sh)
    _lib.LIB_PATH = os.environ["MDG_AB_LIB"]
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(argv[0]) if argv else 4096
L = int(argv[1]) if len(argv) > 1 else 64
s = torch.randn(L, N, N, device="cuda")
r = ops.empty_scores(L, N, N, s.device)
ops.rank_normalize(s[:4], out=r[:4]); torch.cuda.synchronize()            # warm-up (code load, workspace allocation)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e30
for _ in range(3):
    flags = []
    e0.record(); ops.rank_normalize(s, out=r, fallback_flags=flags); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
M = N * (N - 1) // 2
alg = (M * 4.0 + N * N * 4.0) * L
handed = int(sum(int((f != 0).sum()) for f in flags)) if flags else None
if handed:
    print("reasons (1 = bucket beyond its LDS room, 2 = key count, 8 = block shard full):", sorted(set(int(v) for f in flags for v in f.tolist() if v)))
print(f"HIP (MSD={os.environ.get('MDG_RANKS_MSD', '1 (default)')} group={os.environ.get('MDG_RANKS_GROUP', 'default')}): {L} outcomes x {N}x{N}: {best:.2f} ms = "
      f"{best / L * 1e3:.1f} us per outcome, {L * N * N / best / 1e6:.2f} G scores/s, {alg / best / 1e9:.3f} TB/s of algorithmic bytes "
      f"({alg / best / 1e9 / 8.0:.3f} of 8 TB/s); outcomes handed to the LSD sort: {handed}")
if "--no-oracle" in sys.argv or os.environ.get("MDG_AB_LIB"):
    sys.exit(0)
from oracle import madrigal_oracle as O
x = s[0].cpu().numpy()
t = time.perf_counter(); ref = O.rank_normalize(x[None])[0]; dc = time.perf_counter() - t
print(f"CPU oracle (numpy double argsort), one outcome: {dc:.2f} s; bit-identical: {bool(np.array_equal(ref, r[0].cpu().numpy()))}")
