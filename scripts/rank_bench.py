"""Rank-normalisation throughput (notebooks/normalize_scores.py:36-74): [L,N,N] fp32 scores -> normalised ranks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from madrigal_amd import ops, _lib
if os.environ.get("MDG_AB_LIB"):                 # A/B builds (scripts/build_variant.sh)
    _lib.LIB_PATH = os.environ["MDG_AB_LIB"]
N, L = 4096, 32
s = torch.randn(L, N, N, device="cuda")
r = ops.rank_normalize(s); torch.cuda.synchronize()            # warm-up at full size (code load, workspace allocation)
t = time.perf_counter(); r = ops.rank_normalize(s, out=r); torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"HIP: {L} outcomes x {N}x{N}: {dt * 1e3:.1f} ms = {dt / L * 1e3:.2f} ms per outcome, {L * N * N / dt / 1e9:.2f} G scores/s -> 896 outcomes in {896 * dt / L:.2f} s")
if os.environ.get("MDG_AB_LIB"):
    sys.exit(0)
from oracle import madrigal_oracle as O
x = s[0].cpu().numpy()
t = time.perf_counter(); ref = O.rank_normalize(x[None])[0] if hasattr(O, "rank_normalize") else None; dc = time.perf_counter() - t
print(f"CPU oracle (numpy double argsort), one outcome: {dc:.2f} s -> 896 outcomes in {896 * dc:.0f} s; bit-identical: {bool(np.array_equal(ref, r[0].cpu().numpy()))}")
