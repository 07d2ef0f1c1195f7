"""Tie stability of a rank-kernel build (MDG_AB_LIB): heavy ties (7 distinct values), all-equal scores, and random scores against
the oracle (stable in the flat index)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from madrigal_amd import ops, _lib
if os.environ.get("MDG_AB_LIB"):
    _lib.LIB_PATH = os.environ["MDG_AB_LIB"]
from oracle import madrigal_oracle as O
rng = np.random.default_rng(1)
ok = True
for N in (200, 777, 1500):
    for kind in ("ties7", "equal", "random", "ties2"):
        s = {"ties7": lambda: rng.integers(1, 8, size=(2, N, N)).astype(np.float32), "equal": lambda: np.ones((2, N, N), np.float32),
             "random": lambda: rng.standard_normal((2, N, N)).astype(np.float32), "ties2": lambda: rng.integers(1, 3, size=(2, N, N)).astype(np.float32)}[kind]()
        out = ops.rank_normalize(torch.from_numpy(s).cuda()).cpu().numpy()
        same = bool(np.array_equal(out, O.rank_normalize(s)))
        ok &= same
        print(N, kind, "identical to the stable oracle:", same, flush=True)
print("ALL STABLE" if ok else "UNSTABLE")
