"""Host-side workers of bench.py's cpu_baseline legs (test infrastructure: the product path never imports this).

rank_one: the reference's rank normalisation of ONE outcome slice, restated step by step as the reference runs it -- `run_slice`
with interval = 1 (notebooks/normalize_scores.py:62-73): copy the slice, overwrite the upper triangle + diagonal with 1e7, rank
all N^2 entries with one default-kind argsort and its inverse permutation (the shape[0] == 1 branch, :47-51), divide by
N(N-1)/2 (:57), zero the masked entries, add the transpose -- on synthetic scores generated inside the worker, so that a pool
of processes (one outcome per process, the reference's own `Pool().map`, :78-85) moves no large array between them."""
from __future__ import annotations

import time

import numpy as np


def rank_one(args):
    seed, n = args
    s = np.random.default_rng(seed).standard_normal((1, n, n), dtype=np.float32)
    mask = np.vstack(np.triu_indices(n, k=0, m=n))
    t0 = time.perf_counter()
    sl = s.copy()
    sl[:, mask[0], mask[1]] = 1e7
    flat = sl.reshape(1, -1)
    temp = flat.argsort(axis=1)
    flat_rank = np.empty_like(temp)
    flat_rank[0, temp] = np.arange(flat_rank.shape[1]) + 1
    norm = (flat_rank / (n * (n - 1) / 2)).reshape(sl.shape)
    norm[:, mask[0], mask[1]] = 0
    norm = (norm + norm.swapaxes(1, 2)).astype(np.float32)
    dt = time.perf_counter() - t0
    return dt, float(norm[0, 1, 0])
