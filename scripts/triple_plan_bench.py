"""The gathered head's triple plan (ops.triple_plan: label / (label, head) pair / per-drug orders of the labelled triples of one
batch) alone: wall time per plan.  Under rocprofv3 --kernel-trace --stats its kernels are the whole trace.

    python scripts/triple_plan_bench.py [--reps 6]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import data, ops
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=6)
a = ap.parse_args()
sets = [tuple(t.cuda() for t in data.make_labelled_triples(4096, 896, 1_000_000, s)[:3]) for s in range(3)]
ops.triple_plan(*sets[0], 896, 4096, 4096)
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(a.reps):
    p = ops.triple_plan(*sets[i % 3], 896, 4096, 4096)
torch.cuda.synchronize()
print(f"triple plan, {int(sets[0][0].numel())} triples, {p['pairs']['P']} (label, head) pairs: {(time.perf_counter() - t) / a.reps * 1e3:.2f} ms per plan (wall, host round trips included)")
