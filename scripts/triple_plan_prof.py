"""The triple plan alone on fresh sets of labelled triples (what every finetune step of an epoch rebuilds): wall time per plan; run under
rocprofv3 --kernel-trace --stats for its launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import data, ops
sets = [tuple(t.cuda() for t in data.make_labelled_triples(4096, 896, 1_000_000, s)) for s in range(4)]
for i in range(2):
    ops.triple_plan(*sets[i][:3], 896, 4096, 4096)
torch.cuda.synchronize(); t = time.perf_counter()
n = 8
for i in range(n):
    ops.triple_plan(*sets[i % 4][:3], 896, 4096, 4096)
torch.cuda.synchronize()
print(f"triple plan: {(time.perf_counter() - t) / n * 1e3:.2f} ms per plan, {sets[0][0].numel()} triples")
