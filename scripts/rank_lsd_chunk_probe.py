import os, sys
sys.path.insert(0, "/root/repo")
os.environ["MDG_RANKS_MSD"] = "0"
import torch
from madrigal_amd import ops
N, L = 4096, 64
s = torch.randn(L, N, N, device="cuda")
r = ops.empty_scores(L, N, N, s.device)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for ws in (8 << 30, 4 << 30, 2 << 30, 1200 << 20, 600 << 20, 300 << 20, 150 << 20):
    ops.rank_normalize(s[:8], out=r[:8], max_workspace_bytes=ws); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0.record(); ops.rank_normalize(s, out=r, max_workspace_bytes=ws); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"LSD, workspace {ws >> 20} MB: {best / L * 1e3:.1f} us per outcome")
