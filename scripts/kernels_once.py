"""A few launches of the path's main kernels at their bench shapes (for PMC passes: scripts/pmc_kernels.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import ops
torch.manual_seed(0)
x = torch.randn(20480, 2048, device="cuda"); w = torch.randn(6144, 2048, device="cuda") / 45
for prec in ("bf16x3", "bf16", "f32"):
    for _ in range(2):
        ops.linear(x, w, precision=prec)
xs = torch.randn(130000, 128, device="cuda"); ws = torch.randn(384, 128, device="cuda") / 11
for _ in range(2):
    ops.linear(xs, ws, precision="bf16x3")
z = torch.randn(4096, 128, device="cuda"); wl = ops.symmetrize(torch.randn(224, 128, 128, device="cuda") / 11)
out = torch.empty(224, 4096, 4096, device="cuda")
for _ in range(2):
    ops.bilinear_allpairs(z, z, wl, precision="bf16x3", out=out)
# fusion attention at the finetune shape: 700 tiles of 32 live rows, 8 heads x 256
n_t = 700
rs = torch.arange(0, 32 * (n_t + 1), 32, device="cuda", dtype=torch.int64)
qkv = torch.randn(32 * n_t, 3 * 2048, device="cuda"); dout = torch.randn(32 * n_t, 2048, device="cuda")
bits = torch.zeros(32 * n_t, dtype=torch.int32, device="cuda")
for _ in range(2):
    ops.fusion_attention(qkv, n_t, 32, 8, 256, row_start=rs, row_bits=bits)
    ops.fusion_attention_bwd(qkv, dout, n_t, 32, 8, 256, row_start=rs, row_bits=bits)
# rank normalisation of 8 outcomes at the bench's size (the msd_* kernels)
sr = torch.randn(8, 4096, 4096, device="cuda")
for _ in range(2):
    ops.rank_normalize(sr)
torch.cuda.synchronize()
print("done")
