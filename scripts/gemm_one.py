"""A few launches of the dense block at given shapes (for rocprofv3 --kernel-trace):  python scripts/gemm_one.py bf16 22464x2048x2048 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import ops
prec = sys.argv[1]
for a in sys.argv[2:]:
    M, N, K = (int(v) for v in a.split("x"))
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
    img = ops.pack_operand(x, prec)
    for _ in range(6):
        ops.linear_packed(img, M, w, precision=prec)
    torch.cuda.synchronize()
