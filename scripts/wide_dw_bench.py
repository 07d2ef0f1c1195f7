"""Wide weight gradients dW = g^T x of the fusion transformer (K = 22 016 token rows): tile shape / stream-K switches of the tile GEMM."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from madrigal_amd import ops
from madrigal_amd._lib import lib

M = 22016
for N, K in ((6144, 2048), (2048, 2048), (2048, 1024), (1024, 2048)):
    g = torch.randn(M, N, device="cuda")
    x = torch.randn(M, K, device="cuda")
    line = f"dW [{N},{K}] over {M} rows: "
    for name, env in (("default", {}), ("tile128", {"MDG_LINEAR_TILE": "128"}), ("tile256", {"MDG_LINEAR_TILE": "256"}),
                      ("tile256+streamK", {"MDG_LINEAR_TILE": "256", "MDG_LINEAR_STREAMK": "1"})):
        for k in ("MDG_LINEAR_TILE", "MDG_LINEAR_STREAMK"):
            os.environ.pop(k, None)
        os.environ.update(env)
        lib().mdg_tuning_reload()
        _, t_img, _ = ops.linear_backward_pack(g, "bf16", want_bias=False, want_row_image=False)
        for _ in range(20):
            ops.linear_tn_packed_g(t_img, x, N, "bf16")
        torch.cuda.synchronize()
        R = 21
        e = [torch.cuda.Event(enable_timing=True) for _ in range(R + 1)]
        for i in range(R):
            e[i].record()
            dw = ops.linear_tn_packed_g(t_img, x, N, "bf16")
        e[R].record()
        torch.cuda.synchronize()
        t = sorted(e[i].elapsed_time(e[i + 1]) for i in range(R))[R // 2] * 1e3
        line += f"{name} {t:6.1f} us  "
    print(line, "(incl. the transposing pack of x)", flush=True)
