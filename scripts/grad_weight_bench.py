"""Weight-gradient kernel (dW = g^T x) at the shapes of the finetune / pretraining steps: time per arithmetic mode."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from madrigal_amd import ops

shapes = [(106368, 128, 128), (130000, 128, 128), (130000, 384, 128), (65536, 512, 1024), (4096 * 16, 512, 512), (22016, 2048, 128), (4096, 1024, 560)]
for M, N, K in shapes:
    g = torch.randn(M, N, device="cuda")
    x = torch.randn(M, K, device="cuda")
    ref = g.double().T @ x.double()
    line = f"[{M},{N}]^T [{M},{K}]: "
    for prec in ("f32", "bf16x3", "bf16"):
        ops.grad_weight(g, x, prec, want_bias=True)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
        for i in range(7):
            e[i].record()
            dw, db = ops.grad_weight(g, x, prec, want_bias=True)
        e[7].record()
        torch.cuda.synchronize()
        t = sorted(e[i].elapsed_time(e[i + 1]) for i in range(7))[3] * 1e3
        err = float((dw.double() - ref).abs().max() / ref.abs().max())
        line += f"{prec} {t:7.1f} us ({M * (N + K) * 4 / t / 1e6:5.2f} TB/s, err {err:.1e})  "
    print(line, flush=True)
