"""Time mdg_linear on the dense-block shapes of the path (one process per setting: the tile / swizzle switches are read once)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import ops
shapes = [(130000, 128, 128), (106000, 128, 68), (20480, 2048, 128), (50000, 256, 256), (20480, 6144, 2048), (20480, 2048, 2048), (20480, 1024, 2048), (20480, 2048, 1024), (6144, 2048, 20480), (65536, 512, 978),
          (130000, 384, 128), (110000, 128, 128)]
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
for M, N, K in shapes:
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    for cache in (True, False):
        for _ in range(3):
            ops.linear(x, w, precision=prec, cache_weight=cache)
        torch.cuda.synchronize()
        t = time.perf_counter()
        reps = 10
        for _ in range(reps):
            ops.linear(x, w, precision=prec, cache_weight=cache)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        fl = 2.0 * M * N * K
        print(f"{prec} M={M:6d} N={N:5d} K={K:5d} packed_w={cache!s:5}  {dt * 1e3:8.3f} ms  {fl / dt / 1e12:7.1f} TFLOP/s (x3 = {3 * fl / dt / 1e12:6.0f} bf16)", flush=True)
