"""The dense block with and without the row split of its last partial round (MDG_LINEAR_TAIL128): kernel time of y = x W^T.
    python scripts/tail_split_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import ops
from madrigal_amd._lib import lib

def timed(fn, reps=9):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]

for prec in ("bf16", "bf16x3"):
    for M, N, K in ((44928, 2048, 2048), (44928, 1024, 2048), (44928, 6144, 2048), (22016, 6144, 2048), (22016, 1024, 2048), (22016, 2048, 1024)):
        tiles = -(-M // 256) * -(-N // 256)
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5
        img = ops.pack_operand(x, prec)
        out = []
        for sw in ("0", "1"):
            os.environ["MDG_LINEAR_TAIL128"] = sw
            lib().mdg_tuning_reload()
            out.append(timed(lambda: ops.linear_packed(img, M, w, precision=prec)))
        print(f"{prec:7s} [{M}, {N}] x {K}: {tiles} tiles (rem {tiles % 256}): single launch {out[0] * 1e3:7.1f} us, row split {out[1] * 1e3:7.1f} us")
