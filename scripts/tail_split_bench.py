"""The dense block with and without the row split of its last partial round (MDG_LINEAR_TAIL128): kernel time of y = x W^T, the two
settings INTERLEAVED launch by launch (back-to-back blocks of one setting are biased by the clock ramp: the later block wins).
    python scripts/tail_split_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import ops
from madrigal_amd._lib import lib


def once(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for prec, shapes in (("bf16x3", ((22016, 6144, 2048), (22016, 1024, 2048), (22016, 4096, 2048), (22016, 2048, 1024))),
                     ("bf16", ((44928, 2048, 2048), (44928, 6144, 2048), (44928, 1024, 2048), (22464, 2048, 2048)))):
    for M, N, K in shapes:
        tiles = -(-M // 256) * -(-N // 256)
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5
        img = ops.pack_operand(x, prec)
        fn = lambda: ops.linear_packed(img, M, w, precision=prec)
        ts = {"0": [], "1": []}
        for it in range(24):
            for sw in (("0", "1") if it % 2 == 0 else ("1", "0")):
                os.environ["MDG_LINEAR_TAIL128"] = sw
                lib().mdg_tuning_reload()
                t = once(fn)
                if it >= 4:
                    ts[sw].append(t)
        med = {k: sorted(v)[len(v) // 2] * 1e3 for k, v in ts.items()}
        print(f"{prec:7s} [{M}, {N}] x {K}: {tiles} tiles (rem {tiles % 256}): single launch {med['0']:7.1f} us, row split {med['1']:7.1f} us")
os.environ.pop("MDG_LINEAR_TAIL128", None)
