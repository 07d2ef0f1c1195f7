"""Head launch at the BASELINE shape: EPI_STORE (full score tensor) against EPI_TRIKEYS (order keys of the lower triangle only),
and the rank normalisation from each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from madrigal_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = int(sys.argv[2]) if len(sys.argv) > 2 else 896
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
g = torch.Generator().manual_seed(0)
z = torch.randn(N, 128, generator=g).cuda()
w = torch.randn(L, 128, 128, generator=g) / 128 ** 0.5
w = (0.5 * (w + w.transpose(1, 2))).contiguous().cuda()
out = ops.empty_scores(L, N, N, "cuda")
keys = out.view(torch.int32)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for i in range(reps):
        fn()
        e[i + 1].record()
    torch.cuda.synchronize()
    return sorted(e[i].elapsed_time(e[i + 1]) for i in range(reps))[reps // 2]


t_store = timed(lambda: ops.bilinear_allpairs(z, z, w, precision=prec, out=out))
t_keys = timed(lambda: ops.bilinear_allpairs(z, z, w, precision=prec, epilogue=ops.EPI_TRIKEYS, out=keys))
print(f"N={N} L={L} {prec}: head STORE {t_store:.2f} ms, head TRIKEYS {t_keys:.2f} ms ({t_keys / t_store:.2f}x)")
