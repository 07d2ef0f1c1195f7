"""Head launch time per score at aligned and ragged drug counts (HIP events, median of 5): SURVEY 8(d)'s ragged input (4003 x 901),
the reference's real drug count 11 607 (generate_embeddings.ipynb) with a few outcomes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import ops
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
shapes = [(4096, 896), (4003, 901), (4002, 901), (4000, 900), (11607, 64), (11608, 64)]
if len(sys.argv) > 2:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]]
for N, L in shapes:
    g = torch.Generator().manual_seed(0)
    z = torch.randn(N, 128, generator=g).cuda()
    w = ops.symmetrize((torch.randn(L, 128, 128, generator=g) / 128 ** 0.5).cuda())
    out = ops.empty_scores(L, N, N, "cuda") if os.environ.get("PITCHED", "1") == "1" else torch.empty(L, N, N, device="cuda")
    ts = []
    for i in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.bilinear_allpairs(z, z, w, precision=prec, out=out); e1.record(); torch.cuda.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts) // 2]
    print(f"{'pitched' if out.stride(1) != N else 'contig '} {prec} N={N:6d} L={L:4d}: {ms:8.3f} ms  {L * N * N / ms / 1e6:8.1f} G scores/s  {L * N * N * 4 / ms / 1e9:6.2f} TB/s", flush=True)
    del out
    torch.cuda.empty_cache()
