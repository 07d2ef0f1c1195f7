import sys, torch, numpy as np
sys.path.insert(0, '.')
from madrigal_amd import ops
N, L = 4096, 64
g = torch.Generator().manual_seed(20)
z = torch.randn(N, 128, generator=g).cuda()
w = ops.symmetrize((torch.randn(L, 128, 128, generator=g) / 128 ** 0.5).cuda())
ref = ops.bilinear_allpairs(z, z, w, precision="f32")
for it in range(4):
    out = torch.full((L, N, N), float("nan"), device="cuda")
    ops.bilinear_allpairs(z, z, w, precision="bf16x3", out=out)
    bad = ~(torch.abs(out - ref) < 1e-2)
    nb = int(bad.sum())
    print("iter", it, "bad elements", nb, "nan", int(torch.isnan(out).sum()))
    if nb:
        idx = bad.nonzero()
        l, i, j = idx[:, 0], idx[:, 1], idx[:, 2]
        print("  outcomes", torch.unique(l).tolist()[:10], "rows%256 -> wave", torch.unique((i % 256) // 32).tolist(),
              "row blocks", torch.unique(i // 256).tolist()[:10], "col tiles", torch.unique(j // 64).tolist()[:20],
              "cols in tile", torch.unique(j % 64).numel())
        print("  sample", idx[:5].tolist(), out[l[0], i[0], j[0]].item(), ref[l[0], i[0], j[0]].item())
