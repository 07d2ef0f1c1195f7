"""BASELINE configs[4]: synthetic scale-up, 100k drugs x 1k outcomes, bf16 bilinear head, nothing materialised
(row statistics epilogue).  Reports scores/s and the fraction of the dense bf16 MFMA peak (2.5 PFLOP/s)."""
import sys, time, torch
sys.path.insert(0, '.')
from madrigal_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100352
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
g = torch.Generator(device="cuda").manual_seed(0)
z = torch.randn(N, 128, device="cuda", generator=g)
w = ops.symmetrize(torch.randn(L, 128, 128, device="cuda", generator=g) / 128 ** 0.5)
out = torch.empty(L, N, 2, device="cuda")
ops.bilinear_allpairs(z[:4096], z[:4096], w[:8], precision=prec, epilogue=ops.EPI_ROWSTATS)      # warm-up
torch.cuda.synchronize()
t = time.perf_counter()
ops.bilinear_allpairs(z, z, w, precision=prec, epilogue=ops.EPI_ROWSTATS, out=out)
torch.cuda.synchronize()
dt = time.perf_counter() - t
scores = float(L) * N * N
mult = 3 if prec == "bf16x3" else 1
print(f"N={N} L={L} {prec}: {dt*1e3:.1f} ms  {scores/dt:.3e} scores/s  MFMA {scores*256*mult/dt/1e12:.0f} TFLOP/s = {scores*256*mult/dt/2.5e15:.3f} of bf16 peak"
      f"  checksum {float(out[..., 0].double().sum()):.6e} max {float(out[..., 1].max()):.4f}")
# size-independent property: sum_j S[l,i,j] = z_i^T W_l (sum_j z_j)
zs = z.sum(0)
ref = torch.einsum("id,lde,e->li", z[:512].double(), w[:4].double(), zs.double())
err = float((out[:4, :512, 0].double() - ref).abs().max() / ref.abs().max())
print("row-sum identity rel err", err)
