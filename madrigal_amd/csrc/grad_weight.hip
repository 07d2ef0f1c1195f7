// Weight gradient of a dense block:  dW[n,k] = sum_m g[m,n] x[m,k]   (g = dL/dy [M,N], x = layer input [M,K]).
// A "TN" product: both operands are row-major with the REDUCTION index m as their row, the output is small
// (N, K <= a few thousand) and M is the number of token / node / atom rows (10^4 .. 10^6).  The forward GEMM kernel
// would give one workgroup per 128x128 output tile a serial loop over all of M; here the m range is split over
// grid.z so that every launch has >= ~1000 workgroups, each accumulating a 128x128 partial tile on the exact-fp32
// matrix cores (v_mfma_f32_32x32x2_f32: two rows of m per instruction, lane half = row), operands read straight from
// global memory (lane = consecutive column: 128-byte coalesced pieces of the two current rows; neighbouring tiles
// re-read them through L2).  Partials are summed in split order by a second kernel: deterministic, no atomics.
#include "mdg_common.h"

namespace {
#include "tn16.h"

constexpr int GW_U = 8;       // row pairs per batch: 16 rows of both operands in flight per buffer

struct GwBatch { float a0[GW_U], a1[GW_U], b0[GW_U], b1[GW_U]; };

__device__ __forceinline__ void gw_load(GwBatch& t, const float* ga, const float* gb, const float* xa, const float* xb, int64_t ldg,
                                        int64_t ldx, int64_t m, int half) {
#pragma unroll
  for (int u = 0; u < GW_U; ++u) {
    const int64_t r = m + 2 * u + half;
    t.a0[u] = ga[r * ldg];
    t.a1[u] = gb[r * ldg];
    t.b0[u] = xa[r * ldx];
    t.b1[u] = xb[r * ldx];
  }
}

__device__ __forceinline__ void gw_compute(const GwBatch& t, float fa, float fb, f32x16 (&acc)[2][2], float& sa, float& sb) {
#pragma unroll
  for (int u = 0; u < GW_U; ++u) {
    const float va = t.a0[u] * fa, vb = t.a1[u] * fb;
    sa += va;
    sb += vb;
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(va, t.b0[u], acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(va, t.b1[u], acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb, t.b0[u], acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb, t.b1[u], acc[1][1], 0, 0, 0);
  }
}

__global__ __launch_bounds__(256) void grad_weight_kernel(const GwArgs p) {
  const int lane = threadIdx.x & 63, i = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.y * 128 + (wave >> 1) * 64, k0 = blockIdx.x * 128 + (wave & 1) * 64;
  const int64_t m0 = static_cast<int64_t>(blockIdx.z) * p.rows_per_split;
  const int64_t m1 = m0 + p.rows_per_split < p.M ? m0 + p.rows_per_split : p.M;
  // clamp out-of-range columns to column 0 and zero their contribution
  const int na = n0 + i, nb = n0 + 32 + i, ka = k0 + i, kb = k0 + 32 + i;
  const float fa = na < p.N ? 1.f : 0.f, fb = nb < p.N ? 1.f : 0.f;
  const float* ga = p.g + (na < p.N ? na : 0);
  const float* gb = p.g + (nb < p.N ? nb : 0);
  const float* xa = p.x + (ka < p.K ? ka : 0);
  const float* xb = p.x + (kb < p.K ? kb : 0);
  f32x16 acc[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[r][c][v] = 0.f;
  float sa = 0.f, sb = 0.f;                               // column sums of g over this lane's rows (bias gradient)

  // two register buffers: the loads of batch b+1 are in flight while the matrix cores work on batch b
  const int64_t nbatch = (m1 - m0) / (2 * GW_U);
  GwBatch A, B;
  if (nbatch > 0) gw_load(A, ga, gb, xa, xb, p.ldg, p.ldx, m0, half);
  int64_t b = 0;
  while (b < nbatch) {
    if (b + 1 < nbatch) gw_load(B, ga, gb, xa, xb, p.ldg, p.ldx, m0 + (b + 1) * 2 * GW_U, half);
    gw_compute(A, fa, fb, acc, sa, sb);
    if (++b >= nbatch) break;
    if (b + 1 < nbatch) gw_load(A, ga, gb, xa, xb, p.ldg, p.ldx, m0 + (b + 1) * 2 * GW_U, half);
    gw_compute(B, fa, fb, acc, sa, sb);
    ++b;
  }
  for (int64_t m = m0 + nbatch * 2 * GW_U; m < m1; m += 2) {   // tail: the odd last row meets a zero partner
    const int64_t r = m + half;
    const bool ok = r < m1;
    const int64_t rr = ok ? r : m1 - 1;
    const float s = ok ? 1.f : 0.f;
    const float va = ga[rr * p.ldg] * fa * s, vb = gb[rr * p.ldg] * fb * s;
    const float b0 = xa[rr * p.ldx], b1 = xb[rr * p.ldx];
    sa += va;
    sb += vb;
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(va, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(va, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb, b1, acc[1][1], 0, 0, 0);
  }
  // acc[r][c][v]: row n = n0 + 32r + (v&3) + 8(v>>2) + 4*half, column k = k0 + 32c + i
  float* out = p.out + static_cast<int64_t>(blockIdx.z) * p.N * p.K;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int k = k0 + 32 * c + i;
      if (k >= p.K) continue;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int n = n0 + 32 * r + (v & 3) + 8 * (v >> 2) + 4 * half;
        if (n < p.N) out[static_cast<int64_t>(n) * p.K + k] = acc[r][c][v];
      }
    }
  if (p.db && blockIdx.x == 0 && (wave & 1) == 0) {         // one k-tile column of waves owns the bias partials
    sa += __shfl_xor(sa, 32, 64);                           // even rows + odd rows
    sb += __shfl_xor(sb, 32, 64);
    if (half == 0) {
      float* d = p.db + static_cast<int64_t>(blockIdx.z) * p.N;
      if (na < p.N) d[na] = sa;
      if (nb < p.N) d[nb] = sb;
    }
  }
}

// out[i] = sum_k part[k][i] in a fixed order: four interleaved groups of splits per element (thread rows of the block),
// four independent accumulators per thread so that the loads overlap, groups combined through LDS.
// Two arrays in one launch (the weight partials [splits][n] and, behind them, the bias partials [splits][n2]): blocks past
// ceil(n / 64) serve the second one.
__global__ __launch_bounds__(256) void sum_splits_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t n, int splits,
                                                         const float* __restrict__ part2, float* __restrict__ out2, int64_t n2) {
  const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t first = (n + 63) / 64;
  int64_t i = static_cast<int64_t>(blockIdx.x) * 64 + e;
  if (static_cast<int64_t>(blockIdx.x) >= first) {       // block-uniform
    i -= first * 64;
    part = part2;
    out = out2;
    n = n2;
  }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    int k = grp;
    for (; k + 12 < splits; k += 16) {
      s0 += part[static_cast<int64_t>(k) * n + i];
      s1 += part[static_cast<int64_t>(k + 4) * n + i];
      s2 += part[static_cast<int64_t>(k + 8) * n + i];
      s3 += part[static_cast<int64_t>(k + 12) * n + i];
    }
    for (; k < splits; k += 4) s0 += part[static_cast<int64_t>(k) * n + i];
  }
  __shared__ float sh[4][64];
  sh[grp][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && i < n) out[i] = (sh[0][e] + sh[1][e]) + (sh[2][e] + sh[3][e]);
}

// 16-bit form: ~2 workgroups per CU, at least 512 rows per split (a partial tile costs as much traffic as 64 rows of both operands;
// measured: 1-tile outputs 34 -> 31 us with 512 instead of 256 rows, wide ones unchanged)
int pick_splits16(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = mdg_cdiv(N, 128) * mdg_cdiv(K, 128);
  int64_t s = mdg_cdiv(512, tiles);                       // ~2 workgroups per CU
  const int64_t mr = 512;                                 // at least 512 rows per split
  const int64_t max_s = M / mr > 1 ? M / mr : 1;
  if (s > max_s) s = max_s;
  if (s > 4096) s = 4096;
  return static_cast<int>(s < 1 ? 1 : s);
}

bool use16(int precision, const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t N, int64_t K) {
  return (precision == MDG_PREC_BF16 || precision == MDG_PREC_BF16X3) && N % 4 == 0 && K % 4 == 0 && ldg % 4 == 0 && ldx % 4 == 0 &&
         (g == nullptr || mdg_aligned16(g)) && (x == nullptr || mdg_aligned16(x));
}

int pick_splits(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = mdg_cdiv(N, 128) * mdg_cdiv(K, 128);
  int64_t s = mdg_cdiv(1024, tiles);                      // ~4 workgroups per CU; every partial costs N*K*8 bytes of traffic
  const int64_t max_s = mdg_cdiv(M, 128);                 // at least 128 rows (8 batches) per split
  if (s > max_s) s = max_s;
  if (s > 4096) s = 4096;
  return static_cast<int>(s < 1 ? 1 : s);
}

}  // namespace

extern "C" size_t mdg_grad_weight_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int s = pick_splits(M, N, K), s16 = pick_splits16(M, N, K);          // whichever form the call takes
  const int sm = s > s16 ? s : s16;
  return sm == 1 ? 0 : static_cast<size_t>(sm) * N * (K + 1) * sizeof(float);
}

extern "C" int mdg_grad_weight_prec(const float* g, int64_t ldg, const float* x, int64_t ldx, float* dw, float* dbias, int64_t M, int64_t N,
                                    int64_t K, int precision, void* workspace, size_t workspace_bytes, void* stream) {
  MDG_CHECK_ARG(M >= 0 && N > 0 && K > 0 && N <= (1 << 20) && K <= (1 << 20), "mdg_grad_weight: bad shape");
  MDG_CHECK_ARG(dw, "mdg_grad_weight: null dw");
  MDG_CHECK_ARG(precision == MDG_PREC_F32 || precision == MDG_PREC_BF16 || precision == MDG_PREC_BF16X3, "mdg_grad_weight: unknown precision %d", precision);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (M == 0) {
    (void)hipMemsetAsync(dw, 0, static_cast<size_t>(N) * K * sizeof(float), st);
    if (dbias) (void)hipMemsetAsync(dbias, 0, static_cast<size_t>(N) * sizeof(float), st);
    return MDG_OK;
  }
  MDG_CHECK_ARG(g && x && ldg >= N && ldx >= K, "mdg_grad_weight: null operand / short row stride");
  const bool m16 = use16(precision, g, ldg, x, ldx, N, K);
  const int splits = m16 ? pick_splits16(M, N, K) : pick_splits(M, N, K);
  const size_t need = splits == 1 ? 0 : static_cast<size_t>(splits) * N * (K + 1) * sizeof(float);
  if (need && (!workspace || workspace_bytes < need)) {
    mdg_set_error("mdg_grad_weight: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return MDG_EWORKSPACE;
  }
  int64_t rps = mdg_cdiv(M, splits);
  if (m16) rps = mdg_cdiv(rps, G16_ROWS) * G16_ROWS;        // whole 32-row chunks per split
  else rps += rps & 1;                                      // whole row pairs per split
  const int64_t used = mdg_cdiv(M, rps);
  float* part = static_cast<float*>(workspace);
  float* part_db = (splits == 1 || !dbias) ? dbias : part + static_cast<size_t>(splits) * N * K;
  GwArgs a{g, ldg, x, ldx, splits == 1 ? dw : part, part_db, M, rps, static_cast<int>(N), static_cast<int>(K)};
  const dim3 grid(static_cast<unsigned>(mdg_cdiv(K, 128)), static_cast<unsigned>(mdg_cdiv(N, 128)), static_cast<unsigned>(used));
  if (m16 && precision == MDG_PREC_BF16X3) hipLaunchKernelGGL(grad_weight16_kernel<MDG_PREC_BF16X3>, grid, dim3(256), 2 * 2 * 2 * G16_PLANE, st, a);
  else if (m16) hipLaunchKernelGGL(grad_weight16_kernel<MDG_PREC_BF16>, grid, dim3(256), 2 * 2 * G16_PLANE, st, a);
  else hipLaunchKernelGGL(grad_weight_kernel, grid, dim3(256), 0, st, a);
  if (splits > 1) {
    const int64_t blocks = mdg_cdiv(N * K, 64) + (dbias ? mdg_cdiv(N, 64) : 0);
    hipLaunchKernelGGL(sum_splits_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, static_cast<const float*>(workspace), dw, N * K,
                       static_cast<int>(used), static_cast<const float*>(part_db), dbias, dbias ? N : 0);
  }
  MDG_CHECK_LAUNCH("mdg_grad_weight");
  return MDG_OK;
}

extern "C" int mdg_grad_weight(const float* g, int64_t ldg, const float* x, int64_t ldx, float* dw, float* dbias, int64_t M, int64_t N,
                               int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
  return mdg_grad_weight_prec(g, ldg, x, ldx, dw, dbias, M, N, K, MDG_PREC_F32, workspace, workspace_bytes, stream);
}
