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

struct GwArgs {
  const float* g; int64_t ldg;
  const float* x; int64_t ldx;
  float* out;                 // [S, N, K] partials (S > 1) or dW itself (S == 1)
  float* db;                  // [S, N] partial column sums of g (bias gradient) or null
  int64_t M, rows_per_split;
  int N, K;
};

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

// ---- 16-bit matrix-core form (arithmetic modes bf16 / bf16x3) ------------------------------------------------------------------
// The fp32 kernel above feeds every lane with dword loads and is bound by their latency (82 us for [106k,128]^T [106k,128], 109 MB,
// two waves per SIMD); the operand rounding modes of the dense blocks allow the 16x-faster v_mfma_f32_16x16x32_bf16 here too, and
// then the product is bound by how fast g and x arrive.  Per workgroup (256 threads, 128 x 128 output tile, a range of rows m):
//   * chunks of 32 rows: every thread loads 4 + 4 float4 (rows r0 + 8i of both operands, 16-byte coalesced), rounds them to bf16
//     (bf16x3: hi and lo planes) and stores 8-byte pieces into an LDS image [row][128 columns] -- the operands stay row-major,
//     i.e. REDUCTION-index-major, which is the wrong way round for the matrix cores;
//   * ds_read_b64_tr_b16 (the transposing LDS read of gfx950) turns 4 rows x 16 columns into "4 consecutive m of one column" per
//     lane: two of them are one 16x16x32 operand.  Row r of a chunk lives at image row pos(r) = (r&3) + 4((r>>3)&1) + 8((r>>2)&1)
//     + 16(r>>4) with a 288-byte pitch (bank 8 pos mod 64): the 8 rows one 32-lane half reads at once sit on 8 disjoint bank
//     groups (conflict-free, MI355X_MICROARCH.md "LDS");
//   * the next chunk's global loads are in flight (registers) while the matrix cores work; two LDS buffers, one barrier per chunk.
// fp32 accumulation over the workgroup's rows, partial tiles summed in split order by sum_splits_kernel as above.  The bias
// gradient is summed from the fp32 g the loader holds (never from the rounded values).
constexpr int G16_ROWS = 32;
constexpr int G16_PITCH = 288;
constexpr int G16_PLANE = G16_ROWS * G16_PITCH;

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 g16_operand(const char* at) {           // rows 8G+q and 8G+4+q of the lane group: image rows `at`, `at` + 8 rows
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4*>(reinterpret_cast<uintptr_t>(at)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4*>(reinterpret_cast<uintptr_t>(at + 8 * G16_PITCH)));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int MODE>
__global__ __launch_bounds__(256) void grad_weight16_kernel(const GwArgs p) {
  constexpr bool X3 = (MODE == MDG_PREC_BF16X3);
  constexpr int OPB = (X3 ? 2 : 1) * G16_PLANE;              // bytes of one operand's planes
  constexpr int BUF = 2 * OPB;                               // g | x
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int c4 = tid & 31, r0 = tid >> 5;                    // loader role: float4 column, first row (rows r0 + 8 i)
  const int n0 = blockIdx.y * 128, k0 = blockIdx.x * 128;
  const int64_t m0 = static_cast<int64_t>(blockIdx.z) * p.rows_per_split;
  const int64_t m1 = m0 + p.rows_per_split < p.M ? m0 + p.rows_per_split : p.M;
  const bool g_ok = n0 + 4 * c4 < p.N, x_ok = k0 + 4 * c4 < p.K;          // N, K are multiples of 4: a float4 is inside or outside
  const float* gp = p.g + (g_ok ? n0 + 4 * c4 : 0);
  const float* xp = p.x + (x_ok ? k0 + 4 * c4 : 0);
  f32x4 gq[4], xq[4];
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  auto load = [&](int64_t m) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t r = m + r0 + 8 * i;
      const bool ok = r < m1;
      gq[i] = (ok && g_ok) ? *reinterpret_cast<const f32x4*>(gp + r * p.ldg) : zero;
      xq[i] = (ok && x_ok) ? *reinterpret_cast<const f32x4*>(xp + r * p.ldx) : zero;
    }
  };
  const int wpos = (r0 & 3) + 8 * ((r0 >> 2) & 1);            // image row of chunk row r0 + 8 i: wpos + 4 (i & 1) + 16 (i >> 1)
  auto stash = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      char* at = buf + (wpos + 4 * (i & 1) + 16 * (i >> 1)) * G16_PITCH + c4 * 8;
      bsum += gq[i];
#pragma unroll
      for (int op = 0; op < 2; ++op) {
        const f32x4 v = op == 0 ? gq[i] : xq[i];
        bf16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          __bf16 hi, lo;
          mdg_split_bf16(v[e], hi, lo);
          h[e] = hi;
          l[e] = lo;
        }
        *reinterpret_cast<bf16x4*>(at + op * OPB) = h;
        if constexpr (X3) *reinterpret_cast<bf16x4*>(at + op * OPB + G16_PLANE) = l;
      }
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = zero;
  // operand read role: lane 4q+pp of 16-lane group G supplies row 8G + q (then 8G + 4 + q), columns 4pp..4pp+3 of the 16-column tile
  const int l16 = lane & 15, q = l16 >> 2, pp = l16 & 3, G = lane >> 4;
  const int rrow = q + 4 * (G & 1) + 16 * (G >> 1);
  const int a_off = rrow * G16_PITCH + (wr * 64 + 4 * pp) * 2;
  const int b_off = OPB + rrow * G16_PITCH + (wc * 64 + 4 * pp) * 2;
  auto compute = [&](const char* buf) {
    bf16x8 ah[4], bh[4], al[4], bl[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      ah[t] = g16_operand(buf + a_off + 32 * t);
      bh[t] = g16_operand(buf + b_off + 32 * t);
      if constexpr (X3) {
        al[t] = g16_operand(buf + a_off + 32 * t + G16_PLANE);
        bl[t] = g16_operand(buf + b_off + 32 * t + G16_PLANE);
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if constexpr (X3) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
        }
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
      }
  };
  const int64_t nchunk = (m1 - m0 + G16_ROWS - 1) / G16_ROWS;
  load(m0);
  stash(smem);
  if (nchunk > 1) load(m0 + G16_ROWS);
  __syncthreads();
  for (int64_t c = 0; c < nchunk; ++c) {
    compute(smem + (c & 1) * BUF);
    if (c + 1 < nchunk) stash(smem + ((c + 1) & 1) * BUF);
    if (c + 2 < nchunk) load(m0 + (c + 2) * G16_ROWS);
    __syncthreads();
  }
  // acc[a][b][i]: row n = n0 + 64 wr + 16 a + 4 (lane >> 4) + i, column k = k0 + 64 wc + 16 b + (lane & 15)
  float* out = p.out + static_cast<int64_t>(blockIdx.z) * p.N * p.K;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int k = k0 + 64 * wc + 16 * b + l16;
      if (k >= p.K) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = n0 + 64 * wr + 16 * a + 4 * G + i;
        if (n < p.N) out[static_cast<int64_t>(n) * p.K + k] = acc[a][b][i];
      }
    }
  if (p.db && blockIdx.x == 0) {                              // column sums of g: 8 row groups x 128 columns through LDS (all chunks are behind the last barrier)
    float* red = reinterpret_cast<float*>(smem);
    *reinterpret_cast<f32x4*>(red + r0 * 128 + 4 * c4) = bsum;
    __syncthreads();
    if (tid < 128 && n0 + tid < p.N) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) s += red[r * 128 + tid];
      p.db[static_cast<int64_t>(blockIdx.z) * p.N + n0 + tid] = s;
    }
  }
}

// 16-bit form: ~2 workgroups per CU, at least 512 rows per split (a partial tile costs as much traffic as 64 rows of both operands;
// measured: 1-tile outputs 34 -> 31 us with 512 instead of 256 rows, wide ones unchanged)
int pick_splits16(int64_t M, int64_t N, int64_t K) {
  static MdgEnvInt target_sw{"MDG_GW16_TARGET", 512}, minrows_sw{"MDG_GW16_MINROWS", 512};
  const int64_t tiles = mdg_cdiv(N, 128) * mdg_cdiv(K, 128);
  int64_t s = mdg_cdiv(target_sw.get(), tiles);
  const int64_t mr = minrows_sw.get();
  const int64_t max_s = M / mr > 1 ? M / mr : 1;
  if (s > max_s) s = max_s;
  if (s > 4096) s = 4096;
  return static_cast<int>(s < 1 ? 1 : s);
}

bool use16(int precision, const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t N, int64_t K) {
  static MdgEnvInt sw{"MDG_GRAD_WEIGHT_16", 1};
  return (precision == MDG_PREC_BF16 || precision == MDG_PREC_BF16X3) && sw.get() != 0 && N % 4 == 0 && K % 4 == 0 && ldg % 4 == 0 && ldx % 4 == 0 &&
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
