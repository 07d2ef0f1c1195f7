// TN products on the 16-bit matrix cores: out[n][k] = sum_m g[m][n] x[m][k] for row-major fp32 g and x (the reduction index m is
// the ROW of both operands).  Included by grad_weight.hip (weight gradients of the dense blocks) and head_train.hip (dW of the
// gathered bilinear head, where the rows of both operands are gathered through index lists).
#pragma once
#include "mdg_common.h"

struct GwArgs {
  const float* g; int64_t ldg;
  const float* x; int64_t ldx;
  float* out;                 // [S, N, K] partials (S > 1) or dW itself (S == 1)
  float* db;                  // [S, N] partial column sums of g (bias gradient) or null
  int64_t M, rows_per_split;
  int N, K;
  // gathered form (grad_weight16_kernel<MODE, true>): split z covers the rows [chunk_start[z], chunk_start[z + 1]) of the index lists;
  // row r of the operands is g[gidx[r]] * row_scale[r] and x[xidx[r]] (a null list = the identity, a null scale = 1)
  const int64_t* gidx; const int64_t* xidx; const float* row_scale; const int64_t* chunk_start;
};

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

template <int MODE, bool GATHER = false>
__global__ __launch_bounds__(256) void grad_weight16_kernel(const GwArgs p) {
  constexpr bool X3 = (MODE == MDG_PREC_BF16X3);
  constexpr int OPB = (X3 ? 2 : 1) * G16_PLANE;              // bytes of one operand's planes
  constexpr int BUF = 2 * OPB;                               // g | x
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int c4 = tid & 31, r0 = tid >> 5;                    // loader role: float4 column, first row (rows r0 + 8 i)
  const int n0 = blockIdx.y * 128, k0 = blockIdx.x * 128;
  int64_t m0, m1;
  if constexpr (GATHER) {
    m0 = p.chunk_start[blockIdx.z];
    m1 = p.chunk_start[blockIdx.z + 1];
  } else {
    m0 = static_cast<int64_t>(blockIdx.z) * p.rows_per_split;
    m1 = m0 + p.rows_per_split < p.M ? m0 + p.rows_per_split : p.M;
  }
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
      if constexpr (GATHER) {
        const int64_t rr = ok ? r : m1 - 1;
        const int64_t gi = p.gidx ? p.gidx[rr] : rr, xi = p.xidx ? p.xidx[rr] : rr;
        const float sc = ok ? (p.row_scale ? p.row_scale[rr] : 1.f) : 0.f;
        gq[i] = g_ok ? *reinterpret_cast<const f32x4*>(gp + gi * p.ldg) * sc : zero;
        xq[i] = (ok && x_ok) ? *reinterpret_cast<const f32x4*>(xp + xi * p.ldx) : zero;
      } else {
        gq[i] = (ok && g_ok) ? *reinterpret_cast<const f32x4*>(gp + r * p.ldg) : zero;
        xq[i] = (ok && x_ok) ? *reinterpret_cast<const f32x4*>(xp + r * p.ldx) : zero;
      }
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
  if (nchunk > 0) {
    load(m0);
    stash(smem);
    if (nchunk > 1) load(m0 + G16_ROWS);
  }
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

