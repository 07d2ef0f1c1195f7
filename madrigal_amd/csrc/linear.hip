// Fused dense block for gfx950:   Y = alpha * act( (X . W^T + bias) * scale + shift ) + beta * R
//
// X [M,K] (row stride ldx), W [N,K] (torch nn.Linear layout, row stride ldw), Y [M,N] (ldy), all fp32.
// One kernel serves every dense block on the path: the cv MLP (madrigal/models/models.py:178-180),
// the chemCPA encoder Linear+BatchNorm(eval)+ReLU blocks (madrigal/chemcpa/chemCPA/model.py:226-231;
// BN folded into scale/shift), the transformer's in/out projections and FFN with GELU and the
// residual add (nn.TransformerEncoderLayer, models.py:366), the GIN MLPs and the HGT projections.
//
// 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles of
// 32x32), BK = 32.  Tiles are staged global -> registers -> LDS (double buffered, prefetch issued
// before the MFMA phase); in the bf16 modes the fp32 operands are split hi/lo while they sit in
// registers, so LDS holds ready-made MFMA fragments (ds_read_b128, XOR-swizzled => conflict free).
// Arithmetic modes as in the head: exact fp32 MFMA, bf16x3 (fp32-grade), bf16.
#include "mdg_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
constexpr int TILE_BYTES = BM * BK * 4;     // one operand, one stage: 16 KB (fp32) == hi 8 KB + lo 8 KB
constexpr int LO_OFF = BM * BK * 2;

struct LinearArgs {
  const float* x; int64_t ldx;
  const float* w; int64_t ldw;
  float* y; int64_t ldy;
  const float* bias; const float* scale; const float* shift;
  const float* res; int64_t ldr;
  float alpha, beta;
  int act;
  int64_t M, N, K;
};

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case MDG_ACT_RELU: return fmaxf(v, 0.f);
    case MDG_ACT_GELU: return mdg_gelu(v);
    case MDG_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case MDG_ACT_TANH: return tanhf(v);
    case MDG_ACT_LEAKYRELU: return v >= 0.f ? v : 0.01f * v;
    case MDG_ACT_SOFTPLUS: return v > 20.f ? v : log1pf(expf(v));
    case MDG_ACT_SELU: return 1.0507009873554804934193349852946f * (v > 0.f ? v : 1.6732632423543772848170429916717f * (expf(v) - 1.f));
    default: return v;
  }
}

// fp32 tile [128][32]: 128-B rows, 8 chunks; bf16 tile [128][32]: 64-B rows, 4 chunks.
__device__ __forceinline__ int off_f32(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int off_bf16(int row, int cb) { return row * 64 + ((cb ^ ((row >> 2) & 3)) << 4); }

__device__ __forceinline__ void load_tile(const float* base, int64_t ld, int64_t row0, int64_t nrows, int64_t k0, int64_t K,
                                          int tid, f32x4 (&regs)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int g = tid + NT * i, row = g >> 3, c = g & 7;
    int64_t gr = row0 + row;
    gr = gr < nrows ? gr : nrows - 1;
    const int64_t k = k0 + 4 * c;
    const int64_t kk = k < K ? k : 0;             // keep the address in range; value is zeroed below
    f32x4 v = *reinterpret_cast<const f32x4*>(base + gr * ld + kk);
    if (k >= K) v = f32x4{0.f, 0.f, 0.f, 0.f};
    regs[i] = v;
  }
}

template <int MODE>
__device__ __forceinline__ void write_tile(char* lds, int tid, const f32x4 (&regs)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int g = tid + NT * i, row = g >> 3, c = g & 7;
    if constexpr (MODE == MDG_PREC_F32) {
      *reinterpret_cast<f32x4*>(lds + off_f32(row, c)) = regs[i];
    } else {
      bf16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        __bf16 a, b;
        mdg_split_bf16(regs[i][e], a, b);
        hi[e] = a;
        lo[e] = b;
      }
      const int o = off_bf16(row, c >> 1) + (c & 1) * 8;
      *reinterpret_cast<bf16x4*>(lds + o) = hi;
      if constexpr (MODE == MDG_PREC_BF16X3) *reinterpret_cast<bf16x4*>(lds + LO_OFF + o) = lo;
    }
  }
}

template <int MODE>
__device__ __forceinline__ void mma_stage(const char* la, const char* lb, int wr, int wc, int r, int h, f32x16 (&acc)[2][2]) {
  if constexpr (MODE == MDG_PREC_F32) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = *reinterpret_cast<const f32x4*>(la + off_f32(wr * 64 + t * 32 + r, 2 * q + h));
        b[t] = *reinterpret_cast<const f32x4*>(lb + off_f32(wc * 64 + t * 32 + r, 2 * q + h));
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][e], b[nt][e], acc[mt][nt], 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int oa = off_bf16(wr * 64 + t * 32 + r, 2 * s + h), ob = off_bf16(wc * 64 + t * 32 + r, 2 * s + h);
        ah[t] = *reinterpret_cast<const bf16x8*>(la + oa);
        bh[t] = *reinterpret_cast<const bf16x8*>(lb + ob);
        if constexpr (MODE == MDG_PREC_BF16X3) {
          al[t] = *reinterpret_cast<const bf16x8*>(la + LO_OFF + oa);
          bl[t] = *reinterpret_cast<const bf16x8*>(lb + LO_OFF + ob);
        }
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          if constexpr (MODE == MDG_PREC_BF16X3) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
          }
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
        }
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(NT, 2) void linear_kernel(const LinearArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][A tile | B tile]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t col0 = static_cast<int64_t>(blockIdx.x) * BN, row0 = static_cast<int64_t>(blockIdx.y) * BM;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

  const int nk = static_cast<int>((p.K + BK - 1) / BK);
  f32x4 ra[4], rb[4];
  load_tile(p.x, p.ldx, row0, p.M, 0, p.K, tid, ra);
  load_tile(p.w, p.ldw, col0, p.N, 0, p.K, tid, rb);
  write_tile<MODE>(smem, tid, ra);
  write_tile<MODE>(smem + TILE_BYTES, tid, rb);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    char* const cur = smem + (kt & 1) * 2 * TILE_BYTES;
    char* const nxt = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
    // unconditional prefetch (past the end k >= K zero-fills; the tile is never consumed)
    load_tile(p.x, p.ldx, row0, p.M, static_cast<int64_t>(kt + 1) * BK, p.K, tid, ra);
    load_tile(p.w, p.ldw, col0, p.N, static_cast<int64_t>(kt + 1) * BK, p.K, tid, rb);
    mma_stage<MODE>(cur, cur + TILE_BYTES, wr, wc, r, h, acc);
    write_tile<MODE>(nxt, tid, ra);
    write_tile<MODE>(nxt + TILE_BYTES, tid, rb);
    __syncthreads();
  }

  // ---- epilogue: lane = output column, accumulator registers = rows ---------------------
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int64_t n = col0 + wc * 64 + nt * 32 + r;
    const bool n_ok = n < p.N;
    const float bias = (p.bias && n_ok) ? p.bias[n] : 0.f;
    const float scale = (p.scale && n_ok) ? p.scale[n] : 1.f;
    const float shift = (p.shift && n_ok) ? p.shift[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int64_t m = row0 + wr * 64 + mt * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
        if (n_ok && m < p.M) {
          float val = acc[mt][nt][v] + bias;
          if (p.scale) val = val * scale + shift;
          val = apply_act(val, p.act);
          if (p.alpha != 1.0f) val *= p.alpha;
          if (p.res) val += p.beta * p.res[m * p.ldr + n];
          p.y[m * p.ldy + n] = val;
        }
      }
    }
  }
}

// ---- LayerNorm: one wave per row -------------------------------------------------------------
template <int VEC>   // floats per lane = 4*VEC, d <= 256*VEC
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ g,
                                                        const float* __restrict__ b, float* __restrict__ y, int64_t ldy,
                                                        int64_t rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  f32x4 v[VEC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int c = (lane + 64 * i) * 4;
    v[i] = c < d ? *reinterpret_cast<const f32x4*>(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mean = mdg_wave_sum(s) / d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < d) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = v[i][e] - mean;
        q += t * t;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(mdg_wave_sum(q) / d + eps);
  float* yr = y + row * ldy;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < d) {
      const f32x4 gg = *reinterpret_cast<const f32x4*>(g + c), bb = *reinterpret_cast<const f32x4*>(b + c);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
      *reinterpret_cast<f32x4*>(yr + c) = o;
    }
  }
}

}  // namespace

extern "C" int mdg_linear(const float* x, int64_t ldx, const float* w, int64_t ldw, float* y, int64_t ldy, int64_t M, int64_t N,
                          int64_t K, const float* bias, const float* scale, const float* shift, int act, const float* residual,
                          int64_t ldr, float alpha, float beta, int precision, void* stream) {
  MDG_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "mdg_linear: negative size");
  if (M == 0 || N == 0) return MDG_OK;
  MDG_CHECK_ARG(x && w && y, "mdg_linear: null pointer");
  MDG_CHECK_ARG(K > 0 && K % 4 == 0 && ldx % 4 == 0 && ldw % 4 == 0 && ldx >= K && ldw >= K,
                "mdg_linear: K, ldx, ldw must be multiples of 4 with ld >= K (K=%lld ldx=%lld ldw=%lld); zero-pad the inner dimension",
                (long long)K, (long long)ldx, (long long)ldw);
  MDG_CHECK_ARG(mdg_aligned16(x) && mdg_aligned16(w), "mdg_linear: x and w must be 16-byte aligned");
  MDG_CHECK_ARG(ldy >= N && (!residual || ldr >= N || ldr == 0), "mdg_linear: ldy/ldr smaller than N (ldr == 0 broadcasts one row)");
  MDG_CHECK_ARG((scale == nullptr) == (shift == nullptr), "mdg_linear: scale and shift come together");
  MDG_CHECK_ARG(act >= MDG_ACT_NONE && act <= MDG_ACT_SELU, "mdg_linear: unknown activation %d", act);
  MDG_CHECK_ARG(mdg_cdiv(M, BM) <= 65535, "mdg_linear: M too large for one launch");
  LinearArgs a{x, ldx, w, ldw, y, ldy, bias, scale, shift, residual, ldr, alpha, beta, act, M, N, K};
  const dim3 grid(static_cast<unsigned>(mdg_cdiv(N, BN)), static_cast<unsigned>(mdg_cdiv(M, BM)));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t lds = 4 * TILE_BYTES;
  switch (precision) {
    case MDG_PREC_F32: hipLaunchKernelGGL(linear_kernel<MDG_PREC_F32>, grid, dim3(NT), lds, st, a); break;
    case MDG_PREC_BF16X3: hipLaunchKernelGGL(linear_kernel<MDG_PREC_BF16X3>, grid, dim3(NT), lds, st, a); break;
    case MDG_PREC_BF16: hipLaunchKernelGGL(linear_kernel<MDG_PREC_BF16>, grid, dim3(NT), lds, st, a); break;
    default: mdg_set_error("mdg_linear: unknown precision %d", precision); return MDG_EINVAL;
  }
  MDG_CHECK_LAUNCH("mdg_linear");
  return MDG_OK;
}

extern "C" int mdg_layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy,
                             int64_t rows, int64_t d, float eps, void* stream) {
  MDG_CHECK_ARG(rows >= 0 && d > 0, "mdg_layernorm: bad shape");
  if (rows == 0) return MDG_OK;
  MDG_CHECK_ARG(x && gamma && beta && y, "mdg_layernorm: null pointer");
  MDG_CHECK_ARG(d % 4 == 0 && d <= 2048 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= d && ldy >= d,
                "mdg_layernorm: d must be a multiple of 4 and <= 2048 (got %lld)", (long long)d);
  MDG_CHECK_ARG(mdg_aligned16(x) && mdg_aligned16(y) && mdg_aligned16(gamma) && mdg_aligned16(beta), "mdg_layernorm: 16-byte alignment");
  const dim3 grid(static_cast<unsigned>(mdg_cdiv(rows, 4)));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int di = static_cast<int>(d);
  if (d <= 256) hipLaunchKernelGGL(layernorm_kernel<1>, grid, dim3(256), 0, st, x, ldx, gamma, beta, y, ldy, rows, di, eps);
  else if (d <= 512) hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, st, x, ldx, gamma, beta, y, ldy, rows, di, eps);
  else if (d <= 1024) hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, st, x, ldx, gamma, beta, y, ldy, rows, di, eps);
  else hipLaunchKernelGGL(layernorm_kernel<8>, grid, dim3(256), 0, st, x, ldx, gamma, beta, y, ldy, rows, di, eps);
  MDG_CHECK_LAUNCH("mdg_layernorm");
  return MDG_OK;
}
