// Cross-modal fusion kernels for gfx950: token assembly, small-sequence self-attention on the fp32
// matrix cores, and the one-query cross-attention pooling.  (madrigal/models/models.py:401-455,
// 775-853; nn.TransformerEncoderLayer / nn.MultiheadAttention semantics, eval mode.)
#include "mdg_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Token assembly: seq[n, S, D] = [cls?] str kg cv [bottleneck x nb] tx_0..tx_15, optional per-token
// L2 normalisation, then + pe[s] for s < pe_len.  One 32-lane group (float4 per lane) per token.
// ---------------------------------------------------------------------------------------------
struct AssembleArgs {
  const float* str; const float* kg; const float* cv; const float* tx;   // [n_src,128] x3, [16*n_src,128]
  const float* bottleneck; const float* cls; const float* pe;            // [nb,128], [128], [pe_len,128]
  const int64_t* rows;                                                   // [n] source row per output row, or null
  const int64_t* token_index;                                            // [n_tok] drug*S+s of the tokens to emit, or null (= all)
  int64_t n_tok;
  float* seq;
  int64_t n, n_src;
  int nb, has_cls, pe_len, normalize;
};

__global__ __launch_bounds__(256) void assemble_tokens_kernel(const AssembleArgs p) {
  const int S = p.has_cls + 3 + p.nb + 16;
  const int64_t slot = static_cast<int64_t>(blockIdx.x) * 8 + (threadIdx.x >> 5);
  const int sub = threadIdx.x & 31;
  if (slot >= (p.token_index ? p.n_tok : p.n * S)) return;
  const int64_t tok = p.token_index ? p.token_index[slot] : slot;
  const int64_t i = tok / S;
  const int s = static_cast<int>(tok % S);
  const int64_t src = p.rows ? p.rows[i] : i;
  int t = s - p.has_cls;
  const float* row;
  bool learned = false;
  if (t < 0) { row = p.cls; learned = true; }
  else if (t == 0) row = p.str + src * 128;
  else if (t == 1) row = p.kg + src * 128;
  else if (t == 2) row = p.cv + src * 128;
  else if (t < 3 + p.nb) { row = p.bottleneck + (t - 3) * 128; learned = true; }
  else row = p.tx + (static_cast<int64_t>(t - 3 - p.nb) * p.n_src + src) * 128;
  (void)learned;
  f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * sub);
  if (p.normalize) {
    float q = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float inv = 1.0f / fmaxf(sqrtf(q), 1e-12f);      // F.normalize(p=2, eps=1e-12)
    v *= inv;
  }
  if (s < p.pe_len) v += *reinterpret_cast<const f32x4*>(p.pe + s * 128 + 4 * sub);
  *reinterpret_cast<f32x4*>(p.seq + slot * 128 + 4 * sub) = v;
}

// ---------------------------------------------------------------------------------------------
// Self-attention over S <= 32 tokens, one wave per (drug, head), on v_mfma_f32_32x32x2_f32.
//   P^T = softmax_keys( K (Q/sqrt(dh))^T + masks )   computed "swapped" so that the query sits on the
//   lane and the 32 keys of a row sit in the accumulator registers of the two lane halves: the
//   softmax is in-register plus one cross-half exchange.
//   O^T = V^T P^T : the P^T accumulator is consumed as the B operand as it stands (the second product
//   sums over P^T's row index), V is read with the lane on the feature column (coalesced).
// k order of the first product is permuted (k = 8q+4h+e) identically for Q and K so that each lane
// reads its row in 16-byte pieces.
// ---------------------------------------------------------------------------------------------
struct AttnArgs {
  const float* qkv; int64_t ld;      // [n*S, 3d]: q | k | v
  float* out; int64_t ldo;           // [n*S, d]
  const uint32_t* kpm_bits;          // [n] bit j set = key j masked for this drug, or null
  const uint32_t* src_bits;          // [S] bit j set = query i may not attend key j, or null
  float* probs;                      // [n,H,S,S] or null
  const int64_t* row_start;          // compact mode: [n+1] first row of each drug's LIVE tokens (null = dense [n,S])
  const uint32_t* row_bits;          // compact mode: [R] per query row, bit j = its j-th live key is not allowed
  int64_t n;
  int S, H, dh;
  float qscale;
  float p_drop; uint64_t seed;       // training: dropout on the attention weights (nn.MultiheadAttention(dropout=p))
  const float* dout; int64_t lddo;   // backward: gradient of `out`
  float* dqkv; int64_t lddq;         // backward: gradient of qkv, same layout
};

// Operand staging.  The 32x32x2 MFMA wants lane x to hold ROW x of its operand, but a row-per-lane global load touches 32
// rows 24 KB apart for 32 bytes each.  So a wave first copies the 32 rows x 64 floats it is about to consume into its own
// LDS tile with row-major loads (16 lanes x 16 B = one 256-byte row piece, 4 rows per instruction) and reads its row back from
// there.  64 floats per row with the 16-byte piece index XORed by the row: the row reads of 16 consecutive lanes fall on 16
// distinct pieces.  No barrier: a wave's LDS operations execute in order and the tile is private to the wave.
constexpr int kTileFloats = 32 * 64;
__device__ __forceinline__ int tile_off(int row, int piece) { return row * 64 + ((piece ^ (row & 15)) << 2); }

__device__ __forceinline__ void load_rows(const float* mat, int64_t ld, int T, int k0, int dh, int lane, f32x4 (&r)[8]) {
  const int k = k0 + 4 * (lane & 15);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (lane >> 4) + 4 * i;
    const int rr = row < T ? row : T - 1;                      // rows past the tile repeat its last row (they meet zero weights)
    r[i] = k < dh ? *reinterpret_cast<const f32x4*>(mat + static_cast<int64_t>(rr) * ld + k) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

__device__ __forceinline__ void write_rows(const f32x4 (&r)[8], float* tile, int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(tile + tile_off((lane >> 4) + 4 * i, lane & 15)) = r[i];
}

// out[i][c] = sum_j w[i][j] m[j][c] for the 32 x 32 weights a wave holds in the swapped layout (lane x = row i, registers =
// j) and a row-major operand m (32 rows x dh): the P V product of the forward pass, and dS K, dS^T Q, Pd^T dO of the backward
// pass.  m is staged 64 columns at a time, the next piece in flight (in registers) while the current one feeds the MFMAs from
// LDS; with the weights as the A operand the result comes out with the lane on the column, so every store is a 128-byte
// row piece.
__device__ __forceinline__ void weights_rows_product(const f32x16& w, const float* m, int64_t ldm, float* out, int64_t ldo, int T, int dh,
                                                     float* ta, float* tb, int x, int half, int lane) {
  f32x4 pre[8];
  load_rows(m, ldm, T, 0, dh, lane, pre);
  write_rows(pre, ta, lane);
  for (int c0 = 0, it = 0; c0 < dh; c0 += 64, ++it) {
    const float* cur = (it & 1) ? tb : ta;
    float* nxt = (it & 1) ? ta : tb;
    const bool more = c0 + 64 < dh;
    if (more) load_rows(m, ldm, T, c0 + 64, dh, lane, pre);
#pragma unroll
    for (int hc = 0; hc < 2; ++hc) {
      const int col = hc * 32 + x;
      f32x16 o;
#pragma unroll
      for (int v = 0; v < 16; ++v) o[v] = 0.f;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int j = (s & 3) + 8 * (s >> 2) + 4 * half;
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(w[s], cur[tile_off(j, col >> 2) + (col & 3)], o, 0, 0, 0);
      }
      if (c0 + col < dh) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int i = (v & 3) + 8 * (v >> 2) + 4 * half;
          if (i < T) out[static_cast<int64_t>(i) * ldo + c0 + col] = o[v];
        }
      }
    }
    if (more) write_rows(pre, nxt, lane);
  }
}

// acc += B^T-style product of the swapped layout: acc[v] (lane x) += sum_k a[j_v][k] * b[x][k] over the dh columns of two
// row-major operands a (rows -> accumulator registers) and b (rows -> lanes), b scaled by `bscale`.
__device__ __forceinline__ void rows_product(const float* a, int64_t lda, const float* b, int64_t ldb, int T, int dh, float bscale, float* ta,
                                             float* tb, int x, int half, int lane, f32x16& acc) {
  f32x4 ra[8], rb[8];
  load_rows(a, lda, T, 0, dh, lane, ra);
  load_rows(b, ldb, T, 0, dh, lane, rb);
  for (int k0 = 0; k0 < dh; k0 += 64) {
    write_rows(ra, ta, lane);
    write_rows(rb, tb, lane);
    if (k0 + 64 < dh) {                                        // the next 64 columns travel while these feed the MFMAs
      load_rows(a, lda, T, k0 + 64, dh, lane, ra);
      load_rows(b, ldb, T, k0 + 64, dh, lane, rb);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 af = *reinterpret_cast<const f32x4*>(ta + tile_off(x, 2 * q + half));
      const f32x4 bf = *reinterpret_cast<const f32x4*>(tb + tile_off(x, 2 * q + half));
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e] * bscale, acc, 0, 0, 0);
    }
  }
}

// Scores -> attention weights of one (tile, head) wave, in the swapped layout: lane (x = query, half) holds
// acc[v] = P[x][j], j = (v&3) + 8(v>>2) + 4*half.  Shared by the forward and the backward kernel.  qmat / kmat: the tile's
// first row of q / k at this head's columns; ta / tb: the wave's two staging tiles.
__device__ __forceinline__ void attn_weights(const AttnArgs& p, const float* qmat, const float* kmat, int64_t drug, int64_t base, int x, int xr, int T,
                                             int half, int lane, float* ta, float* tb, f32x16& acc) {
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  rows_product(kmat, p.ld, qmat, p.ld, T, p.dh, p.qscale, ta, tb, x, half, lane, acc);
  const uint32_t blocked = p.row_start ? (p.row_bits ? p.row_bits[base + xr] : 0u)
                                       : ((p.kpm_bits ? p.kpm_bits[drug] : 0u) | (p.src_bits ? p.src_bits[xr] : 0u));
  float m = -INFINITY;
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const int j = (v & 3) + 8 * (v >> 2) + 4 * half;
    if (j >= T || ((blocked >> j) & 1u)) acc[v] = -INFINITY;
    m = fmaxf(m, acc[v]);
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    acc[v] = expf(acc[v] - m);         // a fully masked row gives exp(-inf - -inf) = NaN, as torch does
    sum += acc[v];
  }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] *= inv;
}

// keep / (1-p) factor of attention weight (wave gw, query x, key j)
__device__ __forceinline__ float attn_drop_scale(const AttnArgs& p, int64_t gw, int x, int j, uint32_t thr, float keep_scale) {
  return mdg_keep(p.seed, (static_cast<uint64_t>(gw) * 32 + x) * 32 + j, thr) ? keep_scale : 0.f;
}

__global__ __launch_bounds__(256) void fusion_attention_kernel(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) float stage[4][2][kTileFloats];
  const int lane = threadIdx.x & 63, x = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
  const int64_t gw = static_cast<int64_t>(blockIdx.x) * 4 + wave;
  if (gw >= p.n * p.H) return;
  const int64_t drug = gw / p.H;
  const int head = static_cast<int>(gw % p.H);
  const int d = p.H * p.dh;
  // dense: the drug's S tokens are rows drug*S ..; compact: only its live tokens, rows row_start[drug] ..
  const int64_t base = p.row_start ? p.row_start[drug] : drug * p.S;
  const int T = p.row_start ? static_cast<int>(p.row_start[drug + 1] - base) : p.S;
  if (T <= 0) return;
  const int xr = x < T ? x : T - 1;
  const float* qmat = p.qkv + base * p.ld + head * p.dh;

  f32x16 acc;
  attn_weights(p, qmat, qmat + d, drug, base, x, xr, T, half, lane, stage[wave][0], stage[wave][1], acc);

  if (p.probs && x < T) {
    float* pr = p.probs + ((drug * p.H + head) * p.S + x) * p.S;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int j = (v & 3) + 8 * (v >> 2) + 4 * half;
      if (j < T) pr[j] = acc[v];
    }
  }

  if (p.p_drop > 0.f) {
    const uint32_t thr = mdg_drop_threshold(p.p_drop);
    const float ks = 1.0f / (1.0f - p.p_drop);
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] *= attn_drop_scale(p, gw, x, (v & 3) + 8 * (v >> 2) + 4 * half, thr, ks);
  }

  weights_rows_product(acc, qmat + 2 * d, p.ld, p.out + base * p.ldo + head * p.dh, p.ldo, T, p.dh, stage[wave][0], stage[wave][1], x, half, lane);
}

// ---------------------------------------------------------------------------------------------
// Backward of the self-attention above, same one-wave-per-(tile, head) decomposition.  The attention weights
// are recomputed (nothing but qkv and the dropout seed is kept from the forward pass):
//   dPd = dO V^T            (same swapped MFMA shape as the scores: lane = query, registers = keys)
//   Pd  = P o D/(1-p),  dP = dPd o D/(1-p),  dS = P o (dP - rowsum(dP o P))
//   dQ  = qscale dS K       (contraction over keys: the accumulator layout feeds the MFMA B operand directly)
//   dK  = qscale dS^T Q,  dV = Pd^T dO   (contraction over queries: dS / Pd are transposed through a 32x33 LDS tile)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fusion_attention_bwd_kernel(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) float stage[4][2][kTileFloats];       // staging tiles, later the two 32x33 transpose tiles
  static_assert(kTileFloats >= 32 * 33, "the transpose tiles reuse the staging tiles");
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63, x = lane & 31, half = lane >> 5;
  const int64_t gw = static_cast<int64_t>(blockIdx.x) * 4 + wave;
  if (gw >= p.n * p.H) return;
  const int64_t drug = gw / p.H;
  const int head = static_cast<int>(gw % p.H);
  const int d = p.H * p.dh;
  const int64_t base = p.row_start ? p.row_start[drug] : drug * p.S;
  const int T = p.row_start ? static_cast<int>(p.row_start[drug + 1] - base) : p.S;
  if (T <= 0) return;
  const int xr = x < T ? x : T - 1;
  const float* qmat = p.qkv + base * p.ld + head * p.dh;
  float* const ta = stage[wave][0];
  float* const tb = stage[wave][1];

  f32x16 P;
  attn_weights(p, qmat, qmat + d, drug, base, x, xr, T, half, lane, ta, tb, P);

  // dPd[x][j] = sum_c dO[x][c] V[j][c]
  f32x16 dP;
#pragma unroll
  for (int v = 0; v < 16; ++v) dP[v] = 0.f;
  rows_product(qmat + 2 * d, p.ld, p.dout + base * p.lddo + head * p.dh, p.lddo, T, p.dh, 1.0f, ta, tb, x, half, lane, dP);

  f32x16 Pd = P;
  if (p.p_drop > 0.f) {
    const uint32_t thr = mdg_drop_threshold(p.p_drop);
    const float ks = 1.0f / (1.0f - p.p_drop);
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const float f = attn_drop_scale(p, gw, x, (v & 3) + 8 * (v >> 2) + 4 * half, thr, ks);
      Pd[v] *= f;
      dP[v] *= f;
    }
  }
  float delta = 0.f;
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const int j = (v & 3) + 8 * (v >> 2) + 4 * half;
    if (j >= T) dP[v] = 0.f;                       // clamped key rows: P is exactly 0 there, keep 0 * garbage out
    delta += dP[v] * P[v];
  }
  delta += __shfl_xor(delta, 32, 64);
  f32x16 dS;
#pragma unroll
  for (int v = 0; v < 16; ++v) dS[v] = (x < T) ? P[v] * (dP[v] - delta) * p.qscale : 0.f;
  if (x >= T) {
#pragma unroll
    for (int v = 0; v < 16; ++v) Pd[v] = 0.f;      // duplicate query rows must not enter the sums over queries
  }

  // dQ[x][c] = sum_j dS[x][j] K[j][c]   (as the forward P V product, with K in place of V)
  float* const dq = p.dqkv + base * p.lddq + head * p.dh;
  weights_rows_product(dS, qmat + d, p.ld, dq, p.lddq, T, p.dh, ta, tb, x, half, lane);

  // transpose dS and Pd through LDS: afterwards lane (j, half) holds X[x'][j] for x' = (s&3) + 8(s>>2) + 4*half
  float (*tS)[33] = reinterpret_cast<float (*)[33]>(ta);
  float (*tP)[33] = reinterpret_cast<float (*)[33]>(tb);
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const int j = (v & 3) + 8 * (v >> 2) + 4 * half;
    tS[x][j] = dS[v];
    tP[x][j] = Pd[v];
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the LDS writes of this wave have landed
  f32x16 dSt, Pdt;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int xq = (s & 3) + 8 * (s >> 2) + 4 * half;
    dSt[s] = tS[xq][x];
    Pdt[s] = tP[xq][x];
  }

  // dK[j][c] = sum_x dS[x][j] Q[x][c],  dV[j][c] = sum_x Pd[x][j] dO[x][c]   (lane = key j after the transpose)
  weights_rows_product(dSt, qmat, p.ld, dq + d, p.lddq, T, p.dh, ta, tb, x, half, lane);
  weights_rows_product(Pdt, p.dout + base * p.lddo + head * p.dh, p.lddo, dq + 2 * d, p.lddq, T, p.dh, ta, tb, x, half, lane);
}

// ---------------------------------------------------------------------------------------------
// Cross-attention pooling with ONE query shared by every drug: out[i, head] = softmax_j(q.K_ij) V_ij.
// One wave per (drug, head); lanes span the head dimension.  (models.py:422-438; the fixed key mask
// is applied by the caller by passing only the allowed key tokens.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xattn_pool_kernel(const float* __restrict__ q, const float* __restrict__ kv, int64_t ld,
                                                         float* __restrict__ out, int64_t ldo, int64_t n, int Tk, int H, int dh,
                                                         float qscale, float p_drop, uint64_t seed) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (gw >= n * H) return;
  const uint32_t thr = mdg_drop_threshold(p_drop);
  const float ks = 1.0f / (1.0f - p_drop);
  const int64_t drug = gw / H;
  const int head = static_cast<int>(gw % H);
  const int d = H * dh;
  const int per = (dh + 63) / 64;            // <= 4 for dh <= 256
  float qv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = lane + 64 * u;
    qv[u] = (u < per && c < dh) ? q[head * dh + c] * qscale : 0.f;
  }
  // online softmax over the Tk keys (running max m, running sum, running weighted V sum)
  float m = -INFINITY, sum = 0.f;
  float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int j = 0; j < Tk; ++j) {
    const float* kr = kv + (drug * Tk + j) * ld + head * dh;
    const float* vr = kr + d;
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = lane + 64 * u;
      if (u < per && c < dh) s += qv[u] * kr[c];
    }
    s = mdg_wave_sum(s);
    const float mn = fmaxf(m, s);
    const float f = expf(m - mn), e = expf(s - mn);
    sum = sum * f + e;
    const float ed = (p_drop > 0.f && !mdg_keep(seed, static_cast<uint64_t>(gw) * 32 + j, thr)) ? 0.f : (p_drop > 0.f ? e * ks : e);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = lane + 64 * u;
      if (u < per && c < dh) o[u] = o[u] * f + ed * vr[c];
    }
    m = mn;
  }
  const float inv = 1.0f / sum;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = lane + 64 * u;
    if (u < per && c < dh) out[drug * ldo + head * dh + c] = o[u] * inv;
  }
}

// Backward of the one-query cross-attention pooling: one wave per (drug, head); lane j keeps the score, weight and
// weight gradient of key j (Tk <= 32).  dq is written per drug ([n, d]) and reduced over drugs by the caller
// (the query is a shared parameter): fixed summation order, no atomics.
__global__ __launch_bounds__(256) void xattn_pool_bwd_kernel(const float* __restrict__ q, const float* __restrict__ kv, int64_t ld,
                                                             const float* __restrict__ dout, int64_t lddo, float* __restrict__ dkv,
                                                             int64_t lddkv, float* __restrict__ dq_part, int64_t n, int Tk, int H, int dh,
                                                             float qscale, float p_drop, uint64_t seed) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (gw >= n * H) return;
  const int64_t drug = gw / H;
  const int head = static_cast<int>(gw % H);
  const int d = H * dh;
  const int per = (dh + 63) / 64;
  const uint32_t thr = mdg_drop_threshold(p_drop);
  const float ks = 1.0f / (1.0f - p_drop);
  float qv[4], gv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = lane + 64 * u;
    const bool ok = u < per && c < dh;
    qv[u] = ok ? q[head * dh + c] * qscale : 0.f;
    gv[u] = ok ? dout[drug * lddo + head * dh + c] : 0.f;
  }
  float sv = -INFINITY, dpv = 0.f, fv = 1.f;
#pragma unroll 1
  for (int j = 0; j < Tk; ++j) {
    const float* kr = kv + (drug * Tk + j) * ld + head * dh;
    const float* vr = kr + d;
    float s = 0.f, g = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = lane + 64 * u;
      if (u < per && c < dh) { s += qv[u] * kr[c]; g += gv[u] * vr[c]; }
    }
    s = mdg_wave_sum(s);
    g = mdg_wave_sum(g);
    const float f = p_drop > 0.f ? (mdg_keep(seed, static_cast<uint64_t>(gw) * 32 + j, thr) ? ks : 0.f) : 1.f;
    if (lane == j) { sv = s; dpv = g * f; fv = f; }
  }
  const float m = mdg_wave_max(sv);
  const float e = lane < Tk ? expf(sv - m) : 0.f;
  const float P = e / mdg_wave_sum(e);
  const float delta = mdg_wave_sum(P * dpv);
  const float ds = P * (dpv - delta);
  const float pd = P * fv;
  float dq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int j = 0; j < Tk; ++j) {
    const float dsj = __shfl(ds, j, 64), pdj = __shfl(pd, j, 64);
    const float* kr = kv + (drug * Tk + j) * ld + head * dh;
    float* dkr = dkv + (drug * Tk + j) * lddkv + head * dh;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = lane + 64 * u;
      if (u < per && c < dh) {
        dkr[c] = dsj * qv[u];
        dkr[d + c] = pdj * gv[u];
        dq[u] += dsj * kr[c];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = lane + 64 * u;
    if (u < per && c < dh) dq_part[drug * d + head * dh + c] = dq[u] * qscale;
  }
}

// Backward of assemble_tokens: each token's gradient goes back to its source row (through the L2 normalisation when
// it was applied); the learned tokens (cls, bottleneck) and the position table are shared by all drugs: their
// per-drug contributions are written to dense scratch ([n, S, 128], zero-filled by the caller) and summed over drugs
// by mdg_colsum.  Requires rows == null (every modality row feeds exactly one token).
struct AssembleBwdArgs {
  const float* dseq;                                                    // [n_tok or n*S, 128]
  const float* str; const float* kg; const float* cv; const float* tx;  // forward sources (for the normalisation)
  const float* bottleneck; const float* cls;
  const int64_t* token_index; int64_t n_tok;
  float* dstr; float* dkg; float* dcv; float* dtx;                      // zero-filled by the caller (non-live tokens)
  float* dlearned;                                                      // [n, S, 128] scratch or null
  float* dpe;                                                           // [n, S, 128] scratch or null
  int64_t n;
  int nb, has_cls, pe_len, normalize;
};

__global__ __launch_bounds__(256) void assemble_tokens_bwd_kernel(const AssembleBwdArgs p) {
  const int S = p.has_cls + 3 + p.nb + 16;
  const int64_t slot = static_cast<int64_t>(blockIdx.x) * 8 + (threadIdx.x >> 5);
  const int sub = threadIdx.x & 31;
  if (slot >= (p.token_index ? p.n_tok : p.n * S)) return;
  const int64_t tok = p.token_index ? p.token_index[slot] : slot;
  const int64_t i = tok / S;
  const int s = static_cast<int>(tok % S);
  const int t = s - p.has_cls;
  const float* row;
  float* drow;
  if (t < 0) { row = p.cls; drow = p.dlearned + tok * 128; }
  else if (t == 0) { row = p.str + i * 128; drow = p.dstr + i * 128; }
  else if (t == 1) { row = p.kg + i * 128; drow = p.dkg + i * 128; }
  else if (t == 2) { row = p.cv + i * 128; drow = p.dcv + i * 128; }
  else if (t < 3 + p.nb) { row = p.bottleneck + (t - 3) * 128; drow = p.dlearned + tok * 128; }
  else { const int64_t r = static_cast<int64_t>(t - 3 - p.nb) * p.n + i; row = p.tx + r * 128; drow = p.dtx + r * 128; }
  f32x4 g = *reinterpret_cast<const f32x4*>(p.dseq + slot * 128 + 4 * sub);
  if (p.dpe && s < p.pe_len) *reinterpret_cast<f32x4*>(p.dpe + tok * 128 + 4 * sub) = g;
  if (p.normalize) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * sub);
    float q = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    float dot = v[0] * g[0] + v[1] * g[1] + v[2] * g[2] + v[3] * g[3];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { q += __shfl_xor(q, o, 64); dot += __shfl_xor(dot, o, 64); }
    const float nrm = sqrtf(q);
    if (nrm > 1e-12f) g = (g - v * (dot / q)) * (1.0f / nrm);       // d/dx [x/|x|] = (I - y y^T)/|x|
    else g = g * 1e12f;                                              // clamped branch of F.normalize: y = x / eps
  }
  *reinterpret_cast<f32x4*>(drow + 4 * sub) = g;
}

// dx of y = x / max(|x|, 1e-12)
__global__ __launch_bounds__(256) void l2_normalize_bwd_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t ldx,
                                                               float* __restrict__ dx, int64_t lddx, int64_t rows, int d) {
  const int lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  const float* gr = dy + row * lddy;
  float q = 0.f, dot = 0.f;
  for (int c = 4 * lane; c < d; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gr + c);
    q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    dot += (v[0] * g[0] + v[1] * g[1]) + (v[2] * g[2] + v[3] * g[3]);
  }
  q = mdg_wave_sum(q);
  dot = mdg_wave_sum(dot);
  const float nrm = sqrtf(q);
  float* dr = dx + row * lddx;
  for (int c = 4 * lane; c < d; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gr + c);
    *reinterpret_cast<f32x4*>(dr + c) = nrm > 1e-12f ? (g - v * (dot / q)) * (1.0f / nrm) : g * 1e12f;
  }
}

// ---------------------------------------------------------------------------------------------
// Row-wise L2 normalisation (F.normalize, eps 1e-12): one wave per row, two passes over the row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2_normalize_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
                                                           int64_t rows, int d) {
  const int lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  float q = 0.f;
  for (int c = 4 * lane; c < d; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
    q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  const float inv = 1.0f / fmaxf(sqrtf(mdg_wave_sum(q)), 1e-12f);
  float* yr = y + row * ldy;
  for (int c = 4 * lane; c < d; c += 256) *reinterpret_cast<f32x4*>(yr + c) = *reinterpret_cast<const f32x4*>(xr + c) * inv;
}

// ---------------------------------------------------------------------------------------------
// Masked pooling over the tokens of a drug: mean / sum / max over tokens whose mask bit is clear
// (the scatter_mean / scatter_add / scatter_max uses at madrigal/models/models.py:447,451,873,878).
// One 32-lane group (float4 per lane, D = 128) per drug.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void token_pool_kernel(const float* __restrict__ e, const uint32_t* __restrict__ bits,
                                                         float* __restrict__ out, int64_t n, int S, int mode) {
  const int sub = threadIdx.x & 31;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 8 + (threadIdx.x >> 5);
  if (i >= n) return;
  const uint32_t b = bits ? bits[i] : 0u;
  f32x4 acc = (mode == 2) ? f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY} : f32x4{0.f, 0.f, 0.f, 0.f};
  int cnt = 0;
  for (int s = 0; s < S; ++s) {
    if ((b >> s) & 1u) continue;
    const f32x4 v = *reinterpret_cast<const f32x4*>(e + (i * S + s) * 128 + 4 * sub);
    if (mode == 2) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = fmaxf(acc[c], v[c]);
    } else {
      acc += v;
    }
    ++cnt;
  }
  if (mode == 0) acc = acc / static_cast<float>(cnt > 0 ? cnt : 1);
  *reinterpret_cast<f32x4*>(out + i * 128 + 4 * sub) = acc;
}

}  // namespace

extern "C" int mdg_assemble_tokens(const float* str_emb, const float* kg_emb, const float* cv_emb, const float* tx_emb,
                                   const float* bottleneck, const float* cls, const float* pe, const int64_t* rows,
                                   const int64_t* token_index, int64_t n_tok, float* seq, int64_t n, int64_t n_src, int nb,
                                   int has_cls, int pe_len, int normalize, int64_t D, void* stream) {
  MDG_CHECK_ARG(D == 128, "mdg_assemble_tokens: D must be 128 (got %lld)", (long long)D);
  MDG_CHECK_ARG(n >= 0 && n_src >= 0 && nb >= 0 && nb <= 8, "mdg_assemble_tokens: bad sizes");
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(str_emb && kg_emb && cv_emb && tx_emb && seq, "mdg_assemble_tokens: null pointer");
  MDG_CHECK_ARG((nb == 0 || bottleneck) && (!has_cls || cls) && (pe_len == 0 || pe), "mdg_assemble_tokens: missing token table");
  const int S = (has_cls ? 1 : 0) + 3 + nb + 16;
  MDG_CHECK_ARG(S <= 32 && pe_len <= S, "mdg_assemble_tokens: sequence of %d tokens exceeds 32", S);
  AssembleArgs a{str_emb, kg_emb, cv_emb, tx_emb, bottleneck, cls, pe, rows, token_index, n_tok, seq, n, n_src, nb, has_cls ? 1 : 0, pe_len, normalize};
  const int64_t slots = token_index ? n_tok : n * S;
  if (slots == 0) return MDG_OK;
  hipLaunchKernelGGL(assemble_tokens_kernel, dim3(static_cast<unsigned>(mdg_cdiv(slots, 8))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  MDG_CHECK_LAUNCH("mdg_assemble_tokens");
  return MDG_OK;
}

static int attention_check(const char* who, const float* qkv, int64_t ld, const void* out, int64_t ldo, int64_t n, int S, int H, int dh,
                           float p_drop) {
  MDG_CHECK_ARG(n >= 0 && S >= 1 && S <= 32, "%s: S must be in [1,32] (got %d)", who, S);
  MDG_CHECK_ARG(H >= 1 && dh >= 8 && dh % 32 == 0 && dh <= 1024, "%s: head_dim must be a multiple of 32 (got %d)", who, dh);
  MDG_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "%s: dropout p must be in [0,1)", who);
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(qkv && out, "%s: null pointer", who);
  MDG_CHECK_ARG(ld % 4 == 0 && ldo % 4 == 0 && ld >= 3 * H * dh && ldo >= H * dh && mdg_aligned16(qkv) && mdg_aligned16(out),
                "%s: bad strides / alignment", who);
  return MDG_OK;
}

extern "C" int mdg_fusion_attention_dropout(const float* qkv, int64_t ld, float* out, int64_t ldo, const uint32_t* kpm_bits,
                                            const uint32_t* src_bits, float* probs, const int64_t* row_start, const uint32_t* row_bits,
                                            int64_t n, int S, int H, int dh, float p_drop, uint64_t seed, void* stream) {
  if (int rc = attention_check("mdg_fusion_attention", qkv, ld, out, ldo, n, S, H, dh, p_drop)) return rc;
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(!(row_start && probs), "mdg_fusion_attention: attention weights are only produced in dense mode");
  AttnArgs a{qkv, ld, out, ldo, kpm_bits, src_bits, probs, row_start, row_bits, n, S, H, dh, 1.0f / sqrtf(static_cast<float>(dh)),
             p_drop, seed, nullptr, 0, nullptr, 0};
  hipLaunchKernelGGL(fusion_attention_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n * H, 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  MDG_CHECK_LAUNCH("mdg_fusion_attention");
  return MDG_OK;
}

extern "C" int mdg_fusion_attention(const float* qkv, int64_t ld, float* out, int64_t ldo, const uint32_t* kpm_bits,
                                    const uint32_t* src_bits, float* probs, const int64_t* row_start, const uint32_t* row_bits,
                                    int64_t n, int S, int H, int dh, void* stream) {
  return mdg_fusion_attention_dropout(qkv, ld, out, ldo, kpm_bits, src_bits, probs, row_start, row_bits, n, S, H, dh, 0.f, 0, stream);
}

extern "C" int mdg_fusion_attention_bwd(const float* qkv, int64_t ld, const float* dout, int64_t lddo, float* dqkv, int64_t lddq,
                                        const uint32_t* kpm_bits, const uint32_t* src_bits, const int64_t* row_start,
                                        const uint32_t* row_bits, int64_t n, int S, int H, int dh, float p_drop, uint64_t seed,
                                        void* stream) {
  if (int rc = attention_check("mdg_fusion_attention_bwd", qkv, ld, dout, lddo, n, S, H, dh, p_drop)) return rc;
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(dqkv && lddq % 4 == 0 && lddq >= 3 * H * dh && mdg_aligned16(dqkv) && dqkv != qkv, "mdg_fusion_attention_bwd: bad dqkv");
  AttnArgs a{qkv, ld, nullptr, 0, kpm_bits, src_bits, nullptr, row_start, row_bits, n, S, H, dh, 1.0f / sqrtf(static_cast<float>(dh)),
             p_drop, seed, dout, lddo, dqkv, lddq};
  hipLaunchKernelGGL(fusion_attention_bwd_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n * H, 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  MDG_CHECK_LAUNCH("mdg_fusion_attention_bwd");
  return MDG_OK;
}

extern "C" int mdg_xattn_pool_dropout(const float* q_proj, const float* kv_proj, int64_t ld, float* out, int64_t ldo, int64_t n, int Tk,
                                      int H, int dh, float p_drop, uint64_t seed, void* stream) {
  MDG_CHECK_ARG(n >= 0 && Tk >= 1 && Tk <= 32 && H >= 1 && dh >= 1 && dh <= 256, "mdg_xattn_pool: bad shape (Tk=%d H=%d dh=%d)", Tk, H, dh);
  MDG_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "mdg_xattn_pool: dropout p must be in [0,1)");
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(q_proj && kv_proj && out && ld >= 2 * H * dh && ldo >= H * dh, "mdg_xattn_pool: bad pointers / strides");
  hipLaunchKernelGGL(xattn_pool_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n * H, 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), q_proj, kv_proj, ld, out, ldo, n, Tk, H, dh,
                     1.0f / sqrtf(static_cast<float>(dh)), p_drop, seed);
  MDG_CHECK_LAUNCH("mdg_xattn_pool");
  return MDG_OK;
}

extern "C" int mdg_xattn_pool(const float* q_proj, const float* kv_proj, int64_t ld, float* out, int64_t ldo, int64_t n, int Tk,
                              int H, int dh, void* stream) {
  return mdg_xattn_pool_dropout(q_proj, kv_proj, ld, out, ldo, n, Tk, H, dh, 0.f, 0, stream);
}

extern "C" int mdg_xattn_pool_bwd(const float* q_proj, const float* kv_proj, int64_t ld, const float* dout, int64_t lddo, float* dkv,
                                  int64_t lddkv, float* dq_part, int64_t n, int Tk, int H, int dh, float p_drop, uint64_t seed,
                                  void* stream) {
  MDG_CHECK_ARG(n >= 0 && Tk >= 1 && Tk <= 32 && H >= 1 && dh >= 1 && dh <= 256, "mdg_xattn_pool_bwd: bad shape (Tk=%d H=%d dh=%d)", Tk, H, dh);
  MDG_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "mdg_xattn_pool_bwd: dropout p must be in [0,1)");
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(q_proj && kv_proj && dout && dkv && dq_part && ld >= 2 * H * dh && lddkv >= 2 * H * dh && lddo >= H * dh,
                "mdg_xattn_pool_bwd: bad pointers / strides");
  hipLaunchKernelGGL(xattn_pool_bwd_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n * H, 4))), dim3(256), 0, static_cast<hipStream_t>(stream),
                     q_proj, kv_proj, ld, dout, lddo, dkv, lddkv, dq_part, n, Tk, H, dh, 1.0f / sqrtf(static_cast<float>(dh)), p_drop, seed);
  MDG_CHECK_LAUNCH("mdg_xattn_pool_bwd");
  return MDG_OK;
}

extern "C" int mdg_assemble_tokens_bwd(const float* dseq, const float* str_emb, const float* kg_emb, const float* cv_emb, const float* tx_emb,
                                       const float* bottleneck, const float* cls, const int64_t* token_index, int64_t n_tok, float* dstr,
                                       float* dkg, float* dcv, float* dtx, float* dlearned, float* dpe, int64_t n, int nb, int has_cls,
                                       int pe_len, int normalize, int64_t D, void* stream) {
  MDG_CHECK_ARG(D == 128, "mdg_assemble_tokens_bwd: D must be 128 (got %lld)", (long long)D);
  MDG_CHECK_ARG(n >= 0 && nb >= 0 && nb <= 8, "mdg_assemble_tokens_bwd: bad sizes");
  if (n == 0) return MDG_OK;
  const int S = (has_cls ? 1 : 0) + 3 + nb + 16;
  MDG_CHECK_ARG(S <= 32 && pe_len <= S, "mdg_assemble_tokens_bwd: sequence of %d tokens exceeds 32", S);
  MDG_CHECK_ARG(dseq && dstr && dkg && dcv && dtx, "mdg_assemble_tokens_bwd: null pointer");
  MDG_CHECK_ARG(!normalize || (str_emb && kg_emb && cv_emb && tx_emb && (nb == 0 || bottleneck) && (!has_cls || cls)),
                "mdg_assemble_tokens_bwd: the forward sources are needed when tokens were normalised");
  MDG_CHECK_ARG(((nb == 0 && !has_cls) || dlearned) && (pe_len == 0 || dpe), "mdg_assemble_tokens_bwd: missing scratch for shared tokens");
  AssembleBwdArgs a{dseq, str_emb, kg_emb, cv_emb, tx_emb, bottleneck, cls, token_index, n_tok, dstr, dkg, dcv, dtx, dlearned, dpe, n, nb,
                    has_cls ? 1 : 0, pe_len, normalize};
  const int64_t slots = token_index ? n_tok : n * S;
  if (slots == 0) return MDG_OK;
  hipLaunchKernelGGL(assemble_tokens_bwd_kernel, dim3(static_cast<unsigned>(mdg_cdiv(slots, 8))), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  MDG_CHECK_LAUNCH("mdg_assemble_tokens_bwd");
  return MDG_OK;
}

extern "C" int mdg_l2_normalize_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dx, int64_t lddx, int64_t rows,
                                    int64_t d, void* stream) {
  MDG_CHECK_ARG(rows >= 0 && d > 0 && d % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && ldx >= d && lddy >= d && lddx >= d,
                "mdg_l2_normalize_bwd: bad shape");
  if (rows == 0) return MDG_OK;
  MDG_CHECK_ARG(dy && x && dx && mdg_aligned16(dy) && mdg_aligned16(x) && mdg_aligned16(dx), "mdg_l2_normalize_bwd: bad pointers");
  hipLaunchKernelGGL(l2_normalize_bwd_kernel, dim3(static_cast<unsigned>(mdg_cdiv(rows, 4))), dim3(256), 0, static_cast<hipStream_t>(stream),
                     dy, lddy, x, ldx, dx, lddx, rows, static_cast<int>(d));
  MDG_CHECK_LAUNCH("mdg_l2_normalize_bwd");
  return MDG_OK;
}

extern "C" int mdg_l2_normalize(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int64_t d, void* stream) {
  MDG_CHECK_ARG(rows >= 0 && d > 0 && d % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= d && ldy >= d, "mdg_l2_normalize: bad shape");
  if (rows == 0) return MDG_OK;
  MDG_CHECK_ARG(x && y && mdg_aligned16(x) && mdg_aligned16(y), "mdg_l2_normalize: bad pointers");
  hipLaunchKernelGGL(l2_normalize_kernel, dim3(static_cast<unsigned>(mdg_cdiv(rows, 4))), dim3(256), 0, static_cast<hipStream_t>(stream),
                     x, ldx, y, ldy, rows, static_cast<int>(d));
  MDG_CHECK_LAUNCH("mdg_l2_normalize");
  return MDG_OK;
}

extern "C" int mdg_token_pool(const float* tokens, const uint32_t* mask_bits, float* out, int64_t n, int S, int64_t D, int mode,
                              void* stream) {
  MDG_CHECK_ARG(D == 128 && S >= 1 && S <= 32 && n >= 0 && mode >= 0 && mode <= 2, "mdg_token_pool: bad arguments (D=%lld S=%d mode=%d)", (long long)D, S, mode);
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(tokens && out && mdg_aligned16(tokens) && mdg_aligned16(out), "mdg_token_pool: bad pointers");
  hipLaunchKernelGGL(token_pool_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n, 8))), dim3(256), 0, static_cast<hipStream_t>(stream),
                     tokens, mask_bits, out, n, S, mode);
  MDG_CHECK_LAUNCH("mdg_token_pool");
  return MDG_OK;
}
