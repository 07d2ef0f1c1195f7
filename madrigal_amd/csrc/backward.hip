// Backward-pass building blocks for gfx950 (finetune step of train_ddi_batch.py:231-354).
// All streaming / HBM-bound except where noted; reductions are atomic-free with a fixed summation order.
#include "mdg_common.h"

namespace {

// ---- out[c, r] = in[r, c]: 64x64 tiles through LDS (padded), coalesced on both sides ----------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int64_t ldi, float* __restrict__ out,
                                                        int64_t ldo, int64_t rows, int64_t cols) {
  __shared__ float tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * 64, c0 = static_cast<int64_t>(blockIdx.x) * 64;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int64_t r = r0 + ty + 4 * i, c = c0 + tx;
    tile[ty + 4 * i][tx] = (r < rows && c < cols) ? in[r * ldi + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int64_t c = c0 + ty + 4 * i, r = r0 + tx;
    if (c < cols && r < rows) out[c * ldo + r] = tile[tx][ty + 4 * i];
  }
}

// ---- column sums: out[c] (+)= sum_r x[r, c].  Two passes: per-block partial sums over 256-row slabs, then a fixed
// order reduction of the partials.  (bias gradients, LayerNorm gamma/beta gradients, learned-token gradients)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ part,
                                                             int64_t rows, int64_t cols, int64_t slab) {
  const int64_t c = static_cast<int64_t>(blockIdx.x) * 64 + (threadIdx.x & 63);
  const int ty = threadIdx.x >> 6;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * slab;
  const int64_t r1 = r0 + slab < rows ? r0 + slab : rows;
  float s = 0.f;
  if (c < cols)
    for (int64_t r = r0 + ty; r < r1; r += 4) s += x[r * ldx + c];
  __shared__ float sh[4][64];
  sh[ty][threadIdx.x & 63] = s;
  __syncthreads();
  if (ty == 0 && c < cols) part[static_cast<int64_t>(blockIdx.y) * cols + c] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// BatchNorm backward: modes 0 (sum dy -> dbeta) and 2 (sum dy xhat -> dgamma) of the kernel above in ONE pass over dy (same rows per
// thread, same order of additions: the same bits as the two launches).  part0 / part2: [parts][cols] each.
__global__ __launch_bounds__(256) void colreduce_bn_bwd_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t ldx,
                                                               const float* __restrict__ center, const float* __restrict__ rstd, float* __restrict__ part0,
                                                               float* __restrict__ part2, int64_t rows, int64_t cols) {
  const int64_t c = static_cast<int64_t>(blockIdx.x) * 64 + (threadIdx.x & 63);
  const int ty = threadIdx.x >> 6;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * 256;
  float s0 = 0.f, s2 = 0.f;
  if (c < cols) {
    const float mu = center[c], rs = rstd[c];
    for (int i = 0; i < 64; ++i) {
      const int64_t r = r0 + ty + 4 * i;
      if (r >= rows) break;
      const float v = dy[r * lddy + c];
      s0 += v;
      s2 += v * ((x[r * ldx + c] - mu) * rs);
    }
  }
  __shared__ float sh[2][4][64];
  sh[0][ty][threadIdx.x & 63] = s0;
  sh[1][ty][threadIdx.x & 63] = s2;
  __syncthreads();
  if (ty == 0 && c < cols) {
    const int e = threadIdx.x;
    part0[static_cast<int64_t>(blockIdx.y) * cols + c] = (sh[0][0][e] + sh[0][1][e]) + (sh[0][2][e] + sh[0][3][e]);
    part2[static_cast<int64_t>(blockIdx.y) * cols + c] = (sh[1][0][e] + sh[1][1][e]) + (sh[1][2][e] + sh[1][3][e]);
  }
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int64_t ldp, float* __restrict__ out,
                                                           int64_t nparts, int64_t cols, float beta) {
  // 16 columns per block, the partials interleaved over sixteen thread rows (a launch of cols / 64 blocks with four rows walked a few
  // hundred partials per thread on 32-96 workgroups: 19 us of latency chain, 58 times per finetune step), combined through LDS in a
  // fixed order
  const int e = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int64_t c = static_cast<int64_t>(blockIdx.x) * 16 + e;
  float s0 = 0.f, s1 = 0.f;
  if (c < cols) {
    int64_t p = grp;
    for (; p + 16 < nparts; p += 32) {
      s0 += part[p * ldp + c];
      s1 += part[(p + 16) * ldp + c];
    }
    for (; p < nparts; p += 16) s0 += part[p * ldp + c];
  }
  __shared__ float sh[16][17];
  sh[grp][e] = s0 + s1;
  __syncthreads();
  if (grp == 0 && c < cols) {
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = sh[2 * k][e] + sh[2 * k + 1][e];
    const float tot = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    out[c] = (beta != 0.f ? beta * out[c] : 0.f) + tot;
  }
}

// ---- dx = dy * act'(pre) -------------------------------------------------------------------------------------
__device__ __forceinline__ float act_grad(float x, int act) {
  switch (act) {
    case MDG_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case MDG_ACT_GELU: {   // d/dx [x Phi(x)] = Phi(x) + x phi(x)
      const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
      return cdf + x * 0.39894228040143268f * expf(-0.5f * x * x);
    }
    case MDG_ACT_SIGMOID: { const float s = 1.0f / (1.0f + expf(-x)); return s * (1.f - s); }
    case MDG_ACT_TANH: { const float t = tanhf(x); return 1.f - t * t; }
    case MDG_ACT_LEAKYRELU: return x >= 0.f ? 1.f : 0.01f;
    case MDG_ACT_SOFTPLUS: return 1.0f / (1.0f + expf(-x));
    case MDG_ACT_SELU: return 1.0507009873554804934193349852946f * (x > 0.f ? 1.f : 1.6732632423543772848170429916717f * expf(x));
    default: return 1.f;
  }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ dx,
                                                      int64_t n, int act) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) dx[i] = dy[i] * act_grad(pre[i], act);
}

// ---- y = act(pre) (training keeps the pre-activation for the backward pass) ------------------------------------
__device__ __forceinline__ float act_fwd(float v, int act) {
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

__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ pre, float* __restrict__ y, int64_t n, int act) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) y[i] = act_fwd(pre[i], act);
}

// ---- out = alpha * a + beta * b[i mod nb]  (residual adds; nb < n broadcasts a row vector over rows) ------------------
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n,
                                                    int64_t nb, float alpha, float beta) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) out[i] = alpha * a[i] + beta * b[nb == n ? i : i % nb];
}

// ---- out = x * s[0], s on the device (chain-rule scaling by the incoming scalar gradient, no host sync) -------------
__global__ __launch_bounds__(256) void mul_dev_scalar_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ out, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) out[i] = x[i] * s[0];
}

// ---- HGT skip gate: out = s * o + (1 - s) * x,  s = sigmoid(skip[0]) read on the device (no host sync per step) -------
__global__ __launch_bounds__(256) void gated_residual_kernel(const float* __restrict__ o, const float* __restrict__ x, const float* __restrict__ skip,
                                                             float* __restrict__ out, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const float s = 1.0f / (1.0f + expf(-skip[0]));
  out[i] = s * o[i] + (1.0f - s) * x[i];
}

// do = s dout, dx = (1-s) dout, rowdot[v] = s (1-s) sum_c dout[v,c] (o[v,c] - x[v,c])   (d skip = sum_v rowdot[v])
__global__ __launch_bounds__(256) void gated_residual_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ o, const float* __restrict__ x,
                                                                 const float* __restrict__ skip, float* __restrict__ d_o, float* __restrict__ d_x,
                                                                 float* __restrict__ rowdot, int64_t rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int64_t v = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (v >= rows) return;
  const float s = 1.0f / (1.0f + expf(-skip[0]));
  float acc = 0.f;
  for (int c = lane; c < cols; c += 64) {
    const int64_t i = v * cols + c;
    const float g = dout[i];
    d_o[i] = s * g;
    d_x[i] = (1.0f - s) * g;
    acc += g * (o[i] - x[i]);
  }
  acc = mdg_wave_sum(acc);
  if (lane == 0) rowdot[v] = s * (1.0f - s) * acc;
}

// ---- dropout: y = x * keep / (1 - p), keep ~ Bernoulli(1-p) from a counter-based hash of (seed, element index);
// the same (seed, index) reproduces the mask in the backward pass, so no mask tensor is stored. -------------------
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float p,
                                                      uint64_t seed) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const bool keep = mdg_keep(seed, static_cast<uint64_t>(i), mdg_drop_threshold(p));
  y[i] = keep ? x[i] * (1.0f / (1.0f - p)) : 0.f;
}

// Elementwise passes of the training step over 16-byte aligned tensors, four elements per thread (n % 4 == 0).  MODE 0: y = dropout(x);
// 1: y = act(pre); 2: dx = dy act'(pre); 3: y = dropout(act(pre)); 4: dx = act'(pre) dropout-backward(dy).  Same expressions per element as
// the scalar kernels (dropout_kernel, act_fwd_kernel, act_bwd_kernel) and as their compositions: a fused pass equals the two separate ones.
template <int MODE>
__global__ __launch_bounds__(256) void elementwise4_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n4,
                                                           int act, float p, uint64_t seed) {
  const int64_t q = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (q >= n4) return;
  const f32x4 va = reinterpret_cast<const f32x4*>(a)[q];
  f32x4 vb = {0.f, 0.f, 0.f, 0.f};
  if constexpr (MODE == 2 || MODE == 4) vb = reinterpret_cast<const f32x4*>(b)[q];
  const uint32_t thr = mdg_drop_threshold(p);
  const float scale = 1.0f / (1.0f - p);
  f32x4 o;
  uint64_t word = 0;
  if constexpr (MODE == 0 || MODE == 3 || MODE == 4) word = mdg_keep_word(seed, static_cast<uint64_t>(q));      // elements 4q .. 4q+3: one hash
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if constexpr (MODE == 0) o[e] = mdg_keep_field(word, e, thr) ? va[e] * scale : 0.f;
    else if constexpr (MODE == 1) o[e] = act_fwd(va[e], act);
    else if constexpr (MODE == 2) o[e] = va[e] * act_grad(vb[e], act);
    else if constexpr (MODE == 3) { const float y = act_fwd(va[e], act); o[e] = mdg_keep_field(word, e, thr) ? y * scale : 0.f; }
    else { const float g = mdg_keep_field(word, e, thr) ? va[e] * scale : 0.f; o[e] = g * act_grad(vb[e], act); }
  }
  reinterpret_cast<f32x4*>(out)[q] = o;
}

__global__ __launch_bounds__(256) void act_dropout_fwd_kernel(const float* __restrict__ pre, float* __restrict__ y, int64_t n, int act, float p, uint64_t seed) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = act_fwd(pre[i], act);
  y[i] = mdg_keep(seed, static_cast<uint64_t>(i), mdg_drop_threshold(p)) ? v * (1.0f / (1.0f - p)) : 0.f;
}

__global__ __launch_bounds__(256) void act_dropout_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ dx, int64_t n,
                                                              int act, float p, uint64_t seed) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const float g = mdg_keep(seed, static_cast<uint64_t>(i), mdg_drop_threshold(p)) ? dy[i] * (1.0f / (1.0f - p)) : 0.f;
  dx[i] = g * act_grad(pre[i], act);
}

__global__ __launch_bounds__(256) void axpby4_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n4, float alpha,
                                                     float beta) {
  const int64_t q = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (q >= n4) return;
  const f32x4 va = reinterpret_cast<const f32x4*>(a)[q], vb = reinterpret_cast<const f32x4*>(b)[q];
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = alpha * va[e] + beta * vb[e];
  reinterpret_cast<f32x4*>(out)[q] = o;
}

static inline bool vec4_ok(int64_t n, const void* a, const void* b, const void* c) {
  return n % 4 == 0 && mdg_aligned16(a) && (!b || mdg_aligned16(b)) && mdg_aligned16(c);
}


// ---- generalised column reductions for BatchNorm1d in training mode (torchdrug MLP / chemCPA MLP batch_norm) ------
//   mode 0: sum_r x                      mode 1: sum_r (x - center[c])^2
//   mode 2: sum_r x * (y - center[c]) * rstd[c]      (x = dy, y = BN input: the dgamma reduction)
__global__ __launch_bounds__(256) void colreduce_partial_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ y,
                                                                int64_t ldy, const float* __restrict__ center,
                                                                const float* __restrict__ rstd, float* __restrict__ part,
                                                                int64_t rows, int64_t cols, int mode) {
  const int64_t c = static_cast<int64_t>(blockIdx.x) * 64 + (threadIdx.x & 63);
  const int ty = threadIdx.x >> 6;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * 256;
  float s = 0.f;
  if (c < cols) {
    const float mu = center ? center[c] : 0.f, rs = rstd ? rstd[c] : 1.f;
    for (int i = 0; i < 64; ++i) {
      const int64_t r = r0 + ty + 4 * i;
      if (r >= rows) break;
      const float v = x[r * ldx + c];
      if (mode == 0) s += v;
      else if (mode == 1) s += (v - mu) * (v - mu);
      else s += v * ((y[r * ldy + c] - mu) * rs);
    }
  }
  __shared__ float sh[4][64];
  sh[ty][threadIdx.x & 63] = s;
  __syncthreads();
  if (ty == 0 && c < cols) part[static_cast<int64_t>(blockIdx.y) * cols + c] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// stats[0:N]=mean, [N:2N]=rstd, [2N:3N]=scale=gamma*rstd, [3N:4N]=shift=beta-mean*scale, [4N:5N]=biased batch variance (kept
// for the replayed running-statistics update: 1/rstd^2 - eps cancels catastrophically where var << eps); running stats updated
// like nn.BatchNorm1d (momentum, unbiased variance).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ sum, const float* __restrict__ sqsum, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float* __restrict__ stats, double rows,
                                                          int64_t cols, float eps, float momentum, int phase,
                                                          const double* __restrict__ rows_dev = nullptr) {
  const int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (c >= cols) return;
  if (rows_dev) rows = rows_dev[0];
  if (phase == 0) { stats[c] = sum[c] / static_cast<float>(rows); return; }
  const float mean = stats[c];
  const float var = sqsum[c] / static_cast<float>(rows);
  const float rstd = 1.0f / sqrtf(var + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  stats[cols + c] = rstd;
  stats[2 * cols + c] = g * rstd;
  stats[3 * cols + c] = b - mean * g * rstd;
  stats[4 * cols + c] = var;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (rows > 1 ? sqsum[c] / static_cast<float>(rows - 1) : var);
}

// A second momentum update of the running statistics with batch statistics that a training-mode forward already produced
// (stats[0:C] = mean, stats[4C:5C] = biased variance): what nn.BatchNorm1d does when the same module sees the same rows again.
__global__ __launch_bounds__(256) void bn_replay_kernel(const float* __restrict__ stats, float* __restrict__ running_mean,
                                                        float* __restrict__ running_var, double rows, int64_t cols, float eps, float momentum) {
  const int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (c >= cols) return;
  const float mean = stats[c];
  const float var = fmaxf(stats[4 * cols + c], 0.f);                              // biased batch variance, as the forward pass formed it
  const float unbiased = rows > 1 ? var * static_cast<float>(rows / (rows - 1.0)) : var;
  running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
  running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
}

// y = act(x * scale[c] + shift[c])
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ y, int64_t ldy, int64_t rows,
                                                         int64_t cols, int act) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= rows * cols) return;
  const int64_t r = i / cols, c = i - r * cols;
  y[r * ldy + c] = act_fwd(x[r * ldx + c] * scale[c] + (shift ? shift[c] : 0.f), act);
}

// dx = scale[c] * (dy - sum_dy[c]/M - xhat * sum_dy_xhat[c]/M),  xhat = (x - mean[c]) * rstd[c]
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ stats,
                                                           const float* __restrict__ sum_dy, const float* __restrict__ sum_dy_xhat,
                                                           float* __restrict__ dx, int64_t rows, int64_t cols, float count,
                                                           const double* __restrict__ count_dev = nullptr) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= rows * cols) return;
  const int64_t c = i % cols;
  const float inv = 1.0f / (count_dev ? static_cast<float>(count_dev[0]) : count);
  const float xhat = (x[i] - stats[c]) * stats[cols + c];
  dx[i] = stats[2 * cols + c] * (dy[i] - sum_dy[c] * inv - xhat * (sum_dy_xhat[c] * inv));
}

// ---- LayerNorm backward: one wave per row, 64 rows per block; per-block partial dgamma / dbeta -------------------
constexpr int LN_DMAX = 2048;    // 64 lanes x LN_MAXJ columns; instantiated for d <= 128 / 512 / 2048
template <int LN_MAXJ>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ gamma, float* __restrict__ dx, int64_t lddx,
                                                            float* __restrict__ part, int64_t rows, int d, float eps,
                                                            const float* __restrict__ extra, int64_t ldextra) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nj = (d + 63) >> 6;
  float dg[LN_MAXJ], db[LN_MAXJ];
#pragma unroll
  for (int j = 0; j < LN_MAXJ; ++j) dg[j] = db[j] = 0.f;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * 64;
  for (int i = wave; i < 64; i += 4) {
    const int64_t r = r0 + i;
    if (r >= rows) break;
    float xv[LN_MAXJ], gv[LN_MAXJ];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
      const int c = lane + 64 * j;
      const bool ok = j < nj && c < d;
      xv[j] = ok ? x[r * ldx + c] : 0.f;
      gv[j] = ok ? dy[r * lddy + c] : 0.f;
      s += xv[j];
    }
    const float mean = mdg_wave_sum(s) / static_cast<float>(d);
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
      const int c = lane + 64 * j;
      const float t = (j < nj && c < d) ? xv[j] - mean : 0.f;
      v += t * t;
    }
    const float rstd = 1.0f / sqrtf(mdg_wave_sum(v) / static_cast<float>(d) + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
      const int c = lane + 64 * j;
      if (j < nj && c < d) {
        const float xhat = (xv[j] - mean) * rstd;
        const float dxh = gv[j] * gamma[c];
        dg[j] += gv[j] * xhat;
        db[j] += gv[j];
        s1 += dxh;
        s2 += dxh * xhat;
        xv[j] = xhat;
        gv[j] = dxh;
      }
    }
    s1 = mdg_wave_sum(s1) / static_cast<float>(d);
    s2 = mdg_wave_sum(s2) / static_cast<float>(d);
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
      const int c = lane + 64 * j;
      if (j < nj && c < d) dx[r * lddx + c] = rstd * (gv[j] - s1 - xv[j] * s2) + (extra ? extra[r * ldextra + c] : 0.f);
    }
  }
  // block partials: part[block][0][c] = dgamma, part[block][1][c] = dbeta
  extern __shared__ float sh[];       // [4][2][d]
#pragma unroll
  for (int j = 0; j < LN_MAXJ; ++j) {
    const int c = lane + 64 * j;
    if (j < nj && c < d) {
      sh[(wave * 2 + 0) * d + c] = dg[j];
      sh[(wave * 2 + 1) * d + c] = db[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * d; i += 256)
    part[static_cast<int64_t>(blockIdx.x) * 2 * d + i] = (sh[i] + sh[2 * d + i]) + (sh[4 * d + i] + sh[6 * d + i]);
}

// The same with 16 bytes per lane (d, the row strides and the pointers multiples of 4 floats): lane owns columns
// 4*(lane + 64 j) .. +3.  32 rows per block: 22k token rows give ~700 workgroups instead of ~340.
constexpr int LN_VEC_ROWS = 32;
template <int MAXJ>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t ldx,
                                                                const float* __restrict__ gamma, float* __restrict__ dx, int64_t lddx,
                                                                float* __restrict__ part, int64_t rows, int d, float eps,
                                                                const float* __restrict__ extra, int64_t ldextra) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 dg[MAXJ], db[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) dg[j] = db[j] = zero;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * LN_VEC_ROWS;
  const float inv_d = 1.0f / static_cast<float>(d);
  for (int i = wave; i < LN_VEC_ROWS; i += 4) {
    const int64_t r = r0 + i;
    if (r >= rows) break;
    f32x4 xv[MAXJ], gv[MAXJ], ev[MAXJ];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = 4 * (lane + 64 * j);
      const bool ok = c < d;
      xv[j] = ok ? *reinterpret_cast<const f32x4*>(x + r * ldx + c) : zero;
      gv[j] = ok ? *reinterpret_cast<const f32x4*>(dy + r * lddy + c) : zero;
      ev[j] = (ok && extra) ? *reinterpret_cast<const f32x4*>(extra + r * ldextra + c) : zero;      // (travels with the other two rows)
      s += (xv[j][0] + xv[j][1]) + (xv[j][2] + xv[j][3]);
    }
    const float mean = mdg_wave_sum(s) * inv_d;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j)
      if (4 * (lane + 64 * j) < d) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = xv[j][e] - mean;
          v += t * t;
        }
      }
    const float rstd = 1.0f / sqrtf(mdg_wave_sum(v) * inv_d + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = 4 * (lane + 64 * j);
      if (c < d) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xhat = (xv[j][e] - mean) * rstd;
          const float dxh = gv[j][e] * gm[e];
          dg[j][e] += gv[j][e] * xhat;
          db[j][e] += gv[j][e];
          s1 += dxh;
          s2 += dxh * xhat;
          xv[j][e] = xhat;
          gv[j][e] = dxh;
        }
      }
    }
    s1 = mdg_wave_sum(s1) * inv_d;
    s2 = mdg_wave_sum(s2) * inv_d;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int c = 4 * (lane + 64 * j);
      if (c < d) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rstd * (gv[j][e] - s1 - xv[j][e] * s2);
        if (extra) o += ev[j];                              // the gradient of the other consumer of x
        *reinterpret_cast<f32x4*>(dx + r * lddx + c) = o;
      }
    }
  }
  extern __shared__ float sh[];       // [4][2][d]
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int c = 4 * (lane + 64 * j);
    if (c < d) {
      *reinterpret_cast<f32x4*>(sh + (wave * 2 + 0) * d + c) = dg[j];
      *reinterpret_cast<f32x4*>(sh + (wave * 2 + 1) * d + c) = db[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * d; i += 256)
    part[static_cast<int64_t>(blockIdx.x) * 2 * d + i] = (sh[i] + sh[2 * d + i]) + (sh[4 * d + i] + sh[6 * d + i]);
}

}  // namespace

extern "C" int mdg_transpose(const float* in, int64_t ldi, float* out, int64_t ldo, int64_t rows, int64_t cols, void* stream) {
  MDG_CHECK_ARG(rows >= 0 && cols >= 0 && ldi >= cols && ldo >= rows, "mdg_transpose: bad shape");
  if (rows == 0 || cols == 0) return MDG_OK;
  MDG_CHECK_ARG(in && out && in != out, "mdg_transpose: null / aliased pointers");
  hipLaunchKernelGGL(transpose_kernel, dim3(static_cast<unsigned>(mdg_cdiv(cols, 64)), static_cast<unsigned>(mdg_cdiv(rows, 64))),
                     dim3(256), 0, static_cast<hipStream_t>(stream), in, ldi, out, ldo, rows, cols);
  MDG_CHECK_LAUNCH("mdg_transpose");
  return MDG_OK;
}

// rows per partial block: 256, grown so that no column needs more than 128 partials
// rows per partial: enough slabs that (column blocks x slabs) is ~2048 workgroups (a narrow, tall matrix -- a bias gradient
// over 10^5..10^6 rows -- used to get <= 128 workgroups), at least 64 rows each
static int64_t colsum_slab(int64_t rows, int64_t cols) {
  int64_t parts = mdg_cdiv(2048, mdg_cdiv(cols, 64));
  const int64_t most = mdg_cdiv(rows, 64);
  if (parts > most) parts = most;
  if (parts < 1) parts = 1;
  const int64_t slab = (mdg_cdiv(rows, parts) + 3) & ~static_cast<int64_t>(3);
  return slab < 64 ? 64 : slab;      // (rows == 0 included)
}

extern "C" size_t mdg_colsum_workspace_bytes(int64_t rows, int64_t cols) {
  return rows <= 0 || cols <= 0 ? 0 : static_cast<size_t>(mdg_cdiv(rows, colsum_slab(rows, cols))) * cols * sizeof(float);
}

extern "C" int mdg_colsum(const float* x, int64_t ldx, float* out, int64_t rows, int64_t cols, float beta, void* workspace,
                          size_t workspace_bytes, void* stream) {
  MDG_CHECK_ARG(rows >= 0 && cols > 0 && ldx >= cols, "mdg_colsum: bad shape");
  MDG_CHECK_ARG(out, "mdg_colsum: null out");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t slab = colsum_slab(rows, cols);
  const int64_t nparts = mdg_cdiv(rows, slab);
  const size_t need = mdg_colsum_workspace_bytes(rows, cols);
  if (need && (!workspace || workspace_bytes < need)) {
    mdg_set_error("mdg_colsum: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return MDG_EWORKSPACE;
  }
  if (rows > 0) {
    MDG_CHECK_ARG(x, "mdg_colsum: null x");
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(static_cast<unsigned>(mdg_cdiv(cols, 64)), static_cast<unsigned>(nparts)), dim3(256), 0, st,
                       x, ldx, static_cast<float*>(workspace), rows, cols, slab);
  }
  hipLaunchKernelGGL(colsum_final_kernel, dim3(static_cast<unsigned>(mdg_cdiv(cols, 16))), dim3(256), 0, st,
                     static_cast<const float*>(workspace), cols, out, nparts, cols, beta);
  MDG_CHECK_LAUNCH("mdg_colsum");
  return MDG_OK;
}

extern "C" int mdg_activation_fwd(const float* pre, float* y, int64_t n, int activation, void* stream) {
  MDG_CHECK_ARG(n >= 0 && activation >= MDG_ACT_NONE && activation <= MDG_ACT_SELU, "mdg_activation_fwd: bad arguments");
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(pre && y, "mdg_activation_fwd: null pointer");
  if (vec4_ok(n, pre, nullptr, y))
    hipLaunchKernelGGL(elementwise4_kernel<1>, dim3(static_cast<unsigned>(mdg_cdiv(n / 4, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), pre, nullptr, y, n / 4,
                       activation, 0.f, 0);
  else
    hipLaunchKernelGGL(act_fwd_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), pre, y, n, activation);
  MDG_CHECK_LAUNCH("mdg_activation_fwd");
  return MDG_OK;
}

extern "C" int mdg_activation_bwd(const float* dy, const float* pre, float* dx, int64_t n, int activation, void* stream) {
  MDG_CHECK_ARG(n >= 0 && activation >= MDG_ACT_NONE && activation <= MDG_ACT_SELU, "mdg_activation_bwd: bad arguments");
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(dy && pre && dx, "mdg_activation_bwd: null pointer");
  if (vec4_ok(n, dy, pre, dx))
    hipLaunchKernelGGL(elementwise4_kernel<2>, dim3(static_cast<unsigned>(mdg_cdiv(n / 4, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), dy, pre, dx, n / 4,
                       activation, 0.f, 0);
  else
    hipLaunchKernelGGL(act_bwd_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), dy, pre, dx, n, activation);
  MDG_CHECK_LAUNCH("mdg_activation_bwd");
  return MDG_OK;
}

extern "C" int mdg_activation_dropout_fwd(const float* pre, float* y, int64_t n, int activation, float p, uint64_t seed, void* stream) {
  MDG_CHECK_ARG(n >= 0 && activation >= MDG_ACT_NONE && activation <= MDG_ACT_SELU && p >= 0.f && p < 1.f, "mdg_activation_dropout_fwd: bad arguments");
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(pre && y, "mdg_activation_dropout_fwd: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (vec4_ok(n, pre, nullptr, y))
    hipLaunchKernelGGL(elementwise4_kernel<3>, dim3(static_cast<unsigned>(mdg_cdiv(n / 4, 256))), dim3(256), 0, st, pre, nullptr, y, n / 4, activation, p, seed);
  else
    hipLaunchKernelGGL(act_dropout_fwd_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n, 256))), dim3(256), 0, st, pre, y, n, activation, p, seed);
  MDG_CHECK_LAUNCH("mdg_activation_dropout_fwd");
  return MDG_OK;
}

extern "C" int mdg_activation_dropout_bwd(const float* dy, const float* pre, float* dx, int64_t n, int activation, float p, uint64_t seed, void* stream) {
  MDG_CHECK_ARG(n >= 0 && activation >= MDG_ACT_NONE && activation <= MDG_ACT_SELU && p >= 0.f && p < 1.f, "mdg_activation_dropout_bwd: bad arguments");
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(dy && pre && dx, "mdg_activation_dropout_bwd: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (vec4_ok(n, dy, pre, dx))
    hipLaunchKernelGGL(elementwise4_kernel<4>, dim3(static_cast<unsigned>(mdg_cdiv(n / 4, 256))), dim3(256), 0, st, dy, pre, dx, n / 4, activation, p, seed);
  else
    hipLaunchKernelGGL(act_dropout_bwd_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n, 256))), dim3(256), 0, st, dy, pre, dx, n, activation, p, seed);
  MDG_CHECK_LAUNCH("mdg_activation_dropout_bwd");
  return MDG_OK;
}

extern "C" int mdg_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream) {
  MDG_CHECK_ARG(n >= 0 && p >= 0.f && p < 1.f, "mdg_dropout: p must be in [0,1)");
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(x && y, "mdg_dropout: null pointer");
  if (vec4_ok(n, x, nullptr, y))
    hipLaunchKernelGGL(elementwise4_kernel<0>, dim3(static_cast<unsigned>(mdg_cdiv(n / 4, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), x, nullptr, y, n / 4, 0, p,
                       seed);
  else
    hipLaunchKernelGGL(dropout_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n, p, seed);
  MDG_CHECK_LAUNCH("mdg_dropout");
  return MDG_OK;
}

// ------------------------------------------------------------------------------------------------- BatchNorm1d (train)
extern "C" size_t mdg_batchnorm_workspace_bytes(int64_t rows, int64_t cols) {
  return rows <= 0 || cols <= 0 ? 0 : (2 * static_cast<size_t>(mdg_cdiv(rows, 256)) + 2) * cols * sizeof(float);
}

static int colreduce(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* center, const float* rstd, float* out,
                     int64_t rows, int64_t cols, int mode, float* ws, hipStream_t st) {
  const int64_t nparts = mdg_cdiv(rows, 256);
  hipLaunchKernelGGL(colreduce_partial_kernel, dim3(static_cast<unsigned>(mdg_cdiv(cols, 64)), static_cast<unsigned>(nparts)), dim3(256), 0, st,
                     x, ldx, y, ldy, center, rstd, ws, rows, cols, mode);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(static_cast<unsigned>(mdg_cdiv(cols, 16))), dim3(256), 0, st, static_cast<const float*>(ws), cols, out,
                     nparts, cols, 0.f);
  return MDG_OK;
}

extern "C" int mdg_batchnorm_train_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* running_mean,
                                       float* running_var, float* y, int64_t ldy, float* stats, int64_t rows, int64_t cols, float eps,
                                       float momentum, int activation, void* workspace, size_t workspace_bytes, void* stream) {
  MDG_CHECK_ARG(rows > 1 && cols > 0 && ldx >= cols && ldy >= cols, "mdg_batchnorm_train_fwd: needs more than one row (nn.BatchNorm1d raises too)");
  MDG_CHECK_ARG(x && y && stats, "mdg_batchnorm_train_fwd: null pointer");
  MDG_CHECK_ARG(activation >= MDG_ACT_NONE && activation <= MDG_ACT_SELU, "mdg_batchnorm_train_fwd: unknown activation");
  const size_t need = mdg_batchnorm_workspace_bytes(rows, cols);
  if (!workspace || workspace_bytes < need) {
    mdg_set_error("mdg_batchnorm_train_fwd: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return MDG_EWORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* sum = static_cast<float*>(workspace);
  float* sq = sum + cols;
  float* part = sq + cols;
  const unsigned gb = static_cast<unsigned>(mdg_cdiv(cols, 256));
  colreduce(x, ldx, nullptr, 0, nullptr, nullptr, sum, rows, cols, 0, part, st);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(gb), dim3(256), 0, st, sum, sq, gamma, beta, running_mean, running_var, stats, static_cast<double>(rows), cols, eps, momentum, 0);
  colreduce(x, ldx, nullptr, 0, stats, nullptr, sq, rows, cols, 1, part, st);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(gb), dim3(256), 0, st, sum, sq, gamma, beta, running_mean, running_var, stats, static_cast<double>(rows), cols, eps, momentum, 1);
  hipLaunchKernelGGL(affine_act_kernel, dim3(static_cast<unsigned>(mdg_cdiv(rows * cols, 256))), dim3(256), 0, st, x, ldx, stats + 2 * cols,
                     stats + 3 * cols, y, ldy, rows, cols, activation);
  MDG_CHECK_LAUNCH("mdg_batchnorm_train_fwd");
  return MDG_OK;
}

extern "C" int mdg_batchnorm_replay_update(const float* stats, float* running_mean, float* running_var, int64_t rows, int64_t cols, float eps,
                                           float momentum, void* stream) {
  MDG_CHECK_ARG(stats && running_mean && running_var && rows > 0 && cols > 0, "mdg_batchnorm_replay_update: bad argument");
  hipLaunchKernelGGL(bn_replay_kernel, dim3(static_cast<unsigned>(mdg_cdiv(cols, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), stats,
                     running_mean, running_var, static_cast<double>(rows), cols, eps, momentum);
  MDG_CHECK_LAUNCH("mdg_batchnorm_replay_update");
  return MDG_OK;
}

extern "C" int mdg_batchnorm_train_bwd(const float* dy, const float* x, const float* stats, float* dx, float* dgamma, float* dbeta,
                                       int64_t rows, int64_t cols, void* workspace, size_t workspace_bytes, void* stream) {
  MDG_CHECK_ARG(rows > 1 && cols > 0, "mdg_batchnorm_train_bwd: bad shape");
  MDG_CHECK_ARG(dy && x && stats && dx && dgamma && dbeta, "mdg_batchnorm_train_bwd: null pointer");
  const size_t need = mdg_batchnorm_workspace_bytes(rows, cols);
  if (!workspace || workspace_bytes < need) {
    mdg_set_error("mdg_batchnorm_train_bwd: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return MDG_EWORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  // both column reductions from one pass over dy: partials of dbeta in the workspace's first half, of dgamma in its second
  float* part = static_cast<float*>(workspace) + 2 * cols;
  const int64_t nparts = mdg_cdiv(rows, 256);
  if (workspace_bytes >= (2 * static_cast<size_t>(nparts) + 2) * cols * sizeof(float)) {
    float* part2 = part + nparts * cols;
    hipLaunchKernelGGL(colreduce_bn_bwd_kernel, dim3(static_cast<unsigned>(mdg_cdiv(cols, 64)), static_cast<unsigned>(nparts)), dim3(256), 0, st, dy, cols, x, cols,
                       stats, stats + cols, part, part2, rows, cols);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(static_cast<unsigned>(mdg_cdiv(cols, 16))), dim3(256), 0, st, static_cast<const float*>(part), cols, dbeta, nparts,
                       cols, 0.f);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(static_cast<unsigned>(mdg_cdiv(cols, 16))), dim3(256), 0, st, static_cast<const float*>(part2), cols, dgamma, nparts,
                       cols, 0.f);
  } else {
    colreduce(dy, cols, nullptr, 0, nullptr, nullptr, dbeta, rows, cols, 0, part, st);
    colreduce(dy, cols, x, cols, stats, stats + cols, dgamma, rows, cols, 2, part, st);
  }
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(static_cast<unsigned>(mdg_cdiv(rows * cols, 256))), dim3(256), 0, st, dy, x, stats, dbeta, dgamma,
                     dx, rows, cols, static_cast<float>(rows));
  MDG_CHECK_LAUNCH("mdg_batchnorm_train_bwd");
  return MDG_OK;
}

// ------------------------------------------------------------------------------------------------- LayerNorm backward
extern "C" size_t mdg_layernorm_bwd_workspace_bytes(int64_t rows, int64_t d) {
  return rows <= 0 || d <= 0 ? 0 : static_cast<size_t>(mdg_cdiv(rows, LN_VEC_ROWS)) * 2 * d * sizeof(float);
}

static int layernorm_bwd_impl(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, float* dx, int64_t lddx,
                              float* dgamma, float* dbeta, int64_t rows, int64_t d, float eps, void* workspace, size_t workspace_bytes,
                              void* stream, const float* extra, int64_t ldextra) {
  MDG_CHECK_ARG(!extra || ldextra >= d, "mdg_layernorm_bwd: row stride of the added gradient shorter than d");
  MDG_CHECK_ARG(rows >= 0 && d > 0 && d <= LN_DMAX, "mdg_layernorm_bwd: d must be in [1,%d]", LN_DMAX);
  MDG_CHECK_ARG(lddy >= d && ldx >= d && lddx >= d, "mdg_layernorm_bwd: row strides shorter than d");
  MDG_CHECK_ARG(gamma && dgamma && dbeta, "mdg_layernorm_bwd: null parameter pointers");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool vec = d % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && mdg_aligned16(dy) && mdg_aligned16(x) && mdg_aligned16(dx) &&
                   mdg_aligned16(gamma) && (!extra || (ldextra % 4 == 0 && mdg_aligned16(extra)));
  const int64_t nb = mdg_cdiv(rows, vec ? LN_VEC_ROWS : 64);
  const size_t need = mdg_layernorm_bwd_workspace_bytes(rows, d);
  if (need && (!workspace || workspace_bytes < need)) {
    mdg_set_error("mdg_layernorm_bwd: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return MDG_EWORKSPACE;
  }
  if (rows > 0) {
    MDG_CHECK_ARG(dy && x && dx, "mdg_layernorm_bwd: null pointer");
    const size_t lds = static_cast<size_t>(8 * d) * sizeof(float);
    float* part = static_cast<float*>(workspace);
    const dim3 grid(static_cast<unsigned>(nb));
    const int di = static_cast<int>(d);
    if (vec) {
      if (d <= 256) hipLaunchKernelGGL(layernorm_bwd_vec_kernel<1>, grid, dim3(256), lds, st, dy, lddy, x, ldx, gamma, dx, lddx, part, rows, di, eps, extra, ldextra);
      else if (d <= 512) hipLaunchKernelGGL(layernorm_bwd_vec_kernel<2>, grid, dim3(256), lds, st, dy, lddy, x, ldx, gamma, dx, lddx, part, rows, di, eps, extra, ldextra);
      else if (d <= 1024) hipLaunchKernelGGL(layernorm_bwd_vec_kernel<4>, grid, dim3(256), lds, st, dy, lddy, x, ldx, gamma, dx, lddx, part, rows, di, eps, extra, ldextra);
      else hipLaunchKernelGGL(layernorm_bwd_vec_kernel<8>, grid, dim3(256), lds, st, dy, lddy, x, ldx, gamma, dx, lddx, part, rows, di, eps, extra, ldextra);
    } else if (d <= 128)
      hipLaunchKernelGGL(layernorm_bwd_kernel<2>, grid, dim3(256), lds, st, dy, lddy, x, ldx, gamma, dx, lddx, part, rows, di, eps, extra, ldextra);
    else if (d <= 512)
      hipLaunchKernelGGL(layernorm_bwd_kernel<8>, grid, dim3(256), lds, st, dy, lddy, x, ldx, gamma, dx, lddx, part, rows, di, eps, extra, ldextra);
    else
      hipLaunchKernelGGL(layernorm_bwd_kernel<32>, grid, dim3(256), lds, st, dy, lddy, x, ldx, gamma, dx, lddx, part, rows, di, eps, extra, ldextra);
  }
  // dgamma = sum of partial rows [nb, 2d] -> first d columns, dbeta the next d
  hipLaunchKernelGGL(colsum_final_kernel, dim3(static_cast<unsigned>(mdg_cdiv(d, 16))), dim3(256), 0, st, static_cast<const float*>(workspace),
                     2 * d, dgamma, nb, d, 0.f);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(static_cast<unsigned>(mdg_cdiv(d, 16))), dim3(256), 0, st,
                     static_cast<const float*>(workspace) + d, 2 * d, dbeta, nb, d, 0.f);
  MDG_CHECK_LAUNCH("mdg_layernorm_bwd");
  return MDG_OK;
}

extern "C" int mdg_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, float* dx, int64_t lddx,
                                 float* dgamma, float* dbeta, int64_t rows, int64_t d, float eps, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  return layernorm_bwd_impl(dy, lddy, x, ldx, gamma, dx, lddx, dgamma, dbeta, rows, d, eps, workspace, workspace_bytes, stream, nullptr, 0);
}

extern "C" int mdg_layernorm_bwd_add(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* extra,
                                     int64_t ldextra, float* dx, int64_t lddx, float* dgamma, float* dbeta, int64_t rows, int64_t d, float eps,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  MDG_CHECK_ARG(extra, "mdg_layernorm_bwd_add: null gradient to add");
  return layernorm_bwd_impl(dy, lddy, x, ldx, gamma, dx, lddx, dgamma, dbeta, rows, d, eps, workspace, workspace_bytes, stream, extra, ldextra);
}

// y = act(x * scale[c] + shift[c])   (eval-mode BatchNorm under autograd; shift may be null: the backward pass)
extern "C" int mdg_affine_act(const float* x, int64_t ldx, const float* scale, const float* shift, float* y, int64_t ldy, int64_t rows,
                              int64_t cols, int activation, void* stream) {
  MDG_CHECK_ARG(rows >= 0 && cols > 0 && ldx >= cols && ldy >= cols, "mdg_affine_act: bad shape");
  MDG_CHECK_ARG(activation >= MDG_ACT_NONE && activation <= MDG_ACT_SELU, "mdg_affine_act: unknown activation");
  if (rows == 0) return MDG_OK;
  MDG_CHECK_ARG(x && y && scale, "mdg_affine_act: null pointer");
  hipLaunchKernelGGL(affine_act_kernel, dim3(static_cast<unsigned>(mdg_cdiv(rows * cols, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     ldx, scale, shift, y, ldy, rows, cols, activation);
  MDG_CHECK_LAUNCH("mdg_affine_act");
  return MDG_OK;
}

extern "C" int mdg_axpby(const float* a, const float* b, float* out, int64_t n, int64_t nb, float alpha, float beta, void* stream) {
  MDG_CHECK_ARG(n >= 0 && nb > 0 && (n % nb) == 0, "mdg_axpby: numel(b) must divide numel(a)");
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(a && b && out, "mdg_axpby: null pointer");
  if (nb == n && vec4_ok(n, a, b, out))
    hipLaunchKernelGGL(axpby4_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n / 4, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, out, n / 4, alpha, beta);
  else
    hipLaunchKernelGGL(axpby_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, out, n, nb,
                       alpha, beta);
  MDG_CHECK_LAUNCH("mdg_axpby");
  return MDG_OK;
}

extern "C" int mdg_mul_device_scalar(const float* x, const float* scalar, float* out, int64_t n, void* stream) {
  MDG_CHECK_ARG(n >= 0, "mdg_mul_device_scalar: negative size");
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(x && scalar && out, "mdg_mul_device_scalar: null pointer");
  hipLaunchKernelGGL(mul_dev_scalar_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), x, scalar, out, n);
  MDG_CHECK_LAUNCH("mdg_mul_device_scalar");
  return MDG_OK;
}

extern "C" int mdg_gated_residual(const float* o, const float* x, const float* skip, float* out, int64_t n, void* stream) {
  MDG_CHECK_ARG(n >= 0, "mdg_gated_residual: negative size");
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(o && x && skip && out, "mdg_gated_residual: null pointer");
  hipLaunchKernelGGL(gated_residual_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), o, x, skip, out, n);
  MDG_CHECK_LAUNCH("mdg_gated_residual");
  return MDG_OK;
}

extern "C" int mdg_gated_residual_bwd(const float* dout, const float* o, const float* x, const float* skip, float* d_o, float* d_x,
                                      float* rowdot, int64_t rows, int64_t cols, void* stream) {
  MDG_CHECK_ARG(rows >= 0 && cols > 0 && cols <= (1 << 20), "mdg_gated_residual_bwd: bad shape");
  if (rows == 0) return MDG_OK;
  MDG_CHECK_ARG(dout && o && x && skip && d_o && d_x && rowdot, "mdg_gated_residual_bwd: null pointer");
  hipLaunchKernelGGL(gated_residual_bwd_kernel, dim3(static_cast<unsigned>(mdg_cdiv(rows, 4))), dim3(256), 0, static_cast<hipStream_t>(stream), dout, o, x,
                     skip, d_o, d_x, rowdot, rows, static_cast<int>(cols));
  MDG_CHECK_LAUNCH("mdg_gated_residual_bwd");
  return MDG_OK;
}

// ------------------------------------------------------------------------------------------------- BatchNorm1d in phases
// The same kernels as mdg_batchnorm_train_fwd/bwd, one phase per call, so that a data-parallel caller can all-reduce the
// per-column sums between phases (SyncBatchNorm over drug-sharded ranks): statistics over ALL ranks' rows.
extern "C" int mdg_col_reduce(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* center, const float* rstd, float* out,
                              int64_t rows, int64_t cols, int mode, void* workspace, size_t workspace_bytes, void* stream) {
  MDG_CHECK_ARG(rows >= 0 && cols > 0 && ldx >= cols && mode >= 0 && mode <= 2, "mdg_col_reduce: bad arguments");
  MDG_CHECK_ARG(out && (mode != 1 || center) && (mode != 2 || (y && center && rstd && ldy >= cols)), "mdg_col_reduce: missing operand for this mode");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (rows == 0) {
    (void)hipMemsetAsync(out, 0, static_cast<size_t>(cols) * sizeof(float), st);
    return MDG_OK;
  }
  MDG_CHECK_ARG(x, "mdg_col_reduce: null x");
  const size_t need = mdg_batchnorm_workspace_bytes(rows, cols);
  if (!workspace || workspace_bytes < need) {
    mdg_set_error("mdg_col_reduce: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return MDG_EWORKSPACE;
  }
  colreduce(x, ldx, y, ldy, center, rstd, out, rows, cols, mode, static_cast<float*>(workspace), st);
  MDG_CHECK_LAUNCH("mdg_col_reduce");
  return MDG_OK;
}

extern "C" int mdg_batchnorm_finalize(const float* sum, const float* sqsum, const float* gamma, const float* beta, float* running_mean,
                                      float* running_var, float* stats, double count, const double* count_dev, int64_t cols, float eps,
                                      float momentum, int phase, void* stream) {
  MDG_CHECK_ARG(cols > 0 && (count_dev || count > 1.0) && (phase == 0 || phase == 1), "mdg_batchnorm_finalize: bad arguments (needs more than one row in total)");
  MDG_CHECK_ARG(stats && sum && (phase == 0 || sqsum), "mdg_batchnorm_finalize: null pointer");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(static_cast<unsigned>(mdg_cdiv(cols, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), sum, sqsum,
                     gamma, beta, running_mean, running_var, stats, count, cols, eps, momentum, phase, count_dev);
  MDG_CHECK_LAUNCH("mdg_batchnorm_finalize");
  return MDG_OK;
}

extern "C" int mdg_batchnorm_bwd_apply(const float* dy, const float* x, const float* stats, const float* sum_dy, const float* sum_dy_xhat,
                                       float* dx, int64_t rows, int64_t cols, double count, const double* count_dev, void* stream) {
  MDG_CHECK_ARG(rows >= 0 && cols > 0 && (count_dev || count > 1.0), "mdg_batchnorm_bwd_apply: bad shape");
  if (rows == 0) return MDG_OK;
  MDG_CHECK_ARG(dy && x && stats && sum_dy && sum_dy_xhat && dx, "mdg_batchnorm_bwd_apply: null pointer");
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(static_cast<unsigned>(mdg_cdiv(rows * cols, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), dy, x,
                     stats, sum_dy, sum_dy_xhat, dx, rows, cols, static_cast<float>(count), count_dev);
  MDG_CHECK_LAUNCH("mdg_batchnorm_bwd_apply");
  return MDG_OK;
}
