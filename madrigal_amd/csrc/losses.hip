// Loss-side kernels for gfx950: InfoNCE finish (SimCLR_NovelDDI.contrastive_loss) and the
// sigmoid -> gather labelled triples -> BCE step of the finetune loop.
#include "mdg_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// InfoNCE finish (madrigal/models/simclr.py:74-108).  Input: sim = F F^T for the 2B normalised
// features (computed by mdg_l2_normalize + mdg_linear).  Per row i (one wave per row):
//   masked_fill(too_hard_neg.repeat(2,2), -1e9) -> drop the diagonal -> logits = ./T  [2B, 2B-1]
//   labels[i, :] = 1 at the other view of the same drug, else 0                     [2B, 2B-1]
//   row_loss[i] = logsumexp(logits[i]) - logits[i, positive]      (soft-label CE with one positive)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void infonce_rows_kernel(const float* __restrict__ sim, const uint8_t* __restrict__ hard,
                                                           float* __restrict__ logits, float* __restrict__ labels,
                                                           float* __restrict__ row_loss, int B, float inv_T) {
  const int lane = threadIdx.x & 63;
  const int n2 = 2 * B;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n2) return;
  const int ib = i % B;
  const int pos = (i + B) % n2;                      // column (before diagonal removal) of the positive
  const float* srow = sim + static_cast<int64_t>(i) * n2;
  float m = -INFINITY;
  for (int j = lane; j < n2; j += 64) {
    if (j == i) continue;
    float v = srow[j];
    if (hard && hard[static_cast<int64_t>(ib) * B + (j % B)]) v = -1e9f;
    v *= inv_T;
    const int jc = j - (j > i ? 1 : 0);
    if (logits) logits[static_cast<int64_t>(i) * (n2 - 1) + jc] = v;
    if (labels) labels[static_cast<int64_t>(i) * (n2 - 1) + jc] = (j == pos) ? 1.f : 0.f;
    m = fmaxf(m, v);
  }
  m = mdg_wave_max(m);
  float s = 0.f, lp = 0.f;
  for (int j = lane; j < n2; j += 64) {
    if (j == i) continue;
    float v = srow[j];
    if (hard && hard[static_cast<int64_t>(ib) * B + (j % B)]) v = -1e9f;
    v *= inv_T;
    s += expf(v - m);
    if (j == pos) lp = v;
  }
  s = mdg_wave_sum(s);
  lp = mdg_wave_sum(lp);
  if (lane == 0) row_loss[i] = (logf(s) + m) - lp;
}

// Backward of the InfoNCE finish: d loss / d sim[i,j] = dloss / (2B T) * (softmax_j(logits[i,:]) - [j == positive(i)]) for
// j != i, 0 on the diagonal; entries replaced by -1e9 (too-hard negatives) are constants of the graph (masked_fill).
// One wave per row; the row's max / sum are recomputed from sim.  dloss is read from device memory.
__global__ __launch_bounds__(256) void infonce_bwd_rows_kernel(const float* __restrict__ sim, const uint8_t* __restrict__ hard,
                                                               const float* __restrict__ dloss, float* __restrict__ dsim, int B, float inv_T) {
  const int lane = threadIdx.x & 63;
  const int n2 = 2 * B;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n2) return;
  const int ib = i % B;
  const int pos = (i + B) % n2;
  const float* srow = sim + static_cast<int64_t>(i) * n2;
  float m = -INFINITY;
  for (int j = lane; j < n2; j += 64) {
    if (j == i) continue;
    float v = srow[j];
    if (hard && hard[static_cast<int64_t>(ib) * B + (j % B)]) v = -1e9f;
    m = fmaxf(m, v * inv_T);
  }
  m = mdg_wave_max(m);
  float s = 0.f;
  for (int j = lane; j < n2; j += 64) {
    if (j == i) continue;
    float v = srow[j];
    if (hard && hard[static_cast<int64_t>(ib) * B + (j % B)]) v = -1e9f;
    s += expf(v * inv_T - m);
  }
  s = mdg_wave_sum(s);
  const float scale = dloss[0] * inv_T / static_cast<float>(n2);
  float* drow = dsim + static_cast<int64_t>(i) * n2;
  for (int j = lane; j < n2; j += 64) {
    float g = 0.f;
    if (j != i) {
      const bool masked = hard && hard[static_cast<int64_t>(ib) * B + (j % B)];
      if (!masked) g = scale * (expf(srow[j] * inv_T - m) / s - (j == pos ? 1.f : 0.f));
    }
    drow[j] = g;
  }
}

// mean of n floats -> out[0]; single workgroup, fixed summation order (reproducible)
__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n) {
  __shared__ float part[4];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += x[i];
  s = mdg_wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((part[0] + part[1]) + (part[2] + part[3])) / static_cast<float>(n);
}

// ---------------------------------------------------------------------------------------------
// pred[e] = P[label[e], head[e], tail[e]] (P = sigmoid scores or raw logits, apply_sigmoid says which);
// term[e] = -(y log p + (1-y) log(1-p)) with both logs clamped at -100 like nn.BCELoss
// (train_ddi_batch.py:285-288, madrigal/utils.py:616-619).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_bce_kernel(const float* __restrict__ scores, int64_t n_head, int64_t n_tail,
                                                         const int64_t* __restrict__ label, const int64_t* __restrict__ head,
                                                         const int64_t* __restrict__ tail, const float* __restrict__ y,
                                                         float* __restrict__ pred, float* __restrict__ term, int64_t n,
                                                         int apply_sigmoid) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= n) return;
  float p = scores[(label[e] * n_head + head[e]) * n_tail + tail[e]];
  if (apply_sigmoid) p = 1.0f / (1.0f + expf(-p));
  pred[e] = p;
  if (term) {
    const float lp = fmaxf(logf(p), -100.f), l1 = fmaxf(logf(1.0f - p), -100.f);
    term[e] = -(y[e] * lp + (1.0f - y[e]) * l1);
  }
}

}  // namespace

extern "C" int mdg_infonce_finish(const float* sim, const uint8_t* too_hard_neg, float* logits, float* labels, float* row_loss,
                                  float* loss, int64_t B, float temperature, void* stream) {
  MDG_CHECK_ARG(B >= 1 && B <= (1 << 20) && temperature > 0.f, "mdg_infonce_finish: bad B / temperature");
  MDG_CHECK_ARG(sim && row_loss && loss, "mdg_infonce_finish: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(infonce_rows_kernel, dim3(static_cast<unsigned>(mdg_cdiv(2 * B, 4))), dim3(256), 0, st, sim, too_hard_neg, logits,
                     labels, row_loss, static_cast<int>(B), 1.0f / temperature);
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, row_loss, loss, 2 * B);
  MDG_CHECK_LAUNCH("mdg_infonce_finish");
  return MDG_OK;
}

extern "C" int mdg_infonce_bwd(const float* sim, const uint8_t* too_hard_neg, const float* dloss, float* dsim, int64_t B, float temperature,
                               void* stream) {
  MDG_CHECK_ARG(B >= 1 && B <= (1 << 20) && temperature > 0.f, "mdg_infonce_bwd: bad B / temperature");
  MDG_CHECK_ARG(sim && dloss && dsim && sim != dsim, "mdg_infonce_bwd: null / aliased pointer");
  hipLaunchKernelGGL(infonce_bwd_rows_kernel, dim3(static_cast<unsigned>(mdg_cdiv(2 * B, 4))), dim3(256), 0, static_cast<hipStream_t>(stream), sim,
                     too_hard_neg, dloss, dsim, static_cast<int>(B), 1.0f / temperature);
  MDG_CHECK_LAUNCH("mdg_infonce_bwd");
  return MDG_OK;
}

extern "C" int mdg_gather_bce(const float* scores, int64_t n_labels, int64_t n_head, int64_t n_tail, const int64_t* label,
                              const int64_t* head, const int64_t* tail, const float* target, float* pred, float* term, float* loss,
                              int64_t n, int apply_sigmoid, void* stream) {
  MDG_CHECK_ARG(n >= 0 && n_labels >= 0 && n_head >= 0 && n_tail >= 0, "mdg_gather_bce: negative size");
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(scores && label && head && tail && pred, "mdg_gather_bce: null pointer");
  MDG_CHECK_ARG((term == nullptr) == (loss == nullptr) && (!term || target), "mdg_gather_bce: term, loss and target come together");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gather_bce_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n, 256))), dim3(256), 0, st, scores, n_head, n_tail, label,
                     head, tail, target, pred, term, n, apply_sigmoid);
  if (term) hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, term, loss, n);
  MDG_CHECK_LAUNCH("mdg_gather_bce");
  return MDG_OK;
}
