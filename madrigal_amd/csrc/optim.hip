// Multi-tensor AdamW for the finetune step (train_ddi_batch.py:350, madrigal/utils.py:600-613: torch.optim.AdamW over
// parameter groups with their own lr / weight decay).  One launch updates every parameter tensor: the host passes a
// table of 4096-element chunks (param / grad / exp_avg / exp_avg_sq pointers) and per-tensor hyper-parameters.
#include "mdg_common.h"

namespace {

constexpr int OPT_CHUNK = 4096;

// hyper[tensor] = {lr, beta1, beta2, eps, weight_decay, 1/bias_correction1, 1/sqrt(bias_correction2), unused}
__global__ __launch_bounds__(256) void adamw_multi_kernel(const int64_t* __restrict__ ptrs, const int32_t* __restrict__ lens,
                                                          const int32_t* __restrict__ tensor_of_chunk, const float* __restrict__ hyper) {
  const int64_t c = blockIdx.x;
  float* __restrict__ p = reinterpret_cast<float*>(ptrs[4 * c + 0]);
  const float* __restrict__ g = reinterpret_cast<const float*>(ptrs[4 * c + 1]);
  float* __restrict__ m = reinterpret_cast<float*>(ptrs[4 * c + 2]);
  float* __restrict__ v = reinterpret_cast<float*>(ptrs[4 * c + 3]);
  const float* h = hyper + 8 * tensor_of_chunk[c];
  const float lr = h[0], b1 = h[1], b2 = h[2], eps = h[3], wd = h[4], ibc1 = h[5], isbc2 = h[6];
  const int n = lens[c];
  for (int i = threadIdx.x; i < n; i += 256) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float pi = p[i] * (1.0f - lr * wd);                       // decoupled weight decay
    p[i] = pi - (lr * ibc1) * mi / (sqrtf(vi) * isbc2 + eps);
  }
}

}  // namespace

extern "C" int mdg_adamw_chunk_elems(void) { return OPT_CHUNK; }

extern "C" int mdg_adamw_multi(const int64_t* chunk_ptrs, const int32_t* chunk_lens, const int32_t* chunk_tensor, const float* hyper,
                               int64_t n_chunks, void* stream) {
  MDG_CHECK_ARG(n_chunks >= 0 && n_chunks <= 0x7fffffff, "mdg_adamw_multi: bad chunk count");
  if (n_chunks == 0) return MDG_OK;
  MDG_CHECK_ARG(chunk_ptrs && chunk_lens && chunk_tensor && hyper, "mdg_adamw_multi: null table");
  hipLaunchKernelGGL(adamw_multi_kernel, dim3(static_cast<unsigned>(n_chunks)), dim3(256), 0, static_cast<hipStream_t>(stream), chunk_ptrs,
                     chunk_lens, chunk_tensor, hyper);
  MDG_CHECK_LAUNCH("mdg_adamw_multi");
  return MDG_OK;
}
