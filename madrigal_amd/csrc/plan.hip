// Index plumbing of the gathered head's triple plan (ops.triple_plan; reference call site: train_ddi_batch.py:231-354 feeds a NEW
// batch of labelled triples to every step, so the plan is rebuilt per step).  As torch index / scan / repeat_interleave calls the
// plan was ~300 launches of a few microseconds each (3.1 ms of device time, 5.3 ms of wall time per plan with its host round trips);
// the sorts stay with rocprim, everything between them is here: one gather pass, binary-search pointer tables, and the "cut" of a
// CSR list into pieces of bounded length (tiles of 32, chunks of 256 / 512, pieces of 64) as a count pass + a fill pass.
#include "mdg_common.h"

namespace {

__global__ __launch_bounds__(256) void plan_gather_kernel(const int64_t* __restrict__ perm, const int64_t* __restrict__ labels,
                                                          const int64_t* __restrict__ heads, const int64_t* __restrict__ tails, int64_t T,
                                                          int64_t n_labels, int64_t n_head, int64_t n_tail, int64_t* __restrict__ hs,
                                                          int64_t* __restrict__ ts, int64_t* __restrict__ skey, int64_t* __restrict__ inv_perm,
                                                          int32_t* __restrict__ status) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= T) return;
  const int64_t p = perm[i];
  const int64_t l = labels[p], h = heads[p], t = tails[p];
  hs[i] = h;
  ts[i] = t;
  skey[i] = l * n_head + h;
  inv_perm[p] = i;
  int bad = 0;
  if (l < 0 || l >= n_labels) bad |= 1;
  if (h < 0 || h >= n_head || t < 0 || t >= n_tail) bad |= 2;
  if (bad) atomicOr(status, bad);
}

template <class V>
__global__ __launch_bounds__(256) void plan_lower_bounds_kernel(const V* __restrict__ vals, int64_t T, int64_t scale, int64_t n_bounds,
                                                                int64_t* __restrict__ out) {
  const int64_t b = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (b >= n_bounds) return;
  const int64_t want = b * scale;
  int64_t lo = 0, hi = T;                                  // first i with vals[i] >= want
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (static_cast<int64_t>(vals[mid]) < want) lo = mid + 1; else hi = mid;
  }
  out[b] = lo;
}

// one workgroup: first[i] = pieces of the lists before i, first[n] = totals[0] = all pieces, totals[1] = the longest list
__global__ __launch_bounds__(1024) void plan_cut_count_kernel(const int64_t* __restrict__ ptr, int64_t n, int64_t size, int64_t* __restrict__ first,
                                                              int64_t* __restrict__ totals) {
  __shared__ int64_t wsum[16];
  __shared__ int64_t wmax[16];
  __shared__ int64_t carry;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) carry = 0;
  int64_t longest = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += 1024) {
    const int64_t i = base + tid;
    const int64_t c = i < n ? ptr[i + 1] - ptr[i] : 0;
    const int64_t per = (c + size - 1) / size;
    longest = c > longest ? c : longest;
    int64_t inc = per;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int64_t u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int64_t run = carry + inc - per;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    if (i < n) first[i] = run;
    __syncthreads();
    if (tid == 1023) carry = run + per;
    __syncthreads();
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int64_t u = __shfl_xor(longest, o, 64);
    longest = u > longest ? u : longest;
  }
  if (lane == 0) wmax[wave] = longest;
  __syncthreads();
  if (tid == 0) {
    int64_t m = 0;
    for (int w = 0; w < 16; ++w) m = wmax[w] > m ? wmax[w] : m;
    first[n] = carry;
    totals[0] = carry;
    totals[1] = m;
  }
}

__global__ __launch_bounds__(256) void plan_cut_fill_kernel(const int64_t* __restrict__ ptr, const int64_t* __restrict__ first, int64_t n,
                                                            int64_t size, int64_t total, int64_t* __restrict__ which, int64_t* __restrict__ start) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (j == total) start[total] = ptr[n];
  if (j >= total) return;
  int64_t lo = 0, hi = n;                                  // the list whose pieces include j: the last i with first[i] <= j among lists with pieces
  while (hi - lo > 1) {                                    // invariant: first[lo] <= j < first[hi]
    const int64_t mid = (lo + hi) >> 1;
    if (first[mid] <= j) lo = mid; else hi = mid;
  }
  if (which) which[j] = lo;
  start[j] = ptr[lo] + size * (j - first[lo]);
}

// flag[i] = 1 where a new (label, head) pair starts (flag[0] = 0: the inclusive prefix sum is the pair of every triple)
__global__ __launch_bounds__(256) void plan_pair_flags_kernel(const int64_t* __restrict__ skey, int64_t T, int64_t* __restrict__ flag) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= T) return;
  flag[i] = (i > 0 && skey[i] != skey[i - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void plan_pair_table_kernel(const int64_t* __restrict__ skey, const int64_t* __restrict__ pair_of, int64_t T,
                                                              int64_t n_head, int64_t P, int64_t* __restrict__ pair_ptr,
                                                              int64_t* __restrict__ pair_drug) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i == T) pair_ptr[P] = T;
  if (i >= T) return;
  if (i == 0 || skey[i] != skey[i - 1]) {
    const int64_t p = pair_of[i];
    pair_ptr[p] = i;
    pair_drug[p] = skey[i] % n_head;
  }
}

// out[i] = idx[i] < T ? src[idx[i]] : fill
__global__ __launch_bounds__(256) void plan_take_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ idx, int64_t n, int64_t T,
                                                        int64_t fill, int64_t* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t k = idx[i];
  out[i] = (k >= 0 && k < T) ? src[k] : fill;
}

inline unsigned grid_of(int64_t n) { return static_cast<unsigned>(mdg_cdiv(n > 0 ? n : 1, 256)); }

}  // namespace

extern "C" int mdg_plan_gather(const int64_t* perm, const int64_t* labels, const int64_t* heads, const int64_t* tails, int64_t T,
                               int64_t n_labels, int64_t n_head, int64_t n_tail, int64_t* hs, int64_t* ts, int64_t* skey, int64_t* inv_perm,
                               int32_t* status, void* stream) {
  MDG_CHECK_ARG(T >= 0 && n_labels >= 1 && n_head >= 1 && n_tail >= 1, "mdg_plan_gather: bad sizes (T=%lld)", (long long)T);
  if (T == 0) return MDG_OK;
  MDG_CHECK_ARG(perm && labels && heads && tails && hs && ts && skey && inv_perm && status, "mdg_plan_gather: null argument");
  hipLaunchKernelGGL(plan_gather_kernel, dim3(grid_of(T)), dim3(256), 0, static_cast<hipStream_t>(stream), perm, labels, heads, tails, T, n_labels,
                     n_head, n_tail, hs, ts, skey, inv_perm, status);
  MDG_CHECK_LAUNCH("mdg_plan_gather");
  return MDG_OK;
}

extern "C" int mdg_plan_lower_bounds(const void* vals, int val_bytes, int64_t T, int64_t scale, int64_t n_bounds, int64_t* out, void* stream) {
  MDG_CHECK_ARG(T >= 0 && n_bounds >= 0 && scale >= 1 && (val_bytes == 2 || val_bytes == 4 || val_bytes == 8),
                "mdg_plan_lower_bounds: bad arguments (T=%lld, val_bytes=%d)", (long long)T, val_bytes);
  if (n_bounds == 0) return MDG_OK;
  MDG_CHECK_ARG(out && (vals || T == 0), "mdg_plan_lower_bounds: null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (val_bytes == 2)
    hipLaunchKernelGGL(plan_lower_bounds_kernel<int16_t>, dim3(grid_of(n_bounds)), dim3(256), 0, st, static_cast<const int16_t*>(vals), T, scale, n_bounds, out);
  else if (val_bytes == 4)
    hipLaunchKernelGGL(plan_lower_bounds_kernel<int32_t>, dim3(grid_of(n_bounds)), dim3(256), 0, st, static_cast<const int32_t*>(vals), T, scale, n_bounds, out);
  else
    hipLaunchKernelGGL(plan_lower_bounds_kernel<int64_t>, dim3(grid_of(n_bounds)), dim3(256), 0, st, static_cast<const int64_t*>(vals), T, scale, n_bounds, out);
  MDG_CHECK_LAUNCH("mdg_plan_lower_bounds");
  return MDG_OK;
}

extern "C" int mdg_plan_cut_count(const int64_t* ptr, int64_t n, int64_t size, int64_t* first, int64_t* totals, void* stream) {
  MDG_CHECK_ARG(n >= 0 && size >= 1 && ptr && first && totals, "mdg_plan_cut_count: bad arguments (n=%lld, size=%lld)", (long long)n, (long long)size);
  hipLaunchKernelGGL(plan_cut_count_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), ptr, n, size, first, totals);
  MDG_CHECK_LAUNCH("mdg_plan_cut_count");
  return MDG_OK;
}

extern "C" int mdg_plan_cut_fill(const int64_t* ptr, const int64_t* first, int64_t n, int64_t size, int64_t total, int64_t* which, int64_t* start,
                                 void* stream) {
  MDG_CHECK_ARG(n >= 0 && size >= 1 && total >= 0 && ptr && first && start,
                "mdg_plan_cut_fill: bad arguments (n=%lld, total=%lld)", (long long)n, (long long)total);
  hipLaunchKernelGGL(plan_cut_fill_kernel, dim3(grid_of(total + 1)), dim3(256), 0, static_cast<hipStream_t>(stream), ptr, first, n, size, total, which,
                     start);
  MDG_CHECK_LAUNCH("mdg_plan_cut_fill");
  return MDG_OK;
}

extern "C" int mdg_plan_pair_flags(const int64_t* skey, int64_t T, int64_t* flag, void* stream) {
  MDG_CHECK_ARG(T >= 0, "mdg_plan_pair_flags: bad size");
  if (T == 0) return MDG_OK;
  MDG_CHECK_ARG(skey && flag, "mdg_plan_pair_flags: null argument");
  hipLaunchKernelGGL(plan_pair_flags_kernel, dim3(grid_of(T)), dim3(256), 0, static_cast<hipStream_t>(stream), skey, T, flag);
  MDG_CHECK_LAUNCH("mdg_plan_pair_flags");
  return MDG_OK;
}

extern "C" int mdg_plan_pair_table(const int64_t* skey, const int64_t* pair_of, int64_t T, int64_t n_head, int64_t P, int64_t* pair_ptr,
                                   int64_t* pair_drug, void* stream) {
  MDG_CHECK_ARG(T >= 1 && P >= 1 && n_head >= 1 && skey && pair_of && pair_ptr && pair_drug, "mdg_plan_pair_table: bad arguments");
  hipLaunchKernelGGL(plan_pair_table_kernel, dim3(grid_of(T + 1)), dim3(256), 0, static_cast<hipStream_t>(stream), skey, pair_of, T, n_head, P, pair_ptr,
                     pair_drug);
  MDG_CHECK_LAUNCH("mdg_plan_pair_table");
  return MDG_OK;
}

extern "C" int mdg_plan_take(const int64_t* src, const int64_t* idx, int64_t n, int64_t T, int64_t fill, int64_t* out, void* stream) {
  MDG_CHECK_ARG(n >= 0 && T >= 0, "mdg_plan_take: bad sizes");
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(idx && out && (src || T == 0), "mdg_plan_take: null argument");
  hipLaunchKernelGGL(plan_take_kernel, dim3(grid_of(n)), dim3(256), 0, static_cast<hipStream_t>(stream), src, idx, n, T, fill, out);
  MDG_CHECK_LAUNCH("mdg_plan_take");
  return MDG_OK;
}
