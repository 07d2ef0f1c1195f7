// Per-outcome normalised ranks of all-pairs scores for gfx950.
//
// Reference: notebooks/normalize_scores.py:33-74.  Per outcome slice s[N,N]: overwrite the upper triangle
// and diagonal with 1e7, rank all N^2 entries (argsort o argsort, 1-based, ascending), divide by
// N(N-1)/2, zero the masked entries, add the transpose.  For real scores < 1e7 the ranks of the kept
// (strict lower triangle) entries are their ranks among the M = N(N-1)/2 kept entries, so only those
// are sorted here.
//
// Implementation: stable LSD radix sort, 4 passes x 8 bits, of (order-preserving uint32 key, position p
// in the row-major enumeration of the lower triangle), all outcomes of a chunk in one launch per kernel
// (blockIdx.y = outcome).  Per pass: per-tile digit histogram -> exclusive scan in (digit, tile) order ->
// scatter with a stable in-tile rank built from wave ballots.  The last pass does not write the sorted
// pairs: position q of payload p IS rank q+1, which is written to out[i,j] and out[j,i] directly.
// Ties: stable in p (= flat row-major index); numpy's default argsort in the reference is unstable, so
// tie order there is implementation-defined ("Ties in rank normalisation").
// HBM-bound integer work: ~ (4 + 4*16 + 8) bytes moved per score.
#include "mdg_common.h"

namespace {

constexpr int TPB = 256;            // threads per block
constexpr int ITEMS = 16;           // keys per thread
constexpr int TILE = TPB * ITEMS;   // keys per block

__device__ __forceinline__ uint32_t order_key(float f) {
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// p -> (i, j) with i > j, p = i(i-1)/2 + j
__device__ __forceinline__ void tri_decode(int64_t p, int& i, int& j) {
  int64_t r = static_cast<int64_t>((1.0 + sqrt(1.0 + 8.0 * static_cast<double>(p))) * 0.5);
  while (r * (r - 1) / 2 > p) --r;
  while ((r + 1) * r / 2 <= p) ++r;
  i = static_cast<int>(r);
  j = static_cast<int>(p - r * (r - 1) / 2);
}

__global__ __launch_bounds__(TPB) void extract_keys_kernel(const float* __restrict__ scores, uint32_t* __restrict__ keys, int N,
                                                           int64_t M) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * TPB + threadIdx.x;
  if (p >= M) return;
  const int64_t seg = blockIdx.y;
  int i, j;
  tri_decode(p, i, j);
  keys[seg * M + p] = order_key(scores[(seg * N + i) * static_cast<int64_t>(N) + j]);
}

// hist[(seg * 256 + digit) * nblk + blk]
__global__ __launch_bounds__(TPB) void histogram_kernel(const uint32_t* __restrict__ keys, uint32_t* __restrict__ hist, int64_t M,
                                                        int nblk, int shift) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t seg = blockIdx.y;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * TILE;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int64_t p = base + k * TPB + threadIdx.x;
    if (p < M) atomicAdd(&h[(keys[seg * M + p] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(seg * 256 + threadIdx.x) * nblk + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of the 256*nblk counters of one outcome, in place; one workgroup per outcome
__global__ __launch_bounds__(TPB) void scan_kernel(uint32_t* __restrict__ hist, int nblk) {
  __shared__ uint32_t part[TPB];
  uint32_t* h = hist + static_cast<int64_t>(blockIdx.x) * 256 * nblk;
  const int64_t total = static_cast<int64_t>(256) * nblk;
  const int64_t per = (total + TPB - 1) / TPB;
  const int64_t a = threadIdx.x * per, b = (a + per < total) ? a + per : total;
  uint32_t s = 0;
  for (int64_t t = a; t < b; ++t) s += h[t];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int t = 0; t < TPB; ++t) {
      const uint32_t v = part[t];
      part[t] = run;
      run += v;
    }
  }
  __syncthreads();
  uint32_t run = part[threadIdx.x];
  for (int64_t t = a; t < b; ++t) {
    const uint32_t v = h[t];
    h[t] = run;
    run += v;
  }
}

template <bool FIRST, bool LAST>
__global__ __launch_bounds__(TPB) void scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ pay_in,
                                                      uint32_t* __restrict__ keys_out, uint32_t* __restrict__ pay_out,
                                                      const uint32_t* __restrict__ offsets, float* __restrict__ out, int N,
                                                      int64_t M, int nblk, int shift, double denom) {
  __shared__ uint32_t wave_hist[4][256];
  __shared__ uint32_t counter[256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t seg = blockIdx.y;
  counter[tid] = offsets[(seg * 256 + tid) * nblk + blockIdx.x];
#pragma unroll
  for (int w = 0; w < 4; ++w) wave_hist[w][tid] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * TILE;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int k = 0; k < ITEMS; ++k) {
    const int64_t p = base + k * TPB + tid;
    const bool valid = p < M;
    const uint32_t key = valid ? keys_in[seg * M + p] : 0xFFFFFFFFu;
    const uint32_t pay = FIRST ? static_cast<uint32_t>(p) : (valid ? pay_in[seg * M + p] : 0u);
    const uint32_t dg = (key >> shift) & 255u;
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (dg >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    const uint32_t rank_in_wave = __popcll(peers & lt);
    if (valid && rank_in_wave == 0) wave_hist[wave][dg] = __popcll(peers);      // one leader per (wave, digit)
    __syncthreads();
    uint32_t pos = 0;
    if (valid) {
      pos = counter[dg] + rank_in_wave;
      for (int w = 0; w < wave; ++w) pos += wave_hist[w][dg];
    }
    __syncthreads();
    {   // thread t owns digit t: advance the running counter, clear the per-wave counts
      counter[tid] += wave_hist[0][tid] + wave_hist[1][tid] + wave_hist[2][tid] + wave_hist[3][tid];
#pragma unroll
      for (int w = 0; w < 4; ++w) wave_hist[w][tid] = 0;
    }
    if (valid) {
      if constexpr (LAST) {
        int i, j;
        tri_decode(pay, i, j);
        const float v = static_cast<float>(static_cast<double>(pos + 1) / denom);
        float* o = out + seg * static_cast<int64_t>(N) * N;
        o[static_cast<int64_t>(i) * N + j] = v;
        o[static_cast<int64_t>(j) * N + i] = v;
      } else {
        keys_out[seg * M + pos] = key;
        pay_out[seg * M + pos] = pay;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(TPB) void zero_diag_kernel(float* __restrict__ out, int N) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < N) out[(static_cast<int64_t>(blockIdx.y) * N + i) * N + i] = 0.f;
}

inline size_t a256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

// Geometric mean over K <= 8 tensors, elementwise, fp32 like scipy.stats.mstats.gmean on float32 input:
// exp(mean_k(log x_k)).  Entries where any x_k <= 0 (the zero diagonal of normalised ranks; masked by scipy)
// give 0.  (generate_embeddings.ipynb, the 5-seed ensembling cell.)
struct GmeanArgs { const float* in[8]; int K; };

__global__ __launch_bounds__(256) void gmean_kernel(const GmeanArgs a, float* __restrict__ out, int64_t n4) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  bool pos[4] = {true, true, true, true};
  for (int k = 0; k < a.K; ++k) {
    const f32x4 v = reinterpret_cast<const f32x4*>(a.in[k])[i];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      pos[c] = pos[c] && v[c] > 0.f;
      acc[c] += logf(v[c] > 0.f ? v[c] : 1.f);
    }
  }
  f32x4 o;
#pragma unroll
  for (int c = 0; c < 4; ++c) o[c] = pos[c] ? expf(acc[c] / static_cast<float>(a.K)) : 0.f;
  reinterpret_cast<f32x4*>(out)[i] = o;
}

}  // namespace

extern "C" size_t mdg_rank_normalize_workspace_bytes(int64_t n_outcomes, int64_t N) {
  if (n_outcomes <= 0 || N < 2) return 0;
  const size_t M = static_cast<size_t>(N) * (N - 1) / 2;
  const size_t nblk = (M + TILE - 1) / TILE;
  return 4 * a256(static_cast<size_t>(n_outcomes) * M * 4) + a256(static_cast<size_t>(n_outcomes) * 256 * nblk * 4);
}

extern "C" int mdg_rank_normalize(const float* scores, float* out, int64_t n_outcomes, int64_t N, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  MDG_CHECK_ARG(n_outcomes >= 0 && N >= 0 && N <= 65535 && n_outcomes <= 65535, "mdg_rank_normalize: bad sizes (outcomes per call and N <= 65535)");
  if (n_outcomes == 0 || N == 0) return MDG_OK;
  MDG_CHECK_ARG(scores && out, "mdg_rank_normalize: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned L = static_cast<unsigned>(n_outcomes);
  hipLaunchKernelGGL(zero_diag_kernel, dim3(static_cast<unsigned>(mdg_cdiv(N, TPB)), L), dim3(TPB), 0, st, out, static_cast<int>(N));
  if (N < 2) { MDG_CHECK_LAUNCH("mdg_rank_normalize"); return MDG_OK; }
  const int64_t M = N * (N - 1) / 2;
  const int nblk = static_cast<int>(mdg_cdiv(M, TILE));
  const size_t need = mdg_rank_normalize_workspace_bytes(n_outcomes, N);
  if (!workspace || workspace_bytes < need || !mdg_aligned16(workspace)) {
    mdg_set_error("mdg_rank_normalize: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return MDG_EWORKSPACE;
  }
  char* ws = static_cast<char*>(workspace);
  const size_t kb = a256(static_cast<size_t>(n_outcomes) * M * 4);
  uint32_t* k0 = reinterpret_cast<uint32_t*>(ws);
  uint32_t* k1 = reinterpret_cast<uint32_t*>(ws + kb);
  uint32_t* p0 = reinterpret_cast<uint32_t*>(ws + 2 * kb);
  uint32_t* p1 = reinterpret_cast<uint32_t*>(ws + 3 * kb);
  uint32_t* hist = reinterpret_cast<uint32_t*>(ws + 4 * kb);
  const double denom = static_cast<double>(N) * static_cast<double>(N - 1) / 2.0;
  hipLaunchKernelGGL(extract_keys_kernel, dim3(static_cast<unsigned>(mdg_cdiv(M, TPB)), L), dim3(TPB), 0, st, scores, k0, static_cast<int>(N), M);
  const dim3 grid(static_cast<unsigned>(nblk), L);
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 8 * pass;
    uint32_t* kin = (pass & 1) ? k1 : k0;
    uint32_t* kout = (pass & 1) ? k0 : k1;
    uint32_t* pin = (pass & 1) ? p1 : p0;
    uint32_t* pout = (pass & 1) ? p0 : p1;
    hipLaunchKernelGGL(histogram_kernel, grid, dim3(TPB), 0, st, kin, hist, M, nblk, shift);
    hipLaunchKernelGGL(scan_kernel, dim3(L), dim3(TPB), 0, st, hist, nblk);
    if (pass == 0)
      hipLaunchKernelGGL((scatter_kernel<true, false>), grid, dim3(TPB), 0, st, kin, pin, kout, pout, hist, out, static_cast<int>(N), M, nblk, shift, denom);
    else if (pass == 3)
      hipLaunchKernelGGL((scatter_kernel<false, true>), grid, dim3(TPB), 0, st, kin, pin, kout, pout, hist, out, static_cast<int>(N), M, nblk, shift, denom);
    else
      hipLaunchKernelGGL((scatter_kernel<false, false>), grid, dim3(TPB), 0, st, kin, pin, kout, pout, hist, out, static_cast<int>(N), M, nblk, shift, denom);
  }
  MDG_CHECK_LAUNCH("mdg_rank_normalize");
  return MDG_OK;
}

extern "C" int mdg_gmean(const float* const* inputs_host, int K, float* out, int64_t n, void* stream) {
  MDG_CHECK_ARG(inputs_host && out && K >= 1 && K <= 8 && n >= 0 && n % 4 == 0, "mdg_gmean: 1 <= K <= 8 tensors of n (multiple of 4) floats");
  if (n == 0) return MDG_OK;
  GmeanArgs a{};
  a.K = K;
  for (int k = 0; k < K; ++k) {
    MDG_CHECK_ARG(inputs_host[k] && mdg_aligned16(inputs_host[k]), "mdg_gmean: input %d null or not 16-byte aligned", k);
    a.in[k] = inputs_host[k];
  }
  MDG_CHECK_ARG(mdg_aligned16(out), "mdg_gmean: out not 16-byte aligned");
  hipLaunchKernelGGL(gmean_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n / 4, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), a, out, n / 4);
  MDG_CHECK_LAUNCH("mdg_gmean");
  return MDG_OK;
}
