// Parameter-space side of the HGT conv in training: the composite projection weights of every node type from the LIVE parameters,
// and the gradients of the parameters from the gradient of the composite weights.
//
// Reference: PyG 2.3 HGTConv as used at madrigal/models/models.py:76-79, 90-94: per node type t one Linear to K | Q | V (kqv_lin), per
// edge type r = (s, ., d) and head h a [D,D] matrix k_rel / v_rel applied to the SOURCE side (k'_h = k_h @ k_rel[h,r], v'_h = v_h @
// v_rel[h,r]) and p_rel[r][h] / sqrt(D) on the logit.  The conv kernels read q | k'_r v'_r ... of a node straight from ONE GEMM per
// node type, so the relation matrices (with p_rel / sqrt(D) folded into the keys') are multiplied into the K / V projection weights:
//
//   big_w rows of type t:  [ Wq_t (F) | for every used relation r leaving t: K_r (F), V_r (F) ]      (bias column likewise in big_b)
//   K_r[(h,b), c] = sum_a  k_rel[h,r][a,b] * p_rel[r][h] / sqrt(D) * Wk_t[(h,a), c]
//   V_r[(h,b), c] = sum_a  v_rel[h,r][a,b]                          * Wv_t[(h,a), c]
//
// Rounds 1-3 assembled this with ~30 torch ops per conv (stack / bmm / cat / index_select: rocBLAS launches, and a backward graph of
// as many nodes whose per-parameter gradients arrived as views that AccumulateGrad cloned): ~150 launches of the finetune step.
// O(parameters) work, two launches here (forward, backward) + one memset; exact fp32, fixed summation order.
#include "mdg_common.h"

namespace {

constexpr int HP_TPB = 256;
constexpr int HP_SPLIT = 4;         // a (type, K | V, head) job of the backward kernel is cut into HP_SPLIT jobs of D / HP_SPLIT weight-gradient rows
constexpr int HP_ACC = 17;          // outputs per thread of a (type, K | V, head) job of the backward kernel: D (cin + 1) / HP_TPB

struct HgtCompositeArgs {
  const float* const* w;        // [n_types] kqv weight [3F, cin]  (rows: K | Q | V)
  const float* const* b;        // [n_types] kqv bias [3F]
  const float* k_rel;           // [H * R, D, D], index h * R + r
  const float* v_rel;
  const float* const* p;        // [R] p_rel of every edge type, [H] each
  const int* rel_r;             // [n_rel] edge type of the g-th used relation
  const int* rel_src;           // [n_rel] its source type (index into w / b)
  const int* rel_row;           // [n_rel] first row of its K block in big_w (V block: + F)
  const int* type_row;          // [n_types] first row of the type's block (its Wq rows)
  int n_rel, n_types, cin, H, R, F;
};

// grid: n_types (Wq copies) + 2 * H * n_rel (K / V blocks, one head each)
__global__ __launch_bounds__(HP_TPB) void hgt_composite_fwd_kernel(const HgtCompositeArgs A, float* __restrict__ big_w, float* __restrict__ big_b) {
  extern __shared__ float m[];                                    // one head's [D][D] relation matrix
  const int F = A.F, cin = A.cin, D = F / A.H;
  const int job = blockIdx.x;
  if (job < A.n_types) {
    const float* w = A.w[job] + static_cast<int64_t>(F) * cin;    // Q rows
    const float* b = A.b[job] + F;
    float* ow = big_w + static_cast<int64_t>(A.type_row[job]) * cin;
    for (int e = threadIdx.x; e < F * cin; e += HP_TPB) ow[e] = w[e];
    for (int e = threadIdx.x; e < F; e += HP_TPB) big_b[A.type_row[job] + e] = b[e];
    return;
  }
  const int j = job - A.n_types;
  const int g = j / (2 * A.H), kv = (j / A.H) & 1, h = j % A.H, r = A.rel_r[g];
  {
    const float* rel = kv ? A.v_rel : A.k_rel;
    const float sc = kv ? 1.0f : A.p[r][h] / sqrtf(static_cast<float>(D));
    for (int e = threadIdx.x; e < D * D; e += HP_TPB) m[e] = rel[(static_cast<int64_t>(h) * A.R + r) * D * D + e] * sc;      // m[a][b]
  }
  __syncthreads();
  const int src = A.rel_src[g];
  const float* w = A.w[src] + (kv ? 2 : 0) * static_cast<int64_t>(F) * cin + static_cast<int64_t>(h * D) * cin;      // K rows (0) or V rows (2F), head h
  const float* b = A.b[src] + (kv ? 2 : 0) * F + h * D;
  const int row0 = A.rel_row[g] + kv * F + h * D;
  for (int o = threadIdx.x; o < D * (cin + 1); o += HP_TPB) {
    const int bb = o / (cin + 1), c = o - bb * (cin + 1);
    const float* mb = m + bb;                                      // m[a][bb], stride D over a
    float acc = 0.f;
    if (c < cin) {
      const float* wc = w + c;
      for (int a = 0; a < D; ++a) acc += mb[a * D] * wc[static_cast<int64_t>(a) * cin];
      big_w[static_cast<int64_t>(row0 + bb) * cin + c] = acc;
    } else {
      for (int a = 0; a < D; ++a) acc += mb[a * D] * b[a];
      big_b[row0 + bb] = acc;
    }
  }
}

// grads: [type i: dW (3F x cin) | db (3F)] x n_types | dk_rel [H R D D] | dv_rel [H R D D] | dp_rel [R H]   (rel part zeroed by the caller)
// grid: n_types (Q rows: copies) + 2 H n_types (K / V rows of a type's weight gradient, one head each) + 2 H n_rel (relation matrices, one head each)
template <bool STAGED>              // STAGED: a head's dK / dV (and W) rows go through LDS; false (D (cin + 1) tiles beyond 64 KB: one or two heads): read in place
__global__ __launch_bounds__(HP_TPB) void hgt_composite_bwd_kernel(const HgtCompositeArgs A, const float* __restrict__ dbig_w, const float* __restrict__ dbig_b,
                                                                   float* __restrict__ grads) {
  extern __shared__ float m[];                                    // one head's [D][D] relation matrix | HP_TPB partial sums
  const int F = A.F, cin = A.cin, D = F / A.H, H = A.H;
  const int64_t per_type = static_cast<int64_t>(3) * F * cin + 3 * F;
  int job = blockIdx.x;
  if (job < A.n_types) {                                           // Q rows
    const int t = job;
    float* dW = grads + t * per_type + static_cast<int64_t>(F) * cin;
    float* db = grads + t * per_type + static_cast<int64_t>(3) * F * cin + F;
    const float* s = dbig_w + static_cast<int64_t>(A.type_row[t]) * cin;
    for (int e = threadIdx.x; e < F * cin; e += HP_TPB) dW[e] = s[e];
    for (int e = threadIdx.x; e < F; e += HP_TPB) db[e] = dbig_b[A.type_row[t] + e];
    return;
  }
  job -= A.n_types;
  if (job < 2 * H * A.n_types * HP_SPLIT) {
    // dWk_t[(h,a), c] = sum over the relations g that leave t (fixed order) of sum_b M_g[h][a][b] dK_g[(h,b), c]; c = cin: the bias.
    // A job owns D / HP_SPLIT rows a of one head (a type with nine relations was one workgroup's serial loop: the kernel's long pole);
    // the relation's dK rows are staged in LDS once instead of being re-read from global memory for every a.
    const int chunk = job % HP_SPLIT, j2 = job / HP_SPLIT;
    const int t = j2 / (2 * H), kv = (j2 / H) & 1, h = j2 % H;
    float* dW = grads + t * per_type + static_cast<int64_t>(kv ? 2 : 0) * F * cin + static_cast<int64_t>(h * D) * cin;
    float* db = grads + t * per_type + static_cast<int64_t>(3) * F * cin + (kv ? 2 : 0) * F + h * D;
    const int pitch = cin + 1;
    const int rows = D / HP_SPLIT, o_lo = chunk * rows * pitch, n_out = rows * pitch;
    float* const dt = m + D * D;                                   // [D][pitch]: dK_g[(h,b), 0..cin-1] | dbK_g[(h,b)]
    float acc[HP_ACC];
#pragma unroll
    for (int k = 0; k < HP_ACC; ++k) acc[k] = 0.f;
    const float* rel = kv ? A.v_rel : A.k_rel;
    for (int g = 0; g < A.n_rel; ++g) {
      if (A.rel_src[g] != t) continue;
      const int r = A.rel_r[g];
      const float sc = kv ? 1.0f : A.p[r][h] / sqrtf(static_cast<float>(D));
      const int row0 = A.rel_row[g] + kv * F + h * D;
      __syncthreads();
      for (int e = threadIdx.x; e < D * D; e += HP_TPB) m[e] = rel[(static_cast<int64_t>(h) * A.R + r) * D * D + e] * sc;
      if constexpr (STAGED) {
        for (int e = threadIdx.x; e < D * pitch; e += HP_TPB) {
          const int bb = e / pitch, c = e - bb * pitch;
          dt[e] = c < cin ? dbig_w[static_cast<int64_t>(row0 + bb) * cin + c] : dbig_b[row0 + bb];
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < HP_ACC; ++k) {
        const int o = threadIdx.x + k * HP_TPB;
        if (o < n_out) {
          const int a = (o_lo + o) / pitch, c = (o_lo + o) - a * pitch;
          const float* ma = m + a * D;                             // m[a][b]
          float s = 0.f;
          if constexpr (STAGED) {
            for (int bb = 0; bb < D; ++bb) s += ma[bb] * dt[bb * pitch + c];
          } else if (c < cin) {
            const float* dk = dbig_w + static_cast<int64_t>(row0) * cin + c;
            for (int bb = 0; bb < D; ++bb) s += ma[bb] * dk[static_cast<int64_t>(bb) * cin];
          } else {
            for (int bb = 0; bb < D; ++bb) s += ma[bb] * dbig_b[row0 + bb];
          }
          acc[k] += s;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < HP_ACC; ++k) {
      const int o = threadIdx.x + k * HP_TPB;
      if (o < n_out) {
        const int a = (o_lo + o) / pitch, c = (o_lo + o) - a * pitch;
        if (c < cin) dW[static_cast<int64_t>(a) * cin + c] = acc[k];
        else db[a] = acc[k];
      }
    }
    return;
  }
  job -= 2 * H * A.n_types * HP_SPLIT;
  // relation matrices, one head per job: dM[a][b] = sum_c W[(h,a), c] dKV[(h,b), c] + bias[(h,a)] dbKV[(h,b)]
  const int g = job / (2 * H), kv = (job / H) & 1, h = job % H;
  const int r = A.rel_r[g], src = A.rel_src[g];
  const float* w = A.w[src] + (kv ? 2 : 0) * static_cast<int64_t>(F) * cin + static_cast<int64_t>(h * D) * cin;
  const float* b = A.b[src] + (kv ? 2 : 0) * F + h * D;
  const int row0 = A.rel_row[g] + kv * F + h * D;
  const int64_t rel_elems = static_cast<int64_t>(H) * A.R * D * D;
  float* drel = grads + A.n_types * per_type + (kv ? rel_elems : 0) + (static_cast<int64_t>(h) * A.R + r) * D * D;
  float* dp = grads + A.n_types * per_type + 2 * rel_elems;
  const float inv = 1.0f / sqrtf(static_cast<float>(D));
  const float ph = A.p[r][h];
  const float* krel = A.k_rel + (static_cast<int64_t>(h) * A.R + r) * D * D;
  float psum = 0.f;
  // the head's W rows and dK / dV rows staged once, row pitch cin + 1 (a wave reads 32 different dK rows at one c: distinct banks);
  // from global memory every thread walked two 512-byte rows of its own (this job class was most of the kernel's 0.3 ms, and the
  // kernel is the last link of the step's backward chain)
  const int pitch = STAGED ? cin + 1 : cin;
  float* const wt = m;
  float* const dt = m + D * pitch;
  if constexpr (STAGED) {
    for (int e = threadIdx.x; e < D * cin; e += HP_TPB) {
      const int a = e / cin, c = e - a * cin;
      wt[a * pitch + c] = w[static_cast<int64_t>(a) * cin + c];
      dt[a * pitch + c] = dbig_w[static_cast<int64_t>(row0 + a) * cin + c];
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < D * D; e += HP_TPB) {
    const int a = e / D, bb = e - a * D;
    const float* wr = STAGED ? wt + a * pitch : w + static_cast<int64_t>(a) * cin;
    const float* dk = STAGED ? dt + bb * pitch : dbig_w + static_cast<int64_t>(row0 + bb) * cin;
    float s = b[a] * dbig_b[row0 + bb];
    for (int c = 0; c < cin; ++c) s += wr[c] * dk[c];
    if (kv) {
      drel[e] = s;
    } else {
      drel[e] = s * ph * inv;
      psum += s * krel[e] * inv;
    }
  }
  if (!kv) {                                                       // dp_rel[r][h]: the workgroup's partial sums in thread order
    __syncthreads();                                               // (the staged tiles are done with)
    m[threadIdx.x] = psum;
    __syncthreads();
    if (threadIdx.x == 0) {
      float tot = 0.f;
      for (int i = 0; i < HP_TPB; ++i) tot += m[i];
      dp[r * H + h] = tot;
    }
  }
}

}  // namespace

static int hgt_composite_check(int n_rel, int n_types, int cin, int heads, int n_edge_types, int F) {
  MDG_CHECK_ARG(n_rel >= 0 && n_types > 0 && cin > 0 && heads > 0 && n_edge_types > 0 && F > 0 && F % heads == 0 && HP_TPB % heads == 0,
                "mdg_hgt_composite: bad sizes");
  MDG_CHECK_ARG((F / heads) % HP_SPLIT == 0 && static_cast<int64_t>(F / heads / HP_SPLIT) * (cin + 1) <= HP_ACC * HP_TPB,
                "mdg_hgt_composite: (F / heads) * (cin + 1) beyond the backward kernel's register tile");
  return MDG_OK;
}

extern "C" int mdg_hgt_composite_fwd(const void* w_ptrs, const void* b_ptrs, const float* k_rel, const float* v_rel, const void* p_ptrs, const int* rel_r,
                                     const int* rel_src, const int* rel_row, int n_rel, const int* type_row, int n_types, float* big_w, float* big_b,
                                     int cin, int heads, int n_edge_types, int F, void* stream) {
  if (int rc = hgt_composite_check(n_rel, n_types, cin, heads, n_edge_types, F)) return rc;
  MDG_CHECK_ARG(w_ptrs && b_ptrs && k_rel && v_rel && p_ptrs && type_row && big_w && big_b && (n_rel == 0 || (rel_r && rel_src && rel_row)), "mdg_hgt_composite_fwd: null pointer");
  const HgtCompositeArgs A{static_cast<const float* const*>(w_ptrs), static_cast<const float* const*>(b_ptrs), k_rel, v_rel, static_cast<const float* const*>(p_ptrs),
                           rel_r, rel_src, rel_row, type_row, n_rel, n_types, cin, heads, n_edge_types, F};
  const int D = F / heads;
  hipLaunchKernelGGL(hgt_composite_fwd_kernel, dim3(static_cast<unsigned>(n_types + 2 * heads * n_rel)), dim3(HP_TPB), static_cast<size_t>(D) * D * 4,
                     static_cast<hipStream_t>(stream), A, big_w, big_b);
  MDG_CHECK_LAUNCH("mdg_hgt_composite_fwd");
  return MDG_OK;
}

extern "C" int mdg_hgt_composite_bwd(const void* w_ptrs, const void* b_ptrs, const float* k_rel, const float* v_rel, const void* p_ptrs, const int* rel_r,
                                     const int* rel_src, const int* rel_row, int n_rel, const int* type_row, int n_types, const float* dbig_w,
                                     const float* dbig_b, float* grads, int cin, int heads, int n_edge_types, int F, void* stream) {
  if (int rc = hgt_composite_check(n_rel, n_types, cin, heads, n_edge_types, F)) return rc;
  MDG_CHECK_ARG(w_ptrs && b_ptrs && k_rel && v_rel && p_ptrs && type_row && dbig_w && dbig_b && grads && (n_rel == 0 || (rel_r && rel_src && rel_row)),
                "mdg_hgt_composite_bwd: null pointer");
  const HgtCompositeArgs A{static_cast<const float* const*>(w_ptrs), static_cast<const float* const*>(b_ptrs), k_rel, v_rel, static_cast<const float* const*>(p_ptrs),
                           rel_r, rel_src, rel_row, type_row, n_rel, n_types, cin, heads, n_edge_types, F};
  const int D = F / heads;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t per_type = static_cast<int64_t>(3) * F * cin + 3 * F, rel_elems = static_cast<int64_t>(heads) * n_edge_types * D * D;
  (void)hipMemsetAsync(grads + n_types * per_type, 0, static_cast<size_t>(2 * rel_elems + static_cast<int64_t>(n_edge_types) * heads) * 4, st);   // unused relations: zero
  size_t lds = static_cast<size_t>(D) * D * 4 > HP_TPB * 4 ? static_cast<size_t>(D) * D * 4 : HP_TPB * 4;
  const size_t tiles = static_cast<size_t>(2) * D * (cin + 1) * 4;        // W rows | dK / dV rows of one head
  MDG_CHECK_ARG(D % HP_SPLIT == 0, "mdg_hgt_composite_bwd: F / heads must be a multiple of %d", HP_SPLIT);
  const size_t lds2 = (static_cast<size_t>(D) * D + static_cast<size_t>(D) * (cin + 1)) * 4;      // relation matrix | staged dK rows
  const bool staged = tiles <= 64 * 1024 && lds2 <= 64 * 1024;
  if (staged) lds = lds > tiles ? (lds > lds2 ? lds : lds2) : (tiles > lds2 ? tiles : lds2);
  const dim3 grid(static_cast<unsigned>(n_types + 2 * heads * n_types * HP_SPLIT + 2 * heads * n_rel));
  if (staged) hipLaunchKernelGGL(hgt_composite_bwd_kernel<true>, grid, dim3(HP_TPB), lds, st, A, dbig_w, dbig_b, grads);
  else hipLaunchKernelGGL(hgt_composite_bwd_kernel<false>, grid, dim3(HP_TPB), lds, st, A, dbig_w, dbig_b, grads);
  MDG_CHECK_LAUNCH("mdg_hgt_composite_bwd");
  return MDG_OK;
}
