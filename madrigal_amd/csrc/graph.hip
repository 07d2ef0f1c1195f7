// GNN message-passing aggregations for gfx950 (structure GIN and knowledge-graph HGT encoders).
//
// Both are gather-bound (HBM / L2): per edge one F*4-byte neighbour row is read, per destination one
// row is written.  Edges are pre-sorted by destination (CSR, built once per batch/graph by the host
// glue), so every destination is owned by one lane group: no atomics, and the summation order is the
// fixed CSR order => bitwise reproducible results.  Reductions inside a row use wavefront shuffles.
#include "mdg_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// out[v] = self_coef * x_self[v] + sum_{e in [rowptr[v], rowptr[v+1])} w[e] * x[col[e]]      (mean: / count)
//   GIN:      x = h, col = source atom of edge e, self_coef = 1 + eps  (torchdrug GraphIsomorphismConv;
//             call site madrigal/models/models.py:720) and, with F=20, the per-destination sum of bond
//             features that multiplies edge_linear once per layer (linearity of edge_linear);
//   read-out: col == null => contiguous rows [rowptr[v], rowptr[v+1]) of x (node2graph is sorted), mean.
// LPR lanes cooperate on one destination row (float4 per lane), 64/LPR rows per wave.
// ---------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void csr_aggregate_kernel(const float* __restrict__ x, int64_t ldx,
                                                            const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col,
                                                            const float* __restrict__ w, const float* __restrict__ xself,
                                                            int64_t ldself, const float* __restrict__ self_coef_dev,
                                                            float self_coef_add, int mean, float* __restrict__ out, int64_t ldo,
                                                            int64_t n_dst, int F) {
  constexpr int RPB = 256 / LPR;
  const int sub = threadIdx.x % LPR;
  const int64_t v = static_cast<int64_t>(blockIdx.x) * RPB + threadIdx.x / LPR;
  if (v >= n_dst) return;
  const int c = 4 * sub;
  const bool act = c < F;
  const int64_t e0 = rowptr[v], e1 = rowptr[v + 1];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // groups of 4 edges with 4 independent gathers in flight; the last (for a molecule's atoms: the only) group is PREDICATED instead
  // of walked edge by edge: at degree ~2 the edge-by-edge tail was a chain of col -> row -> col -> row dependent loads per atom
  // (round 3: 1.4 TB/s on the GIN aggregation); same summation order, absent edges add 0 * 0
  for (int64_t e = e0; e < e1; e += 4) {
    int64_t s[4];
    float ww[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ok[u] = e + u < e1;
      s[u] = ok[u] ? (col ? col[e + u] : e + u) : 0;
      ww[u] = ok[u] ? (w ? w[e + u] : 1.f) : 0.f;
    }
    f32x4 r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) r[u] = (act && ok[u]) ? *reinterpret_cast<const f32x4*>(x + s[u] * ldx + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += ww[u] * r[u];
  }
  if (mean) {
    const float cnt = static_cast<float>(e1 - e0);
    acc = acc / fmaxf(cnt, 1.f);
  }
  if (xself && act) {
    const float coef = self_coef_add + (self_coef_dev ? self_coef_dev[0] : 0.f);
    acc += coef * *reinterpret_cast<const f32x4*>(xself + v * ldself + c);
  }
  if (act) *reinterpret_cast<f32x4*>(out + v * ldo + c) = acc;
}

// ---------------------------------------------------------------------------------------------
// HGT edge attention (PyG HGTConv message/aggregate; call site madrigal/models/models.py:76-79,
// 90-94).  For destination node i with query q_i [H,D] and incoming edges e -> (row c_e of the
// relation-transformed source table kv = [k' | v'], F = H*D = 128 floats each):
//     a_e[h] = q_i[h] . k'_{c_e}[h]          (p_rel / sqrt(D) is folded into k' by the host glue)
//     alpha  = softmax over ALL incoming edges of i (every edge type), per head
//     out_i  = gelu( sum_e alpha_e[h] v'_{c_e}[h] )
// Work item = (destination, chunk of <= CHUNK edges): heavy-tailed KG degrees are split so that no
// wave owns more than CHUNK edges; each item produces an online-softmax partial (m, l, acc) and a
// second kernel merges the partials of a destination in item order (deterministic).
// A half-wave (32 lanes x float4 = 128 floats) handles one edge; the two halves of a wave walk
// alternate edges with 4 edges each in flight, and are merged at the end.
// ---------------------------------------------------------------------------------------------
// 16-bit mirror of the projection buffer (the step's "bf16" mode): the gathered k' | v' rows are what the kernels move -- 1 KB per
// edge in fp32, half that as bf16 (the values a bf16 GEMM would have handed over anyway; q, the statistics, every sum and every
// gradient stay fp32).  R16 kernels read row r of the mirror at element offset r * 128.
template <bool R16>
__device__ __forceinline__ f32x4 load_row4(const float* __restrict__ f32rows, const uint16_t* __restrict__ b16rows, int64_t elem) {
  if constexpr (R16) {
    const uint2 u = *reinterpret_cast<const uint2*>(b16rows + elem);
    return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u)};
  } else {
    return *reinterpret_cast<const f32x4*>(f32rows + elem);
  }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int64_t n8) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n8) return;
  const f32x4 a = reinterpret_cast<const f32x4*>(x)[2 * i], b = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
  auto rne = [](float v) -> uint32_t {                   // round to nearest even (NaN keeps a quiet payload)
    const uint32_t u = __float_as_uint(v);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
  };
  uint4 o;
  o.x = rne(a[0]) | (rne(a[1]) << 16); o.y = rne(a[2]) | (rne(a[3]) << 16);
  o.z = rne(b[0]) | (rne(b[1]) << 16); o.w = rne(b[2]) | (rne(b[3]) << 16);
  reinterpret_cast<uint4*>(y)[i] = o;
}

struct HgtArgs {
  const float* q; int64_t ldq;          // [n_dst, >=128]
  const float* kv; int64_t ldkv;        // row col[e]: k' at +0 and v' at +128 floats (ldkv 128: v' is the next row)
  const int64_t* col;                   // [nnz] row of kv per edge, sorted by destination
  const int64_t* item_dst; const int64_t* item_begin; const int64_t* item_end;   // [n_items]
  float* part_acc;                      // [n_items,128]
  float* part_ml;                       // [n_items,H,2]
  int64_t n_items;
  int H;
  const int64_t* q_off;                 // optional: float offset of each destination's query row from q (destinations of several
                                        // node types, whose rows differ in width, in ONE launch); null: row dst at q + dst * ldq
  // a destination with ONE work item (<= CHUNK edges: nearly all of them) is finished here - normalised, activated, its softmax
  // statistics stored - instead of going through a partial that hgt_combine_kernel would only copy
  const int64_t* item_ptr; float* out; int64_t ldo; float* stats; int apply_gelu;
  const uint16_t* kv16;                 // R16 kernels: the bf16 mirror of kv (same row numbering, 128 elements per row)
};

__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b) { return (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]); }

template <bool R16>
__global__ __launch_bounds__(256) void hgt_attention_kernel(const HgtArgs p) {
  const int lane = threadIdx.x & 63, sub = lane & 31, half = lane >> 5;
  const int64_t item = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (item >= p.n_items) return;
  const int lph = 32 / p.H;                       // lanes per head
  const int64_t dst = p.item_dst[item], e0 = p.item_begin[item], e1 = p.item_end[item];
  const f32x4 q = *reinterpret_cast<const f32x4*>(p.q + (p.q_off ? p.q_off[dst] : dst * p.ldq) + 4 * sub);
  float m = -INFINITY, l = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // this half takes edges e0+half, e0+half+2, ...; FOUR of them per iteration (8 rows = 8 KB of k'|v' in flight per wave: every
  // step is a dependent col -> row gather, so the rows in flight per wave are what hide its latency)
  for (int64_t e = e0 + half; e < e1; e += 8) {
    bool ok[4];
    int64_t r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ok[u] = (e + 2 * u) < e1;
      r[u] = p.col[ok[u] ? e + 2 * u : e] * (R16 ? 128 : p.ldkv) + 4 * sub;
    }
    f32x4 k[4], v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      k[u] = load_row4<R16>(p.kv, p.kv16, r[u]);
      v[u] = load_row4<R16>(p.kv, p.kv16, r[u] + 128);
    }
    float a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = dot4(q, k[u]);
    for (int o = lph >> 1; o > 0; o >>= 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] += __shfl_xor(a[u], o, 64);
    }
#pragma unroll
    for (int u = 1; u < 4; ++u)
      if (!ok[u]) a[u] = -INFINITY;
    const float mn = fmaxf(fmaxf(m, a[0]), fmaxf(fmaxf(a[1], a[2]), a[3]));
    const float f = expf(m - mn);
    float w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = expf(a[u] - mn);
    l = l * f + ((w[0] + w[1]) + (w[2] + w[3]));
    acc = acc * f + ((w[0] * v[0] + w[1] * v[1]) + (w[2] * v[2] + w[3] * v[3]));
    m = mn;
  }
  // merge the two halves (each lane pairs with lane ^ 32, same feature columns)
  {
    const float mo = __shfl_xor(m, 32, 64), lo = __shfl_xor(l, 32, 64);
    f32x4 ao;
#pragma unroll
    for (int c = 0; c < 4; ++c) ao[c] = __shfl_xor(acc[c], 32, 64);
    const float mn = fmaxf(m, mo);
    const float f0 = (m == -INFINITY) ? 0.f : expf(m - mn), f1 = (mo == -INFINITY) ? 0.f : expf(mo - mn);
    // fixed order (half 0 first) so that both halves compute bit-identical sums
    const float la = half ? lo : l, lb = half ? l : lo;
    const float fa = half ? f1 : f0, fb = half ? f0 : f1;
    const f32x4 xa = half ? ao : acc, xb = half ? acc : ao;
    l = la * fa + lb * fb;
    acc = xa * fa + xb * fb;
    m = mn;
  }
  if (half == 0) {
    const int h = sub / lph;
    if (p.item_ptr[dst + 1] - p.item_ptr[dst] == 1) {          // the same arithmetic as hgt_combine_kernel over one partial
      if (m == -INFINITY) l = 0.f;
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (l > 0.f) o = acc / (l + 1e-16f);
      if (p.stats && sub % lph == 0) {
        p.stats[(dst * p.H + h) * 2 + 0] = m;
        p.stats[(dst * p.H + h) * 2 + 1] = l + 1e-16f;
      }
      if (p.apply_gelu) {
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = mdg_gelu(o[c]);
      }
      *reinterpret_cast<f32x4*>(p.out + dst * p.ldo + 4 * sub) = o;
      return;
    }
    *reinterpret_cast<f32x4*>(p.part_acc + item * 128 + 4 * sub) = acc;
    if (sub % lph == 0) {
      p.part_ml[(item * p.H + h) * 2 + 0] = m;
      p.part_ml[(item * p.H + h) * 2 + 1] = l;
    }
  }
}

// Eight destinations per workgroup, a half-wave each.  A hub (> HGT_COOP items: a drug with 3.6e4 in-edges has 286) is merged by ALL
// eight half-waves -- half-wave g takes items g, g + 8, ... in order, the eight partial states are merged in order g = 0..7 -- instead
// of one half-wave walking its few hundred dependent (m, l, acc) updates while the launch waits for it (the launch's duration WAS its
// largest hub: 100 us in the inference encode's critical chain).  Fixed orders: deterministic; the branch is uniform per workgroup.
constexpr int HGT_COOP = 16;

__device__ __forceinline__ void hgt_merge(float& m, float& l, f32x4& acc, float mi, float li, const f32x4& ai) {
  if (mi == -INFINITY) return;
  const float mn = fmaxf(m, mi);
  const float f = (m == -INFINITY) ? 0.f : expf(m - mn), g = expf(mi - mn);
  l = l * f + li * g;
  acc = acc * f + ai * g;
  m = mn;
}

__global__ __launch_bounds__(256) void hgt_combine_kernel(const float* __restrict__ part_acc, const float* __restrict__ part_ml,
                                                          const int64_t* __restrict__ item_ptr, float* __restrict__ out, int64_t ldo,
                                                          int64_t n_dst, int H, int apply_gelu, float* __restrict__ stats) {
  __shared__ float sh_acc[8][128];
  __shared__ float sh_ml[8][32][2];
  const int sub = threadIdx.x & 31, hw = threadIdx.x >> 5;
  const int h = sub / (32 / H);
  const int64_t v0 = static_cast<int64_t>(blockIdx.x) * 8;
  for (int d = 0; d < 8; ++d) {
    const int64_t v = v0 + d;
    if (v >= n_dst) break;                         // (uniform)
    const int64_t i0 = item_ptr[v], i1 = item_ptr[v + 1];
    const int64_t cnt = i1 - i0;
    if (cnt == 1) continue;                        // finished by hgt_attention_kernel itself
    const bool coop = cnt > HGT_COOP;
    if (!coop && hw != d) continue;
    float m = -INFINITY, l = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // four partials in flight per step (each merge is a dependent exp / multiply chain: loads issued one merge at a time left the
    // walk at a memory latency per item), merged in item order
    const int64_t step = coop ? 8 : 1;
    for (int64_t it = i0 + (coop ? hw : 0); it < i1; it += 4 * step) {
      float mi[4], li[4];
      f32x4 ai[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t iu = it + u * step;
        const bool ok = iu < i1;
        const int64_t ir = ok ? iu : it;
        mi[u] = ok ? part_ml[(ir * H + h) * 2] : -INFINITY;
        li[u] = part_ml[(ir * H + h) * 2 + 1];
        ai[u] = *reinterpret_cast<const f32x4*>(part_acc + ir * 128 + 4 * sub);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) hgt_merge(m, l, acc, mi[u], li[u], ai[u]);
    }
    if (coop) {
      *reinterpret_cast<f32x4*>(&sh_acc[hw][4 * sub]) = acc;
      sh_ml[hw][sub][0] = m;
      sh_ml[hw][sub][1] = l;
      __syncthreads();
      if (hw == 0) {
        m = -INFINITY; l = 0.f; acc = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < 8; ++g) hgt_merge(m, l, acc, sh_ml[g][sub][0], sh_ml[g][sub][1], *reinterpret_cast<const f32x4*>(&sh_acc[g][4 * sub]));
      }
    }
    if (!coop || hw == 0) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (l > 0.f) o = acc / (l + 1e-16f);          // torch_geometric.utils.softmax: exp / (sum + 1e-16)
      if (stats && sub % (32 / H) == 0) {            // kept for the backward pass: alpha_e = exp(a_e - m) / denom
        stats[(v * H + h) * 2 + 0] = m;
        stats[(v * H + h) * 2 + 1] = l + 1e-16f;
      }
      if (apply_gelu) {
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = mdg_gelu(o[c]);
      }
      *reinterpret_cast<f32x4*>(out + v * ldo + 4 * sub) = o;
    }
    if (coop) __syncthreads();                     // the shared partials are free again
  }
}

// ---------------------------------------------------------------------------------------------
// Backward of the HGT edge attention.  With g = d out_i (before the GELU), pre_i = sum_e alpha_e v'_e (kept from the
// forward pass) and delta[h] = g[h] . pre_i[h]  (= sum_e alpha_e dalpha_e):
//     dalpha_e[h] = g[h] . v'_e[h],   da_e[h] = alpha_e[h] (dalpha_e[h] - delta[h])
//     dq_i[h] = sum_e da_e[h] k'_e[h]           (destination-centric: same work items as the forward pass)
//     dk'_r[h] = sum_{e: c_e = r} da_e[h] q_{i(e)}[h],   dv'_r[h] = sum_{e: c_e = r} alpha_e[h] g_{i(e)}[h]
// The second pair is a sum over the edges LEAVING a key row: the edge kernel stores (alpha_e, da_e) per edge and head
// and a second, source-centric kernel walks the reversed edge lists (work items of <= CHUNK edges, partials merged in
// item order).  No atomics: bit-reproducible gradients.
// ---------------------------------------------------------------------------------------------
struct HgtBwdArgs {
  const float* q; int64_t ldq;
  const float* kv; int64_t ldkv;
  const int64_t* col;
  const int64_t* item_dst; const int64_t* item_begin; const int64_t* item_end;
  int64_t n_items;
  const float* g; int64_t ldg;          // [n_dst,128] gradient at the attention output (pre-activation)
  const float* pre; int64_t ldp;        // [n_dst,128] forward output before the activation
  const float* stats;                   // [n_dst,H,2] (max, denominator)
  float* edge_alpha; float* edge_da;    // one array [nnz][alpha: H | da: H] (edge_da = edge_alpha + H): an edge's pair shares a cache line,
                                        // the source-side kernel's gather through the edge id touches one line per edge instead of two
  float* part_dq;                       // [n_items,128]
  int H;
  const int64_t* item_ptr; float* dq; int64_t lddq;     // a destination with one work item writes its dq row itself
  const uint16_t* kv16;
};

template <bool R16>
__global__ __launch_bounds__(256) void hgt_attention_bwd_edge_kernel(const HgtBwdArgs p) {
  const int lane = threadIdx.x & 63, sub = lane & 31, half = lane >> 5;
  const int64_t item = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (item >= p.n_items) return;
  const int lph = 32 / p.H;
  const int h = sub / lph;
  const int64_t dst = p.item_dst[item], e0 = p.item_begin[item], e1 = p.item_end[item];
  const f32x4 q = *reinterpret_cast<const f32x4*>(p.q + dst * p.ldq + 4 * sub);
  const f32x4 g = *reinterpret_cast<const f32x4*>(p.g + dst * p.ldg + 4 * sub);
  const f32x4 pre = *reinterpret_cast<const f32x4*>(p.pre + dst * p.ldp + 4 * sub);
  float delta = dot4(g, pre);
  for (int o = lph >> 1; o > 0; o >>= 1) delta += __shfl_xor(delta, o, 64);
  const float m = p.stats[(dst * p.H + h) * 2], inv = 1.0f / p.stats[(dst * p.H + h) * 2 + 1];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // four edges of this half in flight (8 rows of k' | v'), as in the forward kernel; the sum keeps the edge order
  for (int64_t e = e0 + half; e < e1; e += 8) {
    bool ok[4];
    int64_t r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ok[u] = (e + 2 * u) < e1;
      r[u] = p.col[ok[u] ? e + 2 * u : e] * (R16 ? 128 : p.ldkv) + 4 * sub;
    }
    f32x4 k[4], v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      k[u] = load_row4<R16>(p.kv, p.kv16, r[u]);
      v[u] = load_row4<R16>(p.kv, p.kv16, r[u] + 128);
    }
    float a[4], da[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = dot4(q, k[u]);
      da[u] = dot4(g, v[u]);
    }
    for (int o = lph >> 1; o > 0; o >>= 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] += __shfl_xor(a[u], o, 64);
        da[u] += __shfl_xor(da[u], o, 64);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float alpha = expf(a[u] - m) * inv;
      const float ds = alpha * (da[u] - delta);
      if (ok[u]) {
        acc += ds * k[u];
        if (sub % lph == 0) {
          p.edge_alpha[(e + 2 * u) * 2 * p.H + h] = alpha;
          p.edge_da[(e + 2 * u) * 2 * p.H + h] = ds;
        }
      }
    }
  }
  f32x4 other;
#pragma unroll
  for (int c = 0; c < 4; ++c) other[c] = __shfl_xor(acc[c], 32, 64);
  if (half == 0) {
    const bool single = p.item_ptr[dst + 1] - p.item_ptr[dst] == 1;
    *reinterpret_cast<f32x4*>((single ? p.dq + dst * p.lddq : p.part_dq + item * 128) + 4 * sub) = acc + other;
  }
}

// out[v, 0:width] = sum of part[item, 0:width] over the items of row v, in item order; rows go to out + row_index[v]*ldo
// (row_index null = v).  width = 128 (dq) or 256 (dk' | dv' = two consecutive kv rows).  skip_single: rows with exactly one
// item were written by the kernel that produced the partials.
// A row with more than HGT_COOP items (a hub) is summed by all the workgroup's row groups -- group g takes items g, g + G, ... in
// order, the G partial sums are added in order g = 0..G-1 -- as in hgt_combine_kernel.
__global__ __launch_bounds__(256) void hgt_sum_items_kernel(const float* __restrict__ part, const int64_t* __restrict__ item_ptr,
                                                            const int64_t* __restrict__ row_index, float* __restrict__ out, int64_t ldo,
                                                            int64_t n_rows, int width, int skip_single) {
  __shared__ float sh[8][256];                    // [group][row floats] (G * width = 1024 floats)
  const int lpr = width / 4;                      // lanes per row: 32 or 64
  const int G = 256 / lpr;                        // rows (= row groups) per workgroup: 8 or 4
  const int sub = threadIdx.x % lpr, grp = threadIdx.x / lpr;
  const int64_t v0 = static_cast<int64_t>(blockIdx.x) * G;
  for (int d = 0; d < G; ++d) {
    const int64_t v = v0 + d;
    if (v >= n_rows) break;                        // (uniform)
    const int64_t i0 = item_ptr[v], i1 = item_ptr[v + 1];
    const int64_t cnt = i1 - i0;
    if (skip_single && cnt == 1) continue;         // written by the producing kernel
    const bool coop = cnt > HGT_COOP;
    if (!coop && grp != d) continue;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int64_t step = coop ? G : 1;
    for (int64_t it = i0 + (coop ? grp : 0); it < i1; it += 4 * step) {               // four rows in flight, added in item order
      f32x4 r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t iu = it + u * step;
        r[u] = *reinterpret_cast<const f32x4*>(part + (iu < i1 ? iu : it) * width + 4 * sub);
        if (iu >= i1) r[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += r[u];
    }
    if (coop) {
      *reinterpret_cast<f32x4*>(&sh[grp][4 * sub]) = acc;
      __syncthreads();
      if (grp == 0) {
        acc = *reinterpret_cast<const f32x4*>(&sh[0][4 * sub]);
        for (int g = 1; g < G; ++g) acc += *reinterpret_cast<const f32x4*>(&sh[g][4 * sub]);
      }
    }
    if (!coop || grp == 0) {
      const int64_t row = row_index ? row_index[v] : v;
      *reinterpret_cast<f32x4*>(out + row * ldo + 4 * sub) = acc;
    }
    if (coop) __syncthreads();
  }
}

struct HgtSrcArgs {
  const float* q; int64_t ldq;
  const float* g; int64_t ldg;
  const int64_t* t_edge; const int64_t* t_dst;     // reversed edge list (sorted by key row): forward edge id, destination
  const int64_t* item_begin; const int64_t* item_end;
  int64_t n_items;
  const float* edge_alpha; const float* edge_da;
  float* part;                                     // [n_items,256]: dk' | dv'
  int H;
  // optional (item_row non-null): a key row with one work item writes dk' | dv' itself
  const int64_t* item_row; const int64_t* item_ptr; const int64_t* rows; float* dkv; int64_t lddkv;
};

__global__ __launch_bounds__(256) void hgt_attention_bwd_src_kernel(const HgtSrcArgs p) {
  const int lane = threadIdx.x & 63, sub = lane & 31, half = lane >> 5;
  const int64_t item = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (item >= p.n_items) return;
  const int h = sub / (32 / p.H);
  f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
  const int64_t e1 = p.item_end[item];
  // four reversed edges of this half in flight: each is an (edge id, destination) -> (alpha, da), q row, g row chain
  for (int64_t e = p.item_begin[item] + half; e < e1; e += 8) {
    bool in[4];
    int64_t eid[4], d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      in[u] = (e + 2 * u) < e1;
      const int64_t eu = in[u] ? e + 2 * u : e;
      eid[u] = p.t_edge[eu];
      d[u] = p.t_dst[eu];
    }
    float da[4], al[4];
    f32x4 qr[4], gr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      da[u] = p.edge_da[eid[u] * 2 * p.H + h];
      al[u] = p.edge_alpha[eid[u] * 2 * p.H + h];
      qr[u] = *reinterpret_cast<const f32x4*>(p.q + d[u] * p.ldq + 4 * sub);
      gr[u] = *reinterpret_cast<const f32x4*>(p.g + d[u] * p.ldg + 4 * sub);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (in[u]) {
        ak += da[u] * qr[u];
        av += al[u] * gr[u];
      }
    }
  }
  f32x4 ok, ov;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    ok[c] = __shfl_xor(ak[c], 32, 64);
    ov[c] = __shfl_xor(av[c], 32, 64);
  }
  if (half == 0) {
    float* o = p.part + item * 256;
    if (p.item_row) {
      const int64_t v = p.item_row[item];
      if (p.item_ptr[v + 1] - p.item_ptr[v] == 1) o = p.dkv + p.rows[v] * p.lddkv;
    }
    *reinterpret_cast<f32x4*>(o + 4 * sub) = ak + ok;
    *reinterpret_cast<f32x4*>(o + 128 + 4 * sub) = av + ov;
  }
}

template <int LPR>
void launch_csr(const float* x, int64_t ldx, const int64_t* rowptr, const int64_t* col, const float* w, const float* xself,
                int64_t ldself, const float* coef_dev, float coef_add, int mean, float* out, int64_t ldo, int64_t n_dst, int F,
                hipStream_t st) {
  const int rpb = 256 / LPR;
  hipLaunchKernelGGL(csr_aggregate_kernel<LPR>, dim3(static_cast<unsigned>(mdg_cdiv(n_dst, rpb))), dim3(256), 0, st, x, ldx, rowptr,
                     col, w, xself, ldself, coef_dev, coef_add, mean, out, ldo, n_dst, F);
}

}  // namespace

extern "C" int mdg_csr_aggregate(const float* x, int64_t ldx, const int64_t* rowptr, const int64_t* col, const float* edge_weight,
                                 const float* x_self, int64_t ld_self, const float* self_coef_dev, float self_coef_add, int mean,
                                 float* out, int64_t ldo, int64_t n_dst, int64_t F, void* stream) {
  MDG_CHECK_ARG(n_dst >= 0 && F > 0 && F % 4 == 0 && F <= 256, "mdg_csr_aggregate: F must be a multiple of 4 and <= 256 (got %lld)", (long long)F);
  if (n_dst == 0) return MDG_OK;
  MDG_CHECK_ARG(x && rowptr && out, "mdg_csr_aggregate: null pointer");
  MDG_CHECK_ARG(ldx % 4 == 0 && ldo % 4 == 0 && ldx >= F && ldo >= F && (!x_self || (ld_self % 4 == 0 && ld_self >= F)),
                "mdg_csr_aggregate: row strides must be multiples of 4 and >= F");
  MDG_CHECK_ARG(mdg_aligned16(x) && mdg_aligned16(out) && (!x_self || mdg_aligned16(x_self)), "mdg_csr_aggregate: 16-byte alignment");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int Fi = static_cast<int>(F);
  if (F <= 32) launch_csr<8>(x, ldx, rowptr, col, edge_weight, x_self, ld_self, self_coef_dev, self_coef_add, mean, out, ldo, n_dst, Fi, st);
  else if (F <= 64) launch_csr<16>(x, ldx, rowptr, col, edge_weight, x_self, ld_self, self_coef_dev, self_coef_add, mean, out, ldo, n_dst, Fi, st);
  else if (F <= 128) launch_csr<32>(x, ldx, rowptr, col, edge_weight, x_self, ld_self, self_coef_dev, self_coef_add, mean, out, ldo, n_dst, Fi, st);
  else launch_csr<64>(x, ldx, rowptr, col, edge_weight, x_self, ld_self, self_coef_dev, self_coef_add, mean, out, ldo, n_dst, Fi, st);
  MDG_CHECK_LAUNCH("mdg_csr_aggregate");
  return MDG_OK;
}

extern "C" int mdg_f32_to_bf16(const float* x, void* y, int64_t n, void* stream) {
  MDG_CHECK_ARG(n >= 0 && n % 8 == 0, "mdg_f32_to_bf16: the element count must be a multiple of 8 (got %lld)", (long long)n);
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(x && y && mdg_aligned16(x) && mdg_aligned16(y), "mdg_f32_to_bf16: 16-byte aligned pointers");
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n / 8, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     static_cast<uint16_t*>(y), n / 8);
  MDG_CHECK_LAUNCH("mdg_f32_to_bf16");
  return MDG_OK;
}

extern "C" size_t mdg_hgt_attention_workspace_bytes(int64_t n_items, int heads) {
  if (n_items <= 0) return 0;
  return static_cast<size_t>(n_items) * (128 + 2 * static_cast<size_t>(heads)) * sizeof(float);
}

extern "C" int mdg_hgt_attention_stats(const float* q, int64_t ldq, const float* kv, int64_t ldkv, const int64_t* col,
                                       const int64_t* item_dst, const int64_t* item_begin, const int64_t* item_end, int64_t n_items,
                                       const int64_t* item_ptr, float* out, int64_t ldo, int64_t n_dst, int heads, int64_t F,
                                       int apply_gelu, float* stats, const void* kv16, void* workspace, size_t workspace_bytes, void* stream) {
  MDG_CHECK_ARG(F == 128, "mdg_hgt_attention: hidden size must be 128 (got %lld)", (long long)F);
  MDG_CHECK_ARG(!kv16 || (ldkv == 128 && (reinterpret_cast<uintptr_t>(kv16) & 7u) == 0), "mdg_hgt_attention: the 16-bit mirror needs ldkv == 128 and 8-byte alignment");
  MDG_CHECK_ARG(heads == 1 || heads == 2 || heads == 4 || heads == 8, "mdg_hgt_attention: heads must be 1, 2, 4 or 8 (got %d)", heads);
  MDG_CHECK_ARG(n_dst >= 0 && n_items >= 0, "mdg_hgt_attention: negative size");
  if (n_dst == 0) return MDG_OK;
  MDG_CHECK_ARG(q && out && item_ptr && (n_items == 0 || (kv && col && item_dst && item_begin && item_end)), "mdg_hgt_attention: null pointer");
  MDG_CHECK_ARG(ldq % 4 == 0 && ldkv % 4 == 0 && ldo % 4 == 0 && ldq >= 128 && ldkv >= 128 && ldo >= 128, "mdg_hgt_attention: bad strides");
  MDG_CHECK_ARG(mdg_aligned16(q) && mdg_aligned16(out) && (!kv || mdg_aligned16(kv)), "mdg_hgt_attention: 16-byte alignment");
  const size_t need = mdg_hgt_attention_workspace_bytes(n_items, heads);
  if (need && (!workspace || workspace_bytes < need || !mdg_aligned16(workspace))) {
    mdg_set_error("mdg_hgt_attention: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return MDG_EWORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part_acc = static_cast<float*>(workspace);
  float* part_ml = part_acc ? part_acc + n_items * 128 : nullptr;
  if (n_items > 0) {
    HgtArgs a{q, ldq, kv, ldkv, col, item_dst, item_begin, item_end, part_acc, part_ml, n_items, heads, nullptr, item_ptr, out, ldo, stats, apply_gelu,
              static_cast<const uint16_t*>(kv16)};
    if (kv16) hipLaunchKernelGGL(hgt_attention_kernel<true>, dim3(static_cast<unsigned>(mdg_cdiv(n_items, 4))), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(hgt_attention_kernel<false>, dim3(static_cast<unsigned>(mdg_cdiv(n_items, 4))), dim3(256), 0, st, a);
  }
  hipLaunchKernelGGL(hgt_combine_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n_dst, 8))), dim3(256), 0, st, part_acc, part_ml, item_ptr,
                     out, ldo, n_dst, heads, apply_gelu, stats);
  MDG_CHECK_LAUNCH("mdg_hgt_attention");
  return MDG_OK;
}

extern "C" int mdg_hgt_attention(const float* q, int64_t ldq, const float* kv, int64_t ldkv, const int64_t* col,
                                 const int64_t* item_dst, const int64_t* item_begin, const int64_t* item_end, int64_t n_items,
                                 const int64_t* item_ptr, float* out, int64_t ldo, int64_t n_dst, int heads, int64_t F,
                                 int apply_gelu, void* workspace, size_t workspace_bytes, void* stream) {
  return mdg_hgt_attention_stats(q, ldq, kv, ldkv, col, item_dst, item_begin, item_end, n_items, item_ptr, out, ldo, n_dst, heads, F,
                                 apply_gelu, nullptr, nullptr, workspace, workspace_bytes, stream);
}

// All destination types of a conv in one launch: destinations numbered across the types, query row of destination d at
// q_base + q_off[d] (the types' projection rows differ in width), outputs in one [n_dst,128] buffer.  Inference (no stats).
extern "C" int mdg_hgt_attention_rows(const float* q_base, const int64_t* q_off, const float* kv, int64_t ldkv, const int64_t* col,
                                      const int64_t* item_dst, const int64_t* item_begin, const int64_t* item_end, int64_t n_items,
                                      const int64_t* item_ptr, float* out, int64_t ldo, int64_t n_dst, int heads, int apply_gelu,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  MDG_CHECK_ARG(heads == 1 || heads == 2 || heads == 4 || heads == 8, "mdg_hgt_attention_rows: heads must be 1, 2, 4 or 8 (got %d)", heads);
  MDG_CHECK_ARG(n_dst >= 0 && n_items >= 0, "mdg_hgt_attention_rows: negative size");
  if (n_dst == 0) return MDG_OK;
  MDG_CHECK_ARG(q_base && q_off && out && item_ptr && (n_items == 0 || (kv && col && item_dst && item_begin && item_end)), "mdg_hgt_attention_rows: null pointer");
  MDG_CHECK_ARG(ldkv % 4 == 0 && ldo % 4 == 0 && ldkv >= 128 && ldo >= 128 && mdg_aligned16(q_base) && mdg_aligned16(out) && (!kv || mdg_aligned16(kv)),
                "mdg_hgt_attention_rows: bad strides / alignment (query offsets must be multiples of 4 floats)");
  const size_t need = mdg_hgt_attention_workspace_bytes(n_items, heads);
  if (need && (!workspace || workspace_bytes < need || !mdg_aligned16(workspace))) {
    mdg_set_error("mdg_hgt_attention_rows: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return MDG_EWORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part_acc = static_cast<float*>(workspace);
  float* part_ml = part_acc ? part_acc + n_items * 128 : nullptr;
  if (n_items > 0) {
    HgtArgs a{q_base, 0, kv, ldkv, col, item_dst, item_begin, item_end, part_acc, part_ml, n_items, heads, q_off, item_ptr, out, ldo, nullptr, apply_gelu, nullptr};
    hipLaunchKernelGGL(hgt_attention_kernel<false>, dim3(static_cast<unsigned>(mdg_cdiv(n_items, 4))), dim3(256), 0, st, a);
  }
  hipLaunchKernelGGL(hgt_combine_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n_dst, 8))), dim3(256), 0, st, part_acc, part_ml, item_ptr,
                     out, ldo, n_dst, heads, apply_gelu, static_cast<float*>(nullptr));
  MDG_CHECK_LAUNCH("mdg_hgt_attention_rows");
  return MDG_OK;
}

extern "C" size_t mdg_hgt_attention_bwd_workspace_bytes(int64_t nnz, int64_t n_items, int64_t n_src_items, int heads) {
  if (nnz <= 0) return 0;
  return (static_cast<size_t>(nnz) * 2 * heads + static_cast<size_t>(n_items) * 128 + static_cast<size_t>(n_src_items) * 256) * sizeof(float);
}

extern "C" int mdg_hgt_attention_bwd(const float* q, int64_t ldq, const float* kv, int64_t ldkv, const int64_t* col, int64_t nnz,
                                     const int64_t* item_dst, const int64_t* item_begin, const int64_t* item_end, int64_t n_items,
                                     const int64_t* item_ptr, int64_t n_dst, const float* dout, int64_t lddo, const float* out_pre,
                                     int64_t ldp, const float* stats, int heads, const int64_t* t_edge, const int64_t* t_dst,
                                     const int64_t* t_item_begin, const int64_t* t_item_end, int64_t n_src_items,
                                     const int64_t* t_item_ptr, const int64_t* t_row, int64_t n_src_rows, const int64_t* t_item_row,
                                     float* dq, int64_t lddq, float* dkv, int64_t lddkv, const void* kv16, void* workspace, size_t workspace_bytes, void* stream) {
  MDG_CHECK_ARG(heads == 1 || heads == 2 || heads == 4 || heads == 8, "mdg_hgt_attention_bwd: heads must be 1, 2, 4 or 8 (got %d)", heads);
  MDG_CHECK_ARG(n_dst >= 0 && n_items >= 0 && nnz >= 0 && n_src_items >= 0 && n_src_rows >= 0, "mdg_hgt_attention_bwd: negative size");
  if (n_dst == 0) return MDG_OK;
  MDG_CHECK_ARG(q && dq && item_ptr && dout && out_pre && stats, "mdg_hgt_attention_bwd: null pointer");
  MDG_CHECK_ARG(ldq % 4 == 0 && lddo % 4 == 0 && ldp % 4 == 0 && lddq % 4 == 0 && ldq >= 128 && lddo >= 128 && ldp >= 128 && lddq >= 128 &&
                mdg_aligned16(q) && mdg_aligned16(dout) && mdg_aligned16(out_pre) && mdg_aligned16(dq), "mdg_hgt_attention_bwd: bad strides / alignment");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t need = mdg_hgt_attention_bwd_workspace_bytes(nnz, n_items, n_src_items, heads);
  if (need && (!workspace || workspace_bytes < need || !mdg_aligned16(workspace))) {
    mdg_set_error("mdg_hgt_attention_bwd: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return MDG_EWORKSPACE;
  }
  float* part_dq = static_cast<float*>(workspace);
  float* part_src = part_dq ? part_dq + n_items * 128 : nullptr;
  float* edge_alpha = part_src ? part_src + n_src_items * 256 : nullptr;
  float* edge_da = edge_alpha ? edge_alpha + heads : nullptr;                 // interleaved per edge: [alpha x heads | da x heads]
  if (n_items > 0) {
    MDG_CHECK_ARG(kv && col && item_dst && item_begin && item_end && dkv && t_edge && t_dst && t_item_begin && t_item_end && t_item_ptr && t_row,
                  "mdg_hgt_attention_bwd: null plan pointer");
    MDG_CHECK_ARG(ldkv % 4 == 0 && lddkv % 4 == 0 && ldkv >= 128 && lddkv >= 128 && mdg_aligned16(kv) && mdg_aligned16(dkv), "mdg_hgt_attention_bwd: bad kv strides");
    HgtBwdArgs a{q, ldq, kv, ldkv, col, item_dst, item_begin, item_end, n_items, dout, lddo, out_pre, ldp, stats, edge_alpha, edge_da, part_dq, heads, item_ptr, dq, lddq,
                 static_cast<const uint16_t*>(kv16)};
    MDG_CHECK_ARG(!kv16 || (ldkv == 128 && (reinterpret_cast<uintptr_t>(kv16) & 7u) == 0), "mdg_hgt_attention_bwd: the 16-bit mirror needs ldkv == 128 and 8-byte alignment");
    if (kv16) hipLaunchKernelGGL(hgt_attention_bwd_edge_kernel<true>, dim3(static_cast<unsigned>(mdg_cdiv(n_items, 4))), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(hgt_attention_bwd_edge_kernel<false>, dim3(static_cast<unsigned>(mdg_cdiv(n_items, 4))), dim3(256), 0, st, a);
  }
  hipLaunchKernelGGL(hgt_sum_items_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n_dst, 8))), dim3(256), 0, st, part_dq, item_ptr, nullptr, dq, lddq, n_dst, 128, 1);
  if (n_src_items > 0) {
    HgtSrcArgs s{q, ldq, dout, lddo, t_edge, t_dst, t_item_begin, t_item_end, n_src_items, edge_alpha, edge_da, part_src, heads, t_item_row, t_item_ptr, t_row, dkv, lddkv};
    hipLaunchKernelGGL(hgt_attention_bwd_src_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n_src_items, 4))), dim3(256), 0, st, s);
    // dk' and dv' of key row r are rows r and r+1 of dkv (ldkv == 128 layout): written as one 256-float row
    hipLaunchKernelGGL(hgt_sum_items_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n_src_rows, 4))), dim3(256), 0, st, part_src, t_item_ptr, t_row, dkv,
                       lddkv, n_src_rows, 256, t_item_row ? 1 : 0);
  }
  MDG_CHECK_LAUNCH("mdg_hgt_attention_bwd");
  return MDG_OK;
}
