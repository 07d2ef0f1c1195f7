// Gathered bilinear head for the finetune step (train_ddi_batch.py:285-288): the reference materialises
// sigmoid(model(...)) [L, N, N] and then reads T (label, head, tail) entries of it; here only those T entries are
// ever computed, forward and backward:
//     s_t = z_head[h_t]^T W[l_t] z_tail[t_t]
//     dz_head[h_t] += ds_t W[l_t] z_tail[t_t],  dz_tail[t_t] += ds_t W[l_t]^T z_head[h_t],  dW[l_t] += ds_t z_head[h_t] z_tail[t_t]^T
// Triples are pre-sorted by label (host plan) and cut into tiles of <= 32 triples of one label; one wave per tile runs
// U = Z_tail[tile] W_l^T on v_mfma_f32_32x32x2_f32 (exact fp32) with W_l streamed from L2 and the gathered embedding
// rows held in registers.  No atomics anywhere: per-triple gradient rows are summed per drug by mdg_csr_aggregate and
// per-chunk dW partials are summed per label in a fixed order.
#include "mdg_common.h"
#include <stdlib.h>

namespace {
#include "tn16.h"           // the 16-bit TN kernel with gathered rows: dW of the head on the split-bf16 matrix cores

constexpr int HD = 128;     // embedding width of every shipped config

struct GatherArgs {
  const float* zh; const float* zt;       // [Nh,128], [Nt,128]
  const float* w; const float* wt;        // [L,128,128] and its per-label transpose (same pointer when symmetric)
  const int64_t* head; const int64_t* tail;   // [T] sorted by label
  const int64_t* tile_start;              // [n_tiles+1] first triple of each tile
  const int64_t* tile_label;              // [n_tiles]
  int64_t n_tiles;
  float* score;                           // fwd: [T]
  const float* ds;                        // bwd: [T]
  float* gzh; float* gzt;                 // bwd: per-triple gradient rows [T,128]
};

// acc[v] (lane x = triple, half) <- sum_k W[c0 + i(v,half)][k] z[x][k], for the four 32-column tiles c0 = 0,32,64,96.
// zf holds the lane's share of its z row in the permuted-k order (k = 8q + 4*half + e).
__device__ __forceinline__ void wz_tile(const float* __restrict__ wl, int c0, int x, int half, const f32x4 (&zf)[16], f32x16& acc) {
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  const float* wrow = wl + static_cast<int64_t>(c0 + x) * HD + 4 * half;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const f32x4 wf = *reinterpret_cast<const f32x4*>(wrow + 8 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[e], zf[q][e], acc, 0, 0, 0);
  }
}

__device__ __forceinline__ void load_row_frag(const float* __restrict__ row, int half, f32x4 (&zf)[16]) {
#pragma unroll
  for (int q = 0; q < 16; ++q) zf[q] = *reinterpret_cast<const f32x4*>(row + 8 * q + 4 * half);
}

// ---- rows[t] = W[l] z[index[t]] for tiles of <= 32 rows of one label: the (label, drug) PAIR products of the pair-compressed
// head (V = W^T z_head forward, R = W u backward), 2.4 million rows per call.  The matrix work is the same as in
// bilinear_gather_kernel<2>; what differs is how the operands and the result move.  There a lane loaded ITS z row and ITS W row
// straight from global memory (32 rows x 32 bytes per instruction) and stored ITS result row in 16-byte pieces.  Here the
// wave copies the gathered z rows (64 columns at a time) and the 32 x 64 pieces of W_l into its own LDS tiles with row-major
// loads (one 256-byte row piece per 16 lanes), keeps its z fragments in registers over the four 32-row chunks of W_l, prefetches
// the next W piece while the current one feeds the MFMAs, and takes z as the A operand so that the result has the lane on the
// output feature: every store is a 128-byte row piece.  (Same scheme as the fusion self-attention, fusion.hip.)
__device__ __forceinline__ int mv_tile_off(int row, int piece) { return row * 64 + ((piece ^ (row & 15)) << 2); }

__device__ __forceinline__ void mv_load_w(const float* __restrict__ w, int r0, int k0, int lane, f32x4 (&r)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = *reinterpret_cast<const f32x4*>(w + static_cast<int64_t>(r0 + (lane >> 4) + 4 * i) * HD + k0 + 4 * (lane & 15));
}

__device__ __forceinline__ void mv_write(const f32x4 (&r)[8], float* tile, int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(tile + mv_tile_off((lane >> 4) + 4 * i, lane & 15)) = r[i];
}

__global__ __launch_bounds__(256) void bilinear_matvec_rows_kernel(const GatherArgs p) {
  __shared__ __attribute__((aligned(16))) float stage[4][2][32 * 64];
  const int lane = threadIdx.x & 63, x = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
  const int64_t tile = static_cast<int64_t>(blockIdx.x) * 4 + wave;
  if (tile >= p.n_tiles) return;
  const int64_t t0 = p.tile_start[tile];
  const int cnt = static_cast<int>(p.tile_start[tile + 1] - t0);
  if (cnt <= 0) return;
  const float* wl = p.w + p.tile_label[tile] * HD * HD;
  float* const tz = stage[wave][0];
  float* const tw = stage[wave][1];
  // source row of tile row r (rows past the tile repeat its last row; their results are never stored)
  int64_t src[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = (lane >> 4) + 4 * i;
    const int64_t t = t0 + (r < cnt ? r : cnt - 1);
    src[i] = p.tail ? p.tail[t] : t;
  }
  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
  f32x4 pre[8];
  mv_load_w(wl, 0, 0, lane, pre);
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    {                                                     // the tile's z rows, columns 64 kh .. 64 kh + 63
      f32x4 zr[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) zr[i] = *reinterpret_cast<const f32x4*>(p.zt + src[i] * HD + 64 * kh + 4 * (lane & 15));
      mv_write(zr, tz, lane);
    }
    f32x4 zf[8];                                          // lane x: row x of the z tile, pieces 2q + half
#pragma unroll
    for (int q = 0; q < 8; ++q) zf[q] = *reinterpret_cast<const f32x4*>(tz + mv_tile_off(x, 2 * q + half));
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      mv_write(pre, tw, lane);
      const int nc = c + 1, nk = nc == 4 ? kh + 1 : kh;   // next piece of W_l travels while this one is used
      if (nk < 2) mv_load_w(wl, 32 * (nc & 3), 64 * nk, lane, pre);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x4 wf = *reinterpret_cast<const f32x4*>(tw + mv_tile_off(x, 2 * q + half));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(zf[q][e], wf[e], acc[c], 0, 0, 0);
      }
    }
  }
  // acc[c][v] on lane x = rows[t0 + i_v][32 c + x]
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int i = (v & 3) + 8 * (v >> 2) + 4 * half;
      if (i < cnt) p.gzh[(t0 + i) * HD + 32 * c + x] = acc[c][v];
    }
}

// ---- the same per-pair products on the split-bf16 matrix cores (the step's 16-bit modes; fp32-grade: ~4e-6 of max) ----------
// rows[t][i] = sum_k W_l[i][k] z[index[t]][k] is Z_tile W_l^T with BOTH operands k-contiguous in memory, which is exactly how
// v_mfma_f32_16x16x32_bf16 wants them: no LDS and no barrier.  A wave owns a tile of <= 32 pairs: its z rows are loaded once
// (lane (row, k group): 2 x 16 B of fp32 per 32-k step, split hi / lo in registers, 64 VGPRs), W_l comes from hi / lo bf16 images
// made once per call (16 B per lane and plane: the four k groups of a row are 64 contiguous bytes), three products per tile pair
// (lo.hi + hi.lo + hi.hi).  192 MFMAs of 16 cycles per tile against 256 of 64 cycles for the exact kernel.
__global__ __launch_bounds__(256) void split_rows_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ hi, __bf16* __restrict__ lo, int64_t n4) {
  const int64_t q = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (q >= n4) return;
  const f32x4 v = reinterpret_cast<const f32x4*>(x)[q];
  bf16x4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    __bf16 a, b;
    mdg_split_bf16(v[e], a, b);
    h[e] = a;
    l[e] = b;
  }
  reinterpret_cast<bf16x4*>(hi)[q] = h;
  reinterpret_cast<bf16x4*>(lo)[q] = l;
}

struct Matvec16Args {
  const float* z; const __bf16* whi; const __bf16* wlo;
  const int64_t* row_index; const int64_t* tile_start; const int64_t* tile_label;
  int64_t n_tiles; float* out;
};

// A workgroup takes 16 consecutive tiles (4 per wave).  Tiles are sorted by label, so 5 groups in 6 carry one label: then W_l's
// two images (64 KB) are staged ONCE into LDS for the 16 tiles -- straight from L2 the kernel re-read them per tile (4.9 GB per
// call: it ran at the L2's 8.6 TB/s, not at the matrix cores' rate).  LDS image: row i, 16-byte chunk c at chunk c ^ (i & 15) of
// the row's 256 bytes: the 16 lanes of a ds_read_b128 group (16 rows, one chunk index) cover all 64 banks.
constexpr int MV16_TILES = 16;

template <bool LDS_W>
__device__ __forceinline__ void matvec16_tile(const Matvec16Args& p, int64_t tile, const char* lds_w, int lane) {
  const int c16 = lane & 15, g4 = lane >> 4;
  const int64_t t0 = p.tile_start[tile];
  const int cnt = static_cast<int>(p.tile_start[tile + 1] - t0);
  if (cnt <= 0) return;
  const int64_t l = p.tile_label[tile];
  // A operand: lane (c16, g4) holds row 16 tt + c16 of the tile, k = 32 ks + 8 g4 .. + 7
  bf16x8 ahi[2][4], alo[2][4];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int r = 16 * tt + c16;
    const int64_t t = t0 + (r < cnt ? r : cnt - 1);          // rows past the tile repeat its last row; never stored
    const float* zr = p.z + (p.row_index ? p.row_index[t] : t) * HD + 8 * g4;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(zr + 32 * ks), v1 = *reinterpret_cast<const f32x4*>(zr + 32 * ks + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        __bf16 a, b;
        mdg_split_bf16(e < 4 ? v0[e] : v1[e - 4], a, b);
        ahi[tt][ks][e] = a;
        alo[tt][ks][e] = b;
      }
    }
  }
  // B operand: lane (c16, g4) holds W_l[16 nt + c16][32 ks + 8 g4 .. + 7]
  const __bf16* const wh = p.whi + l * HD * HD + static_cast<int64_t>(c16) * HD + 8 * g4;
  const __bf16* const wl = p.wlo + l * HD * HD + static_cast<int64_t>(c16) * HD + 8 * g4;
  f32x4 acc[2][8];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) acc[tt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    bf16x8 bh[4], bl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if constexpr (LDS_W) {
        const int off = (16 * nt + c16) * 256 + (((4 * ks + g4) ^ c16) << 4);
        bh[ks] = *reinterpret_cast<const bf16x8*>(lds_w + off);
        bl[ks] = *reinterpret_cast<const bf16x8*>(lds_w + 32768 + off);
      } else {
        bh[ks] = *reinterpret_cast<const bf16x8*>(wh + 16 * nt * HD + 32 * ks);
        bl[ks] = *reinterpret_cast<const bf16x8*>(wl + 16 * nt * HD + 32 * ks);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        acc[tt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo[tt][ks], bh[ks], acc[tt][nt], 0, 0, 0);
        acc[tt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[tt][ks], bl[ks], acc[tt][nt], 0, 0, 0);
        acc[tt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[tt][ks], bh[ks], acc[tt][nt], 0, 0, 0);
      }
  }
  // acc[tt][nt][i]: tile row 16 tt + 4 g4 + i, output feature 16 nt + c16
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 16 * tt + 4 * g4 + i;
      if (r < cnt) {
        float* o = p.out + (t0 + r) * HD + c16;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) o[16 * nt] = acc[tt][nt][i];
      }
    }
}

__global__ __launch_bounds__(256) void bilinear_matvec_rows16_kernel(const Matvec16Args p) {
  extern __shared__ __attribute__((aligned(16))) char lds_w[];     // hi image | lo image of one W_l, 32 KB each
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * MV16_TILES;
  const int64_t last = base + MV16_TILES - 1 < p.n_tiles ? base + MV16_TILES - 1 : p.n_tiles - 1;
  const int64_t mine = base + (tid & 15) < p.n_tiles ? base + (tid & 15) : p.n_tiles - 1;
  const int64_t l0 = p.tile_label[base];
  const bool shared_w = __syncthreads_and(p.tile_label[mine] == l0) != 0 && last > base;      // every tile of the group carries label l0
  if (shared_w) {
    const char* gh = reinterpret_cast<const char*>(p.whi + l0 * HD * HD);
    const char* gl = reinterpret_cast<const char*>(p.wlo + l0 * HD * HD);
#pragma unroll
    for (int i = 0; i < 8; ++i) {                          // 2048 chunks of 16 B per image, 8 per thread
      const int q = tid + 256 * i, row = q >> 4, ch = q & 15;
      const int off = row * 256 + ((ch ^ (row & 15)) << 4);
      *reinterpret_cast<u32x4*>(lds_w + off) = *reinterpret_cast<const u32x4*>(gh + q * 16);
      *reinterpret_cast<u32x4*>(lds_w + 32768 + off) = *reinterpret_cast<const u32x4*>(gl + q * 16);
    }
    __syncthreads();
  }
#pragma unroll 1
  for (int j = 0; j < MV16_TILES / 4; ++j) {
    const int64_t tile = base + 4 * j + wave;
    if (tile >= p.n_tiles) break;
    if (shared_w) matvec16_tile<true>(p, tile, lds_w, lane);
    else matvec16_tile<false>(p, tile, lds_w, lane);
  }
}

// MODE 0: scores; MODE 1: both per-triple gradient rows; MODE 2: rows[t] = W[l] z_tail[tail[t]] only (one matrix-vector product
// per "triple" -- the (label, drug) PAIRS of the pair-compressed path, whose inputs are rows of z or of a per-pair sum).
template <int MODE>
__global__ __launch_bounds__(256) void bilinear_gather_kernel(const GatherArgs p) {
  constexpr bool BWD = MODE != 0;
  const int lane = threadIdx.x & 63, x = lane & 31, half = lane >> 5;
  const int64_t tile = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (tile >= p.n_tiles) return;
  const int64_t t0 = p.tile_start[tile];
  const int cnt = static_cast<int>(p.tile_start[tile + 1] - t0);
  if (cnt <= 0) return;
  const int64_t t = t0 + (x < cnt ? x : cnt - 1);
  const int64_t l = p.tile_label[tile];
  const float* wl = p.w + l * HD * HD;
  const float* zt_row = p.zt + (p.tail ? p.tail[t] : t) * HD;           // null index = identity (row t)
  const float* zh_row = MODE == 2 ? zt_row : p.zh + p.head[t] * HD;

  f32x4 zf[16];
  load_row_frag(zt_row, half, zf);
  if (!BWD) {
    float s = 0.f;
    for (int c0 = 0; c0 < HD; c0 += 32) {
      f32x16 acc;
      wz_tile(wl, c0, x, half, zf, acc);                    // acc[v] = (W z_t)[c0 + (v&3) + 8(v>>2) + 4half]
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(zh_row + c0 + 8 * g + 4 * half);
#pragma unroll
        for (int e = 0; e < 4; ++e) s += acc[4 * g + e] * hv[e];
      }
    }
    s += __shfl_xor(s, 32, 64);
    if (half == 0 && x < cnt) p.score[t] = s;
  } else {
    const float g_t = MODE == 2 ? 1.0f : p.ds[t];
    for (int c0 = 0; c0 < HD; c0 += 32) {                   // d z_head[h_t] row: ds * W z_t
      f32x16 acc;
      wz_tile(wl, c0, x, half, zf, acc);
      if (x < cnt) {
        float* r = p.gzh + t * HD + c0 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(r + 8 * g) = f32x4{acc[4 * g] * g_t, acc[4 * g + 1] * g_t, acc[4 * g + 2] * g_t, acc[4 * g + 3] * g_t};
      }
    }
    if constexpr (MODE == 2) return;
    load_row_frag(zh_row, half, zf);
    const float* wtl = p.wt + l * HD * HD;
    for (int c0 = 0; c0 < HD; c0 += 32) {                   // d z_tail[t_t] row: ds * W^T z_h
      f32x16 acc;
      wz_tile(wtl, c0, x, half, zf, acc);
      if (x < cnt) {
        float* r = p.gzt + t * HD + c0 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(r + 8 * g) = f32x4{acc[4 * g] * g_t, acc[4 * g + 1] * g_t, acc[4 * g + 2] * g_t, acc[4 * g + 3] * g_t};
      }
    }
  }
}

// dW partial of one chunk (<= 256 triples of one label): P[a][b] = sum_t ds_t z_head[h_t][a] z_tail[t_t][b].
// Wave w owns rows a in [32w, 32w+32) and all 128 columns (four accumulator tiles); the contraction runs over the
// chunk's triples two at a time (MFMA k = lane half).
__global__ __launch_bounds__(256) void bilinear_gather_dw_kernel(const float* __restrict__ zh, const float* __restrict__ zt,
                                                                 const int64_t* __restrict__ head, const int64_t* __restrict__ tail,
                                                                 const float* __restrict__ ds, const int64_t* __restrict__ chunk_start,
                                                                 float* __restrict__ partial) {
  const int lane = threadIdx.x & 63, x = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
  const int64_t chunk = blockIdx.x;
  const int64_t t0 = chunk_start[chunk], t1 = chunk_start[chunk + 1];
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
  // Batches of DW_U row pairs: the index loads of a batch, then its operand loads (each depends on an index), then its MFMAs,
  // with the NEXT batch's indices and operands already in flight (the plain loop paid two dependent memory latencies per step).
  constexpr int DW_U = 16;
  struct Batch { float a[DW_U], b[DW_U][4]; };
  const auto load = [&](int64_t tb, Batch& o) {
    int64_t hi[DW_U], ti[DW_U];
    float g[DW_U];
#pragma unroll
    for (int u = 0; u < DW_U; ++u) {
      const int64_t t = tb + 2 * u + half;
      const bool ok = t < t1;
      const int64_t tt = ok ? t : t1 - 1;
      hi[u] = head[tt];
      ti[u] = tail ? tail[tt] : tt;                      // null index = identity (row tt)
      g[u] = ok ? (ds ? ds[tt] : 1.f) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < DW_U; ++u) {
      o.a[u] = zh[hi[u] * HD + 32 * wave + x] * g[u];
      const float* zr = zt + ti[u] * HD + x;
#pragma unroll
      for (int j = 0; j < 4; ++j) o.b[u][j] = zr[32 * j];
    }
  };
  const auto compute = [&](const Batch& o) {
#pragma unroll
    for (int u = 0; u < DW_U; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[u], o.b[u][j], acc[j], 0, 0, 0);
  };
  if (t1 > t0) {                                          // (rows past t1 enter with weight 0: both halves take the same steps)
    Batch cur, nxt;
    load(t0, cur);
    for (int64_t tb = t0; tb < t1; tb += 2 * DW_U) {
      const bool more = tb + 2 * DW_U < t1;
      if (more) load(tb + 2 * DW_U, nxt);
      compute(cur);
      if (more) cur = nxt;
    }
  }
  // acc[j][v]: row a = 32*wave + (v&3) + 8(v>>2) + 4*half, column b = 32*j + x
  float* out = partial + chunk * HD * HD;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) out[static_cast<int64_t>(32 * wave + (v & 3) + 8 * (v >> 2) + 4 * half) * HD + 32 * j + x] = acc[j][v];
}

// dW[l] = sum of the partials of label l's chunks, in chunk order; labels without triples get zeros.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, const int64_t* __restrict__ label_chunk_ptr,
                                                              float* __restrict__ dw) {
  const int64_t l = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float s = 0.f;
  for (int64_t c = label_chunk_ptr[l]; c < label_chunk_ptr[l + 1]; ++c) s += partial[c * HD * HD + i];
  dw[l * HD * HD + i] = s;
}

// BCE on probabilities p = sigmoid(s) with nn.BCELoss's log clamp at -100, and its gradient w.r.t. the logit
// (through BCELoss.backward's 1e-12 clamp of p(1-p) and the sigmoid), scaled by `gscale` (1/T for 'mean').
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ s, const float* __restrict__ y, float* __restrict__ term,
                                                         float* __restrict__ ds, int64_t n, float gscale) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const float p = 1.0f / (1.0f + expf(-s[i]));
  const float yy = y[i];
  if (term) term[i] = -(yy * fmaxf(logf(p), -100.f) + (1.0f - yy) * fmaxf(logf(1.0f - p), -100.f));
  if (ds) {
    const float pq = p * (1.0f - p);
    ds[i] = gscale * (p - yy) / fmaxf(pq, 1e-12f) * pq;
  }
}

// dW_original = triu(dW_sym) + triu(dW_sym^T, 1): the upper triangle collects both mirrored entries.
__global__ __launch_bounds__(256) void symmetrize_bwd_kernel(const float* __restrict__ dws, float* __restrict__ dwo, int64_t n_labels, int D) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n_labels * D * D) return;
  const int64_t l = i / (static_cast<int64_t>(D) * D);
  const int r = static_cast<int>((i / D) % D), c = static_cast<int>(i % D);
  const float* m = dws + l * D * D;
  dwo[i] = r == c ? m[r * D + c] : (r < c ? m[r * D + c] + m[c * D + r] : 0.f);
}

// out[t] = a[ia[t]] . b[ib[t]]  (rows of 128 floats): half a wave per entry, 16 bytes per lane, fixed summation order.
__global__ __launch_bounds__(256) void gather_rowdot_kernel(const float* __restrict__ a, const int64_t* __restrict__ ia, const float* __restrict__ b,
                                                            const int64_t* __restrict__ ib, float* __restrict__ out, int64_t n) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 8 + (threadIdx.x >> 5);
  const int x = threadIdx.x & 31;
  if (t >= n) return;
  const f32x4 u = *reinterpret_cast<const f32x4*>(a + ia[t] * HD + 4 * x);
  const f32x4 v = *reinterpret_cast<const f32x4*>(b + ib[t] * HD + 4 * x);
  float s = (u[0] * v[0] + u[1] * v[1]) + (u[2] * v[2] + u[3] * v[3]);
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (x == 0) out[t] = s;
}

}  // namespace

static int gather_check(const char* who, const void* zh, const void* zt, const void* w, const void* head, const void* tail,
                        const void* tile_start, const void* tile_label, int64_t n_tiles, int64_t D) {
  MDG_CHECK_ARG(D == HD, "%s: D must be 128 (got %lld)", who, (long long)D);
  MDG_CHECK_ARG(n_tiles >= 0, "%s: negative tile count", who);
  if (n_tiles == 0) return MDG_OK;
  MDG_CHECK_ARG(zh && zt && w && head && tail && tile_start && tile_label, "%s: null pointer", who);
  MDG_CHECK_ARG(mdg_aligned16(zh) && mdg_aligned16(zt) && mdg_aligned16(w), "%s: operands must be 16-byte aligned", who);
  return MDG_OK;
}

extern "C" int mdg_bilinear_gather(const float* z_head, const float* z_tail, const float* w, const int64_t* head, const int64_t* tail,
                                   const int64_t* tile_start, const int64_t* tile_label, int64_t n_tiles, float* score, int64_t D,
                                   void* stream) {
  if (int rc = gather_check("mdg_bilinear_gather", z_head, z_tail, w, head, tail, tile_start, tile_label, n_tiles, D)) return rc;
  if (n_tiles == 0) return MDG_OK;
  MDG_CHECK_ARG(score, "mdg_bilinear_gather: null score");
  GatherArgs a{z_head, z_tail, w, w, head, tail, tile_start, tile_label, n_tiles, score, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(bilinear_gather_kernel<0>, dim3(static_cast<unsigned>(mdg_cdiv(n_tiles, 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  MDG_CHECK_LAUNCH("mdg_bilinear_gather");
  return MDG_OK;
}

extern "C" int mdg_bilinear_gather_bwd_prec(const float* z_head, const float* z_tail, const float* w, const float* w_t, const int64_t* head,
                                            const int64_t* tail, const int64_t* tile_start, const int64_t* tile_label, int64_t n_tiles,
                                            const int64_t* chunk_start, int64_t n_chunks, const int64_t* label_chunk_ptr, int64_t n_labels,
                                            const float* dscore, float* gz_head_rows, float* gz_tail_rows, float* dw_partial, float* dw,
                                            int64_t D, int precision, void* stream);

extern "C" int mdg_bilinear_gather_bwd(const float* z_head, const float* z_tail, const float* w, const float* w_t, const int64_t* head,
                                       const int64_t* tail, const int64_t* tile_start, const int64_t* tile_label, int64_t n_tiles,
                                       const int64_t* chunk_start, int64_t n_chunks, const int64_t* label_chunk_ptr, int64_t n_labels,
                                       const float* dscore, float* gz_head_rows, float* gz_tail_rows, float* dw_partial, float* dw,
                                       int64_t D, void* stream) {
  return mdg_bilinear_gather_bwd_prec(z_head, z_tail, w, w_t, head, tail, tile_start, tile_label, n_tiles, chunk_start, n_chunks, label_chunk_ptr, n_labels,
                                      dscore, gz_head_rows, gz_tail_rows, dw_partial, dw, D, MDG_PREC_F32, stream);
}

extern "C" int mdg_bilinear_gather_bwd_prec(const float* z_head, const float* z_tail, const float* w, const float* w_t, const int64_t* head,
                                            const int64_t* tail, const int64_t* tile_start, const int64_t* tile_label, int64_t n_tiles,
                                            const int64_t* chunk_start, int64_t n_chunks, const int64_t* label_chunk_ptr, int64_t n_labels,
                                            const float* dscore, float* gz_head_rows, float* gz_tail_rows, float* dw_partial, float* dw,
                                            int64_t D, int precision, void* stream) {
  if (int rc = gather_check("mdg_bilinear_gather_bwd", z_head, z_tail, w, head, tail, tile_start, tile_label, n_tiles, D)) return rc;
  MDG_CHECK_ARG(precision == MDG_PREC_F32 || precision == MDG_PREC_BF16 || precision == MDG_PREC_BF16X3, "mdg_bilinear_gather_bwd: unknown precision %d", precision);
  MDG_CHECK_ARG(n_chunks >= 0 && n_labels >= 0, "mdg_bilinear_gather_bwd: negative size");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_tiles > 0) {
    MDG_CHECK_ARG(w_t && dscore && gz_head_rows && gz_tail_rows && mdg_aligned16(w_t) && mdg_aligned16(gz_head_rows) && mdg_aligned16(gz_tail_rows),
                  "mdg_bilinear_gather_bwd: null / misaligned pointer");
    GatherArgs a{z_head, z_tail, w, w_t, head, tail, tile_start, tile_label, n_tiles, nullptr, dscore, gz_head_rows, gz_tail_rows};
    hipLaunchKernelGGL(bilinear_gather_kernel<1>, dim3(static_cast<unsigned>(mdg_cdiv(n_tiles, 4))), dim3(256), 0, st, a);
  }
  if (dw) {
    MDG_CHECK_ARG(label_chunk_ptr && (n_chunks == 0 || (chunk_start && dw_partial)), "mdg_bilinear_gather_bwd: dW needs the chunk tables and scratch");
    if (n_chunks > 0 && precision != MDG_PREC_F32 && mdg_aligned16(z_head) && mdg_aligned16(z_tail)) {
      // the chunk's outer-product sum as a TN product of gathered rows on the split-bf16 matrix cores (fp32-grade: three products of
      // the hi / lo halves, fp32 accumulation) -- in every 16-bit mode of the step: the head stays fp32-grade
      GwArgs a{z_head, HD, z_tail, HD, dw_partial, nullptr, 0, 0, HD, HD, head, tail, dscore, chunk_start};
      hipLaunchKernelGGL((grad_weight16_kernel<MDG_PREC_BF16X3, true>), dim3(1, 1, static_cast<unsigned>(n_chunks)), dim3(256), 2 * 2 * 2 * G16_PLANE, st, a);
    } else if (n_chunks > 0)
      hipLaunchKernelGGL(bilinear_gather_dw_kernel, dim3(static_cast<unsigned>(n_chunks)), dim3(256), 0, st, z_head, z_tail, head, tail, dscore,
                         chunk_start, dw_partial);
    if (n_labels > 0)
      hipLaunchKernelGGL(reduce_partials_kernel, dim3(HD * HD / 256, static_cast<unsigned>(n_labels)), dim3(256), 0, st, dw_partial,
                         label_chunk_ptr, dw);
  }
  MDG_CHECK_LAUNCH("mdg_bilinear_gather_bwd");
  return MDG_OK;
}

extern "C" int mdg_bilinear_matvec_rows(const float* z, const float* w, const int64_t* row_index, const int64_t* tile_start,
                                        const int64_t* tile_label, int64_t n_tiles, float* rows_out, int64_t D, void* stream) {
  MDG_CHECK_ARG(D == HD, "mdg_bilinear_matvec_rows: D must be 128 (got %lld)", (long long)D);
  MDG_CHECK_ARG(n_tiles >= 0, "mdg_bilinear_matvec_rows: negative tile count");
  if (n_tiles == 0) return MDG_OK;
  MDG_CHECK_ARG(z && w && tile_start && tile_label && rows_out && mdg_aligned16(z) && mdg_aligned16(w) && mdg_aligned16(rows_out),
                "mdg_bilinear_matvec_rows: null / misaligned pointer");
  GatherArgs a{z, z, w, w, row_index, row_index, tile_start, tile_label, n_tiles, nullptr, nullptr, rows_out, nullptr};
  hipLaunchKernelGGL(bilinear_matvec_rows_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n_tiles, 4))), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  MDG_CHECK_LAUNCH("mdg_bilinear_matvec_rows");
  return MDG_OK;
}

extern "C" size_t mdg_bilinear_matvec_rows_workspace_bytes(int64_t n_labels, int precision) {
  return (precision == MDG_PREC_F32 || n_labels <= 0) ? 0 : static_cast<size_t>(n_labels) * HD * HD * 2 * sizeof(__bf16);
}

extern "C" int mdg_bilinear_matvec_rows_prec(const float* z, const float* w, int64_t n_labels, const int64_t* row_index, const int64_t* tile_start,
                                             const int64_t* tile_label, int64_t n_tiles, float* rows_out, int64_t D, int precision, void* workspace,
                                             size_t workspace_bytes, void* stream) {
  MDG_CHECK_ARG(precision == MDG_PREC_F32 || precision == MDG_PREC_BF16 || precision == MDG_PREC_BF16X3, "mdg_bilinear_matvec_rows: unknown precision %d", precision);
  if (precision == MDG_PREC_F32) return mdg_bilinear_matvec_rows(z, w, row_index, tile_start, tile_label, n_tiles, rows_out, D, stream);
  MDG_CHECK_ARG(D == HD, "mdg_bilinear_matvec_rows: D must be 128 (got %lld)", (long long)D);
  MDG_CHECK_ARG(n_tiles >= 0 && n_labels > 0, "mdg_bilinear_matvec_rows: bad sizes");
  if (n_tiles == 0) return MDG_OK;
  MDG_CHECK_ARG(z && w && tile_start && tile_label && rows_out && mdg_aligned16(z) && mdg_aligned16(w) && mdg_aligned16(rows_out),
                "mdg_bilinear_matvec_rows: null / misaligned pointer");
  const size_t need = mdg_bilinear_matvec_rows_workspace_bytes(n_labels, precision);
  if (!workspace || workspace_bytes < need || !mdg_aligned16(workspace)) {
    mdg_set_error("mdg_bilinear_matvec_rows: workspace of %zu bytes (16-byte aligned) required, got %zu", need, workspace_bytes);
    return MDG_EWORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  __bf16* whi = static_cast<__bf16*>(workspace);
  __bf16* wlo = whi + n_labels * HD * HD;
  const int64_t n4 = n_labels * HD * HD / 4;
  hipLaunchKernelGGL(split_rows_bf16_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n4, 256))), dim3(256), 0, st, w, whi, wlo, n4);
  Matvec16Args a{z, whi, wlo, row_index, tile_start, tile_label, n_tiles, rows_out};
  hipLaunchKernelGGL(bilinear_matvec_rows16_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n_tiles, MV16_TILES))), dim3(256), 65536, st, a);
  MDG_CHECK_LAUNCH("mdg_bilinear_matvec_rows");
  return MDG_OK;
}

extern "C" int mdg_gather_rowdot(const float* a, const int64_t* ia, const float* b, const int64_t* ib, float* out, int64_t n, int64_t D,
                                 void* stream) {
  MDG_CHECK_ARG(D == HD && n >= 0, "mdg_gather_rowdot: D must be 128, n >= 0");
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(a && ia && b && ib && out && mdg_aligned16(a) && mdg_aligned16(b), "mdg_gather_rowdot: null / misaligned pointer");
  hipLaunchKernelGGL(gather_rowdot_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n, 8))), dim3(256), 0, static_cast<hipStream_t>(stream), a, ia, b, ib,
                     out, n);
  MDG_CHECK_LAUNCH("mdg_gather_rowdot");
  return MDG_OK;
}

extern "C" int mdg_bce_logits(const float* score, const float* target, float* term, float* dscore, int64_t n, float grad_scale, void* stream) {
  MDG_CHECK_ARG(n >= 0, "mdg_bce_logits: negative size");
  if (n == 0) return MDG_OK;
  MDG_CHECK_ARG(score && target && (term || dscore), "mdg_bce_logits: null pointer");
  hipLaunchKernelGGL(bce_logits_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), score,
                     target, term, dscore, n, grad_scale);
  MDG_CHECK_LAUNCH("mdg_bce_logits");
  return MDG_OK;
}

extern "C" int mdg_symmetrize_bwd(const float* dw_sym, float* dw_original, int64_t n_labels, int64_t D, void* stream) {
  MDG_CHECK_ARG(n_labels >= 0 && D > 0 && D <= 4096, "mdg_symmetrize_bwd: bad shape");
  if (n_labels == 0) return MDG_OK;
  MDG_CHECK_ARG(dw_sym && dw_original && dw_sym != dw_original, "mdg_symmetrize_bwd: null / aliased pointers");
  hipLaunchKernelGGL(symmetrize_bwd_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n_labels * D * D, 256))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), dw_sym, dw_original, n_labels, static_cast<int>(D));
  MDG_CHECK_LAUNCH("mdg_symmetrize_bwd");
  return MDG_OK;
}
