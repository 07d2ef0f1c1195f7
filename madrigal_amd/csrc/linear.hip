// Fused dense block for gfx950:   Y = alpha * act( (X . W^T + bias) * scale + shift ) + beta * R
//
// X [M,K] (row stride ldx), W [N,K] (torch nn.Linear layout, row stride ldw), Y [M,N] (ldy), all fp32.
// One kernel serves every dense block on the path: the cv MLP (madrigal/models/models.py:178-180),
// the chemCPA encoder Linear+BatchNorm(eval)+ReLU blocks (madrigal/chemcpa/chemCPA/model.py:226-231;
// BN folded into scale/shift), the transformer's in/out projections and FFN with GELU and the
// residual add (nn.TransformerEncoderLayer, models.py:366), the GIN MLPs and the HGT projections.
//
// 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles of
// 32x32), BK = 32, two workgroups per CU.  A fused pre-pass writes both operands into the caller's
// workspace with K zero-padded to a multiple of 32 and, in the bf16 modes, already split into hi/lo bf16
// images, so the main loop is pure LDS-DMA (global_load_lds, swizzle on the source address) + ds_read_b128
// + MFMA: no conversion VALU, no ds_write, one raw barrier and one vmcnt wait per k-tile.
// Arithmetic modes as in the head: exact fp32 MFMA, bf16x3 (fp32-grade), bf16.
#include "mdg_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BK = 32;
// Tile shapes: a wave owns MT x NT_ MFMA tiles of 32x32; WM x WN waves per workgroup.
//   small: 2x2 tiles, 2x2 waves -> 128 x 128, 256 threads, 64 KB LDS, two workgroups per CU
//   big  : 4x2 tiles, 2x4 waves -> 256 x 256, 512 threads, 128 KB LDS, one workgroup per CU: each staged operand
//          byte feeds twice as many MFMAs, which is what the large fusion GEMMs (K = 2048) need.
template <int MT_, int NT__, int WM_, int WN_>
struct Shape {
  static constexpr int MT = MT_, NT_ = NT__, WM = WM_, WN = WN_;
  static constexpr int BM = 32 * MT_ * WM_, BN = 32 * NT__ * WN_, THREADS = 64 * WM_ * WN_, WAVES = WM_ * WN_;
  static constexpr int A_BYTES = BM * BK * 4, B_BYTES = BN * BK * 4;     // fp32 image == bf16 hi + lo images
  static constexpr int A_LO = BM * BK * 2, B_LO = BN * BK * 2;
  static constexpr int STAGE = A_BYTES + B_BYTES;
};
using Small = Shape<2, 2, 2, 2>;
using Big = Shape<4, 2, 2, 4>;

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

// fp32 tile [128][32]: 128-B rows, 8 chunks; bf16 tile [128][32]: 64-B rows, 4 chunks.  XOR swizzles chosen so
// that the 16 lanes of a ds_read_b128 group (16 consecutive rows, same chunk) hit 16 distinct 16-B slots.
__device__ __forceinline__ int off_f32(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int off_bf16(int row, int cb) { return row * 64 + ((cb ^ ((row >> 2) & 3)) << 4); }
// bf16 tile read by the 16x16x32 MFMA: the 4 lane groups of a fragment read the 4 chunks of the SAME 16 rows, so the
// conflict-free rotation is by row pair (checked exhaustively against the ds_read_b128 lane grouping of the guide).
__device__ __forceinline__ int off_bf16_m16(int row, int cb) { return row * 64 + ((cb ^ ((row >> 1) & 3)) << 4); }

typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return static_cast<unsigned>(reinterpret_cast<size_t>((lds_void*)p)); }

// LDS-DMA, 16 B per lane, LDS destination = wave-uniform base + lane*16 (inline asm: see bilinear.hip)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst_uniform);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}

// One operand tile (ROWS rows x 32 k) of stage k0 into LDS.  `base` is the operand image (fp32 [rows,ld] or one
// bf16 hi/lo image [rows,ld]); rows past the end clamp to the last row (their results are never stored).
template <int ESIZE, int ROWS, int WAVES, int MF = 32>   // ESIZE 4 = fp32 tile (8 rows per 1-KiB piece), 2 = bf16 tile (16 rows per piece)
__device__ __forceinline__ void dma_tile(const char* base, int64_t ld_bytes, int kb, int64_t row0, int64_t nrows, int64_t k0, char* lds,
                                         int wave, int lane) {
  constexpr int PIECES = ROWS * BK * ESIZE / 1024;
  static_assert(PIECES % WAVES == 0, "pieces must divide evenly over the waves");
#pragma unroll
  for (int i = 0; i < PIECES / WAVES; ++i) {
    const int p = wave + WAVES * i;
    int row, c;
    if constexpr (ESIZE == 4) { row = 8 * p + (lane >> 3); c = (lane & 7) ^ ((row >> 1) & 7); }
    else { row = 16 * p + (lane >> 2); c = (lane & 3) ^ (MF == 16 ? ((row >> 1) & 3) : ((row >> 2) & 3)); }
    int64_t gr = row0 + row;
    gr = gr < nrows ? gr : nrows - 1;
    glds16(base + gr * ld_bytes + k0 * ESIZE + c * 16, lds_addr(lds + p * 1024));
  }
}

// Operand images.  fp32: the tensor itself (or a zero-padded copy), row = Kp * 4 bytes.  bf16: [rows][Kp] bf16, Kp a multiple of 64.
// bf16x3: ONE image, per row and per block of 32 k: 64 B of hi values then 64 B of lo values (row = Kp * 4 bytes), so that a
// k tile of a row is one 128-byte line in both 16-bit modes; p1 = p0 + 64 addresses the lo halves.  kb = bytes a row advances
// per k (2 / 4 / 4).
struct Operand {
  const char* p0;
  const char* p1;
  int64_t ld_bytes;
  int64_t nrows;
  int kb;
};

// LDS layout of one stage.  fp32 / bf16x3: [A tile (fp32, or bf16 hi + lo) | B tile], S::STAGE bytes, two stages.  Single-product
// bf16: only the hi images exist, [A hi | B hi] = half the bytes, so FOUR stages fit in the same LDS: with one bf16 product per
// k step a 32-deep k tile is ~1000 matrix-pipe cycles per SIMD, shorter than the LDS-DMA round trip -- prefetch distance one
// left the matrix cores waiting for operands (0.67 PFLOP/s); three tiles in flight cover it.
template <int MODE, class S> constexpr int kStageBytes = (MODE == MDG_PREC_BF16) ? (S::A_LO + S::B_LO) : S::STAGE;
template <int MODE, class S> constexpr int kBOffset = (MODE == MDG_PREC_BF16) ? S::A_LO : S::A_BYTES;
template <int MODE> constexpr int kStages = (MODE == MDG_PREC_BF16) ? 4 : 2;

// PART 0: the A tiles, 1: the B tiles, -1: both
template <int MODE, class S, int MF = 32, int PART = -1>
__device__ __forceinline__ void dma_stage(const Operand& A, const Operand& B, int64_t row0, int64_t col0, int64_t k0, char* lds,
                                          int wave, int lane) {
  if constexpr (MODE == MDG_PREC_F32) {
    if constexpr (PART != 1) dma_tile<4, S::BM, S::WAVES>(A.p0, A.ld_bytes, 4, row0, A.nrows, k0, lds, wave, lane);
    if constexpr (PART != 0) dma_tile<4, S::BN, S::WAVES>(B.p0, B.ld_bytes, 4, col0, B.nrows, k0, lds + S::A_BYTES, wave, lane);
  } else {
    if constexpr (PART != 1) {
      dma_tile<2, S::BM, S::WAVES, MF>(A.p0, A.ld_bytes, A.kb, row0, A.nrows, k0, lds, wave, lane);
      if constexpr (MODE == MDG_PREC_BF16X3) dma_tile<2, S::BM, S::WAVES, MF>(A.p1, A.ld_bytes, A.kb, row0, A.nrows, k0, lds + S::A_LO, wave, lane);
    }
    if constexpr (PART != 0) {
      dma_tile<2, S::BN, S::WAVES, MF>(B.p0, B.ld_bytes, B.kb, col0, B.nrows, k0, lds + kBOffset<MODE, S>, wave, lane);
      if constexpr (MODE == MDG_PREC_BF16X3) dma_tile<2, S::BN, S::WAVES, MF>(B.p1, B.ld_bytes, B.kb, col0, B.nrows, k0, lds + S::A_BYTES + S::B_LO, wave, lane);
    }
  }
}

// One of the wave's pieces of a bf16 tile, and the 8 issue slots of a k tile on the 16x16x32 path: the LDS-DMA of the next tile
// is spread over the MFMA groups of the current one (slot = A-fragment group), one piece per slot, instead of arriving at the
// memory system as one burst from every CU at the same moment (measured on the 2048-deep blocks: -4..-7 % time).
template <int ROWS, int WAVES, int MF>
__device__ __forceinline__ void dma_piece(const char* base, int64_t ld_bytes, int kb, int64_t row0, int64_t nrows, int64_t k0, char* lds,
                                          int wave, int lane, int i) {
  static_assert(ROWS * BK * 2 / 1024 / WAVES == 2, "two pieces per wave and tile plane");
  const int p = wave + WAVES * i;
  const int row = 16 * p + (lane >> 2);
  const int c = (lane & 3) ^ (MF == 16 ? ((row >> 1) & 3) : ((row >> 2) & 3));
  int64_t gr = row0 + row;
  gr = gr < nrows ? gr : nrows - 1;
  glds16(base + gr * ld_bytes + k0 * kb + c * 16, lds_addr(lds + p * 1024));
}

template <int MODE, class S, int MF>
__device__ __forceinline__ void dma_slot(const Operand& A, const Operand& B, int64_t row0, int64_t col0, int64_t k0, char* lds,
                                         int wave, int lane, int slot) {
  static_assert(MODE != MDG_PREC_F32, "16-bit modes");
  int plane, i;                        // plane 0: A hi, 1: A lo, 2: B hi, 3: B lo
  if constexpr (MODE == MDG_PREC_BF16X3) { plane = slot >> 1; i = slot & 1; }
  else {
    if (slot & 1) return;
    plane = (slot >> 2) * 2;
    i = (slot >> 1) & 1;
  }
  switch (plane) {
    case 0: dma_piece<S::BM, S::WAVES, MF>(A.p0, A.ld_bytes, A.kb, row0, A.nrows, k0, lds, wave, lane, i); break;
    case 1: dma_piece<S::BM, S::WAVES, MF>(A.p1, A.ld_bytes, A.kb, row0, A.nrows, k0, lds + S::A_LO, wave, lane, i); break;
    case 2: dma_piece<S::BN, S::WAVES, MF>(B.p0, B.ld_bytes, B.kb, col0, B.nrows, k0, lds + kBOffset<MODE, S>, wave, lane, i); break;
    default: dma_piece<S::BN, S::WAVES, MF>(B.p1, B.ld_bytes, B.kb, col0, B.nrows, k0, lds + S::A_BYTES + S::B_LO, wave, lane, i); break;
  }
}

template <int MODE, class S, class Mid>
__device__ __forceinline__ void mma_stage(const char* la, const char* lb, int wr, int wc, int r, int h,
                                          f32x16 (&acc)[S::MT][S::NT_], Mid&& mid) {
  constexpr int MT = S::MT, NT_ = S::NT_;
  mid(0);
  if constexpr (MODE == MDG_PREC_F32) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 a[MT], b[NT_];
#pragma unroll
      for (int t = 0; t < MT; ++t) a[t] = *reinterpret_cast<const f32x4*>(la + off_f32(wr * 32 * MT + t * 32 + r, 2 * q + h));
#pragma unroll
      for (int t = 0; t < NT_; ++t) b[t] = *reinterpret_cast<const f32x4*>(lb + off_f32(wc * 32 * NT_ + t * 32 + r, 2 * q + h));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT_; ++nt)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][e], b[nt][e], acc[mt][nt], 0, 0, 0);
      if (q == 1) mid(1);
    }
  } else {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 ah[MT], al[MT], bh[NT_], bl[NT_];
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int oa = off_bf16(wr * 32 * MT + t * 32 + r, 2 * s + h);
        ah[t] = *reinterpret_cast<const bf16x8*>(la + oa);
        if constexpr (MODE == MDG_PREC_BF16X3) al[t] = *reinterpret_cast<const bf16x8*>(la + S::A_LO + oa);
      }
#pragma unroll
      for (int t = 0; t < NT_; ++t) {
        const int ob = off_bf16(wc * 32 * NT_ + t * 32 + r, 2 * s + h);
        bh[t] = *reinterpret_cast<const bf16x8*>(lb + ob);
        if constexpr (MODE == MDG_PREC_BF16X3) bl[t] = *reinterpret_cast<const bf16x8*>(lb + S::B_LO + ob);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT_; ++nt) {
          if constexpr (MODE == MDG_PREC_BF16X3) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
          }
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
        }
      if (s == 0) mid(1);
    }
  }
}

// The same k tile on the 16x16x32 MFMA (16-bit modes): 8 passes for 16 Kflop against 16 passes for 32 Kflop on the 32x32x16 form,
// i.e. the same matrix-pipe cycles, but each instruction switches a quarter of the accumulator registers; under the package
// power limit that buys clock (measured on the row-statistics head, which has this loop shape: +13 %).  A wave's 32*MT x 32*NT
// patch becomes 2MT x 2NT tiles of 16x16; lane (c = lane & 15, g = lane >> 4) holds row c / column c, k chunk g of a fragment
// and rows 4g..4g+3 of column c of an accumulator tile.  B fragments stay in registers over the two halves of the A tiles.
template <int MODE, class S, class Mid>
__device__ __forceinline__ void mma_stage16(const char* la, const char* lb, int wr, int wc, int c, int g,
                                            f32x4 (&acc)[2 * S::MT][2 * S::NT_], Mid&& mid) {
  constexpr int MT2 = 2 * S::MT, NT2 = 2 * S::NT_;
  bf16x8 bh[NT2], bl[NT2];
#pragma unroll
  for (int t = 0; t < NT2; ++t) {
    const int ob = off_bf16_m16(wc * 16 * NT2 + t * 16 + c, g);
    bh[t] = *reinterpret_cast<const bf16x8*>(lb + ob);
    if constexpr (MODE == MDG_PREC_BF16X3) bl[t] = *reinterpret_cast<const bf16x8*>(lb + S::B_LO + ob);
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    bf16x8 ah[MT2 / 2], al[MT2 / 2];
#pragma unroll
    for (int t = 0; t < MT2 / 2; ++t) {
      const int oa = off_bf16_m16(wr * 16 * MT2 + (half * (MT2 / 2) + t) * 16 + c, g);
      ah[t] = *reinterpret_cast<const bf16x8*>(la + oa);
      if constexpr (MODE == MDG_PREC_BF16X3) al[t] = *reinterpret_cast<const bf16x8*>(la + S::A_LO + oa);
    }
#pragma unroll
    for (int t = 0; t < MT2 / 2; ++t) {
#pragma unroll
      for (int u = 0; u < 8 / MT2; ++u) mid((half * (MT2 / 2) + t) * (8 / MT2) + u);      // 8 issue slots per k tile
#pragma unroll
      for (int nt = 0; nt < NT2; ++nt) {
        f32x4& a = acc[half * (MT2 / 2) + t][nt];
        if constexpr (MODE == MDG_PREC_BF16X3) {
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[t], bh[nt], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bl[nt], a, 0, 0, 0);
        }
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bh[nt], a, 0, 0, 0);
      }
    }
  }
}

struct LinearArgs {
  Operand A, B;            // K is a multiple of 32 in both images (zero padded by the pre-pass when needed)
  float* y; int64_t ldy;
  const float* bias; const float* scale; const float* shift;
  const float* res; int64_t ldr;
  float alpha, beta;
  int act;
  int64_t M, N, K;
  int tiles_x, tiles_y, swizzle;   // swizzle: XCD-aware tile order (see tile_of)
  int ty_base;                     // the launch covers tile rows ty_base .. ty_base + tiles_y - 1 of the problem (row split of one product
                                   // over two launches: whole rounds of 256-tiles, the rest as 128-tiles; see launch_linear_core)
  int vec_y, vec_r;                // y / residual rows may be accessed 16 B at a time (alignment and row stride)
  const int64_t* tiles;            // grouped launch: [n_tiles][kGroupTileWords] tile descriptors (device), else null
  const float* a_raw;              // 128-tile kernel, 16-bit modes: x itself (fp32 rows, stride a_ldx floats, a_k valid columns) when no
  int64_t a_ldx, a_k;              // operand image of it exists -- the kernel rounds / splits it while staging (no pre-pass over x)
  int sk_tiles;                    // 256-tile kernel, stream-K hybrid: the last sk_tiles tiles are shared k tile by k tile (0: one tile per workgroup)
  char* sk_ws;                     // its workspace: accumulator slots | flags
  // training: dropout of the activated output inside the epilogue (before the residual is added): element (m, n) is kept iff
  // mdg_keep(drop_seed, m * N + n, drop_thr) -- the mask mdg_dropout applies to a contiguous [M, N] tensor -- and scaled by drop_scale
  uint64_t drop_seed; uint32_t drop_thr; float drop_scale;
  // 128-tile kernel, split-K (ks_count > 1, grid.y = splits): split z multiplies the k range [z * ks_len, min(K, (z + 1) * ks_len)) and
  // writes the plain partial product to y + z * ks_stride (no bias / activation / residual: mdg_linear_tn sums the partials afterwards)
  int ks_count; int64_t ks_len, ks_stride;
};

// Workgroups are dealt to the 8 XCDs round-robin in dispatch order, and each XCD has its own L2.  With the plain
// blockIdx -> tile map the ~32 workgroups resident on one XCD are spread over the whole tile grid and share few operand
// bands.  Swizzled map: XCD x owns a contiguous run of the tile sequence, and that sequence walks the grid in groups of
// 8 tile-rows, column by column, so that the resident set is an ~8 x 4 patch: 12 operand bands feed 32 tiles.
__device__ __forceinline__ bool tile_of(const LinearArgs& p, int id, int& tx, int& ty) {
  if (!p.swizzle) { tx = id % p.tiles_x; ty = id / p.tiles_x; return ty < p.tiles_y; }
  const int total = p.tiles_x * p.tiles_y;
  const int per_xcd = (total + 7) >> 3;
  const int t = (id & 7) * per_xcd + (id >> 3);
  if ((id >> 3) >= per_xcd || t >= total) return false;
  constexpr int GM = 8;
  const int group = t / (GM * p.tiles_x), in = t - group * GM * p.tiles_x;
  const int rows = p.tiles_y - group * GM < GM ? p.tiles_y - group * GM : GM;
  ty = group * GM + in % rows;
  tx = in / rows;
  return true;
}

// The same walk for the persistent (stream-K) launch, where every id in [0, total) must name a tile: the XCD runs are cut
// bijectively (the first total % 8 XCDs own one more tile).
__device__ __forceinline__ void tile_of_bijective(const LinearArgs& p, int id, int& tx, int& ty) {
  const int total = p.tiles_x * p.tiles_y;
  const int q = total >> 3, r = total & 7, xcd = id & 7;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  constexpr int GM = 8;
  const int group = t / (GM * p.tiles_x), in = t - group * GM * p.tiles_x;
  const int rows = p.tiles_y - group * GM < GM ? p.tiles_y - group * GM : GM;
  ty = group * GM + in % rows;
  tx = in / rows;
}

// the part of the next tile's LDS-DMA that goes out at issue slot `slot` of the current tile's MFMA sequence
template <int MODE, class S, int MF>
__device__ __forceinline__ void issue(const Operand& A, const Operand& B, int64_t row0, int64_t col0, int64_t k0, char* lds, int wave,
                                      int lane, int slot) {
  if constexpr (MF == 16 && MODE == MDG_PREC_BF16X3) {
    // two stages only: the pieces must be out early enough to land before the next tile starts -- two per slot over the
    // first half of the MFMA sequence (all eight over the whole sequence was slower again)
    if (slot < 4) {
      dma_slot<MODE, S, MF>(A, B, row0, col0, k0, lds, wave, lane, 2 * slot);
      dma_slot<MODE, S, MF>(A, B, row0, col0, k0, lds, wave, lane, 2 * slot + 1);
    }
  } else if constexpr (MF == 16) dma_slot<MODE, S, MF>(A, B, row0, col0, k0, lds, wave, lane, slot);
  else if (slot == 0) dma_stage<MODE, S, MF, 0>(A, B, row0, col0, k0, lds, wave, lane);
  else dma_stage<MODE, S, MF, 1>(A, B, row0, col0, k0, lds, wave, lane);
}

// Pipeline: one raw barrier per k-tile.  Top of tile kt: wait for this wave's DMA pieces of tile kt (the only
// vector-memory ops in flight), barrier (=> every wave's pieces landed, every wave finished reading tile kt-1),
// issue the DMA of tile kt+1 into the other buffer, then the MFMAs of tile kt run under that DMA.
// the B half of `issue` (the A operand comes from registers in the raw-x variant)
template <int MODE, class S, int MF>
__device__ __forceinline__ void issue_b(const Operand& A, const Operand& B, int64_t col0, int64_t k0, char* lds, int wave, int lane, int slot) {
  if constexpr (MF == 16 && MODE == MDG_PREC_BF16X3) { if (slot < 4) dma_slot<MODE, S, MF>(A, B, 0, col0, k0, lds, wave, lane, 4 + slot); }
  else if constexpr (MF == 16) { if (slot < 2) dma_slot<MODE, S, MF>(A, B, 0, col0, k0, lds, wave, lane, 4 + 2 * slot); }
  else if (slot == 0) dma_stage<MODE, S, MF, 1>(A, B, 0, col0, k0, lds, wave, lane);
}

template <int MODE, class S, int MF, bool ARAW, class Mma>
__device__ __forceinline__ void k_loop(const LinearArgs& p, int64_t row0, int64_t col0, char* smem, int wave, int lane, Mma&& mma) {
  const int nk = static_cast<int>(p.K / BK);
  if constexpr (ARAW) {
    // x itself is the A operand: thread (row = tid / 2, half = tid & 1) loads its 16 floats of the next k tile straight from
    // the fp32 rows while the current tile multiplies, then rounds / splits them (what the pre-pass would have written: same
    // values, same results bit for bit) into the other stage's bf16 tile.  The weights stay an LDS-DMA of their cached image.
    static_assert(S::THREADS == 256 && S::BM == 128 && MODE != MDG_PREC_F32, "one half row per thread");
    const int tid = wave * 64 + lane, arow = tid >> 1, ahalf = tid & 1;
    int64_t gr = row0 + arow;
    gr = gr < p.A.nrows ? gr : p.A.nrows - 1;
    const float* const src = p.a_raw + gr * p.a_ldx + 16 * ahalf;
    const int64_t kvalid = p.a_k - 16 * ahalf;               // floats of this thread's half row that exist
    f32x4 st[4];
    const auto load_a = [&](int kt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t k = static_cast<int64_t>(kt) * BK + 4 * i;
        st[i] = k < kvalid ? *reinterpret_cast<const f32x4*>(src + k) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    const auto store_a = [&](char* la) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bf16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          __bf16 a_, b_;
          mdg_split_bf16(st[2 * c + (e >> 2)][e & 3], a_, b_);
          hi[e] = a_;
          lo[e] = b_;
        }
        const int off = MF == 16 ? off_bf16_m16(arow, 2 * ahalf + c) : off_bf16(arow, 2 * ahalf + c);
        *reinterpret_cast<bf16x8*>(la + off) = hi;
        if constexpr (MODE == MDG_PREC_BF16X3) *reinterpret_cast<bf16x8*>(la + S::A_LO + off) = lo;
      }
    };
    load_a(0);
    dma_stage<MODE, S, MF, 1>(p.A, p.B, 0, col0, 0, smem, wave, lane);
    store_a(smem);
    for (int kt = 0; kt < nk; ++kt) {
      char* const cur = smem + (kt & 1) * S::STAGE;
      char* const nxt = smem + ((kt + 1) & 1) * S::STAGE;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's weight DMA and its A-tile writes
      __builtin_amdgcn_s_barrier();
      const bool more = kt + 1 < nk;
      const int64_t kn = static_cast<int64_t>(kt + 1) * BK;
      if (more) load_a(kt + 1);
      mma(cur, cur + kBOffset<MODE, S>, [&](int slot) { if (more) issue_b<MODE, S, MF>(p.A, p.B, col0, kn, nxt, wave, lane, slot); });
      if (more) store_a(nxt);
    }
    return;
  }
  if constexpr (kStages<MODE> == 4) {

    // four-stage ring, three k tiles in flight.  Every wave issues exactly 4 LDS-DMA pieces per tile (static_assert), its only
    // vector-memory operations, and they retire in issue order: `vmcnt(8)` at the top of tile kt leaves the pieces of tiles
    // kt+1 and kt+2 in flight.  The buffer refilled at the top of kt held tile kt-1, which every wave finished reading before
    // it arrived at this barrier.
    static_assert((S::BM + S::BN) * BK * 2 / 1024 / S::WAVES == 4, "counted waits below assume 4 pieces per wave and tile");
    constexpr int SB = kStageBytes<MODE, S>;
    for (int t = 0; t < 3 && t < nk; ++t) dma_stage<MODE, S, MF>(p.A, p.B, row0, col0, static_cast<int64_t>(t) * BK, smem + t * SB, wave, lane);
    for (int kt = 0; kt < nk; ++kt) {
      char* const cur = smem + (kt & 3) * SB;
      if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const bool more = kt + 3 < nk;
      char* const dst = smem + ((kt + 3) & 3) * SB;
      const int64_t kn = static_cast<int64_t>(kt + 3) * BK;
      mma(cur, cur + kBOffset<MODE, S>, [&](int slot) { if (more) issue<MODE, S, MF>(p.A, p.B, row0, col0, kn, dst, wave, lane, slot); });
    }
  } else {
    dma_stage<MODE, S, MF>(p.A, p.B, row0, col0, 0, smem, wave, lane);
    for (int kt = 0; kt < nk; ++kt) {
      char* const cur = smem + (kt & 1) * S::STAGE;
      char* const nxt = smem + ((kt + 1) & 1) * S::STAGE;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const bool more = kt + 1 < nk;
      const int64_t kn = static_cast<int64_t>(kt + 1) * BK;
      mma(cur, cur + S::A_BYTES, [&](int slot) { if (more) issue<MODE, S, MF>(p.A, p.B, row0, col0, kn, nxt, wave, lane, slot); });
    }
  }
}

// ---- epilogue -----------------------------------------------------------------------------------------------------------------
// The accumulators hold columns across lanes (a store straight from them writes 64- or 128-byte row pieces, and every workgroup
// of a launch reaches its epilogue at about the same time, so the write burst is paid in full).  Instead each wave turns its
// patch through LDS -- the staging buffers are free once the k loop is over: 16 KB per wave = 64 rows x 64 columns fp32 -- and
// writes row-major, 16 B per lane, 256 contiguous bytes per row per instruction; bias / BN scale+shift / residual are read the
// same way.  Slab layout: row-major with the 16-float column block XORed by (row >> 2) & 3, conflict-free for both accumulator
// layouts on the way in (a 32x32 tile writes 32 columns of one row per half-wave, a 16x16 tile 16 columns of two rows four
// apart) and for the float4 rows on the way out.
__device__ __forceinline__ int slab_off(int row, int col) { return row * 64 + ((((col >> 4) ^ ((row >> 2) & 3)) << 4) | (col & 15)); }

__device__ __forceinline__ float finish(const LinearArgs& p, float v, float bias, float scale, float shift) {
  float val = v + bias;
  if (p.scale) val = val * scale + shift;
  val = apply_act(val, p.act);
  if (p.alpha != 1.0f) val *= p.alpha;
  return val;
}

// 64 x 64 slab -> y[m0.., n0..]
__device__ __forceinline__ void slab_to_global(const LinearArgs& p, const float* slab, int64_t m0, int64_t n0, int lane) {
  const int c4 = lane & 15;
  const int64_t n = n0 + 4 * c4;
  if (n >= p.N) return;
  const bool full = n + 3 < p.N;
  float bias[4] = {0.f, 0.f, 0.f, 0.f}, scale[4] = {1.f, 1.f, 1.f, 1.f}, shift[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (n + e < p.N) {
      if (p.bias) bias[e] = p.bias[n + e];
      if (p.scale) { scale[e] = p.scale[n + e]; shift[e] = p.shift[n + e]; }
    }
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int row = it * 4 + (lane >> 4);
    const int64_t m = m0 + row;
    const f32x4 a = *reinterpret_cast<const f32x4*>(slab + slab_off(row, 4 * c4));
    if (m >= p.M) continue;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = finish(p, a[e], bias[e], scale[e], shift[e]);
    if (p.drop_scale != 0.f) {
      const uint64_t i0 = static_cast<uint64_t>(m) * static_cast<uint64_t>(p.N) + static_cast<uint64_t>(n);
      if ((p.N & 3) == 0) {                               // n is a multiple of 4: the lane's four columns are one hash group
        const uint64_t word = mdg_keep_word(p.drop_seed, i0 >> 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = mdg_keep_field(word, e, p.drop_thr) ? o[e] * p.drop_scale : 0.f;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = mdg_keep(p.drop_seed, i0 + e, p.drop_thr) ? o[e] * p.drop_scale : 0.f;
      }
    }
    if (p.res) {
      const float* rr = p.res + m * p.ldr + n;
      if (full && p.vec_r) {
        const f32x4 rv = *reinterpret_cast<const f32x4*>(rr);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += p.beta * rv[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.N) o[e] += p.beta * rr[e];
      }
    }
    float* yr = p.y + m * p.ldy + n;
    if (full && p.vec_y) *reinterpret_cast<f32x4*>(yr) = o;
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < p.N) yr[e] = o[e];
    }
  }
}

// MF = edge of the MFMA tile: 32 (fp32 always; 16-bit modes on request) or 16 (16-bit modes).
// Grouped launch (GROUPED): G independent products y_g = x_g W_g^T that share K and one launch.  The x rows of all groups are
// stacked in one operand image and so are the W rows (and the biases); a device table lists the output tiles, one descriptor
// each.  Used where a layer exists once per node type of the knowledge graph (ten small GEMMs per HGT conv otherwise).
//   words: 0 a_row0  1 a_row_end  2 b_row0  3 b_row_end   first row / end of the group's rows in the stacked images (tile origin in 0, 2)
//          4 m_base  5 n_base     the group's first x row / first W row: output element (m, n) is y[y_off + (m - m_base) * ldy + (n - n_base)]
//          6 y_off   7 ldy   8 res_off (-1: none)  9 ldr   10 alpha (float bits) | beta (float bits) << 32
constexpr int kGroupTileWords = 12;

template <int MODE, class S, int MF, bool GROUPED = false, bool ARAW = false>
__global__ __launch_bounds__(S::THREADS, 2) void linear_kernel(const LinearArgs pk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [stages][A tile | B tile]; reused as [wave][64][64] fp32 by the epilogue
  constexpr int MT = S::MT, NT_ = S::NT_;
  static_assert(MF == 32 || (MF == 16 && MODE != MDG_PREC_F32), "the 16x16x32 form exists for the 16-bit modes only");
  static_assert(NT_ == 2 && MT % 2 == 0 && S::WAVES * 16384 <= 2 * S::STAGE, "the epilogue turns 64 x 64 patches through 16 KB of LDS per wave");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wr = wave / S::WN, wc = wave % S::WN;
  LinearArgs p = pk;
  int64_t col0, row0;
  if constexpr (GROUPED) {
    const int64_t* e = pk.tiles + static_cast<int64_t>(blockIdx.x) * kGroupTileWords;      // workgroup-uniform (scalar loads)
    row0 = e[0];
    col0 = e[2];
    p.A.nrows = e[1];
    p.B.nrows = e[3];
    p.M = e[1];
    p.N = e[3];
    p.ldy = e[7];
    p.y = pk.y + e[6] - e[4] * e[7] - e[5];              // so that y[m * ldy + n] with the stacked (m, n) lands in the group's block
    p.ldr = e[9];
    p.res = (pk.res && e[8] >= 0) ? pk.res + e[8] - e[4] * e[9] - e[5] : nullptr;
    const unsigned long long ab = static_cast<unsigned long long>(e[10]);
    p.alpha = __builtin_bit_cast(float, static_cast<unsigned>(ab & 0xffffffffull));
    p.beta = __builtin_bit_cast(float, static_cast<unsigned>(ab >> 32));
  } else {
    int tx, ty;
    if (!tile_of(p, static_cast<int>(blockIdx.x), tx, ty)) return;   // workgroup-uniform
    ty += p.ty_base;
    col0 = static_cast<int64_t>(tx) * S::BN;
    row0 = static_cast<int64_t>(ty) * S::BM;
    if (pk.ks_count > 1) {                                           // split-K: this workgroup's k range and partial result
      const int64_t z = blockIdx.y, k_begin = z * pk.ks_len;
      p.K = pk.K - k_begin < pk.ks_len ? pk.K - k_begin : pk.ks_len;
      p.A.p0 += k_begin * p.A.kb;
      p.B.p0 += k_begin * p.B.kb;
      if (p.A.p1) p.A.p1 += k_begin * p.A.kb;
      if (p.B.p1) p.B.p1 += k_begin * p.B.kb;
      p.y += z * pk.ks_stride;
    }
  }
  float* const slab = reinterpret_cast<float*>(smem) + wave * 4096;
  const int64_t pm0 = row0 + wr * 32 * MT, pn0 = col0 + wc * 64;

  if constexpr (MF == 16) {
    const int c = lane & 15, g = lane >> 4;
    f32x4 acc[2 * MT][2 * NT_];
#pragma unroll
    for (int a = 0; a < 2 * MT; ++a)
#pragma unroll
      for (int b = 0; b < 2 * NT_; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    k_loop<MODE, S, MF, ARAW>(p, row0, col0, smem, wave, lane,
                              [&](const char* la, const char* lb, auto&& mid) { mma_stage16<MODE, S>(la, lb, wr, wc, c, g, acc, mid); });
    __syncthreads();                                      // every wave is done with the staging buffers
#pragma unroll
    for (int pass = 0; pass < MT / 2; ++pass) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int v = 0; v < 4; ++v) slab[slab_off(mt * 16 + 4 * g + v, nt * 16 + c)] = acc[pass * 4 + mt][nt][v];
      slab_to_global(p, slab, pm0 + pass * 64, pn0, lane);
    }
  } else {
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[MT][NT_];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT_; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
    k_loop<MODE, S, MF, ARAW && MODE != MDG_PREC_F32>(p, row0, col0, smem, wave, lane,
                                                      [&](const char* la, const char* lb, auto&& mid) { mma_stage<MODE, S>(la, lb, wr, wc, r, h, acc, mid); });
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < MT / 2; ++pass) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int v = 0; v < 16; ++v)
            slab[slab_off(mt * 32 + (v & 3) + 8 * (v >> 2) + 4 * h, nt * 32 + r)] = acc[pass * 2 + mt][nt][v];
      slab_to_global(p, slab, pm0 + pass * 64, pn0, lane);
    }
  }
}

// ---- 256 x 256 "ping-pong" dense block (16-bit modes) -----------------------------------------------------------------------
// The large blocks of the path (fusion transformer at d = 2048, chemCPA / cv first layers over 65k rows).  One 8-wave workgroup
// per CU, waves as 2 (M) x 4 (N), each wave a 128 x 64 patch = 8 x 4 accumulator tiles of the 16x16x32 MFMA (128 VGPRs).
//
// LDS: two stages of four 16-KB HALF-TILES, each 128 rows x 128 B: A0 | B0 | B1 | A1.  A row holds one k tile of an image row:
// bf16: 64 k; bf16x3: 32 k as 64 B hi | 64 B lo (the image interleaves them, so both modes stage identical bytes and differ
// only in the MFMA sequence: 2 products over k 0..31, 32..63 against lo.hi + hi.lo + hi.hi).  Half mh of A holds rows
// 128 wr + 64 mh + (0..63) of the tile for wr = 0, 1 (so every wave reads every half: rows 64 wr .. of it), half nh of B the
// columns 64 wc + 32 nh + (0..31): a wave's patch is contiguous, and quadrant (mh, nh) of it needs exactly A[mh] and B[nh].
// 16-byte chunk j of LDS row r sits at chunk j ^ ((r >> 1) & 7): the 16 lanes of every ds_read_b128 lane group then cover all
// 64 banks (checked exhaustively against the guide's lane grouping, scripts/lds_swizzle_check.py); the LDS-DMA writes rows
// linearly, so the XOR is applied to each lane's SOURCE chunk.
//
// Schedule per k tile t (stage t & 1), four phases, one quadrant of MFMAs each (16 / 24 instructions):
//   P1  read B0, A0 -> (0,0)     issues B1[t+1]          P3  read A1 -> (1,1)     issues A0[t+2]
//   P2  read B1     -> (0,1)     issues A1[t+1]          P4  (b0 kept) -> (1,0)   issues B0[t+2], waits vmcnt(4)
// Each phase is  { fragment reads, LDS-DMA issue } s_barrier { MFMAs } s_barrier.  Waves 4-7 run ONE barrier behind waves 0-3
// (an extra barrier in front, one fewer behind): a SIMD hosts one wave of each half, so while one feeds the matrix pipe the
// other issues its LDS reads and DMA, and the pipe sees back-to-back MFMA clusters.
// Hazards.  A half-tile is refilled two phases (four barriers) after the phase that read it, so every wave's reads have been
// waited for (the MFMAs of the reading phase consumed them) two barriers before the first refill is issued.  Every wave issues
// exactly two LDS-DMA pieces per half-tile, in order H(g + 6) at phase g; P4's counted wait (after its own issue) leaves the
// youngest two half-tiles (A0, B0 of tile t+2) in flight and retires all of tile t+1; both wave halves have passed that wait
// before the barrier that opens tile t+1's first read (the late half waits one barrier later and reads one barrier later).
namespace pp {
constexpr int HT = 16384, STAGE = 4 * HT, LDS_BYTES = 2 * STAGE;
constexpr int OFF_A0 = 0, OFF_B0 = HT, OFF_B1 = 2 * HT, OFF_A1 = 3 * HT;
constexpr int BM = 256, BN = 256, THREADS = 512;

// LDS-DMA with the global address as SGPR base + 32-bit lane offset: one VGPR per piece instead of two
__device__ __forceinline__ void glds16_so(const char* sbase, unsigned voff, unsigned lds_dst_uniform) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst_uniform);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(dst)
               : "memory");
}
}  // namespace pp

// Work decomposition.  One workgroup per CU and a tile that takes tens of microseconds: with T tiles over P workgroups the last
// of ceil(T / P) rounds runs partly empty (704 tiles of the [22464, 2048] blocks: 2.75 rounds cost 3; the 352 tiles of the FFN's
// first block: 1.4 cost 2).  Stream-K hybrid (p.sk_tiles > 0, grid = P persistent workgroups): the first T - sk_tiles tiles
// -- whole rounds -- are dealt out tile by tile, then the k tiles of the remaining sk_tiles tiles are laid end to end and cut into
// P equal runs.  A workgroup whose run starts INSIDE a tile computes that piece first and leaves its accumulators in its slot
// of the workspace (the only piece it ever publishes); the workgroup that owns a tile's k = 0 computes its piece last, adds the
// slots of the workgroups after it that cover the rest of the tile -- they finished those pieces long before, at the head of
// their runs -- and runs the epilogue.  A workgroup only ever waits for HIGHER-numbered workgroups, none of which waits for
// it; the grid is at most the number of CUs (one workgroup fits a CU), so every workgroup is resident as soon as a CU is free.
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility; cdna_hip_programming.md, in-launch split-K reduction): write-through
// (sc1) 16-byte stores of the slot -> every wave waits vmcnt(0) -> barrier -> one lane: relaxed agent-scope flag store; consumer:
// one lane polls the flag (relaxed, agent scope, s_sleep between polls, bounded), agent-scope acquire fence, vmcnt(0), barrier,
// plain loads.
// The split tiles' sums are grouped differently from an unsplit tile's (fp32 rounding of the last place); the cut points depend
// on the shape and P only, so a call is reproducible bit for bit.
namespace pp {
constexpr int SLOT_FLOATS = BM * BN;                       // one workgroup's accumulators
constexpr int MAX_WG = 256;
constexpr unsigned SPIN_LIMIT = 1u << 21;                  // ~2 s of polling: a lost hand-off must not hang the card
inline size_t sk_bytes() { return static_cast<size_t>(MAX_WG) * SLOT_FLOATS * 4 + 4096; }   // slots | flags[MAX_WG] | error word
}  // namespace pp

template <int MODE, bool SK>
__global__ __launch_bounds__(pp::THREADS, 2) void linear_pp_kernel(const LinearArgs pk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][A0 | B0 | B1 | A1]; reused as [wave][64][64] fp32 by the epilogue
  static_assert(MODE == MDG_PREC_BF16 || MODE == MDG_PREC_BF16X3, "16-bit operand modes");
  const LinearArgs& p = pk;
  const int nk = static_cast<int>(MODE == MDG_PREC_BF16 ? p.K / 64 : p.K / 32);
  using Q0 = std::integral_constant<int, 0>; using Q1 = std::integral_constant<int, 1>;
  using Q2 = std::integral_constant<int, 2>; using Q3 = std::integral_constant<int, 3>;
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  // Lane-dependent values (fragment offsets, DMA offsets, roles) are derived afresh inside every piece from an opaque copy of the
  // thread id: kept live across the persistent launch's outer loop they cost 30 spilled VGPRs, and a kernel that touches scratch
  // ran its tiles at half speed.
  struct Lane { int tid, lane, wave, wr, wc, c, g, fo0, fo1, fa, fb; unsigned lds0; };
  auto lane_roles = [&]() {
    Lane L;
    int t = threadIdx.x;
    if constexpr (SK) asm volatile("" : "+v"(t));
    L.tid = t; L.lane = t & 63;
    L.wave = __builtin_amdgcn_readfirstlane(t >> 6);
    L.wr = L.wave >> 2; L.wc = L.wave & 3;
    L.c = L.lane & 15; L.g = L.lane >> 4;                     // lane (c = row / column in a 16-wide tile, g = 16-byte k chunk)
    const int x = (L.c >> 1) & 7;
    L.fo0 = L.c * 128 + ((L.g ^ x) << 4); L.fo1 = L.c * 128 + (((4 + L.g) ^ x) << 4);   // chunk g / 4 + g of row c, swizzled
    L.fa = (64 * L.wr) * 128; L.fb = (32 * L.wc) * 128;         // + OFF_{A,B}{half} + tile * 2048
    L.lds0 = lds_addr(smem) + static_cast<unsigned>(L.wave) * 2048u;
    return L;
  };

  f32x4 acc[8][4];

  // accumulators of tile (tx, ty) over k tiles [kb, ke)
  auto run_piece = [&](int tx, int ty, int kb, int ke) {
    const Lane L = lane_roles();
    const int lane = L.lane, wave = L.wave, wr = L.wr, fo0 = L.fo0, fo1 = L.fo1, fa = L.fa, fb = L.fb;
    const unsigned lds0 = L.lds0;
    const int64_t col0 = static_cast<int64_t>(tx) * pp::BN, row0 = static_cast<int64_t>(ty) * pp::BM;
    // LDS-DMA sources: piece i of this wave covers LDS rows 8 (2 wave + i) .. + 7 of a half-tile; lane -> (row, chunk).  Offsets
    // are relative to the tile's first image row (32 bits: 256 rows x a row of at most a few hundred KB)
    unsigned off_a[2][2], off_b[2][2];                        // [half][piece]
    const int64_t a_rows = p.A.nrows - row0, b_rows = p.B.nrows - col0;   // rows of the image at / below the tile origin (>= 1)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 8 * (2 * wave + i) + (lane >> 3);
      const unsigned chunk = static_cast<unsigned>((lane & 7) ^ ((r >> 1) & 7)) << 4;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int64_t ta = 128 * (r >> 6) + 64 * h + (r & 63), tb = 64 * (r >> 5) + 32 * h + (r & 31);
        ta = ta < a_rows ? ta : a_rows - 1;                   // rows past the end re-read the last row (never stored)
        tb = tb < b_rows ? tb : b_rows - 1;
        off_a[h][i] = static_cast<unsigned>(ta * p.A.ld_bytes) + chunk;
        off_b[h][i] = static_cast<unsigned>(tb * p.B.ld_bytes) + chunk;
      }
    }
    const char* const a_base = p.A.p0 + row0 * p.A.ld_bytes;
    const char* const b_base = p.B.p0 + col0 * p.B.ld_bytes;
    // half-tile q (0 A0, 1 B0, 2 B1, 3 A1) of k tile kt; the stage is the parity of the tile's index WITHIN the piece
    auto dma = [&](auto qc, int kt) {
      constexpr int q = decltype(qc)::value;
      constexpr int off = q == 0 ? pp::OFF_A0 : q == 1 ? pp::OFF_B0 : q == 2 ? pp::OFF_B1 : pp::OFF_A1;
      const char* sb = ((q == 0 || q == 3) ? a_base : b_base) + static_cast<int64_t>(kt) * 128;
      const unsigned dst = lds0 + static_cast<unsigned>(((kt - kb) & 1) * pp::STAGE + off);
      const unsigned o0 = q == 0 ? off_a[0][0] : q == 3 ? off_a[1][0] : q == 1 ? off_b[0][0] : off_b[1][0];
      const unsigned o1 = q == 0 ? off_a[0][1] : q == 3 ? off_a[1][1] : q == 1 ? off_b[0][1] : off_b[1][1];
      pp::glds16_so(sb, o0, dst);
      pp::glds16_so(sb, o1, dst + 1024u);
    };
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[4][2], b0f[2][2], b1f[2][2];                    // [tile][k chunk group: bf16 k 0..31 / 32..63, bf16x3 hi / lo]
    auto read_a = [&](const char* st, int off) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        af[mt][0] = *reinterpret_cast<const bf16x8*>(st + off + fa + mt * 2048 + fo0);
        af[mt][1] = *reinterpret_cast<const bf16x8*>(st + off + fa + mt * 2048 + fo1);
      }
    };
    auto read_b = [&](const char* st, int off, bf16x8 (&bf)[2][2]) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        bf[nt][0] = *reinterpret_cast<const bf16x8*>(st + off + fb + nt * 2048 + fo0);
        bf[nt][1] = *reinterpret_cast<const bf16x8*>(st + off + fb + nt * 2048 + fo1);
      }
    };
    auto quadrant = [&](auto mhc, auto nhc, const bf16x8 (&bf)[2][2]) {
      constexpr int mh = decltype(mhc)::value, nh = decltype(nhc)::value;
      __builtin_amdgcn_s_setprio(1);
      if constexpr (MODE == MDG_PREC_BF16) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[4 * mh + mt][2 * nh + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt][ks], bf[nt][ks], acc[4 * mh + mt][2 * nh + nt], 0, 0, 0);
      } else {
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)                        // lo.hi, hi.lo, hi.hi: per element the order of the 128-tile kernel
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[4 * mh + mt][2 * nh + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt][pr == 0 ? 1 : 0], bf[nt][pr == 1 ? 1 : 0],
                                                                                      acc[4 * mh + mt][2 * nh + nt], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    };

    // prologue: all of the first k tile, A0 / B0 of the second
    dma(Q0{}, kb); dma(Q1{}, kb); dma(Q2{}, kb); dma(Q3{}, kb);
    if (kb + 1 < ke) { dma(Q0{}, kb + 1); dma(Q1{}, kb + 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();                // the late half runs one barrier behind

    for (int t = kb; t < ke; ++t) {
      const char* const st = smem + ((t - kb) & 1) * pp::STAGE;
      // P1
      read_b(st, pp::OFF_B0, b0f);
      __builtin_amdgcn_sched_barrier(0);
      read_a(st, pp::OFF_A0);
      if (t + 1 < ke) dma(Q2{}, t + 1);
      __builtin_amdgcn_s_barrier();
      quadrant(I0{}, I0{}, b0f);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      // P2
      read_b(st, pp::OFF_B1, b1f);
      if (t + 1 < ke) dma(Q3{}, t + 1);
      __builtin_amdgcn_s_barrier();
      quadrant(I0{}, I1{}, b1f);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      // P3
      read_a(st, pp::OFF_A1);
      if (t + 2 < ke) dma(Q0{}, t + 2);
      __builtin_amdgcn_s_barrier();
      quadrant(I1{}, I1{}, b1f);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      // P4
      if (t + 2 < ke) { dma(Q1{}, t + 2); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      quadrant(I1{}, I0{}, b0f);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    __syncthreads();                                          // every wave is done with the stages (the epilogue / the next piece reuses them)
  };

  // epilogue: as the 128-tile kernel (64 x 64 patches through 16 KB of LDS per wave, row-major 16-byte stores)
  auto epilogue = [&](int tx, int ty) {
    const Lane L = lane_roles();
    const int lane = L.lane, wave = L.wave, wr = L.wr, wc = L.wc, c = L.c, g = L.g;
    float* const slab = reinterpret_cast<float*>(smem) + wave * 4096;
    const int64_t pm0 = static_cast<int64_t>(ty) * pp::BM + wr * 128, pn0 = static_cast<int64_t>(tx) * pp::BN + wc * 64;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int v = 0; v < 4; ++v) slab[slab_off(mt * 16 + 4 * g + v, nt * 16 + c)] = acc[pass * 4 + mt][nt][v];
      slab_to_global(p, slab, pm0 + pass * 64, pn0, lane);
    }
  };

  if constexpr (!SK) {                                        // one tile per workgroup
    int tx, ty;
    if (!tile_of(p, static_cast<int>(blockIdx.x), tx, ty)) return;          // workgroup-uniform
    ty += p.ty_base;
    run_piece(tx, ty, 0, nk);
    epilogue(tx, ty);
    return;
  }
  // ---- stream-K launch: the workgroup's pieces are whole tiles first (me, me + P, ...), then its run of the shared k tiles.
  // ONE instance of the main loop serves all of them. ----
  const int P = static_cast<int>(gridDim.x), me = static_cast<int>(blockIdx.x);
  const int total = p.tiles_x * p.tiles_y, dp_tiles = total - p.sk_tiles;
  unsigned* const flags = reinterpret_cast<unsigned*>(p.sk_ws + static_cast<size_t>(pp::MAX_WG) * pp::SLOT_FLOATS * 4);
  const int64_t units = static_cast<int64_t>(p.sk_tiles) * nk;
  const auto run_start = [&](int w) { return units * w / P; };
  int64_t u = run_start(me);
  const int64_t u_end = run_start(me + 1);
  int dp_id = me;
  for (;;) {
    int tx, ty, kb = 0, ke = nk, st = 0;
    if (dp_id < dp_tiles) {
      tile_of_bijective(p, dp_id, tx, ty);
      dp_id += P;
    } else if (u < u_end) {
      st = static_cast<int>(u / nk);
      kb = static_cast<int>(u - static_cast<int64_t>(st) * nk);
      const int64_t left = u_end - static_cast<int64_t>(st) * nk;
      ke = static_cast<int>(left < nk ? left : nk);
      tile_of_bijective(p, dp_tiles + st, tx, ty);
      u += ke - kb;
    } else {
      break;
    }
    run_piece(tx, ty, kb, ke);
    const int tid = lane_roles().tid;
    if (kb != 0) {
      // not the tile's first piece: publish the accumulators (this workgroup's only published piece: its run starts here).
      // Write-through (sc1) stores: no release fence -- which would write back every dirty line of this XCD's L2, the output
      // tiles of its 32 workgroups included -- just the stores' own completion before the flag.
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.sk_ws + static_cast<size_t>(me) * pp::SLOT_FLOATS * 4, 0, pp::SLOT_FLOATS * 4, 0x00020000);
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[a][b]), rs, static_cast<unsigned>(((a * 4 + b) * pp::THREADS + tid) * 16), 0, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(&flags[me], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      // the tile's first piece; when it stops short of the tile's end, add the pieces of the workgroups behind this one
      int covered = ke;
      for (int w = me + 1; covered < nk && w < P; ++w) {
        const int64_t ws0 = run_start(w), ws1 = run_start(w + 1);
        const int64_t tile_end = static_cast<int64_t>(st + 1) * nk;
        const int len = static_cast<int>((ws1 < tile_end ? ws1 : tile_end) - ws0);
        if (len <= 0) continue;                               // an empty run (more workgroups than k tiles)
        if (tid == 0) {
          unsigned spins = 0;
          while (__hip_atomic_load(&flags[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && ++spins < pp::SPIN_LIMIT) __builtin_amdgcn_s_sleep(32);
          if (spins >= pp::SPIN_LIMIT) __hip_atomic_store(&flags[pp::MAX_WG], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // error word
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const f32x4* theirs = reinterpret_cast<const f32x4*>(p.sk_ws) + static_cast<size_t>(w) * (pp::SLOT_FLOATS / 4);
#pragma unroll
        for (int a = 0; a < 8; ++a)                           // (fully unrolled: a runtime index would put acc in scratch)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] += theirs[(a * 4 + b) * pp::THREADS + tid];
        covered += len;
      }
      epilogue(tx, ty);
    }
    __syncthreads();                                          // slab / slot traffic done before the next piece's LDS-DMA
  }
}

// ---- pre-pass: both operands -> K padded to a multiple of 32 (zeros), split hi/lo bf16 or copied as fp32 ------
// grid.y = 0: x rows, 1: w rows.  One thread per 4 consecutive k.  (16-bit images pad K to a multiple of 64.)
// byte offset of the hi value of (row, k) in a bf16x3 image: per row and block of 32 k, 64 B of hi values then 64 B of lo values
__host__ __device__ __forceinline__ int64_t img_off_x3(int64_t row, int64_t k, int64_t Kp) { return row * Kp * 4 + (k >> 5) * 128 + (k & 31) * 2; }

struct PrepArgs {
  const float* src[2]; int64_t ld[2]; int64_t rows[2];
  char* dst0[2];                      // fp32 copy / bf16 image / bf16x3 image (hi | lo interleaved per 32 k)
  int64_t K, Kp;
  int bf16, x3;
};

__global__ __launch_bounds__(256) void prep_operands_kernel(const PrepArgs p) {
  const int which = blockIdx.y;
  const int64_t q = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;       // index of a 4-element group
  const int64_t per_row = p.Kp / 4;
  if (q >= p.rows[which] * per_row) return;
  const int64_t row = q / per_row, k = (q % per_row) * 4;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (k < p.K) v = *reinterpret_cast<const f32x4*>(p.src[which] + row * p.ld[which] + k);
  if (p.bf16) {
    bf16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      __bf16 a, b;
      mdg_split_bf16(v[e], a, b);
      hi[e] = a;
      lo[e] = b;
    }
    if (p.x3) {
      char* d = p.dst0[which] + img_off_x3(row, k, p.Kp);
      *reinterpret_cast<bf16x4*>(d) = hi;
      *reinterpret_cast<bf16x4*>(d + 64) = lo;
    } else {
      *reinterpret_cast<bf16x4*>(p.dst0[which] + (row * p.Kp + k) * 2) = hi;
    }
  } else {
    *reinterpret_cast<f32x4*>(p.dst0[which] + (row * p.Kp + k) * 4) = v;
  }
}

// Transposing pack: source [M rows][C columns] row-major (row stride ld); image of the TRANSPOSED operand, rows c = 0..C-1 with
// the inner index m padded to Mp (a multiple of 32), as fp32 or as bf16 hi (+ lo) planes.  64 x 64 tiles through LDS: global
// reads run along the source rows, image writes along the image rows (8 bytes per lane, bf16).  grid = (Mp/64 tiles, C tiles, operands).
struct PrepTArgs {
  const float* src[2]; int64_t ld[2]; int64_t C[2];
  char* dst0[2];
  int64_t M, Mp;
  int bf16, x3;
};

__global__ __launch_bounds__(256) void prep_transposed_kernel(const PrepTArgs p) {
  __shared__ float tile[64][65];
  const int which = blockIdx.z;
  const int64_t c0 = static_cast<int64_t>(blockIdx.y) * 64, m0 = static_cast<int64_t>(blockIdx.x) * 64;
  if (c0 >= p.C[which]) return;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;            // 64 x 4
  const float* src = p.src[which];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int64_t m = m0 + ty + 4 * i, c = c0 + tx;
    tile[ty + 4 * i][tx] = (m < p.M && c < p.C[which]) ? src[m * p.ld[which] + c] : 0.f;
  }
  __syncthreads();
  // image row c, inner index m: thread handles 4 consecutive m of one c
  const int cm = threadIdx.x >> 4, mq = (threadIdx.x & 15) * 4;        // 16 rows x 16 quads per pass
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cl = cm + 16 * i;
    const int64_t c = c0 + cl;
    if (c >= p.C[which]) continue;
    const f32x4 v = {tile[mq][cl], tile[mq + 1][cl], tile[mq + 2][cl], tile[mq + 3][cl]};
    const int64_t off = c * p.Mp + m0 + mq;
    if (p.bf16) {
      bf16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        __bf16 a, b;
        mdg_split_bf16(v[e], a, b);
        hi[e] = a;
        lo[e] = b;
      }
      if (p.x3) {
        char* d = p.dst0[which] + img_off_x3(c, m0 + mq, p.Mp);
        *reinterpret_cast<bf16x4*>(d) = hi;
        *reinterpret_cast<bf16x4*>(d + 64) = lo;
      } else {
        *reinterpret_cast<bf16x4*>(p.dst0[which] + off * 2) = hi;
      }
    } else {
      *reinterpret_cast<f32x4*>(p.dst0[which] + off * 4) = v;
    }
  }
}

// Backward pre-pass of a wide dense block, ONE read of g = dL/dy [M, N]: (a) its row image (A operand of dx = g W: inner index n,
// padded to Np), (b) its transposed image (A operand of dW = g^T x: rows n, inner index m padded to Mp), (c) partial column sums
// (bias gradient) per 64-row block: part[block][n].  The separate launches read g three times (prep_operands_kernel,
// prep_transposed_kernel, colsum_partial_kernel).  64 x 64 tiles through LDS; grid = (Mp / 64, Np / 64).
struct PrepBArgs {
  const float* g; int64_t ld;
  int64_t M, N, Np, Mp;
  char* row_img; char* t_img; float* part;
  int x3;
  uint64_t drop_seed; uint32_t drop_thr; float drop_scale;      // g = dropout-backward of the incoming gradient (mask of element m * N + n), applied on load
};

__global__ __launch_bounds__(256) void prep_backward_kernel(const PrepBArgs p) {
  __shared__ float tile[64][65];
  __shared__ float red[4][64];
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * 64, c0 = static_cast<int64_t>(blockIdx.y) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  float cs = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int64_t m = m0 + ty + 4 * i, c = c0 + tx;
    float v = (m < p.M && c < p.N) ? p.g[m * p.ld + c] : 0.f;
    if (p.drop_scale != 0.f) v = mdg_keep(p.drop_seed, static_cast<uint64_t>(m) * static_cast<uint64_t>(p.N) + static_cast<uint64_t>(c), p.drop_thr) ? v * p.drop_scale : 0.f;
    tile[ty + 4 * i][tx] = v;
    cs += v;
  }
  red[ty][tx] = cs;
  __syncthreads();
  if (p.part && ty == 0 && c0 + tx < p.N) p.part[static_cast<int64_t>(blockIdx.x) * p.N + c0 + tx] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
  const int rl = threadIdx.x >> 4, q4 = (threadIdx.x & 15) * 4;          // 16 rows x 16 quads per pass
  auto put = [&](char* img, int64_t row, int64_t inner, int64_t pitch, const f32x4 v) {
    bf16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      __bf16 a, b;
      mdg_split_bf16(v[e], a, b);
      hi[e] = a;
      lo[e] = b;
    }
    if (p.x3) {
      char* d = img + img_off_x3(row, inner, pitch);
      *reinterpret_cast<bf16x4*>(d) = hi;
      *reinterpret_cast<bf16x4*>(d + 64) = lo;
    } else {
      *reinterpret_cast<bf16x4*>(img + (row * pitch + inner) * 2) = hi;
    }
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int l = rl + 16 * i;
    if (p.row_img && m0 + l < p.M) {                                  // row image: row m, 4 consecutive n (zeros in the padding up to Np)
      const f32x4 v = {tile[l][q4], tile[l][q4 + 1], tile[l][q4 + 2], tile[l][q4 + 3]};
      put(p.row_img, m0 + l, c0 + q4, p.Np, v);
    }
    if (c0 + l < p.N) {                                               // transposed image: row n, 4 consecutive m
      const f32x4 v = {tile[q4][l], tile[q4 + 1][l], tile[q4 + 2][l], tile[q4 + 3][l]};
      put(p.t_img, c0 + l, m0 + q4, p.Mp, v);
    }
  }
}

// out[n] = sum over blocks of part[block][n], fixed order: 16 columns per workgroup, the blocks interleaved over sixteen thread rows
// (one thread per column walked all Mp / 64 partial rows alone, on N / 256 workgroups: 48-90 us of latency chain per bias gradient)
__global__ __launch_bounds__(256) void prep_backward_bias_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t nblocks, int64_t N) {
  const int e = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int64_t n = static_cast<int64_t>(blockIdx.x) * 16 + e;
  float s0 = 0.f, s1 = 0.f;
  if (n < N) {
    int64_t b = grp;
    for (; b + 16 < nblocks; b += 32) {
      s0 += part[b * N + n];
      s1 += part[(b + 16) * N + n];
    }
    for (; b < nblocks; b += 16) s0 += part[b * N + n];
  }
  __shared__ float sh[16][17];
  sh[grp][e] = s0 + s1;
  __syncthreads();
  if (grp == 0 && n < N) {
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = sh[2 * k][e] + sh[2 * k + 1][e];
    out[n] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  }
}

inline size_t al256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }
inline int64_t pad64(int64_t k) { return (k + 63) / 64 * 64; }
inline int64_t pad32(int64_t k) { return (k + 31) / 32 * 32; }

// ---- LayerNorm: one wave per row -------------------------------------------------------------
template <int VEC>   // floats per lane = 4*VEC, d <= 256*VEC
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ g,
                                                        const float* __restrict__ b, float* __restrict__ y, int64_t ldy,
                                                        int64_t rows, int d, float eps, char* __restrict__ img_hi, int img_x3) {
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
      if (y) *reinterpret_cast<f32x4*>(yr + c) = o;
      if (img_hi) {                                      // the next dense block's operand image (what its pre-pass would write)
        bf16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          __bf16 a_, b_;
          mdg_split_bf16(o[e], a_, b_);
          hi[e] = a_;
          lo[e] = b_;
        }
        if (img_x3) {
          char* o_ = img_hi + img_off_x3(row, c, d);
          *reinterpret_cast<bf16x4*>(o_) = hi;
          *reinterpret_cast<bf16x4*>(o_ + 64) = lo;
        } else {
          *reinterpret_cast<bf16x4*>(img_hi + (row * d + c) * 2) = hi;
        }
      }
    }
  }
}

}  // namespace

// inner length of an operand image: fp32 pads K to 32, the 16-bit images to 64 (the 256-tile kernel's bf16 k tile)
static int64_t pad_k(int64_t K, int precision) { return precision == MDG_PREC_F32 ? pad32(K) : pad64(K); }

static size_t image_bytes(int64_t rows, int64_t K, int precision) {
  if (rows <= 0 || K <= 0) return 0;
  const size_t Kp = static_cast<size_t>(pad_k(K, precision));
  if (precision == MDG_PREC_F32) return (K % 32 == 0) ? 0 : al256(static_cast<size_t>(rows) * Kp * 4);
  return al256(static_cast<size_t>(rows) * Kp * (precision == MDG_PREC_BF16X3 ? 4 : 2));
}

static void launch_prep(const float* s0, int64_t ld0, int64_t rows0, char* dst0, const float* s1, int64_t ld1, int64_t rows1,
                        char* dst1, int64_t K, int precision, hipStream_t st) {
  const int64_t Kp = pad_k(K, precision);
  PrepArgs pa{};
  pa.src[0] = s0; pa.ld[0] = ld0; pa.rows[0] = rows0; pa.dst0[0] = dst0;
  pa.src[1] = s1; pa.ld[1] = ld1; pa.rows[1] = s1 ? rows1 : 0; pa.dst0[1] = dst1;
  pa.K = K; pa.Kp = Kp; pa.bf16 = precision != MDG_PREC_F32 ? 1 : 0; pa.x3 = precision == MDG_PREC_BF16X3 ? 1 : 0;
  const int64_t groups = (rows0 > pa.rows[1] ? rows0 : pa.rows[1]) * (Kp / 4);
  hipLaunchKernelGGL(prep_operands_kernel, dim3(static_cast<unsigned>(mdg_cdiv(groups, 256)), s1 ? 2 : 1), dim3(256), 0, st, pa);
}

static void set_image(Operand& o, const char* image, int64_t rows, int64_t Kp, int precision) {
  o.nrows = rows;
  o.p0 = image;
  o.kb = precision == MDG_PREC_BF16 ? 2 : 4;
  o.p1 = precision == MDG_PREC_BF16X3 ? image + 64 : nullptr;
  o.ld_bytes = Kp * static_cast<int64_t>(o.kb);
}

static void set_operand(Operand& o, const float* raw, int64_t ld, const char* image, int64_t rows, int64_t K, int precision) {
  if (!image) { o.nrows = rows; o.p0 = reinterpret_cast<const char*>(raw); o.p1 = nullptr; o.ld_bytes = ld * 4; o.kb = 4; return; }
  set_image(o, image, rows, pad_k(K, precision), precision);
}

// workgroups a persistent launch may count on being resident together: one 128-KB-LDS workgroup per CU
static int resident_workgroups() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = pp::MAX_WG;
    return cus < pp::MAX_WG ? cus : pp::MAX_WG;
  }();
  return n;
}

static bool pp_shape(int precision, int64_t M, int64_t N) {
  return precision != MDG_PREC_F32 && N >= 256 && M >= 1024 && mdg_cdiv(M, pp::BM) * mdg_cdiv(N, pp::BN) >= 192;
}

// tile shape: the 256x256 / 8-wave shape once the problem fills the chip with it, else 128x128 / 4 waves
// (measured on the fusion GEMMs: bf16x3 -9 % time with the big tile, fp32 +6 %: the fp32 MFMA wants two workgroups per CU)
static bool choose_big(int precision, int64_t M, int64_t N) {
  bool big = pp_shape(precision, M, N);
  static MdgEnvInt tile_sw{"MDG_LINEAR_TILE", 0};
  if (tile_sw.get() == 256) big = precision != MDG_PREC_F32;
  else if (tile_sw.get() == 128) big = false;
  return big;
}

// 128-tile kernel, 16-bit modes: x is rounded / split while it is staged, no pre-pass over it (MDG_LINEAR_RAWX=0: pre-pass)
static bool raw_x_ok(int precision, bool big) {
  static MdgEnvInt raw_sw{"MDG_LINEAR_RAWX", 1};
  return !big && precision != MDG_PREC_F32 && raw_sw.get() != 0;
}

// sk_ws / sk_avail: what is left of the caller's workspace behind the operand images (the stream-K slots and flags)
static void launch_linear_core(LinearArgs& a, int precision, int64_t M, int64_t N, hipStream_t st, char* sk_ws = nullptr, size_t sk_avail = 0) {
  a.vec_y = mdg_aligned16(a.y) && a.ldy % 4 == 0;
  a.vec_r = a.res && mdg_aligned16(a.res) && a.ldr % 4 == 0;
  const bool big = choose_big(precision, M, N);
  // 1-D grid; the kernel maps the linear workgroup id to a tile (XCD-aware order once there are enough tiles to matter)
  constexpr int swz_env = -1;                              // XCD-aware tile order once there are 64 tiles
  const auto grid_for = [&](int bm, int bn) {
    a.tiles_x = static_cast<int>(mdg_cdiv(N, bn));
    a.tiles_y = static_cast<int>(mdg_cdiv(M, bm));
    const int total = a.tiles_x * a.tiles_y;
    a.swizzle = swz_env >= 0 ? swz_env : (total >= 64 ? 1 : 0);
    return dim3(static_cast<unsigned>(a.swizzle ? 8 * ((total + 7) / 8) : total));
  };
  // 16-bit modes run on the 16x16x32 MFMA (the 32x32x16 form of the 128-tile kernel lost: MI355X_MICROARCH.md, DVFS give-back item 7)
  constexpr bool m16 = true;
  if (big) {
    dim3 grid = grid_for(pp::BM, pp::BN);
    // stream-K hybrid when the last round of tiles would leave CUs idle (see the kernel): needs the slots in the workspace
    // Measured: OFF by default.  A persistent workgroup pays for what the hardware's own dispatch hides --
    // the epilogue's store drain in front of the next tile's first counted wait (stores and LDS-DMA share vmcnt), the pipeline
    // fill of every piece, the slot round trip -- about +11 us per 65-us tile, which is what the saved part of a round is worth
    // at these sizes ([22464, 2048] x 2048: 216-233 us plain, 251 us hybrid).  MDG_LINEAR_STREAMK=1 switches it on, 2 runs the
    // persistent loop without shared tiles.
    struct { int get() const { return 0; } } sk_sw;          // the stream-K hybrid stays OFF (measured, above); its kernel variant is not dispatched
    const int total = a.tiles_x * a.tiles_y, P = resident_workgroups();
    const int rem = total % P;
    const int64_t nk = a.K / (precision == MDG_PREC_BF16 ? 64 : 32);
    a.sk_tiles = 0;
    a.sk_ws = nullptr;
    if (sk_sw.get() && sk_ws && sk_avail >= pp::sk_bytes() && (reinterpret_cast<uintptr_t>(sk_ws) & 15u) == 0 && rem != 0 && rem * 20 <= P * 17 &&
        nk >= 8 && static_cast<int64_t>(rem) * nk >= P) {
      a.sk_tiles = rem;
      a.sk_ws = sk_ws;
      (void)hipMemsetAsync(sk_ws + static_cast<size_t>(pp::MAX_WG) * pp::SLOT_FLOATS * 4, 0, 4096, st);       // flags | error word
      grid = dim3(static_cast<unsigned>(P));
    }
    const bool persistent = a.sk_tiles != 0 || (sk_sw.get() == 2 && sk_ws && sk_avail >= pp::sk_bytes());   // 2: persistent loop without shared tiles (diagnostics)
    if (persistent && !a.sk_tiles) { a.sk_ws = sk_ws; grid = dim3(static_cast<unsigned>(P)); }
    // Row split of a product whose last round of 256-tiles would run mostly empty (one workgroup per CU; [22016, 6144]: 2064 tiles =
    // 8 rounds + 16 tiles that cost a ninth): the tile rows that fill whole rounds go to this kernel, the remaining rows to the
    // 128-tile kernel in a second launch (same operand images, same k order per element: the result is bit-identical to either
    // kernel alone).  Taken when the remainder is at most half a round -- four 128-tiles per 256-tile at ~0.65 of its rate only pay
    // below that.  MDG_LINEAR_TAIL128=0 keeps the single launch.
    static MdgEnvInt tail_sw{"MDG_LINEAR_TAIL128", 1};
    if (!persistent && tail_sw.get() && total > P && rem != 0 && a.tiles == nullptr) {
      const int full_rows = (total - rem) / a.tiles_x, tail_rows = a.tiles_y - full_rows;
      if (full_rows > 0 && tail_rows > 0 && tail_rows * a.tiles_x * 2 <= P) {
        LinearArgs head = a, tail = a;
        head.tiles_y = full_rows;
        head.ty_base = 0;
        const int n_head = head.tiles_x * head.tiles_y;
        head.swizzle = swz_env >= 0 ? swz_env : (n_head >= 64 ? 1 : 0);
        const dim3 ghead(static_cast<unsigned>(head.swizzle ? 8 * ((n_head + 7) / 8) : n_head));
        if (precision == MDG_PREC_BF16X3) hipLaunchKernelGGL((linear_pp_kernel<MDG_PREC_BF16X3, false>), ghead, dim3(pp::THREADS), pp::LDS_BYTES, st, head);
        else hipLaunchKernelGGL((linear_pp_kernel<MDG_PREC_BF16, false>), ghead, dim3(pp::THREADS), pp::LDS_BYTES, st, head);
        using S = Small;
        const int64_t row_base = static_cast<int64_t>(full_rows) * pp::BM;
        tail.tiles_x = static_cast<int>(mdg_cdiv(N, S::BN));
        tail.tiles_y = static_cast<int>(mdg_cdiv(M - row_base, S::BM));
        tail.ty_base = static_cast<int>(row_base / S::BM);
        const int n_tail = tail.tiles_x * tail.tiles_y;
        tail.swizzle = swz_env >= 0 ? swz_env : (n_tail >= 64 ? 1 : 0);
        tail.a_raw = nullptr;
        tail.ks_count = 0;
        const dim3 gtail(static_cast<unsigned>(tail.swizzle ? 8 * ((n_tail + 7) / 8) : n_tail));
        const size_t lds = 2 * S::STAGE;
        if (precision == MDG_PREC_BF16X3) hipLaunchKernelGGL((linear_kernel<MDG_PREC_BF16X3, S, 16>), gtail, dim3(S::THREADS), lds, st, tail);
        else hipLaunchKernelGGL((linear_kernel<MDG_PREC_BF16, S, 16>), gtail, dim3(S::THREADS), lds, st, tail);
        return;
      }
    }
    if (persistent) {
      if (precision == MDG_PREC_BF16X3) hipLaunchKernelGGL((linear_pp_kernel<MDG_PREC_BF16X3, true>), grid, dim3(pp::THREADS), pp::LDS_BYTES, st, a);
      else hipLaunchKernelGGL((linear_pp_kernel<MDG_PREC_BF16, true>), grid, dim3(pp::THREADS), pp::LDS_BYTES, st, a);
    } else {
      if (precision == MDG_PREC_BF16X3) hipLaunchKernelGGL((linear_pp_kernel<MDG_PREC_BF16X3, false>), grid, dim3(pp::THREADS), pp::LDS_BYTES, st, a);
      else hipLaunchKernelGGL((linear_pp_kernel<MDG_PREC_BF16, false>), grid, dim3(pp::THREADS), pp::LDS_BYTES, st, a);
    }
    return;
  }
  using S = Small;
  dim3 grid = grid_for(S::BM, S::BN);
  if (a.ks_count > 1) grid.y = static_cast<unsigned>(mdg_cdiv(a.K, a.ks_len));
  const size_t lds = 2 * S::STAGE;
  if (precision == MDG_PREC_F32) hipLaunchKernelGGL((linear_kernel<MDG_PREC_F32, S, 32>), grid, dim3(S::THREADS), lds, st, a);
  else if (a.a_raw) {                                         // x straight from its fp32 rows
    if (precision == MDG_PREC_BF16X3) {
      if (m16) hipLaunchKernelGGL((linear_kernel<MDG_PREC_BF16X3, S, 16, false, true>), grid, dim3(S::THREADS), lds, st, a);
      else hipLaunchKernelGGL((linear_kernel<MDG_PREC_BF16X3, S, 32, false, true>), grid, dim3(S::THREADS), lds, st, a);
    } else {
      if (m16) hipLaunchKernelGGL((linear_kernel<MDG_PREC_BF16, S, 16, false, true>), grid, dim3(S::THREADS), lds, st, a);
      else hipLaunchKernelGGL((linear_kernel<MDG_PREC_BF16, S, 32, false, true>), grid, dim3(S::THREADS), lds, st, a);
    }
  } else if (precision == MDG_PREC_BF16X3) {
    if (m16) hipLaunchKernelGGL((linear_kernel<MDG_PREC_BF16X3, S, 16>), grid, dim3(S::THREADS), lds, st, a);
    else hipLaunchKernelGGL((linear_kernel<MDG_PREC_BF16X3, S, 32>), grid, dim3(S::THREADS), lds, st, a);
  } else {
    if (m16) hipLaunchKernelGGL((linear_kernel<MDG_PREC_BF16, S, 16>), grid, dim3(S::THREADS), lds, st, a);
    else hipLaunchKernelGGL((linear_kernel<MDG_PREC_BF16, S, 32>), grid, dim3(S::THREADS), lds, st, a);
  }
}

extern "C" size_t mdg_pack_operand_bytes(int64_t rows, int64_t K, int precision) { return image_bytes(rows, K, precision); }

extern "C" int mdg_pack_operand(const float* src, int64_t ld, int64_t rows, int64_t K, int precision, void* dst, size_t dst_bytes,
                                void* stream) {
  MDG_CHECK_ARG(rows >= 0 && K > 0 && K % 4 == 0 && ld % 4 == 0 && ld >= K, "mdg_pack_operand: K and ld must be multiples of 4, ld >= K");
  const size_t need = image_bytes(rows, K, precision);
  if (need == 0) return MDG_OK;
  MDG_CHECK_ARG(src && mdg_aligned16(src), "mdg_pack_operand: src must be 16-byte aligned");
  if (!dst || dst_bytes < need || !mdg_aligned16(dst)) {
    mdg_set_error("mdg_pack_operand: destination of %zu bytes required, got %zu", need, dst_bytes);
    return MDG_EWORKSPACE;
  }
  launch_prep(src, ld, rows, static_cast<char*>(dst), nullptr, 0, 0, nullptr, K, precision, static_cast<hipStream_t>(stream));
  MDG_CHECK_LAUNCH("mdg_pack_operand");
  return MDG_OK;
}

// Operand image of src^T straight from src [rows, cols] (16-bit modes): what mdg_pack_operand(transpose(src)) writes, in one
// transposing pass -- the backward pass multiplies by W^T (dx = g W as a dense block with weight W^T [K, N]) and W changes every step.
extern "C" int mdg_pack_operand_transposed(const float* src, int64_t ld, int64_t rows, int64_t cols, int precision, void* dst, size_t dst_bytes,
                                           void* stream) {
  MDG_CHECK_ARG(rows > 0 && cols > 0 && ld >= cols, "mdg_pack_operand_transposed: bad shape");
  MDG_CHECK_ARG(precision == MDG_PREC_BF16 || precision == MDG_PREC_BF16X3, "mdg_pack_operand_transposed: a 16-bit operand mode");
  MDG_CHECK_ARG(pad64(rows) / 64 < (1ll << 31) && mdg_cdiv(cols, 64) < 65536, "mdg_pack_operand_transposed: too many tiles");
  const size_t need = image_bytes(cols, rows, precision);
  MDG_CHECK_ARG(src, "mdg_pack_operand_transposed: null source");
  if (!dst || dst_bytes < need || !mdg_aligned16(dst)) {
    mdg_set_error("mdg_pack_operand_transposed: destination of %zu bytes required, got %zu", need, dst_bytes);
    return MDG_EWORKSPACE;
  }
  PrepTArgs pa{};
  pa.src[0] = src; pa.ld[0] = ld; pa.C[0] = cols; pa.dst0[0] = static_cast<char*>(dst);
  pa.src[1] = src; pa.ld[1] = ld; pa.C[1] = 0; pa.dst0[1] = static_cast<char*>(dst);
  pa.M = rows; pa.Mp = pad64(rows); pa.bf16 = 1; pa.x3 = precision == MDG_PREC_BF16X3 ? 1 : 0;
  hipLaunchKernelGGL(prep_transposed_kernel, dim3(static_cast<unsigned>(pa.Mp / 64), static_cast<unsigned>(mdg_cdiv(cols, 64)), 1), dim3(256), 0,
                     static_cast<hipStream_t>(stream), pa);
  MDG_CHECK_LAUNCH("mdg_pack_operand_transposed");
  return MDG_OK;
}

extern "C" size_t mdg_linear_workspace_bytes(int64_t M, int64_t N, int64_t K, int precision, int w_is_packed) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return image_bytes(M, K, precision) + (w_is_packed ? 0 : image_bytes(N, K, precision)) + (pp_shape(precision, M, N) ? pp::sk_bytes() : 0);
}

extern "C" size_t mdg_linear_packed_x_workspace_bytes(int64_t M, int64_t N, int64_t K, int precision, int w_is_packed) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return (w_is_packed ? 0 : image_bytes(N, K, precision)) + (pp_shape(precision, M, N) ? pp::sk_bytes() : 0);
}

static int linear_impl(const float* x, int64_t ldx, const float* w, int64_t ldw, const void* w_packed, float* y, int64_t ldy,
                       int64_t M, int64_t N, int64_t K, const float* bias, const float* scale, const float* shift, int act,
                       const float* residual, int64_t ldr, float alpha, float beta, int precision, void* workspace,
                       size_t workspace_bytes, void* stream, float drop_p, uint64_t drop_seed) {
  MDG_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "mdg_linear: negative size");
  MDG_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "mdg_linear: dropout p must be in [0,1)");
  if (M == 0 || N == 0) return MDG_OK;
  MDG_CHECK_ARG(x && (w || w_packed) && y, "mdg_linear: null pointer");
  MDG_CHECK_ARG(K > 0 && K % 4 == 0 && ldx % 4 == 0 && ldx >= K && (w_packed || (ldw % 4 == 0 && ldw >= K)),
                "mdg_linear: K, ldx, ldw must be multiples of 4 with ld >= K (K=%lld ldx=%lld ldw=%lld); zero-pad the inner dimension",
                (long long)K, (long long)ldx, (long long)ldw);
  MDG_CHECK_ARG(mdg_aligned16(x) && (!w || mdg_aligned16(w)) && (!w_packed || mdg_aligned16(w_packed)), "mdg_linear: x, w, w_packed must be 16-byte aligned");
  MDG_CHECK_ARG(ldy >= N && (!residual || ldr >= N || ldr == 0), "mdg_linear: ldy/ldr smaller than N (ldr == 0 broadcasts one row)");
  MDG_CHECK_ARG((scale == nullptr) == (shift == nullptr), "mdg_linear: scale and shift come together");
  MDG_CHECK_ARG(act >= MDG_ACT_NONE && act <= MDG_ACT_SELU, "mdg_linear: unknown activation %d", act);
  MDG_CHECK_ARG(mdg_cdiv(M, Small::BM) * mdg_cdiv(N, Small::BN) < (1ll << 31) - 8, "mdg_linear: too many tiles for one launch");
  MDG_CHECK_ARG(precision == MDG_PREC_F32 || precision == MDG_PREC_BF16X3 || precision == MDG_PREC_BF16, "mdg_linear: unknown precision %d", precision);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool rawx = raw_x_ok(precision, choose_big(precision, M, N));
  const size_t xb = rawx ? 0 : image_bytes(M, K, precision), wb = image_bytes(N, K, precision);
  const bool w_ready = w_packed != nullptr || wb == 0;      // fp32 with K % 32 == 0 needs no image at all
  MDG_CHECK_ARG(w || w_ready, "mdg_linear: raw w missing");
  const size_t need = xb + (w_ready ? 0 : wb);          // (+ the stream-K slots where the shape takes them: optional, used when present)
  if (need && (!workspace || workspace_bytes < need || !mdg_aligned16(workspace))) {
    mdg_set_error("mdg_linear: workspace of %zu bytes (16-byte aligned) required, got %zu", need, workspace_bytes);
    return MDG_EWORKSPACE;
  }
  char* ws = static_cast<char*>(workspace);
  char* ximg = xb ? ws : nullptr;
  char* wimg = w_ready ? nullptr : ws + xb;
  if (ximg || wimg) {
    if (ximg) launch_prep(x, ldx, M, ximg, wimg ? w : nullptr, ldw, N, wimg, K, precision, st);
    else launch_prep(w, ldw, N, wimg, nullptr, 0, 0, nullptr, K, precision, st);
  }
  LinearArgs a{};
  a.y = y; a.ldy = ldy; a.bias = bias; a.scale = scale; a.shift = shift; a.res = residual; a.ldr = ldr;
  a.alpha = alpha; a.beta = beta; a.act = act; a.M = M; a.N = N; a.K = pad_k(K, precision);
  set_operand(a.A, x, ldx, ximg, M, K, precision);
  if (rawx) { a.a_raw = x; a.a_ldx = ldx; a.a_k = K; a.A.nrows = M; }
  set_operand(a.B, w, ldw, wb == 0 ? nullptr : (w_packed ? static_cast<const char*>(w_packed) : wimg), N, K, precision);
  if (drop_p > 0.f) {
    a.drop_thr = mdg_drop_threshold(drop_p);
    a.drop_seed = drop_seed;
    a.drop_scale = 1.0f / (1.0f - drop_p);
  }
  launch_linear_core(a, precision, M, N, st, workspace ? ws + need : nullptr, workspace_bytes > need ? workspace_bytes - need : 0);
  MDG_CHECK_LAUNCH("mdg_linear");
  return MDG_OK;
}

extern "C" int mdg_linear(const float* x, int64_t ldx, const float* w, int64_t ldw, const void* w_packed, float* y, int64_t ldy,
                          int64_t M, int64_t N, int64_t K, const float* bias, const float* scale, const float* shift, int act,
                          const float* residual, int64_t ldr, float alpha, float beta, int precision, void* workspace,
                          size_t workspace_bytes, void* stream) {
  return linear_impl(x, ldx, w, ldw, w_packed, y, ldy, M, N, K, bias, scale, shift, act, residual, ldr, alpha, beta, precision, workspace, workspace_bytes,
                     stream, 0.f, 0);
}

extern "C" int mdg_linear_dropout(const float* x, int64_t ldx, const float* w, int64_t ldw, const void* w_packed, float* y, int64_t ldy,
                                  int64_t M, int64_t N, int64_t K, const float* bias, int act, const float* residual, int64_t ldr, float beta,
                                  float drop_p, uint64_t drop_seed, int precision, void* workspace, size_t workspace_bytes, void* stream) {
  return linear_impl(x, ldx, w, ldw, w_packed, y, ldy, M, N, K, bias, nullptr, nullptr, act, residual, ldr, 1.0f, beta, precision, workspace,
                     workspace_bytes, stream, drop_p, drop_seed);
}

// ---- grouped launch: see the GROUPED kernel variant -------------------------------------------------------------------------
extern "C" int mdg_linear_group_tile_words(void) { return kGroupTileWords; }

extern "C" size_t mdg_linear_grouped_workspace_bytes(int64_t rows_total, int64_t K, int precision) { return image_bytes(rows_total, K, precision); }

extern "C" int mdg_linear_grouped(const float* x, int64_t ldx, int64_t rows_total, int64_t K, const float* w, int64_t ldw, const void* w_packed,
                                  int64_t w_rows_total, const float* bias, const int64_t* tiles, int64_t n_tiles, float* y,
                                  const float* residual, int act, int precision, void* workspace, size_t workspace_bytes, void* stream) {
  MDG_CHECK_ARG(rows_total >= 0 && w_rows_total >= 0 && n_tiles >= 0, "mdg_linear_grouped: negative size");
  if (n_tiles == 0 || rows_total == 0 || w_rows_total == 0) return MDG_OK;
  MDG_CHECK_ARG(x && y && tiles && (w || w_packed), "mdg_linear_grouped: null pointer");
  MDG_CHECK_ARG(K > 0 && K % 4 == 0 && ldx % 4 == 0 && ldx >= K, "mdg_linear_grouped: K and ldx must be multiples of 4, ldx >= K");
  MDG_CHECK_ARG(mdg_aligned16(x) && mdg_aligned16(y) && (!residual || mdg_aligned16(residual)) && (!w_packed || mdg_aligned16(w_packed)),
                "mdg_linear_grouped: x, y, residual, w_packed must be 16-byte aligned (offsets and row strides in the table: multiples of 4)");
  MDG_CHECK_ARG(act >= MDG_ACT_NONE && act <= MDG_ACT_SELU, "mdg_linear_grouped: unknown activation %d", act);
  MDG_CHECK_ARG(precision == MDG_PREC_F32 || precision == MDG_PREC_BF16X3 || precision == MDG_PREC_BF16, "mdg_linear_grouped: unknown precision %d", precision);
  MDG_CHECK_ARG(n_tiles < (1ll << 31), "mdg_linear_grouped: too many tiles");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t xb = image_bytes(rows_total, K, precision), wb = image_bytes(w_rows_total, K, precision);
  MDG_CHECK_ARG(wb == 0 ? (w != nullptr && ldw >= K && ldw % 4 == 0 && mdg_aligned16(w)) : w_packed != nullptr,
                "mdg_linear_grouped: the stacked weights must come packed (mdg_pack_operand), or raw where the mode needs no image");
  if (xb && !raw_x_ok(precision, false) && (!workspace || workspace_bytes < xb || !mdg_aligned16(workspace))) {
    mdg_set_error("mdg_linear_grouped: workspace of %zu bytes (16-byte aligned) required, got %zu", xb, workspace_bytes);
    return MDG_EWORKSPACE;
  }
  const bool rawx = raw_x_ok(precision, false);
  char* ximg = (xb && !rawx) ? static_cast<char*>(workspace) : nullptr;
  if (ximg) launch_prep(x, ldx, rows_total, ximg, nullptr, 0, 0, nullptr, K, precision, st);
  LinearArgs a{};
  a.y = y; a.ldy = 0; a.bias = bias; a.res = residual; a.ldr = 0; a.alpha = 1.f; a.beta = 1.f; a.act = act;
  a.M = rows_total; a.N = w_rows_total; a.K = pad_k(K, precision);
  a.vec_y = 1; a.vec_r = residual ? 1 : 0;               // the table's offsets and strides are multiples of 4 floats (checked by the caller's builder)
  a.tiles = tiles;
  set_operand(a.A, x, ldx, ximg, rows_total, K, precision);
  if (rawx) { a.a_raw = x; a.a_ldx = ldx; a.a_k = K; }
  set_operand(a.B, w, ldw, wb == 0 ? nullptr : static_cast<const char*>(w_packed), w_rows_total, K, precision);
  const dim3 grid(static_cast<unsigned>(n_tiles));
  const size_t lds = 2 * Small::STAGE;
  if (precision == MDG_PREC_F32) hipLaunchKernelGGL((linear_kernel<MDG_PREC_F32, Small, 32, true>), grid, dim3(Small::THREADS), lds, st, a);
  else if (rawx && precision == MDG_PREC_BF16X3) hipLaunchKernelGGL((linear_kernel<MDG_PREC_BF16X3, Small, 16, true, true>), grid, dim3(Small::THREADS), lds, st, a);
  else if (rawx) hipLaunchKernelGGL((linear_kernel<MDG_PREC_BF16, Small, 16, true, true>), grid, dim3(Small::THREADS), lds, st, a);
  else if (precision == MDG_PREC_BF16X3) hipLaunchKernelGGL((linear_kernel<MDG_PREC_BF16X3, Small, 16, true>), grid, dim3(Small::THREADS), lds, st, a);
  else hipLaunchKernelGGL((linear_kernel<MDG_PREC_BF16, Small, 16, true>), grid, dim3(Small::THREADS), lds, st, a);
  MDG_CHECK_LAUNCH("mdg_linear_grouped");
  return MDG_OK;
}

// ---- TN product: y[N,K] = g^T x for row-major g [M,N], x [M,K] (weight gradients of the wide layers) -----------------------
static size_t image_bytes_t(int64_t rows, int64_t inner, int precision) {      // always a full image (the inner index is re-laid out)
  if (rows <= 0 || inner <= 0) return 0;
  const size_t Mp = static_cast<size_t>(pad64(inner));
  return al256(static_cast<size_t>(rows) * Mp * (precision == MDG_PREC_BF16 ? 2 : 4));
}

namespace {
// y[i] = sum_z part[z][i] (split-K partials of mdg_linear_tn), 16 bytes per thread, fixed order
__global__ __launch_bounds__(256) void ksplit_sum_kernel(const float* __restrict__ part, float* __restrict__ y, int64_t n4, int64_t stride, int splits) {
  const int64_t q = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (q >= n4) return;
  f32x4 s = reinterpret_cast<const f32x4*>(part)[q];
  for (int z = 1; z < splits; ++z) s += reinterpret_cast<const f32x4*>(part + z * stride)[q];
  reinterpret_cast<f32x4*>(y)[q] = s;
}

}  // namespace

// Weight-gradient products with few output tiles (dW [2048, 1024] over 22 016 rows: 128 tiles of 128 x 128, half the CUs idle for the
// whole K loop): the K range is split over grid.y so that ~256 workgroups run, partial products summed afterwards.
static int tn_splits(int precision, int64_t N, int64_t K, int64_t Mp) {
  if (choose_big(precision, N, K) || K % 4 != 0) return 1;
  const int64_t tiles = mdg_cdiv(N, Small::BM) * mdg_cdiv(K, Small::BN);
  if (tiles >= 192 || Mp < 4096) return 1;
  int64_t s = 256 / tiles;
  if (s > 8) s = 8;
  while (s > 1 && Mp / s < 1024) --s;
  return static_cast<int>(s < 1 ? 1 : s);
}
static size_t tn_split_bytes(int precision, int64_t N, int64_t K, int64_t Mp) {
  const int s = tn_splits(precision, N, K, Mp);
  return s > 1 ? al256(static_cast<size_t>(s) * N * K * sizeof(float)) : 0;
}
// launches the TN product a (A = g^T image, B = x^T image, K = Mp) into y [N, ldy = K]: split over k when that fills the chip
static void launch_tn(LinearArgs& a, int precision, int64_t N, int64_t K, int64_t Mp, hipStream_t st, char* ws, size_t ws_bytes) {
  const int s = tn_splits(precision, N, K, Mp);
  const size_t pb = s > 1 ? al256(static_cast<size_t>(s) * N * K * sizeof(float)) : 0;
  if (s > 1 && ws && ws_bytes >= pb && a.ldy == K && mdg_aligned16(ws)) {
    float* const y = a.y;
    a.y = reinterpret_cast<float*>(ws);
    a.ks_count = s;
    a.ks_len = pad64(mdg_cdiv(Mp, s));
    a.ks_stride = N * K;
    launch_linear_core(a, precision, N, K, st, nullptr, 0);
    hipLaunchKernelGGL(ksplit_sum_kernel, dim3(static_cast<unsigned>(mdg_cdiv(N * K / 4, 256))), dim3(256), 0, st, reinterpret_cast<const float*>(ws), y, N * K / 4,
                       N * K, static_cast<int>(mdg_cdiv(Mp, a.ks_len)));
    return;
  }
  launch_linear_core(a, precision, N, K, st, ws, ws_bytes);
}

extern "C" size_t mdg_linear_tn_workspace_bytes(int64_t M, int64_t N, int64_t K, int precision) {
  return image_bytes_t(N, M, precision) + image_bytes_t(K, M, precision) + (pp_shape(precision, N, K) ? pp::sk_bytes() : 0) + tn_split_bytes(precision, N, K, pad64(M));
}

extern "C" int mdg_linear_tn(const float* g, int64_t ldg, const float* x, int64_t ldx, float* y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                             int precision, void* workspace, size_t workspace_bytes, void* stream) {
  MDG_CHECK_ARG(M > 0 && N > 0 && K > 0, "mdg_linear_tn: empty operand");
  MDG_CHECK_ARG(g && x && y && ldg >= N && ldx >= K && ldy >= K, "mdg_linear_tn: null pointer / short row stride");
  MDG_CHECK_ARG(precision == MDG_PREC_F32 || precision == MDG_PREC_BF16X3 || precision == MDG_PREC_BF16, "mdg_linear_tn: unknown precision %d", precision);
  MDG_CHECK_ARG(mdg_cdiv(N, Small::BM) * mdg_cdiv(K, Small::BN) < (1ll << 31) - 8 && mdg_cdiv(N, 64) < 65536, "mdg_linear_tn: too many tiles");
  const size_t ab = image_bytes_t(N, M, precision), bb = image_bytes_t(K, M, precision);
  if (!workspace || workspace_bytes < ab + bb || !mdg_aligned16(workspace)) {
    mdg_set_error("mdg_linear_tn: workspace of %zu bytes (16-byte aligned) required, got %zu", ab + bb, workspace_bytes);
    return MDG_EWORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* aimg = static_cast<char*>(workspace);
  char* bimg = aimg + ab;
  const int64_t Mp = pad64(M);
  PrepTArgs pa{};
  pa.src[0] = g; pa.ld[0] = ldg; pa.C[0] = N; pa.dst0[0] = aimg;
  pa.src[1] = x; pa.ld[1] = ldx; pa.C[1] = K; pa.dst0[1] = bimg;
  pa.M = M; pa.Mp = Mp; pa.bf16 = precision != MDG_PREC_F32 ? 1 : 0; pa.x3 = precision == MDG_PREC_BF16X3 ? 1 : 0;
  const int64_t cmax = N > K ? N : K;
  hipLaunchKernelGGL(prep_transposed_kernel, dim3(static_cast<unsigned>(Mp / 64), static_cast<unsigned>(mdg_cdiv(cmax, 64)), 2), dim3(256), 0, st, pa);
  LinearArgs a{};
  a.y = y; a.ldy = ldy; a.alpha = 1.f; a.beta = 0.f; a.act = MDG_ACT_NONE; a.M = N; a.N = K; a.K = Mp;
  set_image(a.A, aimg, N, Mp, precision);
  set_image(a.B, bimg, K, Mp, precision);
  if (precision == MDG_PREC_F32) launch_linear_core(a, precision, N, K, st, aimg + ab + bb, workspace_bytes - ab - bb);
  else launch_tn(a, precision, N, K, Mp, st, aimg + ab + bb, workspace_bytes - ab - bb);
  MDG_CHECK_LAUNCH("mdg_linear_tn");
  return MDG_OK;
}

// ---- backward of a wide dense block from ONE pass over g (see prep_backward_kernel) -------------------------------------------
extern "C" size_t mdg_linear_backward_pack_bytes(int64_t M, int64_t N, int precision, int which) {
  if (M <= 0 || N <= 0 || (precision != MDG_PREC_BF16 && precision != MDG_PREC_BF16X3)) return 0;
  if (which == 0) return image_bytes(M, N, precision);                    // row image of g
  if (which == 1) return image_bytes_t(N, M, precision);                  // image of g^T
  return al256(static_cast<size_t>(pad64(M) / 64) * N * sizeof(float));   // workspace: partial column sums
}

extern "C" int mdg_linear_backward_pack(const float* g, int64_t ldg, int64_t M, int64_t N, int precision, void* row_image, void* t_image,
                                        float* dbias, float drop_p, uint64_t drop_seed, void* workspace, size_t workspace_bytes, void* stream) {
  MDG_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "mdg_linear_backward_pack: dropout p must be in [0,1)");
  MDG_CHECK_ARG(M > 0 && N > 0 && g && ldg >= N, "mdg_linear_backward_pack: empty operand / short row stride");
  MDG_CHECK_ARG(precision == MDG_PREC_BF16 || precision == MDG_PREC_BF16X3, "mdg_linear_backward_pack: a 16-bit operand mode");
  MDG_CHECK_ARG(t_image && mdg_aligned16(t_image) && (!row_image || mdg_aligned16(row_image)), "mdg_linear_backward_pack: null / misaligned image");
  MDG_CHECK_ARG(pad64(M) / 64 < (1ll << 31) && pad64(N) / 64 < 65536, "mdg_linear_backward_pack: too many tiles");
  const size_t need = dbias ? mdg_linear_backward_pack_bytes(M, N, precision, 2) : 0;
  if (need && (!workspace || workspace_bytes < need)) {
    mdg_set_error("mdg_linear_backward_pack: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return MDG_EWORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  PrepBArgs a{g, ldg, M, N, pad64(N), pad64(M), static_cast<char*>(row_image), static_cast<char*>(t_image), dbias ? static_cast<float*>(workspace) : nullptr,
              precision == MDG_PREC_BF16X3 ? 1 : 0, 0, 0, 0.f};
  if (drop_p > 0.f) {
    a.drop_thr = mdg_drop_threshold(drop_p);
    a.drop_seed = drop_seed;
    a.drop_scale = 1.0f / (1.0f - drop_p);
  }
  hipLaunchKernelGGL(prep_backward_kernel, dim3(static_cast<unsigned>(a.Mp / 64), static_cast<unsigned>(a.Np / 64)), dim3(256), 0, st, a);
  if (dbias)
    hipLaunchKernelGGL(prep_backward_bias_kernel, dim3(static_cast<unsigned>(mdg_cdiv(N, 16))), dim3(256), 0, st, static_cast<const float*>(workspace), dbias,
                       a.Mp / 64, N);
  MDG_CHECK_LAUNCH("mdg_linear_backward_pack");
  return MDG_OK;
}

extern "C" size_t mdg_linear_tn_packed_g_workspace_bytes(int64_t M, int64_t N, int64_t K, int precision) {
  return image_bytes_t(K, M, precision) + (pp_shape(precision, N, K) ? pp::sk_bytes() : 0) + tn_split_bytes(precision, N, K, pad64(M));
}

// dW [N, K] = g^T x with the image of g^T already made (mdg_linear_backward_pack); x [M, K] is re-laid out here.
extern "C" int mdg_linear_tn_packed_g(const void* gt_image, const float* x, int64_t ldx, float* y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                                      int precision, void* workspace, size_t workspace_bytes, void* stream) {
  MDG_CHECK_ARG(M > 0 && N > 0 && K > 0, "mdg_linear_tn_packed_g: empty operand");
  MDG_CHECK_ARG(gt_image && mdg_aligned16(gt_image) && x && y && ldx >= K && ldy >= K, "mdg_linear_tn_packed_g: null pointer / short row stride");
  MDG_CHECK_ARG(precision == MDG_PREC_BF16X3 || precision == MDG_PREC_BF16, "mdg_linear_tn_packed_g: a 16-bit operand mode");
  MDG_CHECK_ARG(mdg_cdiv(N, Small::BM) * mdg_cdiv(K, Small::BN) < (1ll << 31) - 8 && mdg_cdiv(K, 64) < 65536, "mdg_linear_tn_packed_g: too many tiles");
  const size_t bb = image_bytes_t(K, M, precision);
  if (!workspace || workspace_bytes < bb || !mdg_aligned16(workspace)) {
    mdg_set_error("mdg_linear_tn_packed_g: workspace of %zu bytes (16-byte aligned) required, got %zu", bb, workspace_bytes);
    return MDG_EWORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* bimg = static_cast<char*>(workspace);
  const int64_t Mp = pad64(M);
  PrepTArgs pa{};
  pa.src[0] = x; pa.ld[0] = ldx; pa.C[0] = K; pa.dst0[0] = bimg;
  pa.src[1] = x; pa.ld[1] = ldx; pa.C[1] = 0; pa.dst0[1] = bimg;
  pa.M = M; pa.Mp = Mp; pa.bf16 = 1; pa.x3 = precision == MDG_PREC_BF16X3 ? 1 : 0;
  hipLaunchKernelGGL(prep_transposed_kernel, dim3(static_cast<unsigned>(Mp / 64), static_cast<unsigned>(mdg_cdiv(K, 64)), 1), dim3(256), 0, st, pa);
  LinearArgs a{};
  a.y = y; a.ldy = ldy; a.alpha = 1.f; a.beta = 0.f; a.act = MDG_ACT_NONE; a.M = N; a.N = K; a.K = Mp;
  set_image(a.A, static_cast<const char*>(gt_image), N, Mp, precision);
  set_image(a.B, bimg, K, Mp, precision);
  launch_tn(a, precision, N, K, Mp, st, bimg + bb, workspace_bytes - bb);
  MDG_CHECK_LAUNCH("mdg_linear_tn_packed_g");
  return MDG_OK;
}

static int layernorm_impl(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy, int64_t rows, int64_t d,
                          float eps, int precision, void* y_packed, size_t y_packed_bytes, void* stream) {
  MDG_CHECK_ARG(rows >= 0 && d > 0, "mdg_layernorm: bad shape");
  if (rows == 0) return MDG_OK;
  MDG_CHECK_ARG(x && gamma && beta && (y || y_packed), "mdg_layernorm: null pointer");
  MDG_CHECK_ARG(d % 4 == 0 && d <= 2048 && ldx % 4 == 0 && ldx >= d && (!y || (ldy % 4 == 0 && ldy >= d)),
                "mdg_layernorm: d must be a multiple of 4 and <= 2048 (got %lld)", (long long)d);
  MDG_CHECK_ARG(mdg_aligned16(x) && (!y || mdg_aligned16(y)) && mdg_aligned16(gamma) && mdg_aligned16(beta), "mdg_layernorm: 16-byte alignment");
  char* hi = nullptr;
  int x3 = 0;
  if (y_packed) {
    MDG_CHECK_ARG(precision == MDG_PREC_BF16X3 || precision == MDG_PREC_BF16, "mdg_layernorm_packed: a 16-bit operand mode (the fp32 mode reads y itself)");
    MDG_CHECK_ARG(d % 64 == 0 && mdg_aligned16(y_packed), "mdg_layernorm_packed: d must be a multiple of 64 (the image has no padding to fill)");
    const size_t need = image_bytes(rows, d, precision);
    if (y_packed_bytes < need) {
      mdg_set_error("mdg_layernorm_packed: image of %zu bytes required, got %zu", need, y_packed_bytes);
      return MDG_EWORKSPACE;
    }
    hi = static_cast<char*>(y_packed);
    x3 = precision == MDG_PREC_BF16X3 ? 1 : 0;
  }
  const dim3 grid(static_cast<unsigned>(mdg_cdiv(rows, 4)));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int di = static_cast<int>(d);
  if (d <= 256) hipLaunchKernelGGL(layernorm_kernel<1>, grid, dim3(256), 0, st, x, ldx, gamma, beta, y, ldy, rows, di, eps, hi, x3);
  else if (d <= 512) hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, st, x, ldx, gamma, beta, y, ldy, rows, di, eps, hi, x3);
  else if (d <= 1024) hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, st, x, ldx, gamma, beta, y, ldy, rows, di, eps, hi, x3);
  else hipLaunchKernelGGL(layernorm_kernel<8>, grid, dim3(256), 0, st, x, ldx, gamma, beta, y, ldy, rows, di, eps, hi, x3);
  MDG_CHECK_LAUNCH("mdg_layernorm");
  return MDG_OK;
}

extern "C" int mdg_layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy,
                             int64_t rows, int64_t d, float eps, void* stream) {
  return layernorm_impl(x, ldx, gamma, beta, y, ldy, rows, d, eps, MDG_PREC_F32, nullptr, 0, stream);
}

// LayerNorm that also writes y as the operand image of the dense block that consumes it (mdg_pack_operand_bytes(rows, d, precision)
// bytes; identical to what mdg_linear's own pre-pass would write): that block then starts from mdg_linear_packed_x.
extern "C" int mdg_layernorm_packed(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy, int64_t rows,
                                    int64_t d, float eps, int precision, void* y_packed, size_t y_packed_bytes, void* stream) {
  MDG_CHECK_ARG(y_packed, "mdg_layernorm_packed: null image");
  return layernorm_impl(x, ldx, gamma, beta, y, ldy, rows, d, eps, precision, y_packed, y_packed_bytes, stream);
}

// mdg_linear on an x that already exists as an operand image (written by mdg_layernorm_packed / mdg_pack_operand): no pre-pass.
extern "C" int mdg_linear_packed_x(const void* x_packed, int64_t M, int64_t K, const float* w, int64_t ldw, const void* w_packed, float* y,
                                   int64_t ldy, int64_t N, const float* bias, int act, const float* residual, int64_t ldr, float alpha,
                                   float beta, int precision, void* workspace, size_t workspace_bytes, void* stream) {
  MDG_CHECK_ARG(M >= 0 && N >= 0 && K > 0, "mdg_linear_packed_x: bad size");
  if (M == 0 || N == 0) return MDG_OK;
  MDG_CHECK_ARG(precision == MDG_PREC_BF16X3 || precision == MDG_PREC_BF16, "mdg_linear_packed_x: a 16-bit operand mode");
  MDG_CHECK_ARG(x_packed && y && (w || w_packed) && mdg_aligned16(x_packed), "mdg_linear_packed_x: null / misaligned pointer");
  MDG_CHECK_ARG(K % 64 == 0, "mdg_linear_packed_x: K must be a multiple of 64");
  MDG_CHECK_ARG(ldy >= N && (!residual || ldr >= N || ldr == 0), "mdg_linear_packed_x: ldy/ldr smaller than N");
  MDG_CHECK_ARG(act >= MDG_ACT_NONE && act <= MDG_ACT_SELU, "mdg_linear_packed_x: unknown activation %d", act);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t wb = image_bytes(N, K, precision);
  char* wimg = nullptr;
  if (!w_packed) {
    MDG_CHECK_ARG(w && ldw % 4 == 0 && ldw >= K && mdg_aligned16(w), "mdg_linear_packed_x: raw w must be 16-byte aligned with ldw >= K");
    if (!workspace || workspace_bytes < wb || !mdg_aligned16(workspace)) {
      mdg_set_error("mdg_linear_packed_x: workspace of %zu bytes required for the weight image, got %zu", wb, workspace_bytes);
      return MDG_EWORKSPACE;
    }
    wimg = static_cast<char*>(workspace);
    launch_prep(w, ldw, N, wimg, nullptr, 0, 0, nullptr, K, precision, st);
  }
  LinearArgs a{};
  a.y = y; a.ldy = ldy; a.bias = bias; a.res = residual; a.ldr = ldr; a.alpha = alpha; a.beta = beta; a.act = act; a.M = M; a.N = N; a.K = K;
  set_operand(a.A, nullptr, 0, static_cast<const char*>(x_packed), M, K, precision);
  set_operand(a.B, w, ldw, w_packed ? static_cast<const char*>(w_packed) : wimg, N, K, precision);
  {
    const size_t used = w_packed ? 0 : wb;
    char* wsb = static_cast<char*>(workspace);
    launch_linear_core(a, precision, M, N, st, (wsb && workspace_bytes > used) ? wsb + used : nullptr, workspace_bytes > used ? workspace_bytes - used : 0);
  }
  MDG_CHECK_LAUNCH("mdg_linear_packed_x");
  return MDG_OK;
}
