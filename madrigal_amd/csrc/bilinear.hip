// All-pairs symmetric bilinear DDI head for gfx950.
//
//   S[l,i,j] = z_head[i]^T W_sym[l] z_tail[j]        (madrigal/models/models.py:537-547)
//
// One workgroup (8 waves) owns a slab of 256 head rows of ONE outcome l and sweeps every tail
// drug.  Prologue: T = z_head[rows] . W_sym[l] (the reference's inner matmul, rounded to fp32) is
// formed on the matrix cores, bounced once through LDS so that each wave holds its 32 rows of T as
// the MFMA *A* operand in registers for the whole sweep.  Main loop: 64 tail rows at a time arrive in
// LDS by LDS-DMA (z_tail is 2 MB at N=4096: it lives in L2 / Infinity Cache), each wave multiplies its
// resident T rows against the staged tile (2 accumulator tiles of 32x32) and the epilogue streams
// the scores to HBM as full 128-byte lines (lane = column, so one store instruction writes two
// complete row segments).  Algorithmic traffic is 4 B written per score (STORE) and ~0 read, so the
// kernel is bound by HBM writes for the bf16 products and by the fp32 matrix pipe for MDG_PREC_F32.
// Synchronisation: one raw s_barrier per stage and a counted `s_waitcnt vmcnt(32)` that retires the LDS-DMA of
// the next tile while the 32 score stores of the stage stay in flight (details at the main loop).
//
// LDS tile layout: [64 tail rows][D] with the 16-byte chunks of a row XOR-swizzled by (row & 15), so
// the MFMA B-operand reads (lane = tail row, ds_read_b128) are bank-conflict free.
// k ordering: fp32 MFMA consumes k = 8q+4h+e (q: 16-byte chunk pair, h: lane half, e: element) for
// BOTH operands -- any permutation of k is legal as long as A and B agree -- which turns the
// per-MFMA scalar operand into one ds_read_b128 per four MFMAs.
#include "mdg_common.h"
#include <stdlib.h>

namespace {

constexpr int D = 128;
constexpr int BN = 64;             // tail rows per stage
constexpr int STAGE_BYTES = BN * D * 4;   // fp32 tile, or bf16 hi (16 KB) + lo (16 KB)
constexpr int LO_OFF = BN * D * 2;

struct TileSrc {
  const float* f32;
  const __bf16* hi;
  const __bf16* lo;
  int64_t nrows;
};

struct BilinearArgs {
  const float* z_head;
  TileSrc zt;
  TileSrc w;          // W_sym (all labels); nrows = D
  float* out;
  int64_t n_head, n_tail, n_labels;
  int64_t ldo;          // row pitch of out in floats (n_tail)
  const float* zt_raw;  // z_tail and W_sym as the caller passed them (fp32): the column strip of the symmetric sweep rounds them itself
  const float* w_raw;
  int stagger;          // per-workgroup sweep start (HBM channel spreading)
  int stagger_waves;    // counted pipeline: younger half of the waves stores one stage late
  int loaders;          // conservative pipeline: waves that issue the LDS-DMA (1, 2 or 4)
  int pipeline;         // 0 = counted waits (default), 1 = conservative
  int symmetric;        // z_head and z_tail are the same matrix (same pointer, same row count)
  unsigned long long* stamps;   // diagnostics only (mdg_debug_bilinear_stamps): per workgroup {shader cycles, 100 MHz ticks} of the sweep
};

// 16-bit operand images travel as bf16x8 containers; MDG_PREC_F16 stores IEEE half bits in them and casts at the MFMA.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

template <int MODE> struct AFrag;
template <> struct AFrag<MDG_PREC_F32> { float a[64]; };
template <> struct AFrag<MDG_PREC_BF16X3> { bf16x8 hi[8]; bf16x8 lo[8]; };
template <> struct AFrag<MDG_PREC_BF16> { bf16x8 hi[8]; };
template <> struct AFrag<MDG_PREC_F16> { bf16x8 hi[8]; };

// one rounded product per k-step (operands rounded to bf16 / fp16 once)
template <int MODE> constexpr bool kSingle16 = (MODE == MDG_PREC_BF16 || MODE == MDG_PREC_F16);

template <int MODE>
__device__ __forceinline__ f32x16 mma16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (MODE == MDG_PREC_F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <int ROWB>
__device__ __forceinline__ int tile_off(int row, int chunk) {
  return row * ROWB + ((chunk ^ (row & 15)) << 4);
}

// ---- global -> registers -> LDS staging of one 64-row tile ---------------------------------
template <int MODE, int NW>
__device__ __forceinline__ void stage_load(const TileSrc& s, int64_t row0, int tid, u32x4 (&regs)[32 / NW]) {
  constexpr int NTHREADS = 64 * NW;
  if constexpr (MODE == MDG_PREC_F32) {
#pragma unroll
    for (int i = 0; i < 32 / NW; ++i) {
      const int g = tid + NTHREADS * i, row = g >> 5, c = g & 31;
      int64_t gr = row0 + row;
      gr = gr < s.nrows ? gr : s.nrows - 1;
      regs[i] = *reinterpret_cast<const u32x4*>(s.f32 + gr * D + c * 4);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16 / NW; ++i) {
      const int g = tid + NTHREADS * i, row = g >> 4, c = g & 15;
      int64_t gr = row0 + row;
      gr = gr < s.nrows ? gr : s.nrows - 1;
      regs[i] = *reinterpret_cast<const u32x4*>(s.hi + gr * D + c * 8);
      if constexpr (MODE == MDG_PREC_BF16X3) regs[16 / NW + i] = *reinterpret_cast<const u32x4*>(s.lo + gr * D + c * 8);
    }
  }
}

template <int MODE, int NW>
__device__ __forceinline__ void stage_write(char* lds, int tid, const u32x4 (&regs)[32 / NW]) {
  constexpr int NTHREADS = 64 * NW;
  if constexpr (MODE == MDG_PREC_F32) {
#pragma unroll
    for (int i = 0; i < 32 / NW; ++i) {
      const int g = tid + NTHREADS * i, row = g >> 5, c = g & 31;
      *reinterpret_cast<u32x4*>(lds + tile_off<512>(row, c)) = regs[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16 / NW; ++i) {
      const int g = tid + NTHREADS * i, row = g >> 4, c = g & 15;
      *reinterpret_cast<u32x4*>(lds + tile_off<256>(row, c)) = regs[i];
      if constexpr (MODE == MDG_PREC_BF16X3) *reinterpret_cast<u32x4*>(lds + LO_OFF + tile_off<256>(row, c)) = regs[16 / NW + i];
    }
  }
}


// ---- global -> LDS staging by LDS-DMA (no staging registers, asynchronous) ------------------
// One wave-instruction moves 1 KiB: LDS destination = wave-uniform base + lane*16 (linear), the
// per-lane SOURCE address carries the chunk swizzle (cdna guide rule 21: linear dest + swizzled
// source + the same swizzle on the read).  Completion is tracked by the issuing wave's vmcnt.
// Issued as inline asm so that hipcc's waitcnt pass does not see it (with the builtin it drains
// vmcnt(0) -- i.e. every score store in flight -- before the next ds_read); the waits are counted
// by hand in the kernel.  M0 carries the LDS base and is restored (cdna guide 5.7).
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return static_cast<unsigned>(reinterpret_cast<size_t>((lds_void*)p));
}

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst_uniform);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}

// `nload` waves (0..nload-1) share the pieces of a tile; the other waves issue nothing.
template <int MODE>
__device__ __forceinline__ void stage_dma(const TileSrc& s, int64_t row0, char* lds, int wave, int lane, int nload) {
  if constexpr (MODE == MDG_PREC_F32) {
    for (int p = wave; p < 32; p += nload) {
      const int row = 2 * p + (lane >> 5), c = (lane & 31) ^ (row & 15);
      int64_t gr = row0 + row;
      gr = gr < s.nrows ? gr : s.nrows - 1;
      glds16(s.f32 + gr * D + c * 4, lds_addr(lds + p * 1024));
    }
  } else {
    for (int p = wave; p < 16; p += nload) {
      const int row = 4 * p + (lane >> 4), c = (lane & 15) ^ (row & 15);
      int64_t gr = row0 + row;
      gr = gr < s.nrows ? gr : s.nrows - 1;
      glds16(s.hi + gr * D + c * 8, lds_addr(lds + p * 1024));
      if constexpr (MODE == MDG_PREC_BF16X3) glds16(s.lo + gr * D + c * 8, lds_addr(lds + LO_OFF + p * 1024));
    }
  }
}

// ---- A fragment from 8 consecutive fp32 values --------------------------------------------
template <int MODE>
__device__ __forceinline__ void split8(const float4& x0, const float4& x1, bf16x8& hi, bf16x8& lo) {
  const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
  if constexpr (MODE == MDG_PREC_F16) {
    f16x8 t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = static_cast<_Float16>(v[j]);       // round to nearest even
    hi = __builtin_bit_cast(bf16x8, t);
    lo = hi;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __bf16 a, b;
      mdg_split_bf16(v[j], a, b);
      hi[j] = a;
      lo[j] = b;
    }
  }
}

// rows of z_head straight from global memory (one-time, 512 B per lane)
template <int MODE>
__device__ __forceinline__ void afrag_from_global(AFrag<MODE>& A, const float* row, int h) {
  if constexpr (MODE == MDG_PREC_F32) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(row + 8 * q + 4 * h);
      A.a[4 * q + 0] = v.x; A.a[4 * q + 1] = v.y; A.a[4 * q + 2] = v.z; A.a[4 * q + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float4 v0 = *reinterpret_cast<const float4*>(row + 16 * s + 8 * h);
      const float4 v1 = *reinterpret_cast<const float4*>(row + 16 * s + 8 * h + 4);
      bf16x8 hi, lo;
      split8<MODE>(v0, v1, hi, lo);
      A.hi[s] = hi;
      if constexpr (MODE == MDG_PREC_BF16X3) A.lo[s] = lo;
    }
  }
}

// half of the T fragment (k in [64*st, 64*st+64)) from this wave's [32][64] fp32 slab in LDS
template <int MODE>
__device__ __forceinline__ void afrag_from_slab(AFrag<MODE>& A, const char* slab, int st, int r, int h) {
  if constexpr (MODE == MDG_PREC_F32) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(slab + tile_off<256>(r, 2 * q + h));
      const int o = 4 * (8 * st + q);
      A.a[o + 0] = v.x; A.a[o + 1] = v.y; A.a[o + 2] = v.z; A.a[o + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float4 v0 = *reinterpret_cast<const float4*>(slab + tile_off<256>(r, 4 * s + 2 * h));
      const float4 v1 = *reinterpret_cast<const float4*>(slab + tile_off<256>(r, 4 * s + 2 * h + 1));
      bf16x8 hi, lo;
      split8<MODE>(v0, v1, hi, lo);
      A.hi[4 * st + s] = hi;
      if constexpr (MODE == MDG_PREC_BF16X3) A.lo[4 * st + s] = lo;
    }
  }
}

// ---- 32 rows (A, registers) x 64 staged tail rows (B, LDS) -> two 32x32 accumulators --------
template <int MODE>
__device__ __forceinline__ void compute_tile(const AFrag<MODE>& A, const char* lds, int r, int h, f32x16 (&acc)[2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int j = 32 * t + r;
    if constexpr (MODE == MDG_PREC_F32) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float4 b = *reinterpret_cast<const float4*>(lds + tile_off<512>(j, 2 * q + h));
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.a[4 * q + 0], b.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.a[4 * q + 1], b.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.a[4 * q + 2], b.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.a[4 * q + 3], b.w, acc[t], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(lds + tile_off<256>(j, 2 * s + h));
        if constexpr (MODE == MDG_PREC_BF16X3) {
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(lds + LO_OFF + tile_off<256>(j, 2 * s + h));
          acc[t] = mma16<MODE>(A.lo[s], bh, acc[t]);
          acc[t] = mma16<MODE>(A.hi[s], bl, acc[t]);
        }
        acc[t] = mma16<MODE>(A.hi[s], bh, acc[t]);
      }
    }
  }
}

// The same products with the 32 score stores of the PREVIOUS tile (held in registers by the caller) spread evenly between
// the MFMAs instead of issued as one burst: `store_k(k)`, k = 0..31, issues store k.  B fragments are fetched one step
// ahead by hand because the scheduling barriers that pin the store positions also stop the compiler from hoisting them.
template <int MODE, typename StoreFn>
__device__ __forceinline__ void compute_tile_spread(const AFrag<MODE>& A, const char* lds, int r, int h, f32x16 (&acc)[2],
                                                    StoreFn&& store_k) {
  if constexpr (MODE == MDG_PREC_F32) {
    float4 b = *reinterpret_cast<const float4*>(lds + tile_off<512>(r, h));
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int t = i >> 4, q = i & 15;
      float4 nb = b;
      if (i + 1 < 32) nb = *reinterpret_cast<const float4*>(lds + tile_off<512>(32 * ((i + 1) >> 4) + r, 2 * ((i + 1) & 15) + h));
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.a[4 * q + 0], b.x, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.a[4 * q + 1], b.y, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.a[4 * q + 2], b.z, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.a[4 * q + 3], b.w, acc[t], 0, 0, 0);
      store_k(i);
      __builtin_amdgcn_sched_barrier(0);
      b = nb;
    }
  } else {
    bf16x8 bh = *reinterpret_cast<const bf16x8*>(lds + tile_off<256>(r, h)), bl = bh;
    if constexpr (MODE == MDG_PREC_BF16X3) bl = *reinterpret_cast<const bf16x8*>(lds + LO_OFF + tile_off<256>(r, h));
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = i >> 3, s = i & 7;
      bf16x8 nbh = bh, nbl = bl;
      if (i + 1 < 16) {
        const int j = 32 * ((i + 1) >> 3) + r, c = 2 * ((i + 1) & 7) + h;
        nbh = *reinterpret_cast<const bf16x8*>(lds + tile_off<256>(j, c));
        if constexpr (MODE == MDG_PREC_BF16X3) nbl = *reinterpret_cast<const bf16x8*>(lds + LO_OFF + tile_off<256>(j, c));
      }
      if constexpr (MODE == MDG_PREC_BF16X3) {
        acc[t] = mma16<MODE>(A.lo[s], bh, acc[t]);
        store_k(2 * i);
        acc[t] = mma16<MODE>(A.hi[s], bl, acc[t]);
        store_k(2 * i + 1);
        acc[t] = mma16<MODE>(A.hi[s], bh, acc[t]);
      } else {
        acc[t] = mma16<MODE>(A.hi[s], bh, acc[t]);
        store_k(2 * i);
        store_k(2 * i + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      bh = nbh;
      bl = nbl;
    }
  }
}

// accumulator register v of lane (r,h) is element [row (v&3)+8(v>>2)+4h][col r] of the 32x32 tile
__device__ __forceinline__ int acc_row(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }

// RB = 32-row blocks of z_head per wave.  RB = 2 (row statistics in bf16 only): every B fragment read from LDS feeds two
// MFMAs instead of one — with a single bf16 product per k-step the sweep is otherwise bound by the LDS operand reads
// (one ds_read_b128 per MFMA), not by the matrix cores.
// VAR = 1: every wave keeps the finished tile in registers for one stage and issues its 32 stores one or two at a time
// between the MFMAs of the next tile (compute_tile_spread), instead of the burst-per-stage of VAR = 0.
// VAR = 2: the finished 32 x 64 tile of a wave is transposed through a wave-private 8 KB LDS slab (32 x ds_write_b32, lane =
// column, then 8 x ds_read_b128, lane = 4 consecutive columns of one row) and leaves as 8 x buffer_store_dwordx4 -- 1 KB per
// store instruction (4 whole 256-byte row segments) instead of 256 B -- issued between the MFMAs of the next tile.  Needs
// n_tail % 4 == 0 and a 16-byte aligned output (the launcher falls back to VAR 0 otherwise).
template <int MODE, int EPI, int NW, int RB = 1, int VAR = 0>
__global__ __launch_bounds__(64 * NW, 2) void bilinear_allpairs_kernel(const BilinearArgs p) {
  static_assert(NW == 4 || NW == 8, "the counted waits below are written for 4 or 8 waves per workgroup");
  constexpr int BM = 32 * NW * RB;   // head rows per workgroup
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const buf0 = smem;
  char* const buf1 = smem + STAGE_BYTES;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int64_t l = blockIdx.y;
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * BM;

  // ---------------- prologue: T = z_head[rows] . W_sym[l], kept as the A operand -------------
  AFrag<MODE> Ats[RB];
  AFrag<MODE>& At = Ats[0];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    AFrag<MODE> Az;
    int64_t zr = row0 + (wave * RB + rb) * 32 + r;
    zr = zr < p.n_head ? zr : p.n_head - 1;
    afrag_from_global<MODE>(Az, p.z_head + zr * D, h);
    TileSrc ws = p.w;
    if constexpr (MODE == MDG_PREC_F32) ws.f32 += l * D * D;
    else { ws.hi += l * D * D; if constexpr (MODE == MDG_PREC_BF16X3) ws.lo += l * D * D; }
    char* const slab = smem + wave * 8192;       // [32 rows][64 cols] fp32, chunk-swizzled
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      u32x4 regs[32 / NW];
      stage_load<MODE, NW>(ws, 64 * st, tid, regs);
      __syncthreads();                            // slabs of the previous half are consumed
      stage_write<MODE, NW>(buf0, tid, regs);
      __syncthreads();
      f32x16 acc[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
      compute_tile<MODE>(Az, buf0, r, h, acc);
      __syncthreads();                            // every wave is done reading buf0
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int row = acc_row(v, h), n = 32 * t + r;
          *reinterpret_cast<float*>(slab + tile_off<256>(row, n >> 2) + (n & 3) * 4) = acc[t][v];
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      afrag_from_slab<MODE>(Ats[rb], slab, st, r, h);
    }
    __syncthreads();
  }

  // ---------------- main sweep over the tail drugs -----------------------------------------
  unsigned long long stamp_c = 0, stamp_r = 0;          // diagnostic build path: clock held during the sweep (never in outputs)
  if (p.stamps && tid == 0) { stamp_c = __builtin_amdgcn_s_memtime(); stamp_r = __builtin_amdgcn_s_memrealtime(); }
  const int nst = static_cast<int>((p.n_tail + BN - 1) / BN);
  const int64_t slab_rows = (p.n_head - row0) < BM ? (p.n_head - row0) : BM;
  float* const out_slab = (EPI == MDG_EPI_ROWSTATS) ? nullptr : p.out + (l * p.n_head + row0) * p.ldo;
  __amdgpu_buffer_rsrc_t rsrc;
  if constexpr (EPI != MDG_EPI_ROWSTATS)
    rsrc = __builtin_amdgcn_make_buffer_rsrc(out_slab, 0, static_cast<int>(slab_rows * p.ldo * 4), 0x00020000);
  f32x16 rsum, rmax, rsum2, rmax2;       // (second pair: row block 1 when RB == 2)
  if constexpr (EPI == MDG_EPI_ROWSTATS) {
#pragma unroll
    for (int v = 0; v < 16; ++v) { rsum[v] = 0.f; rmax[v] = -INFINITY; rsum2[v] = 0.f; rmax2[v] = -INFINITY; }
  }

  // Pipeline (default, p.pipeline == 0): ONE raw s_barrier per stage and a COUNTED wait.  Per stage and wave the
  // vector-memory stream is  [LDS-DMA pieces of tile s+1] [32 score stores]; `s_waitcnt vmcnt(32)` at the top of the next
  // stage retires the DMA (older) and leaves the 32 stores (younger) in flight.  That relies on loads, stores -- also
  // the out-of-range ones the buffer descriptor drops -- and LDS-DMA retiring in ISSUE ORDER (MI355X_MICROARCH.md:
  // "`s_waitcnt vmcnt(N)` waits until all but the wave's N youngest vector-memory operations are done. Loads, stores,
  // atomics and LDS-DMA count together, in issue order"; re-checked on the card by scripts/micro/vmcnt_order.hip) and
  // on EXACTLY 32 stores per wave and stage: ragged rows / columns are stored out of range (epilogue()), never skipped.
  //   s_waitcnt vmcnt(32)  this wave's DMA pieces of tile s have landed
  //   s_barrier            ... and everybody else's; every wave has finished reading tile s-1 (its buffer is reused)
  //   issue DMA(s+1)       asynchronous, lands under the MFMA phase
  //   MFMA(s) / stores     VAR 0: "early" waves MFMA(s) then the 32 stores of tile s; "late" waves (the younger half, which
  //                        shares each SIMD with an early wave) first the 32 stores of tile s-1, held in registers, then
  //                        MFMA(s): one partner stores while the other computes.  VAR 1: every wave is late and its stores
  //                        are spread between its MFMAs (compute_tile_spread).
  // A conservative pipeline without counted waits (full vmcnt(0) per stage) is kept behind MDG_BILINEAR_PIPELINE=1.
  // Each workgroup starts its sweep at a different tail tile (and wraps): co-resident workgroups otherwise write
  // addresses that differ only by multiples of the row / slab strides (16 KB, 4 MB at N=4096) at the same instant,
  // which piles them onto a few HBM channels.
  const int start = p.stagger ? static_cast<int>((blockIdx.x * 5u + blockIdx.y * 3u) % static_cast<unsigned>(nst)) : 0;
  auto tile_of = [&](int s) { int t = s + start; return t >= nst ? t - nst : t; };
  auto epilogue = [&](const f32x16 (&acc)[2], int64_t tcol0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t col = tcol0 + 32 * t + r;
      const bool col_ok = col < p.n_tail;
      if constexpr (EPI == MDG_EPI_ROWSTATS) {
        if (tcol0 + BN <= p.n_tail) {          // whole tile in range (wave-uniform): no per-element masking
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            rsum[v] += acc[t][v];
            rmax[v] = fmaxf(rmax[v], acc[t][v]);
          }
        } else {
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            rsum[v] += col_ok ? acc[t][v] : 0.f;
            rmax[v] = fmaxf(rmax[v], col_ok ? acc[t][v] : -INFINITY);
          }
        }
      } else {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int64_t e = static_cast<int64_t>(wave * 32 + acc_row(v, h)) * p.ldo + col;
          const unsigned off = col_ok ? static_cast<unsigned>(e * 4) : 0xFFFFFFFFu;   // out of range => dropped
          float val = acc[t][v];
          if constexpr (EPI == MDG_EPI_STORE_SIGMOID) val = 1.0f / (1.0f + expf(-val));
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rsrc, off, 0, 0);
        }
      }
    }
  };
  f32x16 held[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) held[t][v] = 0.f;
  int64_t held_col0 = 0;
  if (p.pipeline == 0) {
    // ---- counted pipeline (default).  Every wave issues its share of the LDS-DMA.  Per stage and wave the
    // vector-memory stream is  [DMA(s+1) x NDMA] [32 score stores]  (in either order of stores / MFMAs, see
    // below), so at the top of the next stage `s_waitcnt vmcnt(32)` retires the DMA while leaving the 32 stores
    // in flight.  This is sound because loads, stores (also the out-of-range ones that are dropped) and LDS-DMA
    // retire in issue order on gfx950 -- checked by scripts/micro/vmcnt_order.hip: 0 stale LDS reads in 131072
    // trials behind 32 dropped or real stores, 131072 of 131072 without them.  The store count per stage must
    // therefore be exactly 32 for every wave: ragged columns / rows are stored out of range, never skipped.
    // Stagger: the two waves sharing a SIMD (w, w + NW/2) run the same code between the same barriers and would
    // do their MFMAs together and their stores together; the younger half issues the stores of the PREVIOUS
    // tile (held in registers) before its MFMAs, so one partner stores while the other computes.
    const bool late = (EPI != MDG_EPI_ROWSTATS) && p.stagger_waves && (__builtin_amdgcn_readfirstlane(wave) >= NW / 2);
    if constexpr (EPI == MDG_EPI_ROWSTATS) {
      // Nothing is stored, so a stage is only the MFMAs of 64 tail rows (~1000 matrix-pipe cycles per SIMD): shorter
      // than the LDS-DMA latency.  Three buffers, prefetch distance two; the vector-memory stream of a wave holds
      // loads only (in order), so `vmcnt(NDMA)` retires tile s and leaves tile s+1 in flight.
      constexpr int NDMA = (kSingle16<MODE> ? 16 : 32) / NW;             // LDS-DMA instructions per wave and tile
      static_assert(NDMA == 2 || NDMA == 4 || NDMA == 8, "s_waitcnt immediates below cover NDMA in {2,4,8}");
      stage_dma<MODE>(p.zt, static_cast<int64_t>(tile_of(0)) * BN, smem, wave, lane, NW);
      stage_dma<MODE>(p.zt, static_cast<int64_t>(tile_of(1 < nst ? 1 : 0)) * BN, smem + STAGE_BYTES, wave, lane, NW);
      int cur = 0;
      for (int s = 0; s < nst; ++s) {
        const int64_t tcol0 = static_cast<int64_t>(tile_of(s)) * BN;
        if constexpr (NDMA == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // = NDMA: tile s+1 stays in flight
        else if constexpr (NDMA == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // tile s landed for every wave; every wave finished reading tile s-1
        const int nxt2 = cur == 0 ? 2 : cur - 1;                           // (cur + 2) % 3 = buffer of tile s-1
        const int s2 = s + 2 < nst ? s + 2 : nst - 1;                      // past the end: a copy nobody consumes
        stage_dma<MODE>(p.zt, static_cast<int64_t>(tile_of(s2)) * BN, smem + nxt2 * STAGE_BYTES, wave, lane, NW);
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
        if constexpr (RB == 2) {
          f32x16 acc1[2];
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc1[t][v] = 0.f;
          const char* lds = smem + cur * STAGE_BYTES;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
              const bf16x8 bh = *reinterpret_cast<const bf16x8*>(lds + tile_off<256>(32 * t + r, 2 * s + h));
              acc[t] = mma16<MODE>(Ats[0].hi[s], bh, acc[t]);
              acc1[t] = mma16<MODE>(Ats[RB - 1].hi[s], bh, acc1[t]);
            }
          epilogue(acc, tcol0);
          // row block 1: same reduction into the second accumulator pair
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            if (tcol0 + BN <= p.n_tail) {                 // whole tile in range (wave-uniform): no per-element masking
#pragma unroll
              for (int v = 0; v < 16; ++v) {
                rsum2[v] += acc1[t][v];
                rmax2[v] = fmaxf(rmax2[v], acc1[t][v]);
              }
            } else {
              const bool col_ok = tcol0 + 32 * t + r < p.n_tail;
#pragma unroll
              for (int v = 0; v < 16; ++v) {
                rsum2[v] += col_ok ? acc1[t][v] : 0.f;
                rmax2[v] = fmaxf(rmax2[v], col_ok ? acc1[t][v] : -INFINITY);
              }
            }
          }
        } else {
          compute_tile<MODE>(At, smem + cur * STAGE_BYTES, r, h, acc);
          epilogue(acc, tcol0);
        }
        cur = cur == 2 ? 0 : cur + 1;
      }
    } else if constexpr (VAR == 2) {
      char* const stg = smem + 2 * STAGE_BYTES + wave * 8192;          // [32 rows][64 columns] fp32, this wave's tile
      stage_dma<MODE>(p.zt, static_cast<int64_t>(tile_of(0)) * BN, buf0, wave, lane, NW);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int srow = lane >> 4, scol = 4 * (lane & 15);            // store role: row within a group of 4, first of 4 columns
      u32x4 o[8];
      auto out_store = [&](int q, int64_t col0) {                     // rows 4q .. 4q+3 of the wave's tile, 1 KB
        const int64_t col = col0 + scol;
        const int64_t e = static_cast<int64_t>(wave * 32 + 4 * q + srow) * p.ldo + col;
        const unsigned off = col < p.n_tail ? static_cast<unsigned>(e * 4) : 0xFFFFFFFFu;       // out of range => dropped
        u32x4 v = o[q];
        if constexpr (EPI == MDG_EPI_STORE_SIGMOID) {
          const f32x4 x = __builtin_bit_cast(f32x4, v);
          f32x4 y;
          y[0] = 1.0f / (1.0f + expf(-x[0])); y[1] = 1.0f / (1.0f + expf(-x[1]));
          y[2] = 1.0f / (1.0f + expf(-x[2])); y[3] = 1.0f / (1.0f + expf(-x[3]));
          v = __builtin_bit_cast(u32x4, y);
        }
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, off, 0, 0);
      };
      held_col0 = p.n_tail;                                           // first stage: the slab is empty, 8 dropped stores
      // Per stage and wave the vector-memory stream is [LDS-DMA of tile s+1][8 stores of tile s-1]: vmcnt(8) at the top of
      // the next stage retires the DMA and leaves the stores in flight (same in-order argument as VAR 0, 8 instead of 32).
      for (int s = 0; s < nst; ++s) {
        const int64_t tcol0 = static_cast<int64_t>(tile_of(s)) * BN;
        char* const cur = (s & 1) ? buf1 : buf0;
        char* const nxt = (s & 1) ? buf0 : buf1;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage_dma<MODE>(p.zt, static_cast<int64_t>(tile_of(s + 1 < nst ? s + 1 : s)) * BN, nxt, wave, lane, NW);
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
        const int64_t hc0 = held_col0;
        compute_tile_spread<MODE>(At, cur, r, h, acc, [&](int k) {
          if (k < 8) o[k] = *reinterpret_cast<const u32x4*>(stg + (4 * k + srow) * 256 + scol * 4);
          else if (k >= 16 && (k & 1) == 0) out_store((k - 16) >> 1, hc0);
        });
        // tile s -> the slab (the 8 reads above are older LDS operations of this wave: the LDS executes them first)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int v = 0; v < 16; ++v)
            *reinterpret_cast<float*>(stg + acc_row(v, h) * 256 + (32 * t + r) * 4) = acc[t][v];
        held_col0 = tcol0;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = *reinterpret_cast<const u32x4*>(stg + (4 * q + srow) * 256 + scol * 4);
#pragma unroll
      for (int q = 0; q < 8; ++q) out_store(q, held_col0);
    } else if constexpr (VAR == 1) {
      stage_dma<MODE>(p.zt, static_cast<int64_t>(tile_of(0)) * BN, buf0, wave, lane, NW);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      held_col0 = p.n_tail;                                       // first stage: 32 out-of-range (dropped) stores
      for (int s = 0; s < nst; ++s) {
        const int64_t tcol0 = static_cast<int64_t>(tile_of(s)) * BN;
        char* const cur = (s & 1) ? buf1 : buf0;
        char* const nxt = (s & 1) ? buf0 : buf1;
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage_dma<MODE>(p.zt, static_cast<int64_t>(tile_of(s + 1 < nst ? s + 1 : s)) * BN, nxt, wave, lane, NW);
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
        const int64_t hc0 = held_col0;
        compute_tile_spread<MODE>(At, cur, r, h, acc, [&](int k) {
          const int t = k >> 4, v = k & 15;
          const int64_t col = hc0 + 32 * t + r;
          const int64_t e = static_cast<int64_t>(wave * 32 + acc_row(v, h)) * p.ldo + col;
          const unsigned off = col < p.n_tail ? static_cast<unsigned>(e * 4) : 0xFFFFFFFFu;    // out of range => dropped
          float val = held[t][v];
          if constexpr (EPI == MDG_EPI_STORE_SIGMOID) val = 1.0f / (1.0f + expf(-val));
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rsrc, off, 0, 0);
        });
#pragma unroll
        for (int t = 0; t < 2; ++t) held[t] = acc[t];
        held_col0 = tcol0;
      }
      epilogue(held, held_col0);
    } else {
    stage_dma<MODE>(p.zt, static_cast<int64_t>(tile_of(0)) * BN, buf0, wave, lane, NW);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int s = 0; s < nst; ++s) {
      const int64_t tcol0 = static_cast<int64_t>(tile_of(s)) * BN;
      char* const cur = (s & 1) ? buf1 : buf0;
      char* const nxt = (s & 1) ? buf0 : buf1;
      asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // past the end the tile index repeats the last one; that copy is never consumed
      stage_dma<MODE>(p.zt, static_cast<int64_t>(tile_of(s + 1 < nst ? s + 1 : s)) * BN, nxt, wave, lane, NW);
      f32x16 acc[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
      if (late) {
        epilogue(held, s > 0 ? held_col0 : p.n_tail);          // s == 0: all 32 stores out of range (dropped)
        compute_tile<MODE>(At, cur, r, h, acc);
#pragma unroll
        for (int t = 0; t < 2; ++t) held[t] = acc[t];
        held_col0 = tcol0;
      } else {
        compute_tile<MODE>(At, cur, r, h, acc);
        epilogue(acc, tcol0);
      }
    }
    if (late) epilogue(held, held_col0);
    }
  } else {
    // ---- conservative pipeline (MDG_BILINEAR_PIPELINE=1): no counted waits.  Only the first `nload` waves issue
    // LDS-DMA and only they wait (a full vmcnt(0) once per stage, after their MFMAs and before their stores); the
    // other waves never wait for their stores inside the loop.
    const int nload = p.loaders;
    const bool loader = __builtin_amdgcn_readfirstlane(wave) < nload;
    if (loader) {
      stage_dma<MODE>(p.zt, static_cast<int64_t>(tile_of(0)) * BN, buf0, wave, lane, nload);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (int s = 0; s < nst; ++s) {
      const int64_t tcol0 = static_cast<int64_t>(tile_of(s)) * BN;
      char* const cur = (s & 1) ? buf1 : buf0;
      char* const nxt = (s & 1) ? buf0 : buf1;
      __builtin_amdgcn_s_barrier();
      if (loader && s + 1 < nst) stage_dma<MODE>(p.zt, static_cast<int64_t>(tile_of(s + 1)) * BN, nxt, wave, lane, nload);
      f32x16 acc[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
      compute_tile<MODE>(At, cur, r, h, acc);
      if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      epilogue(acc, tcol0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (p.stamps && tid == 0) {
    const size_t wg = static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x;
    p.stamps[2 * wg] = __builtin_amdgcn_s_memtime() - stamp_c;
    p.stamps[2 * wg + 1] = __builtin_amdgcn_s_memrealtime() - stamp_r;
  }

  if constexpr (EPI == MDG_EPI_ROWSTATS) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        float sv = rb == 0 ? rsum[v] : rsum2[v], mv = rb == 0 ? rmax[v] : rmax2[v];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
          sv += __shfl_xor(sv, o, 64);
          mv = fmaxf(mv, __shfl_xor(mv, o, 64));
        }
        const int64_t row = row0 + (wave * RB + rb) * 32 + acc_row(v, h);
        if (r == 0 && row < p.n_head) {
          float* o2 = p.out + (l * p.n_head + row) * 2;
          o2[0] = sv;
          o2[1] = mv;
        }
      }
  }
}

// ---- row statistics on the 16x16x32 matrix instruction (single-product 16-bit modes) --------------------------------------
// Same work as bilinear_allpairs_kernel<MODE, ROWSTATS, 8, 2> -- 64 head rows per wave, 64 tail rows per stage, three LDS
// buffers with prefetch distance two -- issued as v_mfma_f32_16x16x32 instead of 32x32x16: this loop is bound by the power
// envelope (MI355X_MICROARCH.md, DVFS give-back (7): the 16x16x32 form delivers ~1.15x the FLOP/s of 32x32x16 at equal
// cycles per FLOP because the card holds a higher clock under it).  Per stage and wave: 16 ds_read_b128 (one per 16-column
// tile and 32-deep k step), each feeding the 4 row tiles: 64 MFMAs.
typedef __attribute__((ext_vector_type(4))) float f32x4v;

template <int MODE>
__device__ __forceinline__ f32x4v mma16x16(const bf16x8& a, const bf16x8& b, const f32x4v& c) {
  if constexpr (MODE == MDG_PREC_F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// A operand of the 16x16x32 form for the 32 rows of a wave, from the wave's [32][64] fp32 slab of T (columns 64 st .. +63):
// lane (c16, g4) holds row 16 rt + c16, k = 32 ks + 8 g4 .. +7.
template <int MODE>
__device__ __forceinline__ void afrag16_from_slab(bf16x8 (&Ahi)[2][4], bf16x8 (&Alo)[2][4], const char* slab, int st, int c16, int g4) {
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ksl = 0; ksl < 2; ++ksl) {
      const int row = 16 * rt + c16, chunk = (32 * ksl + 8 * g4) >> 2;
      const float4 v0 = *reinterpret_cast<const float4*>(slab + tile_off<256>(row, chunk));
      const float4 v1 = *reinterpret_cast<const float4*>(slab + tile_off<256>(row, chunk + 1));
      bf16x8 hi, lo;
      split8<MODE>(v0, v1, hi, lo);
      Ahi[rt][2 * st + ksl] = hi;
      Alo[rt][2 * st + ksl] = lo;
    }
}

// 32 rows (A, registers) x 64 staged tail rows (B, LDS) on v_mfma_f32_16x16x32: acc[rt][ct] is the 16 x 16 block of rows
// 16 rt .., columns 16 ct ..; element i of lane (c16, g4) = row 16 rt + 4 g4 + i, column 16 ct + c16.  `hook(k)`, k = 0..31,
// is called twice per (column tile, k step) between the MFMAs (store / LDS traffic of the previous tile); B fragments are
// fetched one step ahead by hand (the scheduling barriers pin everything in place).
template <int MODE, typename Hook>
__device__ __forceinline__ void compute_tile_spread16(const bf16x8 (&Ahi)[2][4], const bf16x8 (&Alo)[2][4], const char* lds, int c16, int g4,
                                                      f32x4v (&acc)[2][4], Hook&& hook) {
  static_assert(MODE != MDG_PREC_F32, "16-bit operand modes");
  bf16x8 bh = *reinterpret_cast<const bf16x8*>(lds + tile_off<256>(c16, g4)), bl = bh;
  if constexpr (MODE == MDG_PREC_BF16X3) bl = *reinterpret_cast<const bf16x8*>(lds + LO_OFF + tile_off<256>(c16, g4));
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int ct = i >> 2, ks = i & 3;
    bf16x8 nbh = bh, nbl = bl;
    if (i + 1 < 16) {
      const int j = 16 * ((i + 1) >> 2) + c16, c = 4 * ((i + 1) & 3) + g4;
      nbh = *reinterpret_cast<const bf16x8*>(lds + tile_off<256>(j, c));
      if constexpr (MODE == MDG_PREC_BF16X3) nbl = *reinterpret_cast<const bf16x8*>(lds + LO_OFF + tile_off<256>(j, c));
    }
    if constexpr (MODE == MDG_PREC_BF16X3) {
      acc[0][ct] = mma16x16<MODE>(Alo[0][ks], bh, acc[0][ct]);
      acc[1][ct] = mma16x16<MODE>(Alo[1][ks], bh, acc[1][ct]);
      hook(2 * i);
      acc[0][ct] = mma16x16<MODE>(Ahi[0][ks], bl, acc[0][ct]);
      acc[1][ct] = mma16x16<MODE>(Ahi[1][ks], bl, acc[1][ct]);
      hook(2 * i + 1);
      acc[0][ct] = mma16x16<MODE>(Ahi[0][ks], bh, acc[0][ct]);
      acc[1][ct] = mma16x16<MODE>(Ahi[1][ks], bh, acc[1][ct]);
    } else {
      acc[0][ct] = mma16x16<MODE>(Ahi[0][ks], bh, acc[0][ct]);
      hook(2 * i);
      acc[1][ct] = mma16x16<MODE>(Ahi[1][ks], bh, acc[1][ct]);
      hook(2 * i + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    bh = nbh;
    bl = nbl;
  }
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void bilinear_rowstats16_kernel(const BilinearArgs p) {
  static_assert(kSingle16<MODE>, "one rounded 16-bit product per k step");
  constexpr int NW = 8, BM = 512;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const buf0 = smem;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5, c16 = lane & 15, g4 = lane >> 4;
  const int64_t l = blockIdx.y;
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * BM;
  // ---- prologue: T = z_head[rows] . W_sym[l] (32x32x16 products, as every other path), re-laid out for 16x16x32 ----
  bf16x8 A16[4][4];                                  // [row tile of 16][k step of 32]: lane (c16, g4) holds row c16, k = 32 ks + 8 g4 ..+7
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    AFrag<MODE> Az;
    int64_t zr = row0 + (wave * 2 + rb) * 32 + r;
    zr = zr < p.n_head ? zr : p.n_head - 1;
    afrag_from_global<MODE>(Az, p.z_head + zr * D, h);
    TileSrc ws = p.w;
    ws.hi += l * D * D;
    char* const slab = smem + wave * 8192;             // [32 rows][64 cols] fp32, chunk-swizzled
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      u32x4 regs[32 / NW];
      stage_load<MODE, NW>(ws, 64 * st, tid, regs);
      __syncthreads();
      stage_write<MODE, NW>(buf0, tid, regs);
      __syncthreads();
      f32x16 acc[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
      compute_tile<MODE>(Az, buf0, r, h, acc);
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int row = acc_row(v, h), n = 32 * t + r;
          *reinterpret_cast<float*>(slab + tile_off<256>(row, n >> 2) + (n & 3) * 4) = acc[t][v];
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
        for (int ksl = 0; ksl < 2; ++ksl) {
          const int row = 16 * rt2 + c16, chunk = (32 * ksl + 8 * g4) >> 2;         // 4-float chunks of the 64-column half
          const float4 v0 = *reinterpret_cast<const float4*>(slab + tile_off<256>(row, chunk));
          const float4 v1 = *reinterpret_cast<const float4*>(slab + tile_off<256>(row, chunk + 1));
          bf16x8 hi, lo;
          split8<MODE>(v0, v1, hi, lo);
          A16[2 * rb + rt2][2 * st + ksl] = hi;
        }
    }
    __syncthreads();
  }
  // ---- sweep ----
  const int nst = static_cast<int>((p.n_tail + BN - 1) / BN);
  const int start = p.stagger ? static_cast<int>((blockIdx.x * 5u + blockIdx.y * 3u) % static_cast<unsigned>(nst)) : 0;
  auto tile_of = [&](int s) { int t = s + start; return t >= nst ? t - nst : t; };
  float rsum[4][4], rmax[4][4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int i = 0; i < 4; ++i) { rsum[rt][i] = 0.f; rmax[rt][i] = -INFINITY; }
  constexpr int NDMA = 16 / NW;                       // LDS-DMA instructions per wave and tile (2)
  stage_dma<MODE>(p.zt, static_cast<int64_t>(tile_of(0)) * BN, smem, wave, lane, NW);
  stage_dma<MODE>(p.zt, static_cast<int64_t>(tile_of(1 < nst ? 1 : 0)) * BN, smem + STAGE_BYTES, wave, lane, NW);
  int cur = 0;
  for (int s = 0; s < nst; ++s) {
    const int64_t tcol0 = static_cast<int64_t>(tile_of(s)) * BN;
    static_assert(NDMA == 2, "vmcnt immediate below");
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");           // = NDMA: tile s landed, tile s+1 stays in flight
    __builtin_amdgcn_s_barrier();
    const int nxt2 = cur == 0 ? 2 : cur - 1;
    const int s2 = s + 2 < nst ? s + 2 : nst - 1;
    stage_dma<MODE>(p.zt, static_cast<int64_t>(tile_of(s2)) * BN, smem + nxt2 * STAGE_BYTES, wave, lane, NW);
    const char* lds = smem + cur * STAGE_BYTES;
    const bool whole = tcol0 + BN <= p.n_tail;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      f32x4v acc[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(lds + tile_off<256>(16 * ct + c16, 4 * ks + g4));
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = mma16x16<MODE>(A16[rt][ks], b, acc[rt]);
      }
      const bool col_ok = whole || (tcol0 + 16 * ct + c16 < p.n_tail);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          rsum[rt][i] += col_ok ? acc[rt][i] : 0.f;
          rmax[rt][i] = fmaxf(rmax[rt][i], col_ok ? acc[rt][i] : -INFINITY);
        }
    }
    cur = cur == 2 ? 0 : cur + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // accumulator element i of lane (c16, g4) of row tile rt is row 16 rt + 4 g4 + i, column c16: reduce over the 16 columns
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float sv = rsum[rt][i], mv = rmax[rt][i];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        sv += __shfl_xor(sv, o, 64);
        mv = fmaxf(mv, __shfl_xor(mv, o, 64));
      }
      const int64_t row = row0 + wave * 64 + 16 * rt + 4 * g4 + i;
      if (c16 == 0 && row < p.n_head) {
        float* o2 = p.out + (l * p.n_head + row) * 2;
        o2[0] = sv;
        o2[1] = mv;
      }
    }
}

// ---- symmetric sweep: z_head and z_tail are the SAME matrix ------------------------------------------------------
// All-pairs scoring of one drug set (generate_embeddings.ipynb / predict.py:428 call decoder(z, z, ...)): W_sym is
// symmetric, so S[l,i,j] = S[l,j,i] up to fp32 rounding of the two association orders (z_i W) z_j and (z_j W) z_i.
// The kernel is bound by the card's power envelope, not by the store stream (MI355X holds 1.40 GHz under the full
// bf16x3 sweep, 1.74 GHz with half the matrix work: DESIGN.md 4), so the matrix work is halved: only tiles on or right
// of the block diagonal are computed; an off-diagonal tile is stored twice, as computed and transposed.  The 256 x 256
// blocks ON the diagonal are computed in full (both triangles, as the general kernel does).
// Work split: row block a does nb - a column blocks, so workgroup x handles the pair of row blocks (x, nb-1-x), one after
// the other: every workgroup sweeps nb + 1 column blocks.  Stores: the wave's 32 x 64 tile goes through its 8 KB LDS slab
// twice -- column-major (8 x ds_write_b128) to leave as 64 rows x 128 B of the mirrored block, then row-major
// (32 x ds_write_b32) to leave as 32 rows x 256 B -- always as 16-byte-per-lane stores of whole 128-byte lines.
// SC1: the score stores carry the sc1 bit (write-through: the line is not kept in the XCD's L2, MI355X_MICROARCH.md "stores of
// each flavour"), so the 60 GB store stream does not evict the 2 MB of z_tail images the LDS-DMA re-reads from L2.
template <int MODE, int EPI, int NW, int SC1 = 0>
__global__ __launch_bounds__(64 * NW, 2) void bilinear_allpairs_sym_kernel(const BilinearArgs p) {
  static_assert(NW == 8, "256-row blocks: 8 waves of 32 rows");
  constexpr int AUX = SC1 ? 16 : 0;                     // gfx940+ cache-policy immediate: bit 4 = sc1
  static_assert(EPI == MDG_EPI_STORE || EPI == MDG_EPI_STORE_SIGMOID || EPI == MDG_EPI_TRIKEYS, "materialising epilogues only");
  // TRIKEYS: the sort keys of the strict lower triangle.  A tile right of the diagonal block leaves through its mirrored stores only
  // (those ARE the lower triangle); the diagonal block is written in full, its upper half is ignored by the reader
  constexpr bool TRI = (EPI == MDG_EPI_TRIKEYS);
  constexpr int BM = 32 * NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const buf0 = smem;
  char* const buf1 = smem + STAGE_BYTES;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5, c16 = lane & 15, g4 = lane >> 4;
  // 16-bit operand modes run the sweep on v_mfma_f32_16x16x32 (the loop is bound by the power envelope: the card holds a higher
  // clock under that form, MI355X_MICROARCH.md DVFS give-back (7)); exact fp32 stays on 32x32x2
  constexpr bool M16 = (MODE != MDG_PREC_F32);
  char* const stg = smem + 2 * STAGE_BYTES + wave * 8192;
  const int64_t l = blockIdx.y, N = p.n_tail, ld = p.ldo;
  // Every store writes 4 consecutive columns.  With a row pitch that leaves room (ld >= N rounded up to 4: the pitched layout,
  // rows 128-byte aligned) the last group of a row runs into the padding.  In the contiguous layout (ld == N) with N % 4 != 0
  // that group is dropped here and the N % 4 last columns come from bilinear_strip_kernel; rows then start at any 4-byte
  // alignment, the 16-byte stores are unaligned and straddle cache lines (correct, 2-3x slower: partial-line writes).
  const int64_t N4 = (ld >= ((N + 3) & ~static_cast<int64_t>(3))) ? N : (N & ~static_cast<int64_t>(3));
  const int nst = static_cast<int>((N + BN - 1) / BN);
  const int nb = static_cast<int>((N + BM - 1) / BM);
  const int srow = lane >> 4, scol = 4 * (lane & 15);            // row-major store role: row in a group of 4, first of 4 columns
  const int mrow = lane >> 3, mchunk = lane & 7;                 // mirrored store role: tile column in a group of 8, 16-B chunk of 32 rows
  float* const out_l = p.out + l * N * ld;
  // whole [N,ld] slab of this outcome (N * ld * 4 < 2^32 is checked by the launcher): the mirrored stores land anywhere in it
  const __amdgpu_buffer_rsrc_t rs_all = __builtin_amdgcn_make_buffer_rsrc(out_l, 0, static_cast<int>(static_cast<unsigned>(N * ld * 4)), 0x00020000);
  auto sig = [&](u32x4 v) {
    if constexpr (EPI == MDG_EPI_STORE_SIGMOID) {
      const f32x4 x = __builtin_bit_cast(f32x4, v);
      f32x4 y;
      y[0] = 1.0f / (1.0f + expf(-x[0])); y[1] = 1.0f / (1.0f + expf(-x[1]));
      y[2] = 1.0f / (1.0f + expf(-x[2])); y[3] = 1.0f / (1.0f + expf(-x[3]));
      return __builtin_bit_cast(u32x4, y);
    } else if constexpr (TRI) {
      const f32x4 x = __builtin_bit_cast(f32x4, v);
      return u32x4{mdg_order_key(x[0]), mdg_order_key(x[1]), mdg_order_key(x[2]), mdg_order_key(x[3])};
    } else {
      return v;
    }
  };
  unsigned long long stamp_c = 0, stamp_r = 0;
  if (p.stamps && tid == 0) { stamp_c = __builtin_amdgcn_s_memtime(); stamp_r = __builtin_amdgcn_s_memrealtime(); }

  for (int half = 0; half < 2; ++half) {
    const int rbk = half == 0 ? static_cast<int>(blockIdx.x) : nb - 1 - static_cast<int>(blockIdx.x);
    if (half == 1 && rbk == static_cast<int>(blockIdx.x)) break;          // odd number of row blocks: the middle one once
    const int64_t row0 = static_cast<int64_t>(rbk) * BM;
    // ---- prologue: T = z[rows] . W_sym[l] as the A operand (same arithmetic as the general kernel) ----
    AFrag<MODE> At;
    bf16x8 A16hi[2][4], A16lo[2][4];
    {
      AFrag<MODE> Az;
      int64_t zr = row0 + wave * 32 + r;
      zr = zr < N ? zr : N - 1;
      afrag_from_global<MODE>(Az, p.z_head + zr * D, h);
      TileSrc ws = p.w;
      if constexpr (MODE == MDG_PREC_F32) ws.f32 += l * D * D;
      else { ws.hi += l * D * D; if constexpr (MODE == MDG_PREC_BF16X3) ws.lo += l * D * D; }
      char* const slab = smem + wave * 8192;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        u32x4 regs[32 / NW];
        stage_load<MODE, NW>(ws, 64 * st, tid, regs);
        __syncthreads();
        stage_write<MODE, NW>(buf0, tid, regs);
        __syncthreads();
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
        compute_tile<MODE>(Az, buf0, r, h, acc);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int row = acc_row(v, h), n = 32 * t + r;
            *reinterpret_cast<float*>(slab + tile_off<256>(row, n >> 2) + (n & 3) * 4) = acc[t][v];
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if constexpr (M16) afrag16_from_slab<MODE>(A16hi, A16lo, slab, st, c16, g4);
        else afrag_from_slab<MODE>(At, slab, st, r, h);
      }
      __syncthreads();
    }
    // ---- sweep: column tiles t0 .. nst-1 (the diagonal block first), rotated per workgroup (HBM channel spreading) ----
    const int t0 = rbk * (BM / BN);
    const int t_diag_end = t0 + BM / BN;                            // tiles below this index lie inside the diagonal block
    const int nt = nst - t0;
    const int start = p.stagger ? static_cast<int>((blockIdx.x * 5u + blockIdx.y * 3u) % static_cast<unsigned>(nt)) : 0;
    auto tile_of = [&](int s) { int t = s + start; return t0 + (t >= nt ? t - nt : t); };
    const int64_t slab_rows = (N - row0) < BM ? (N - row0) : BM;
    const __amdgpu_buffer_rsrc_t rs_rows = __builtin_amdgcn_make_buffer_rsrc(out_l + row0 * ld, 0, static_cast<int>(slab_rows * ld * 4), 0x00020000);
    u32x4 o[8];
    auto store_rows = [&](int q, int64_t col0, bool on) {             // rows 4q..4q+3 of the wave's tile: 4 x 256 B
      const int64_t col = col0 + scol;
      const int64_t e = static_cast<int64_t>(wave * 32 + 4 * q + srow) * ld + col;
      const unsigned off = (on && col < N4) ? static_cast<unsigned>(e * 4) : 0xFFFFFFFFu;     // out of range => dropped
      __builtin_amdgcn_raw_buffer_store_b128(sig(o[q]), rs_rows, off, 0, AUX);
    };
    auto store_mirror = [&](int q, int64_t col0, bool on) {           // tile columns 8q..8q+7 as rows of the mirrored block: 8 x 128 B
      const int64_t mr = col0 + 8 * q + mrow, mc = row0 + wave * 32 + 4 * mchunk;
      const unsigned off = (on && mr < N && mc < N4) ? static_cast<unsigned>((mr * ld + mc) * 4) : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_buffer_store_b128(sig(o[q]), rs_all, off, 0, AUX);
    };
    stage_dma<MODE>(p.zt, static_cast<int64_t>(tile_of(0)) * BN, buf0, wave, lane, NW);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int64_t prev_col0 = N;                                            // first stage: the slab is empty, 8 dropped stores
    bool prev_rows = true;                                            // TRIKEYS: the previous tile has row stores (diagonal block)
    bool stage_rows = true;                                           // ... and the last stage issued them
    // Per stage and wave the vector-memory stream is [LDS-DMA of tile s+1][8 row stores of tile s-1][8 mirrored stores of
    // tile s]: `s_waitcnt vmcnt(16)` at the top of the next stage retires the DMA and leaves 16 stores in flight (loads,
    // stores -- dropped ones included -- and LDS-DMA retire in issue order; exactly 16 stores per wave and stage).
    for (int s = 0; s < nt; ++s) {
      const int tile = tile_of(s);
      const int64_t tcol0 = static_cast<int64_t>(tile) * BN;
      char* const cur = (s & 1) ? buf1 : buf0;
      char* const nxt = (s & 1) ? buf0 : buf1;
      // TRIKEYS: a stage whose previous tile lies right of the diagonal block has no row stores at all (8 stores per wave, not 16)
      if (TRI && !stage_rows) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      stage_dma<MODE>(p.zt, static_cast<int64_t>(tile_of(s + 1 < nt ? s + 1 : s)) * BN, nxt, wave, lane, NW);
      const int64_t pc0 = prev_col0;
      const bool mirror = tile >= t_diag_end;
      const bool pr = prev_rows;
      stage_rows = pr;
      auto hook = [&](int k) {
        if (TRI && !pr) return;
        if (k < 8) o[k] = *reinterpret_cast<const u32x4*>(stg + (4 * k + srow) * 256 + scol * 4);
        else if (k >= 16 && (k & 1) == 0) store_rows((k - 16) >> 1, pc0, true);
      };
      if constexpr (M16) {
        f32x4v acc[2][4];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
        compute_tile_spread16<MODE>(A16hi, A16lo, cur, c16, g4, acc, hook);
        // tile s, column-major ([64 columns][32 rows]): a lane's 4 consecutive rows of one column are 16 contiguous bytes
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct)
            *reinterpret_cast<f32x4v*>(stg + (16 * ct + c16) * 128 + (16 * rt + 4 * g4) * 4) = acc[rt][ct];
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = *reinterpret_cast<const u32x4*>(stg + (8 * q + mrow) * 128 + mchunk * 16);
#pragma unroll
        for (int q = 0; q < 8; ++q) store_mirror(q, tcol0, mirror);
        // tile s, row-major, for the row stores of the next stage (the reads above are older LDS operations of this wave)
        if (!TRI || !mirror) {
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
              for (int i = 0; i < 4; ++i)
                *reinterpret_cast<float*>(stg + (16 * rt + 4 * g4 + i) * 256 + (16 * ct + c16) * 4) = acc[rt][ct][i];
        }
      } else {
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
        compute_tile_spread<MODE>(At, cur, r, h, acc, hook);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v4 = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
            *reinterpret_cast<f32x4*>(stg + (32 * t + r) * 128 + (8 * g + 4 * h) * 4) = v4;
          }
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = *reinterpret_cast<const u32x4*>(stg + (8 * q + mrow) * 128 + mchunk * 16);
#pragma unroll
        for (int q = 0; q < 8; ++q) store_mirror(q, tcol0, mirror);
        if (!TRI || !mirror) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v)
              *reinterpret_cast<float*>(stg + acc_row(v, h) * 256 + (32 * t + r) * 4) = acc[t][v];
        }
      }
      prev_col0 = tcol0;
      if constexpr (TRI) prev_rows = !mirror;
    }
    if (!TRI || prev_rows) {
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = *reinterpret_cast<const u32x4*>(stg + (4 * q + srow) * 256 + scol * 4);
#pragma unroll
      for (int q = 0; q < 8; ++q) store_rows(q, prev_col0, true);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                  // the next row block's prologue reuses the stage buffers
  }
  if (p.stamps && tid == 0) {
    const size_t wg = static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x;
    p.stamps[2 * wg] = __builtin_amdgcn_s_memtime() - stamp_c;
    p.stamps[2 * wg + 1] = __builtin_amdgcn_s_memrealtime() - stamp_r;
  }
}

// ---- the N % 4 last columns of the symmetric sweep's matrix ------------------------------------------------------------
// The sweep stores groups of 4 columns; for N % 4 != 0 the columns [N4, N) of every row are left to this kernel:
// S[l, i, t] = z_i . (W_l z_t) -- W_l is symmetric, so this is the association the sweep's MIRRORED entries carry.  One workgroup
// per (256 rows, outcome); U_t = W_l z_t (at most 3 vectors of 128) is recomputed by every workgroup, then one row per thread.
// Operand roundings of the arithmetic mode are applied as the sweep applies them (16-bit modes: W, z rounded once, the
// intermediate vector rounded once; the split-bf16 mode is fp32-grade and runs in plain fp32); the fp32 summation order differs
// from the matrix cores', i.e. the strip agrees with the general kernel to the mode's accumulation noise, not bit for bit.
template <int MODE>
__device__ __forceinline__ float round_operand(float v) {
  if constexpr (MODE == MDG_PREC_BF16) return static_cast<float>(static_cast<__bf16>(v));
  else if constexpr (MODE == MDG_PREC_F16) return static_cast<float>(static_cast<_Float16>(v));
  else return v;
}

template <int MODE, int EPI>
__global__ __launch_bounds__(256) void bilinear_strip_kernel(const BilinearArgs p) {
  __shared__ float U[3][D];
  const int64_t l = blockIdx.y, N = p.n_tail, N4 = N & ~static_cast<int64_t>(3);
  const int nt = static_cast<int>(N - N4), tid = threadIdx.x;
  const float* W = p.w_raw + l * D * D;
  for (int e = tid; e < nt * D; e += 256) {
    const int t = e / D, k = e - t * D;
    const float* wr = W + static_cast<int64_t>(k) * D;
    const float* zt = p.zt_raw + (N4 + t) * D;
    float s = 0.f;
#pragma unroll 8
    for (int j = 0; j < D; j += 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(wr + j), b = *reinterpret_cast<const f32x4*>(zt + j);
#pragma unroll
      for (int c = 0; c < 4; ++c) s += round_operand<MODE>(a[c]) * round_operand<MODE>(b[c]);
    }
    U[t][k] = round_operand<MODE>(s);
  }
  __syncthreads();
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + tid;
  if (i >= N) return;
  const float* zi = p.z_head + i * D;
  float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll 8
  for (int k = 0; k < D; k += 4) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(zi + k);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float av = round_operand<MODE>(a[c]);
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] += av * U[t < nt ? t : 0][k + c];
    }
  }
  float* o = p.out + (l * N + i) * p.ldo + N4;
  for (int t = 0; t < nt; ++t) o[t] = (EPI == MDG_EPI_STORE_SIGMOID) ? 1.0f / (1.0f + expf(-acc[t])) : acc[t];
}

// ---- pre-passes ---------------------------------------------------------------------------
__global__ void symmetrize_kernel(const float* __restrict__ w, float* __restrict__ ws, int64_t L, int D_) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t per = static_cast<int64_t>(D_) * D_;
  if (idx >= L * per) return;
  const int64_t l = idx / per;
  const int a = static_cast<int>((idx % per) / D_), b = static_cast<int>(idx % D_);
  // read both before writing: in-place is allowed (each thread owns exactly one output element,
  // and the element it may read besides its own, [b][a] with a > b, is an upper-triangle entry
  // that its owner rewrites with the same value).
  const float v = (a <= b) ? w[idx] : w[l * per + static_cast<int64_t>(b) * D_ + a];
  ws[idx] = v;
}

__global__ void split_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ hi, __bf16* __restrict__ lo,
                                  int64_t n4) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 v = reinterpret_cast<const float4*>(x)[i];
  bf16x4 h, l;
  __bf16 a, b;
  mdg_split_bf16(v.x, a, b); h[0] = a; l[0] = b;
  mdg_split_bf16(v.y, a, b); h[1] = a; l[1] = b;
  mdg_split_bf16(v.z, a, b); h[2] = a; l[2] = b;
  mdg_split_bf16(v.w, a, b); h[3] = a; l[3] = b;
  reinterpret_cast<bf16x4*>(hi)[i] = h;
  if (lo) reinterpret_cast<bf16x4*>(lo)[i] = l;
}

__global__ void round_f16_kernel(const float* __restrict__ x, _Float16* __restrict__ y, int64_t n4) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 v = reinterpret_cast<const float4*>(x)[i];
  typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
  f16x4 o;
  o[0] = static_cast<_Float16>(v.x); o[1] = static_cast<_Float16>(v.y); o[2] = static_cast<_Float16>(v.z); o[3] = static_cast<_Float16>(v.w);
  reinterpret_cast<f16x4*>(y)[i] = o;
}

inline size_t align256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

template <int MODE, int NW>
int launch_allpairs_nw(const BilinearArgs& a, int epilogue, hipStream_t st) {
  const dim3 grid(static_cast<unsigned>(mdg_cdiv(a.n_head, 32 * NW)), static_cast<unsigned>(a.n_labels));
  const dim3 block(64 * NW);
  const size_t lds = 2 * STAGE_BYTES;
  // (store schedule of the general sweep: one burst of 32 dword stores per wave and stage, early / late halves.  Rounds 2-3 also had
  // the stores spread between the next tile's MFMAs and a tile transposed through LDS for 16-byte stores: neither won, removed)
  const size_t lds2 = lds + static_cast<size_t>(NW) * 8192;
  // Symmetric sweep (z_head and z_tail are the same matrix): half the matrix work, every off-diagonal tile stored twice.
  // Default for that case; MDG_BILINEAR_SYMMETRIC=0 switches it off.
  if constexpr (NW == 8) {
    static MdgEnvInt sym_sw{"MDG_BILINEAR_SYMMETRIC", 1};
    const bool want = sym_sw.get() != 0;
    if (epilogue == MDG_EPI_TRIKEYS) {        // keys of the lower triangle: a product of the symmetric sweep only
      MDG_CHECK_ARG(a.symmetric, "mdg_bilinear_allpairs: MDG_EPI_TRIKEYS needs z_head == z_tail (one drug set against itself)");
      MDG_CHECK_ARG((a.ldo & 3) == 0 && a.ldo >= ((a.n_tail + 3) & ~static_cast<int64_t>(3)) && a.n_tail * a.ldo * 4 < (int64_t(1) << 32),
                    "mdg_bilinear_allpairs: MDG_EPI_TRIKEYS needs a row pitch that is a multiple of 4 >= n_tail and n_tail * pitch < 2^30 (got %lld, %lld)",
                    (long long)a.n_tail, (long long)a.ldo);
      const int nb = static_cast<int>(mdg_cdiv(a.n_tail, 256));
      const dim3 gsym(static_cast<unsigned>((nb + 1) / 2), static_cast<unsigned>(a.n_labels));
      const bool sc1 = MODE == MDG_PREC_BF16X3;
      if (sc1) hipLaunchKernelGGL((bilinear_allpairs_sym_kernel<MODE, MDG_EPI_TRIKEYS, 8, 1>), gsym, block, lds2, st, a);
      else hipLaunchKernelGGL((bilinear_allpairs_sym_kernel<MODE, MDG_EPI_TRIKEYS, 8, 0>), gsym, block, lds2, st, a);
      MDG_CHECK_LAUNCH("mdg_bilinear_allpairs(lower-triangle keys)");
      return MDG_OK;
    }
    if (want && a.symmetric && a.pipeline == 0 && (epilogue == MDG_EPI_STORE || epilogue == MDG_EPI_STORE_SIGMOID) &&
        a.n_tail * a.ldo * 4 < (int64_t(1) << 32)) {
      const int nb = static_cast<int>(mdg_cdiv(a.n_tail, 256));
      const dim3 gsym(static_cast<unsigned>((nb + 1) / 2), static_cast<unsigned>(a.n_labels));
      // write-through score stores keep z_tail L2-resident (FETCH 11.3 -> 0.8 GB per launch at 4096^2 x 896); measured faster for
      // the three-product mode (11.78 -> 11.53 ms), slower for the single-product 16-bit modes (10.70 -> 11.11 ms)
      const bool sc1 = MODE == MDG_PREC_BF16X3;
      if (epilogue == MDG_EPI_STORE) {
        if (sc1) hipLaunchKernelGGL((bilinear_allpairs_sym_kernel<MODE, MDG_EPI_STORE, 8, 1>), gsym, block, lds2, st, a);
        else hipLaunchKernelGGL((bilinear_allpairs_sym_kernel<MODE, MDG_EPI_STORE, 8, 0>), gsym, block, lds2, st, a);
      } else {
        hipLaunchKernelGGL((bilinear_allpairs_sym_kernel<MODE, MDG_EPI_STORE_SIGMOID, 8>), gsym, block, lds2, st, a);
      }
      MDG_CHECK_LAUNCH("mdg_bilinear_allpairs(symmetric)");
      if ((a.n_tail & 3) && a.ldo < ((a.n_tail + 3) & ~static_cast<int64_t>(3))) {      // contiguous rows: the last N % 4 columns of every row
        const dim3 gs(static_cast<unsigned>(mdg_cdiv(a.n_tail, 256)), static_cast<unsigned>(a.n_labels));
        if (epilogue == MDG_EPI_STORE) hipLaunchKernelGGL((bilinear_strip_kernel<MODE, MDG_EPI_STORE>), gs, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((bilinear_strip_kernel<MODE, MDG_EPI_STORE_SIGMOID>), gs, dim3(256), 0, st, a);
        MDG_CHECK_LAUNCH("mdg_bilinear_allpairs(strip)");
      }
      return MDG_OK;
    }
  }
  switch (epilogue) {
    case MDG_EPI_STORE:
      hipLaunchKernelGGL((bilinear_allpairs_kernel<MODE, MDG_EPI_STORE, NW>), grid, block, lds, st, a);
      break;
    case MDG_EPI_STORE_SIGMOID:
      hipLaunchKernelGGL((bilinear_allpairs_kernel<MODE, MDG_EPI_STORE_SIGMOID, NW>), grid, block, lds, st, a);
      break;
    case MDG_EPI_ROWSTATS: {    // three-buffer ring (prefetch distance two)
      if constexpr (kSingle16<MODE> && NW == 8) {              // 64 rows per wave (halves the LDS operand reads per MFMA) on v_mfma_f32_16x16x32
        const dim3 grid2(static_cast<unsigned>(mdg_cdiv(a.n_head, 32 * NW * 2)), static_cast<unsigned>(a.n_labels));
        hipLaunchKernelGGL((bilinear_rowstats16_kernel<MODE>), grid2, block, 3 * STAGE_BYTES, st, a);
      } else {
        hipLaunchKernelGGL((bilinear_allpairs_kernel<MODE, MDG_EPI_ROWSTATS, NW>), grid, block, 3 * STAGE_BYTES, st, a);
      }
      break;
    }
    default:
      mdg_set_error("mdg_bilinear_allpairs: unknown epilogue %d", epilogue);
      return MDG_EINVAL;
  }
  MDG_CHECK_LAUNCH("mdg_bilinear_allpairs");
  return MDG_OK;
}

// Workgroup shape: 8 waves x 32 rows (4 waves x 128 rows, two workgroups per CU, was the alternative of rounds 1-2)
template <int MODE>
int launch_allpairs(const BilinearArgs& a, int epilogue, hipStream_t st) {
  return launch_allpairs_nw<MODE, 8>(a, epilogue, st);
}

}  // namespace

// Diagnostics (scripts/head_variants.py): a device buffer of `entries` uint64 that receives {shader cycles, 100 MHz ticks} per
// workgroup of every following mdg_bilinear_allpairs launch with at most entries / 2 workgroups; NULL / 0 switches it off.
static unsigned long long* g_stamps = nullptr;
static int64_t g_stamp_entries = 0;
extern "C" void mdg_debug_bilinear_stamps(void* buffer, int64_t entries) {
  g_stamps = static_cast<unsigned long long*>(buffer);
  g_stamp_entries = buffer ? entries : 0;
}

extern "C" int mdg_symmetrize(const float* w_original, float* w_sym, int64_t L, int64_t D_, void* stream) {
  MDG_CHECK_ARG(w_original && w_sym, "mdg_symmetrize: null pointer");
  MDG_CHECK_ARG(L >= 0 && D_ > 0 && D_ <= 4096, "mdg_symmetrize: bad shape L=%lld D=%lld", (long long)L, (long long)D_);
  const int64_t n = L * D_ * D_;
  if (n == 0) return MDG_OK;
  hipLaunchKernelGGL(symmetrize_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n, 256))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), w_original, w_sym, L, static_cast<int>(D_));
  MDG_CHECK_LAUNCH("mdg_symmetrize");
  return MDG_OK;
}

extern "C" size_t mdg_bilinear_allpairs_workspace_bytes(int64_t n_head, int64_t n_tail, int64_t n_labels, int64_t D_,
                                                        int precision) {
  (void)n_head;
  if (precision == MDG_PREC_F32 || n_tail <= 0 || n_labels <= 0) return 0;
  const size_t z = align256(static_cast<size_t>(n_tail) * D_ * 2), w = align256(static_cast<size_t>(n_labels) * D_ * D_ * 2);
  return precision == MDG_PREC_BF16X3 ? 2 * (z + w) : (z + w);
}

extern "C" int mdg_bilinear_allpairs(const float* z_head, const float* z_tail, const float* w_sym, float* out,
                                     int64_t n_head, int64_t n_tail, int64_t n_labels, int64_t D_, int precision,
                                     int epilogue, void* workspace, size_t workspace_bytes, void* stream) {
  return mdg_bilinear_allpairs_ld(z_head, z_tail, w_sym, out, n_tail, n_head, n_tail, n_labels, D_, precision, epilogue, workspace,
                                  workspace_bytes, stream);
}

extern "C" int mdg_bilinear_allpairs_ld(const float* z_head, const float* z_tail, const float* w_sym, float* out, int64_t ldo,
                                        int64_t n_head, int64_t n_tail, int64_t n_labels, int64_t D_, int precision,
                                        int epilogue, void* workspace, size_t workspace_bytes, void* stream) {
  MDG_CHECK_ARG(D_ == D, "mdg_bilinear_allpairs: D must be %d (got %lld)", D, (long long)D_);
  MDG_CHECK_ARG(epilogue == MDG_EPI_ROWSTATS || ldo >= n_tail, "mdg_bilinear_allpairs: row pitch %lld < n_tail %lld", (long long)ldo, (long long)n_tail);
  MDG_CHECK_ARG(n_head >= 0 && n_tail >= 0 && n_labels >= 0, "mdg_bilinear_allpairs: negative size");
  MDG_CHECK_ARG(n_labels <= 65535, "mdg_bilinear_allpairs: n_labels %lld > 65535 per call", (long long)n_labels);
  MDG_CHECK_ARG((ldo > n_tail ? ldo : n_tail) * 256 * 4 < (int64_t(1) << 31), "mdg_bilinear_allpairs: n_tail / row pitch %lld too large", (long long)ldo);
  if (n_head == 0 || n_tail == 0 || n_labels == 0) return MDG_OK;
  MDG_CHECK_ARG(z_head && z_tail && w_sym && out, "mdg_bilinear_allpairs: null pointer");
  MDG_CHECK_ARG(mdg_aligned16(z_head) && mdg_aligned16(z_tail) && mdg_aligned16(w_sym),
                "mdg_bilinear_allpairs: z_head, z_tail and w_sym must be 16-byte aligned");
  MDG_CHECK_ARG(precision == MDG_PREC_F32 || precision == MDG_PREC_BF16X3 || precision == MDG_PREC_BF16 || precision == MDG_PREC_F16,
                "mdg_bilinear_allpairs: unknown precision %d", precision);
  hipStream_t st = static_cast<hipStream_t>(stream);
  BilinearArgs a{};
  a.z_head = z_head;
  a.out = out;
  a.n_head = n_head; a.n_tail = n_tail; a.n_labels = n_labels;
  a.ldo = epilogue == MDG_EPI_ROWSTATS ? n_tail : ldo;
  a.zt_raw = z_tail;
  a.w_raw = w_sym;
  a.zt.nrows = n_tail;
  a.symmetric = (z_head == z_tail && n_head == n_tail) ? 1 : 0;
  a.stagger = 1;
  a.stagger_waves = 1;
  a.loaders = 4;
  a.pipeline = 0;
  // diagnostics: clock stamps per workgroup, only into a buffer handed over through mdg_debug_bilinear_stamps and only when it is large enough
  {
    const int64_t wgs = mdg_cdiv(n_head, 128) * n_labels;          // the largest grid any variant launches
    a.stamps = (g_stamps && g_stamp_entries >= 2 * wgs) ? g_stamps : nullptr;
  }
  a.w.nrows = D;
  if (precision == MDG_PREC_F32) {
    a.zt.f32 = z_tail;
    a.w.f32 = w_sym;
    return launch_allpairs<MDG_PREC_F32>(a, epilogue, st);
  }
  const size_t need = mdg_bilinear_allpairs_workspace_bytes(n_head, n_tail, n_labels, D_, precision);
  if (!workspace || workspace_bytes < need || !mdg_aligned16(workspace)) {
    mdg_set_error("mdg_bilinear_allpairs: workspace of %zu bytes (16-byte aligned) required, got %zu", need, workspace_bytes);
    return MDG_EWORKSPACE;
  }
  const bool x3 = precision == MDG_PREC_BF16X3;
  char* ws = static_cast<char*>(workspace);
  const size_t zb = align256(static_cast<size_t>(n_tail) * D * 2), wb = align256(static_cast<size_t>(n_labels) * D * D * 2);
  __bf16* zhi = reinterpret_cast<__bf16*>(ws);
  __bf16* whi = reinterpret_cast<__bf16*>(ws + zb);
  __bf16* zlo = x3 ? reinterpret_cast<__bf16*>(ws + zb + wb) : nullptr;
  __bf16* wlo = x3 ? reinterpret_cast<__bf16*>(ws + 2 * zb + wb) : nullptr;
  const int64_t z4 = n_tail * D / 4, w4 = n_labels * D * D / 4;
  if (precision == MDG_PREC_F16) {           // operands rounded to IEEE half once (BASELINE configs[4]: "fp16 bilinear head")
    hipLaunchKernelGGL(round_f16_kernel, dim3(static_cast<unsigned>(mdg_cdiv(z4, 256))), dim3(256), 0, st, z_tail, reinterpret_cast<_Float16*>(zhi), z4);
    hipLaunchKernelGGL(round_f16_kernel, dim3(static_cast<unsigned>(mdg_cdiv(w4, 256))), dim3(256), 0, st, w_sym, reinterpret_cast<_Float16*>(whi), w4);
    MDG_CHECK_LAUNCH("mdg_bilinear_allpairs(round)");
    a.zt.hi = zhi; a.zt.lo = nullptr;
    a.w.hi = whi; a.w.lo = nullptr;
    return launch_allpairs<MDG_PREC_F16>(a, epilogue, st);
  }
  hipLaunchKernelGGL(split_bf16_kernel, dim3(static_cast<unsigned>(mdg_cdiv(z4, 256))), dim3(256), 0, st, z_tail, zhi, zlo, z4);
  hipLaunchKernelGGL(split_bf16_kernel, dim3(static_cast<unsigned>(mdg_cdiv(w4, 256))), dim3(256), 0, st, w_sym, whi, wlo, w4);
  MDG_CHECK_LAUNCH("mdg_bilinear_allpairs(split)");
  a.zt.hi = zhi; a.zt.lo = zlo;
  a.w.hi = whi; a.w.lo = wlo;
  return x3 ? launch_allpairs<MDG_PREC_BF16X3>(a, epilogue, st) : launch_allpairs<MDG_PREC_BF16>(a, epilogue, st);
}
