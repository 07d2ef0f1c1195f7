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
// (blockIdx.y = outcome).  Tiles of 8192 keys per 512-thread workgroup.  Per pass: per-tile digit histogram
// (the first one inside the key-extraction pass) -> exclusive scan in (digit, tile) order -> scatter.
// The scatter SORTS ITS TILE IN LDS first (stable): every wave owns 1024 consecutive keys and ranks them
// 64 at a time against wave-private digit counters -- lanes holding the same digit find each other with 8
// ballots, no barrier inside the loop -- then the tile leaves as runs of equal digits, consecutive lanes
// writing consecutive addresses (32 keys = 128 B per run on average) instead of one 4-byte store per lane
// and cache line.  The last pass does not write the sorted pairs: position q of payload p IS rank q+1, which is
// written to out[i,j] and out[j,i] directly.
// Ties: stable in p (= flat row-major index); numpy's default argsort in the reference is unstable, so
// tie order there is implementation-defined ("Ties in rank normalisation").
// HBM-bound integer work, per key: extract 4 + 4, passes 4 + (4|8) + 8 each, last pass 8 of scattered
// rank stores = 80 B (algorithmic bytes are the M key reads and the N^2 rank stores).
#include "mdg_common.h"

namespace {

// Tile shape of the sort: the scatter leaves a tile as runs of equal digits, TILE / 256 keys long on average -- the larger the
// tile, the longer the contiguous stores (measured per 4096^2 outcome: 4096-key tiles 0.42 ms, 8192 0.31 ms, 16384 0.28 ms).
// Big (passes 0-2): one 1024-thread workgroup per CU with 146 KB of LDS.  The last pass -- no digit sort, LDS holds the pair
// exchange and the block tables instead -- always runs on 8192-key tiles (two workgroups per CU).
template <int TPB_, int ITEMS_>
struct RankCfg {
  static constexpr int TPB = TPB_, ITEMS = ITEMS_, WAVES = TPB_ / 64, TILE = TPB_ * ITEMS_, WSPAN = TILE / WAVES;
};
using CfgBig = RankCfg<1024, 16>;
using CfgStd = RankCfg<512, 16>;
#define MDG_RANK_USING(C) constexpr int TPB = C::TPB, ITEMS = C::ITEMS, WAVES = C::WAVES, TILE = C::TILE, WSPAN = C::WSPAN; (void)TPB; (void)ITEMS; (void)WAVES; (void)TILE; (void)WSPAN
constexpr uint32_t NO_PAY = 0xFFFFFFFFu;   // payload of the padding behind the last key of the last tile

// p -> (i, j) with i > j, p = i(i-1)/2 + j
__device__ __forceinline__ void tri_decode(int64_t p, int& i, int& j) {
  int64_t r = static_cast<int64_t>((1.0 + sqrt(1.0 + 8.0 * static_cast<double>(p))) * 0.5);
  while (r * (r - 1) / 2 > p) --r;
  while ((r + 1) * r / 2 <= p) ++r;
  i = static_cast<int>(r);
  j = static_cast<int>(p - r * (r - 1) / 2);
}

struct TriWalk {                          // wave-uniform walk along the rows of the strict lower triangle, 64 positions per step
  int wi, wj;
  __device__ __forceinline__ void start(int64_t p, int64_t M) {
    int i, j;
    tri_decode(p < M ? p : M - 1, i, j);
    wi = __builtin_amdgcn_readfirstlane(i);
    wj = __builtin_amdgcn_readfirstlane(j);
  }
  __device__ __forceinline__ void lane_pos(int lane, int& i, int& j) const {
    i = wi;
    j = wj + lane;
    while (j >= i) { j -= i; ++i; }
  }
  __device__ __forceinline__ void step() {
    wj += 64;
    while (wj >= wi) { wj -= wi; ++wi; }
  }
};

// lanes of the wave that hold the same 8-bit digit as this lane (all 64 lanes take part)
__device__ __forceinline__ uint64_t match_digit(uint32_t dg) {
  uint64_t peers = ~0ull;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const bool bit = (dg >> b) & 1u;
    const uint64_t bal = __ballot(bit);
    peers &= bit ? bal : ~bal;
  }
  return peers;
}

// Digit counts of the wave's 1024 keys into its private counters cnt[256]; rank_out[k] = number of EARLIER keys of the wave
// (in p order) with the same digit.  Wave-private LDS, in-order LDS queue: no barrier.
template <bool WANT_RANK, int ITEMS>
__device__ __forceinline__ void wave_digit_ranks(const uint32_t (&key)[ITEMS], int shift, uint32_t* cnt, int lane, uint32_t (&rank_out)[ITEMS]) {
#ifdef MDG_RANK_ATOMIC_ORDER
  // EXPERIMENT: one returning LDS atomic per key.  Stable only if the LDS serves the lanes of one instruction that hit the same
  // counter in ascending lane order (not an architectural promise) -- see DESIGN.md 4c for what the tie tests said.
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const uint32_t old = __hip_atomic_fetch_add(&cnt[(key[k] >> shift) & 255u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (WANT_RANK) rank_out[k] = old;
    __builtin_amdgcn_wave_barrier();
  }
  (void)lane;
#else
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const uint32_t dg = (key[k] >> shift) & 255u;
    const uint64_t peers = match_digit(dg);
    const uint32_t before = __popcll(peers & lt);
    const uint32_t old = cnt[dg];
    if (before == 0) cnt[dg] = old + __popcll(peers);      // one leader per digit
    if (WANT_RANK) rank_out[k] = old + before;
    __builtin_amdgcn_wave_barrier();                       // keep the rounds' counter updates in program order
  }
#endif
}

// Digit counts only (no ranks): no-return LDS atomics on the wave's private counters.  Lanes sharing a digit serialise inside one
// instruction (n lanes on a counter = n LDS cycles), which even for a wave of equal digits costs less than the 8-ballot match.
template <int ITEMS>
__device__ __forceinline__ void wave_digit_counts(const uint32_t (&key)[ITEMS], int shift, uint32_t* cnt) {
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) __hip_atomic_fetch_add(&cnt[(key[k] >> shift) & 255u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// hist[(seg * 256 + digit) * nblk + blk] from the per-wave counters (valid keys only: the padding of the last tile was counted
// as digit 255 and is taken out again)
template <class C>
__device__ __forceinline__ void store_tile_histogram(const uint32_t (*cnt)[256], uint32_t* __restrict__ hist, int64_t seg, int nblk, int64_t base, int64_t M) {
  MDG_RANK_USING(C);
  const int d = threadIdx.x;
  if (d < 256) {
    uint32_t c = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) c += cnt[w][d];
    if (d == 255 && base + TILE > M) c -= static_cast<uint32_t>(base + TILE - M);
    hist[(seg * 256 + d) * nblk + blockIdx.x] = c;
  }
}

// The LSD kernels sort either every outcome of the call (only == nullptr: outcome = the grid's y / x index) or, behind the MSD path, the
// outcomes it flagged: only[0] = how many, only[1..] = which (msd_flag_list_kernel); the grid then has one or two rows that walk the list,
// so that a call in which nothing was flagged launches a few hundred workgroups that leave at once instead of one per tile and outcome
// (LIST is a template parameter: without a list the body is straight-line code -- as a loop it cost the sort 17 %: registers, occupancy)
#define MDG_SEG_BEGIN(LIST, only, IDX, STRIDE)                                                                                                         \
  for (int64_t seg_it_ = (IDX); LIST ? seg_it_ < static_cast<int64_t>((only)[0]) : seg_it_ == static_cast<int64_t>(IDX); seg_it_ += (STRIDE)) {       \
    const int64_t seg = LIST ? static_cast<int64_t>((only)[1 + seg_it_]) : seg_it_;
#define MDG_SEG_END(LIST) \
    if (LIST) __syncthreads(); \
    else break;            \
  }

// keys of the strict lower triangle in p order + the tile histogram of the lowest digit
template <bool LIST, class C>
__global__ __launch_bounds__(C::TPB) void extract_keys_kernel(const float* __restrict__ scores, int64_t lds, uint32_t* __restrict__ keys,
                                                           uint32_t* __restrict__ hist, int N, int64_t M, int nblk, int src_is_keys,
                                                           const uint32_t* __restrict__ only) {
  MDG_RANK_USING(C);
  __shared__ uint32_t cnt[WAVES][256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  MDG_SEG_BEGIN(LIST, only, blockIdx.y, gridDim.y)
  for (int i = tid; i < WAVES * 256; i += TPB) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * TILE;
  const float* sc = scores + seg * static_cast<int64_t>(N) * lds;
  uint32_t key[ITEMS];
  // (i, j) of the wave's first position by the closed form, once; every later position by stepping along the rows
  int wi, wj;
  {
    const int64_t pw = base + wave * WSPAN < M ? base + wave * WSPAN : M - 1;
    tri_decode(pw, wi, wj);
    wi = __builtin_amdgcn_readfirstlane(wi);
    wj = __builtin_amdgcn_readfirstlane(wj);
  }
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int64_t p = base + wave * WSPAN + k * 64 + lane;
    key[k] = 0xFFFFFFFFu;
    int i = wi, j = wj + lane;                             // row i holds i entries
    while (j >= i) { j -= i; ++i; }
    if (p < M) {
      const float v = sc[static_cast<int64_t>(i) * lds + j];
      key[k] = src_is_keys ? __builtin_bit_cast(uint32_t, v) : mdg_order_key(v);      // MDG_EPI_TRIKEYS wrote the keys themselves
      keys[seg * M + p] = key[k];
    }
    wj += 64;                                              // the wave's next 64 positions (wave-uniform walk)
    while (wj >= wi) { wj -= wi; ++wi; }
  }
  wave_digit_counts(key, 0, cnt[wave]);
  __syncthreads();
  store_tile_histogram<C>(cnt, hist, seg, nblk, base, M);
  MDG_SEG_END(LIST)
}

// KT: what the previous pass left of the key -- the bits below the current digit are sorted already and are not carried along:
// pass 1 writes the upper 16 bits (uint16_t), pass 2 the upper 8 (uint8_t); `shift` = position of the current digit inside a KT
template <bool LIST, class C, class KT>
__global__ __launch_bounds__(C::TPB) void histogram_kernel(const KT* __restrict__ keys, uint32_t* __restrict__ hist, int64_t M,
                                                           int nblk, int shift, const uint32_t* __restrict__ only) {
  MDG_RANK_USING(C);
  __shared__ uint32_t cnt[WAVES][256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  MDG_SEG_BEGIN(LIST, only, blockIdx.y, gridDim.y)
  for (int i = tid; i < WAVES * 256; i += TPB) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * TILE;
  uint32_t key[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int64_t p = base + wave * WSPAN + k * 64 + lane;
    key[k] = p < M ? static_cast<uint32_t>(keys[seg * M + p]) : 0xFFFFFFFFu;
  }
  wave_digit_counts(key, shift, cnt[wave]);
  __syncthreads();
  store_tile_histogram<C>(cnt, hist, seg, nblk, base, M);
  MDG_SEG_END(LIST)
}

// exclusive scan over the 256 digits held one per thread by threads 0..255 (4 waves): step 1 inside a wave ...
__device__ __forceinline__ uint32_t wave_inclusive(uint32_t v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = __shfl_up(v, o, 64);
    if (lane >= o) v += u;
  }
  return v;
}

// wave64 inclusive scan / reduction in registers (DPP row shifts and row broadcasts: no trip through the LDS crossbar, which a
// __shfl step is -- six dependent ones cost more than the phase they sat in).  `ident` is op's identity.
#define MDG_DPP(old, v, ctrl, rows) static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(old), static_cast<int>(v), ctrl, rows, 0xf, false))
template <class Op>
__device__ __forceinline__ uint32_t wave_scan_dpp(uint32_t v, uint32_t ident, Op op) {
  v = op(v, MDG_DPP(ident, v, 0x111, 0xf));                // row_shr:1, :2, :4, :8 -- inclusive within each row of 16 lanes
  v = op(v, MDG_DPP(ident, v, 0x112, 0xf));
  v = op(v, MDG_DPP(ident, v, 0x114, 0xf));
  v = op(v, MDG_DPP(ident, v, 0x118, 0xf));
  v = op(v, MDG_DPP(ident, v, 0x142, 0xa));                // row_bcast:15 into rows 1 and 3
  v = op(v, MDG_DPP(ident, v, 0x143, 0xc));                // row_bcast:31 into rows 2 and 3
  return v;
}
template <class Op>
__device__ __forceinline__ uint32_t wave_reduce_dpp(uint32_t v, uint32_t ident, Op op) {
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(wave_scan_dpp(v, ident, op)), 63));
}


// exclusive scan of the 256*nblk counters of one outcome, in place; one workgroup per outcome, coalesced: every wave owns a
// contiguous range and walks it 256 counters (16 bytes per lane) at a time, the next group's load issued before the scan of this one
template <bool LIST>
__global__ __launch_bounds__(1024) void scan_kernel(uint32_t* __restrict__ hist, int nblk, const uint32_t* __restrict__ only) {
  __shared__ uint32_t part[16];
  MDG_SEG_BEGIN(LIST, only, blockIdx.x, gridDim.x)
  u32x4* h = reinterpret_cast<u32x4*>(hist + seg * 256 * nblk);
  const int total = 64 * nblk;                                  // groups of 4 counters
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int per = ((total + 15) / 16 + 63) / 64 * 64;           // groups per wave, a multiple of 64
  const int a = wave * per, b = (a + per < total) ? a + per : total;
  uint32_t s = 0;
  for (int t = a + lane; t < b; t += 64) {
    const u32x4 v = h[t];
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) part[wave] = s;
  __syncthreads();
  uint32_t run = 0;
  for (int w = 0; w < wave; ++w) run += part[w];
  u32x4 nxt = (a + lane < b) ? h[a + lane] : u32x4{0u, 0u, 0u, 0u};
  for (int t0 = a; t0 < b; t0 += 64) {
    const int t = t0 + lane;
    const u32x4 v = nxt;
    if (t0 + 64 < b) nxt = (t + 64 < b) ? h[t + 64] : u32x4{0u, 0u, 0u, 0u};
    const uint32_t own = (v[0] + v[1]) + (v[2] + v[3]);
    uint32_t inc = own;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    const uint32_t e0 = run + inc - own;
    if (t < b) h[t] = u32x4{e0, e0 + v[0], e0 + v[0] + v[1], e0 + v[0] + v[1] + v[2]};
    run += __shfl(inc, 63, 64);
  }
  MDG_SEG_END(LIST)
}

// KIN / KOUT: key representation read / written (see histogram_kernel): the output drops the digit this pass sorts by when
// KOUT is narrower than KIN (`shift` must then be 0 within KIN ... the current digit is its low byte, or bits 8..15 for pass 1)
template <bool LIST, class C, bool FIRST, bool LAST, class KIN = uint32_t, class KOUT = uint32_t>
__global__ __launch_bounds__(C::TPB) void scatter_kernel(const KIN* __restrict__ keys_in, const uint32_t* __restrict__ pay_in,
                                                      KOUT* __restrict__ keys_out, uint32_t* __restrict__ pay_out,
                                                      const uint32_t* __restrict__ offsets, float* __restrict__ out, int64_t ldo, int N,
                                                      int64_t M, int nblk, int shift, double denom, const uint32_t* __restrict__ only) {
  MDG_RANK_USING(C);
  __shared__ uint32_t cnt[WAVES][256];     // per-wave digit counts, then their exclusive prefix over the waves
  __shared__ uint32_t dstart[256];         // first slot of digit d in the sorted tile
  __shared__ uint32_t gofs[256];           // global position of slot 0 of digit d's run, minus dstart[d]
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t skey[TILE];
  __shared__ uint32_t spay[TILE];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  MDG_SEG_BEGIN(LIST, only, blockIdx.y, gridDim.y)
  for (int i = tid; i < WAVES * 256; i += TPB) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * TILE;
  uint32_t key[ITEMS], pay[ITEMS], rk[ITEMS];
  // payload = (i << 16) | j of the entry (round 4; rounds 1-3 carried the triangle index p and took a double-precision square root per
  // key to get (i, j) back in the last pass): the first pass walks the triangle's rows once per wave, as the key extraction does
  TriWalk tw;
  if (FIRST) tw.start(base + wave * WSPAN, M);
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int64_t p = base + wave * WSPAN + k * 64 + lane;
    const bool valid = p < M;
    key[k] = valid ? static_cast<uint32_t>(keys_in[seg * M + p]) : 0xFFFFFFFFu;   // padding: digit 255 in every pass, behind every real key of the tile
    if (FIRST) {
      int i, j;
      tw.lane_pos(lane, i, j);
      pay[k] = valid ? ((static_cast<uint32_t>(i) << 16) | static_cast<uint32_t>(j)) : NO_PAY;
      tw.step();
    } else {
      pay[k] = valid ? pay_in[seg * M + p] : NO_PAY;
    }
  }
  wave_digit_ranks<true, ITEMS>(key, shift, cnt[wave], lane, rk);
  __syncthreads();
  uint32_t run = 0;                        // thread d < 256: the tile's count of digit d
  if (tid < 256) {                         // thread d: prefix over the waves, then over the digits
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      const uint32_t c = cnt[w][tid];
      cnt[w][tid] = run;
      run += c;
    }
    const uint32_t inc = wave_inclusive(run, lane);
    if (lane == 63) wsum[wave] = inc;
    dstart[tid] = inc - run;               // exclusive within the wave's 64 digits
  }
  __syncthreads();
  if (tid < 256) {
    uint32_t add = 0;
    for (int w = 0; w < wave; ++w) add += wsum[w];
    const uint32_t ds = dstart[tid] + add;
    dstart[tid] = ds;
    gofs[tid] = offsets[(seg * 256 + tid) * nblk + blockIdx.x] - ds;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const uint32_t dg = (key[k] >> shift) & 255u;
    const uint32_t pos = dstart[dg] + cnt[wave][dg] + rk[k];
    skey[pos] = key[k];
    spay[pos] = pay[k];
  }
  __syncthreads();
  float* o = LAST ? out + seg * static_cast<int64_t>(N) * ldo : nullptr;
#pragma unroll 4
  for (int k = 0; k < ITEMS; ++k) {
    const int idx = k * TPB + tid;
    const uint32_t kk = skey[idx], pp = spay[idx];
    if (pp == NO_PAY) continue;
    const uint32_t g = gofs[(kk >> shift) & 255u] + static_cast<uint32_t>(idx);
    if constexpr (LAST) {
      const int i = static_cast<int>(pp >> 16), j = static_cast<int>(pp & 0xFFFFu);
      const float v = static_cast<float>(static_cast<double>(g + 1u) / denom);
      o[static_cast<int64_t>(i) * ldo + j] = v;
      o[static_cast<int64_t>(j) * ldo + i] = v;
    } else {
      keys_out[seg * M + g] = static_cast<KOUT>(kk >> (8 * (static_cast<int>(sizeof(KIN)) - static_cast<int>(sizeof(KOUT)))));
      pay_out[seg * M + g] = pp;
    }
  }
  MDG_SEG_END(LIST)
}

// ---- last pass, blocked: ranks delivered to 128 x 128 blocks of the lower triangle instead of one 4-byte store per entry ----
// The last digit's scatter needs no data movement: the global position g of an element IS its rank - 1.  Writing it straight to
// out[i,j] and out[j,i] costs two random 4-byte stores per pair (a 32-64 byte memory transaction each: 0.21 of the 0.49 ms per
// 4096^2 outcome).  Instead the tile's (g, position inside the block) pairs are sorted IN LDS by the 128 x 128 block (bi >= bj) of
// (i, j) they belong to -- no order needed inside a block, so LDS atomics hand out the slots -- and appended to that block's
// region of the pair buffer (room reserved with one global atomic per tile and non-empty block; every block's size is known in
// closed form, so the regions need no histogram).  A second kernel owns one block: it places the block's ranks in an LDS tile
// and writes the rows of out[i, j] and of the mirrored out[j, i] as whole 512-byte pieces (the diagonal of out included).
constexpr int BB = 128;                    // block edge
constexpr int FILL_STRIDE = 1;            // words between two blocks' fill counters (one 128-byte line each, 32, was tried: 64 lanes on 64 lines are slower than the contended lines)
constexpr int MAX_BLOCKS = 8192;           // LDS budget of the sorting kernel: N <= 16256; larger N take the direct scatter

// first pair slot of block (bi, bj), bj <= bi: all rows above block row bi, then the blocks left of it
__device__ __forceinline__ uint32_t block_base(int bi, int bj, int N) {
  const int64_t r0 = static_cast<int64_t>(bi) * BB;
  const int rcount = N - r0 < BB ? static_cast<int>(N - r0) : BB;
  return static_cast<uint32_t>(r0 * (r0 - 1) / 2 + static_cast<int64_t>(bj) * rcount * BB);
}

template <bool LIST, class C, class KIN>
__global__ __launch_bounds__(C::TPB) void rank_blocks_kernel(const KIN* __restrict__ keys_in, const uint32_t* __restrict__ pay_in,
                                                          const uint32_t* __restrict__ offsets, u32x2* __restrict__ pairs,
                                                          uint32_t* __restrict__ fill, int N, int64_t M, int nblk, int n_blocks,
                                                          const uint32_t* __restrict__ only) {
  MDG_RANK_USING(C);
  extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];            // [TILE] pairs (u32x2) | bcnt[n_blocks] | bdst[n_blocks]
  __shared__ uint32_t cnt[WAVES][256];
  __shared__ uint32_t gbase[256];
  __shared__ uint32_t wsum[WAVES];
  u32x2* spair = reinterpret_cast<u32x2*>(dyn);
  uint32_t* bcnt = dyn + 2 * TILE;
  uint32_t* bdst = bcnt + n_blocks;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  MDG_SEG_BEGIN(LIST, only, blockIdx.y, gridDim.y)
  for (int i = tid; i < WAVES * 256; i += TPB) (&cnt[0][0])[i] = 0;
  for (int i = tid; i < n_blocks; i += TPB) bcnt[i] = 0;
  __syncthreads();
  const int64_t base = static_cast<int64_t>(blockIdx.x) * TILE;
  uint32_t key[ITEMS], pay[ITEMS], rk[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int64_t p = base + wave * WSPAN + k * 64 + lane;
    const bool valid = p < M;
    key[k] = valid ? static_cast<uint32_t>(keys_in[seg * M + p]) : 0xFFFFFFFFu;
    pay[k] = valid ? pay_in[seg * M + p] : NO_PAY;
  }
  constexpr int SH = 8 * (static_cast<int>(sizeof(KIN)) - 1);          // the last digit = the top byte of what is left of the key
  wave_digit_ranks<true, ITEMS>(key, SH, cnt[wave], lane, rk);
  __syncthreads();
  if (tid < 256) {
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      const uint32_t c = cnt[w][tid];
      cnt[w][tid] = run;
      run += c;
    }
    gbase[tid] = offsets[(seg * 256 + tid) * nblk + blockIdx.x];
  }
  __syncthreads();
  uint32_t blk[ITEMS], slot[ITEMS];                       // key[] is reused for g, pay[] for the position inside the block
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    blk[k] = 0xFFFFFFFFu;
    if (pay[k] == NO_PAY) continue;
    const uint32_t dg = (key[k] >> SH) & 255u;
    key[k] = gbase[dg] + cnt[wave][dg] + rk[k];
    const int i = static_cast<int>(pay[k] >> 16), j = static_cast<int>(pay[k] & 0xFFFFu);
    const int bi = i >> 7, bj = j >> 7;
    blk[k] = static_cast<uint32_t>(bi * (bi + 1) / 2 + bj);
    pay[k] = static_cast<uint32_t>(((i & 127) << 7) | (j & 127));
    slot[k] = atomicAdd(&bcnt[blk[k]], 1u);
  }
  __syncthreads();
  // exclusive scan of bcnt in place (a contiguous run of blocks per thread) + room in every non-empty block's region
  {
    const int per = (n_blocks + TPB - 1) / TPB;
    const int a = tid * per, b = a + per < n_blocks ? a + per : n_blocks;
    uint32_t s = 0;
    for (int t = a; t < b; ++t) s += bcnt[t];
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t run = inc - s;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    for (int t = a; t < b; ++t) {
      const uint32_t c = bcnt[t];
      bcnt[t] = run;
      if (c) {
        int bi = static_cast<int>((sqrtf(8.0f * static_cast<float>(t) + 1.0f) - 1.0f) * 0.5f);
        while (bi * (bi + 1) / 2 > t) --bi;
        while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
        const int bj = t - bi * (bi + 1) / 2;
        bdst[t] = block_base(bi, bj, N) + atomicAdd(&fill[(seg * n_blocks + t) * FILL_STRIDE], c) - run;
      }
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ITEMS; ++k)
    if (blk[k] != 0xFFFFFFFFu) spair[bcnt[blk[k]] + slot[k]] = u32x2{key[k], pay[k] | (blk[k] << 14)};
  __syncthreads();
  const int n_valid = static_cast<int>(M - base < TILE ? M - base : TILE);
  u32x2* dst = pairs + seg * M;
#pragma unroll 4
  for (int k = 0; k < ITEMS; ++k) {
    const int idx = k * TPB + tid;
    if (idx >= n_valid) break;
    const u32x2 v = spair[idx];
    dst[bdst[v[1] >> 14] + static_cast<uint32_t>(idx)] = u32x2{v[0], v[1] & 16383u};
  }
  MDG_SEG_END(LIST)
}

// ================================================================================================================================
// DEFAULT path up to N = 5793 (round 5): ONE MSD partition with an EXACT layout + an in-LDS counting sort per bucket, instead of four
// global LSD passes.  Every position in memory comes from a count and a scan -- nothing has a capacity a score tensor could overflow
// (round 4's capacity-based layouts did, on row-structured scores: DESIGN.md 4e).  The layout is in units of the OUTPUT: a count /
// partition tile is one 128 x 128 block of the lower triangle, so a bucket's range is already in runs by output block:
//
//   msd_minmax / msd_hist1 / msd_level1 / msd_hist2 / msd_table   (all outcomes of the call, small launches)  a 1-in-S sample of
//                        64-key chunks, hashed over the whole triangle (every row and column band is seen: scores are
//                        row-structured), gives the outcome's key range and a two-level histogram: aligned slices of the key
//                        space, cut finer where the sample is dense.  bucket(key) = ((keys of the sample below the key's bin) +
//                        sub-range(key) * (keys per sub-range)) * buckets / samples: monotone in the key, ~8192 keys per bucket.
//                        The sample only steers the BALANCE; the layout below is exact for any table.
//   msd_count_kernel     per tile (= output block): how many of its keys fall in each bucket (u16 counters)
//   msd_scan_kernel      per bucket: exclusive prefix of the tile counts (= where each tile's run starts inside the bucket) + total
//   msd_base_kernel      per outcome: exclusive prefix of the bucket totals = the rank of each bucket's first key; the list of buckets
//                        beyond the bucket sort's LDS room; the zeroed work counter of the bucket sort; an outcome with a bucket of
//                        65 536 keys or more (a point mass of equal keys) is flagged for the LSD kernels
//   msd_partition_[pipe_]kernel   the tile is sorted by bucket in LDS (slots from LDS atomics, no digit matching) and leaves as one
//                        run per bucket, (key, position) pairs, at base[b] + offs[tile][b]
//   msd_bucket_kernel    persistent; big buckets first (msd_big_bucket, through global memory), then buckets drawn from a counter:
//                        counting sort on 16 384 fine bins of the bucket's own key range, keys that share a fine bin ordered by
//                        (key, position) explicitly -- this makes ties stable although neither the partition nor the counting
//                        preserves arrival order -- and ONE WORD per key written at the key's own place: rank in bucket | cell of the block
//   msd_block_gather_kernel   a block gathers its run of every bucket (start = base + offs[tile], length = counts[tile]), ranks into
//                        an LDS tile, whole rows of out[i, j] and of the mirrored block
//
// Bytes per key: 4 + 4 (count, partition reads) + 8 + 8 (pairs out / in) + 4 + 4 (words out / in) + 8 (ranks) = 40 + ~3 of tables
// (measured past the L2: 3.65 x the algorithmic 12), against ~80 for the LSD passes.  Payload q = (i << 16) | j: ordered like the
// triangle index, no square root to get (i, j) back.
// Handed to the LSD kernels (flag per outcome; same bits): a bucket of 65 536 keys or more, more than MSD_BIG_MAX buckets beyond
// MSD_CAP keys, a fine bin with more than MSD_TIE_LIMIT keys -- point masses, heavy ties.
constexpr int MSD_N1 = 512, MSD_NC = 4096;   // level-1 slices and level-2 bins of the bucket function
constexpr int MSD_HDR = 16;               // header words in front of an outcome's tables
constexpr int MSD_TABLE_WORDS = MSD_HDR + MSD_N1 + MSD_NC;
constexpr int MSD_NB_MAX = 2048;          // buckets per outcome: a 16 384-key tile then leaves as runs of >= 8 pairs = 64 B on average even at the largest N (measured,
                                          // scripts/micro/run_scatter_bw.hip: runs of 64 B and longer store at 4.7-6.4 TB/s, runs of 32 B at 1.6-2.3)
constexpr int MSD_QLG = 13;               // ~8192 keys per bucket
constexpr int MSD_NF = 16384;             // fine bins of the bucket sort
constexpr int MSD_CAP = 12288;            // keys a bucket may hold (LDS room of the bucket sort): 1.5 x the mean
constexpr int MSD_TILE = BB * BB;         // keys per count / partition tile: one 128 x 128 output block (1024 threads x 16)
constexpr int MSD_TIE_LIMIT = 1024;       // keys per fine bin ordered in place (c probes each: a bin of 170 keys -- seen in 1 to 2 % of the bench tensor's outcomes, in the
                                          // bucket around zero -- costs 30 000 LDS reads); more: LSD fallback
constexpr int MSD_SAMPLE_CHUNKS = 4096;   // 64-key chunks sampled per outcome (262 144 keys: 128 per bucket at N = 4096)
constexpr int MSD_SAMPLE_WGS = 32;        // 256-thread workgroups per outcome in the two sampling sweeps
constexpr int MSD_BIG_MAX = 63, MSD_BIG_WORDS = 72;   // buckets beyond MSD_CAP keys per outcome that msd_big_bucket takes (count + ids), then
constexpr int MSD_BIG_QUEUE = 64;                     // the counter the bucket sort's workgroups draw their buckets from
constexpr int MSD_BIG_NF = 32768, MSD_BIG_TIES = 4096;
constexpr uint32_t MSD_F_BUCKET = 1u, MSD_F_TOTAL = 2u, MSD_F_TIES = 4u;      // why an outcome was handed back
constexpr uint32_t MSD_SKIP = 0xFFFFFFFFu;

struct MsdMap { uint32_t lo, hi, s1, e0, mul, nbt; };

// per-outcome table: header (lo, hi, s1, multiplier, buckets, e0) | level 1 (MSD_N1 words) | level 2 (MSD_NC words)
__device__ __forceinline__ MsdMap msd_load_map(const uint32_t* __restrict__ hdr) { return MsdMap{hdr[0], hdr[1], hdr[2], hdr[5], hdr[3], hdr[4]}; }

// Level-1 bins: aligned 2^s1-key slices of the key space ((key >> s1) - (lo >> s1)), the smallest s1 that covers [lo, hi] with at
// most MSD_N1 of them.  Scores that span many binades get s1 = 23: the slices ARE the binades, so the factor-two steps of the key
// density at binade boundaries fall on bin boundaries; a narrow score range gets finer slices of its one or two binades.
__device__ __forceinline__ uint32_t msd_s1_of(uint32_t lo, uint32_t hi) {
  uint32_t s = 0;
  while (s < 23u && (hi >> s) - (lo >> s) >= static_cast<uint32_t>(MSD_N1)) ++s;
  return s;
}

// level 1: t1[e] = first level-2 bin of slice e | log2(level-2 bins of slice e) << 16 (a slice with many sample keys is cut finer)
// level 2: t2[c] = (4 x sample keys in the bins before c) << 11 | (4 x sample keys per sub-range of c) << 5 | log2(sub-ranges of c)
// bucket = (4 x sample keys below the key, taking every bin's keys as evenly spread over its sub-ranges) * buckets / (4 x samples)
__device__ __forceinline__ uint32_t msd_coarse_of(uint32_t key, const uint32_t* t1, const MsdMap& m, uint32_t& low2, uint32_t& se) {
  const uint32_t k = key < m.lo ? m.lo : (key > m.hi ? m.hi : key);
  const uint32_t e1 = t1[(k >> m.s1) - m.e0];
  se = m.s1 - (e1 >> 16);
  const uint32_t low1 = k & ((1u << m.s1) - 1u);
  low2 = low1 & ((1u << se) - 1u);
  return (e1 & 0xFFFFu) + (low1 >> se);
}

__device__ __forceinline__ uint32_t msd_bucket_of(uint32_t key, const uint32_t* t1, const uint32_t* t2, const MsdMap& m) {
  uint32_t low2, se;
  const uint32_t ent = t2[msd_coarse_of(key, t1, m, low2, se)];
  const uint32_t lg = ent & 31u, per = (ent >> 5) & 63u;
  const uint32_t sub = low2 >> (se - lg);                  // lg <= se; lg = 0: low2 >> se = 0
  const uint32_t b = __umulhi((ent >> 11) + sub * per, m.mul);
  return b < m.nbt ? b : m.nbt - 1u;
}

__device__ __forceinline__ int64_t msd_sample_chunk(int64_t g, int64_t S, int64_t n_chunks) {
  const uint32_t h = static_cast<uint32_t>(g) * 2654435761u;
  const int64_t c = g * S + static_cast<int64_t>((h >> 7) % static_cast<uint32_t>(S));
  return c < n_chunks ? c : n_chunks - 1;
}

// the sampled keys of one outcome, dealt to `n_waves` waves; eight chunk loads in flight per wave
template <class F>
__device__ __forceinline__ void msd_sample_sweep(const float* __restrict__ sc, int64_t lds, int64_t M, int src_is_keys, int wave_global, int n_waves,
                                                 int lane, F&& f) {
  const int64_t n_chunks = (M + 63) >> 6;
  const int64_t S = (n_chunks + MSD_SAMPLE_CHUNKS - 1) / MSD_SAMPLE_CHUNKS;
  const int64_t n_groups = (n_chunks + S - 1) / S;
  for (int64_t g0 = wave_global; g0 < n_groups; g0 += 8 * n_waves) {
    float v[8];
    bool ok[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int64_t g = g0 + static_cast<int64_t>(e) * n_waves;
      ok[e] = false;
      v[e] = 0.f;
      if (g < n_groups) {
        const int64_t c = msd_sample_chunk(g, S, n_chunks);
        TriWalk w;
        w.start(c * 64, M);
        int i, j;
        w.lane_pos(lane, i, j);
        ok[e] = c * 64 + lane < M;
        if (ok[e]) v[e] = sc[static_cast<int64_t>(i) * lds + j];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (ok[e]) f(src_is_keys ? __builtin_bit_cast(uint32_t, v[e]) : mdg_order_key(v[e]));
  }
}

// mm[2 * outcome] = ~0, mm[2 * outcome + 1] = 0
__global__ __launch_bounds__(256) void msd_init_minmax_kernel(uint32_t* __restrict__ mm, int L) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l < L) { mm[2 * l] = 0xFFFFFFFFu; mm[2 * l + 1] = 0u; }
}

// mm[2 * outcome] = smallest sampled key, mm[2 * outcome + 1] = largest
__global__ __launch_bounds__(256) void msd_minmax_kernel(const float* __restrict__ scores, int64_t lds, uint32_t* __restrict__ mm, int N, int64_t M,
                                                        int src_is_keys) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t seg = blockIdx.y;
  const float* sc = scores + seg * static_cast<int64_t>(N) * lds;
  uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
  msd_sample_sweep(sc, lds, M, src_is_keys, static_cast<int>(blockIdx.x) * 4 + wave, MSD_SAMPLE_WGS * 4, lane, [&](uint32_t key) {
    kmin = kmin < key ? kmin : key;
    kmax = kmax > key ? kmax : key;
  });
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t a = __shfl_xor(kmin, o, 64), c = __shfl_xor(kmax, o, 64);
    kmin = kmin < a ? kmin : a;
    kmax = kmax > c ? kmax : c;
  }
  // one pair of global atomics per workgroup, not per wave (4 096 waves of an outcome on one address serialise in the L2)
  __shared__ uint32_t wg_mm[2];
  if (tid == 0) { wg_mm[0] = 0xFFFFFFFFu; wg_mm[1] = 0u; }
  __syncthreads();
  if (lane == 0 && kmin <= kmax) {
    atomicMin(&wg_mm[0], kmin);
    atomicMax(&wg_mm[1], kmax);
  }
  __syncthreads();
  if (tid == 0 && wg_mm[0] <= wg_mm[1]) {
    atomicMin(&mm[2 * seg], wg_mm[0]);
    atomicMax(&mm[2 * seg + 1], wg_mm[1]);
  }
}

// hist1[outcome * MSD_N1 + level-1 slice of the sampled key]
__global__ __launch_bounds__(256) void msd_hist1_kernel(const float* __restrict__ scores, int64_t lds, const uint32_t* __restrict__ mm,
                                                       uint32_t* __restrict__ hist1, int N, int64_t M, int src_is_keys) {
  __shared__ uint32_t cnt[MSD_N1];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t seg = blockIdx.y;
  for (int c = tid; c < MSD_N1; c += 256) cnt[c] = 0;
  __syncthreads();
  const float* sc = scores + seg * static_cast<int64_t>(N) * lds;
  const uint32_t lo = mm[2 * seg], hi = mm[2 * seg + 1];
  const uint32_t s1 = msd_s1_of(lo, hi), e0 = lo >> s1;
  msd_sample_sweep(sc, lds, M, src_is_keys, static_cast<int>(blockIdx.x) * 4 + wave, MSD_SAMPLE_WGS * 4, lane, [&](uint32_t key) {
    __hip_atomic_fetch_add(&cnt[(key >> s1) - e0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  });
  __syncthreads();
  uint32_t* h = hist1 + seg * MSD_N1;
  for (int c = tid; c < MSD_N1; c += 256) {
    const uint32_t v = cnt[c];
    if (v) atomicAdd(&h[c], v);
  }
}

// header + level-1 table of one outcome: a slice with more than 256 sample keys is cut into 2^k level-2 bins of <= 256 sample keys
// each (at most MSD_N1 + 2 * 262144 / 256 = 2560 <= MSD_NC bins in all)
__global__ __launch_bounds__(MSD_N1) void msd_level1_kernel(const uint32_t* __restrict__ hist1, const uint32_t* __restrict__ mm, uint32_t* __restrict__ tables) {
  __shared__ uint32_t wsum[MSD_N1 / 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t seg = blockIdx.x;
  uint32_t lo = mm[2 * seg], hi = mm[2 * seg + 1];
  if (lo > hi) lo = hi = 0u;
  const uint32_t s1 = msd_s1_of(lo, hi);
  const uint32_t n = hist1[seg * MSD_N1 + tid];
  uint32_t lgb = 0;
  if (n > 256u) {
    lgb = 32u - static_cast<uint32_t>(__builtin_clz((n + 255u) / 256u - 1u));
    lgb = lgb > s1 ? s1 : lgb;
  }
  const uint32_t bins = 1u << lgb;
  const uint32_t inc = wave_inclusive(bins, lane);
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t run = inc - bins;
  for (int w = 0; w < wave; ++w) run += wsum[w];
  uint32_t* t = tables + seg * MSD_TABLE_WORDS;
  t[MSD_HDR + tid] = run | (lgb << 16);
  if (tid == 0) {
    t[0] = lo;
    t[1] = hi;
    t[2] = s1;
    t[5] = lo >> s1;
  }
}

// hist2[outcome * MSD_NC + level-2 bin of the sampled key]
__global__ __launch_bounds__(256) void msd_hist2_kernel(const float* __restrict__ scores, int64_t lds, const uint32_t* __restrict__ tables,
                                                       uint32_t* __restrict__ hist2, int N, int64_t M, int src_is_keys) {
  __shared__ uint32_t cnt[MSD_NC];
  __shared__ uint32_t t1[MSD_N1];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t seg = blockIdx.y;
  const uint32_t* t = tables + seg * MSD_TABLE_WORDS;
  for (int c = tid; c < MSD_NC; c += 256) cnt[c] = 0;
  for (int c = tid; c < MSD_N1; c += 256) t1[c] = t[MSD_HDR + c];
  const MsdMap m{t[0], t[1], t[2], t[5], 0u, 0u};
  __syncthreads();
  const float* sc = scores + seg * static_cast<int64_t>(N) * lds;
  msd_sample_sweep(sc, lds, M, src_is_keys, static_cast<int>(blockIdx.x) * 4 + wave, MSD_SAMPLE_WGS * 4, lane, [&](uint32_t key) {
    uint32_t low2, se;
    __hip_atomic_fetch_add(&cnt[msd_coarse_of(key, t1, m, low2, se)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  });
  __syncthreads();
  uint32_t* h = hist2 + seg * MSD_NC;
  for (int c = tid; c < MSD_NC; c += 256) {
    const uint32_t v = cnt[c];
    if (v) atomicAdd(&h[c], v);
  }
}

// level-2 table of one outcome from its sample histogram (+ the multiplier and the bucket count in the header)
__global__ __launch_bounds__(1024) void msd_table_kernel(const uint32_t* __restrict__ hist2, uint32_t* __restrict__ tables, int nbt) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t t1[MSD_N1];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t seg = blockIdx.x;
  uint32_t* t = tables + seg * MSD_TABLE_WORDS;
  if (tid < MSD_N1) t1[tid] = t[MSD_HDR + tid];
  const uint32_t s1 = t[2];
  const u32x4 n4 = reinterpret_cast<const u32x4*>(hist2 + seg * MSD_NC)[tid];
  const uint32_t n[4] = {n4[0], n4[1], n4[2], n4[3]};
  const uint32_t tot = (n[0] + n[1]) + (n[2] + n[3]);
  const uint32_t inc = wave_inclusive(tot, lane);
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t run = inc - tot, ns = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) run += wsum[w];
    ns += wsum[w];
  }
  uint32_t unit = ns / static_cast<uint32_t>(nbt) / 8u;      // sub-ranges of at most an eighth of a bucket, and at most 15 sample keys
  unit = unit < 1u ? 1u : (unit > 15u ? 15u : unit);
  u32x4 ent;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t c = 4u * static_cast<uint32_t>(tid) + static_cast<uint32_t>(e);
    int a = 0, b = MSD_N1 - 1;                             // the level-1 slice that owns bin c: the last one whose first bin is <= c
    while (a < b) {
      const int mid = (a + b + 1) >> 1;
      if ((t1[mid] & 0xFFFFu) <= c) a = mid; else b = mid - 1;
    }
    const uint32_t se = s1 - (t1[a] >> 16);                // key bits below bin c
    uint32_t lg = 0;
    if (n[e] > unit) {
      const uint32_t parts = (n[e] + unit - 1u) / unit;
      lg = 32u - static_cast<uint32_t>(__builtin_clz(parts - 1u));          // ceil(log2(parts)), parts >= 2
      lg = lg > se ? se : lg;
    }
    uint32_t per = (4u * n[e]) >> lg;
    per = per > 63u ? 63u : per;                           // (only when the bin ran out of key bits: heavy ties, handed back anyway)
    ent[e] = ((4u * run) << 11) | (per << 5) | lg;
    run += n[e];
  }
  reinterpret_cast<u32x4*>(t + MSD_HDR + MSD_N1)[tid] = ent;
  if (tid == 0) {
    t[3] = (nbt > 1 && ns > 0u) ? static_cast<uint32_t>((static_cast<uint64_t>(nbt) << 32) / (4ull * ns)) : 0u;      // bucket = umulhi(x, mul), x < 4 ns
    t[4] = static_cast<uint32_t>(nbt);
  }
}

// A count / partition TILE is one 128 x 128 block of the lower triangle -- the unit the output leaves in.  A bucket's keys therefore sit
// in the pair buffer in runs by output block (the partition writes tile t's keys of bucket b at base[b] + offs[t][b]), and the table
// of the exact layout -- counts / offs, one row per tile -- IS the directory the block gather reads its runs from: the bucket sort
// neither regroups its pairs by block nor writes a directory (round 5's first version did both, with a second returning LDS atomic
// per key, a second trip through LDS and a transposing kernel behind it).
struct MsdTileGeom {
  int r0, c0;
  __device__ __forceinline__ void of(int t) {
    int bi = static_cast<int>((sqrtf(8.0f * static_cast<float>(t) + 1.0f) - 1.0f) * 0.5f);
    while (bi * (bi + 1) / 2 > t) --bi;
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    r0 = bi * BB;
    c0 = (t - bi * (bi + 1) / 2) * BB;
  }
  // item k of thread tid (1024 threads x 16 items): row k * 8 + tid / 128, column tid % 128 -- a wave reads 256 contiguous bytes
  __device__ __forceinline__ bool cell(int k, int tid, int N, int& i, int& j) const {
    i = r0 + k * 8 + (tid >> 7);
    j = c0 + (tid & 127);
    return i < N && j < i;
  }
  __device__ __forceinline__ int keys(int N) const {      // cells of the tile in the strict lower triangle
    const int rc = N - r0 < BB ? N - r0 : BB;
    return r0 == c0 ? rc * (rc - 1) / 2 : rc * BB;
  }
};

// counts[(outcome * n_tiles + tile) * nbs + bucket] (u16): keys of the tile per bucket
__global__ __launch_bounds__(1024) void msd_count_kernel(const float* __restrict__ scores, int64_t lds, const uint32_t* __restrict__ tables,
                                                        uint16_t* __restrict__ counts, int N, int nbs, int src_is_keys) {
  constexpr int TPB = 1024, ITEMS = 16;
  __shared__ __attribute__((aligned(16))) uint32_t tab[MSD_N1 + MSD_NC];      // level 1 | level 2
  __shared__ uint32_t cnt[MSD_NB_MAX];
  const int tid = threadIdx.x;
  const int64_t seg = blockIdx.y;
  const float* sc = scores + seg * static_cast<int64_t>(N) * lds;
  const uint32_t* t = tables + seg * MSD_TABLE_WORDS;
  MsdTileGeom g;
  g.of(static_cast<int>(blockIdx.x));
  float raw[ITEMS];
  bool ok[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    int i, j;
    ok[k] = g.cell(k, tid, N, i, j);
    raw[k] = ok[k] ? sc[static_cast<int64_t>(i) * lds + j] : 0.f;
  }
  for (int c = tid; c < (MSD_N1 + MSD_NC) / 4; c += TPB) reinterpret_cast<u32x4*>(tab)[c] = reinterpret_cast<const u32x4*>(t + MSD_HDR)[c];
  const MsdMap m = msd_load_map(t);
  for (int b = tid; b < nbs; b += TPB) cnt[b] = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ITEMS; ++k)
    if (ok[k]) {
      const uint32_t key = src_is_keys ? __builtin_bit_cast(uint32_t, raw[k]) : mdg_order_key(raw[k]);
      __hip_atomic_fetch_add(&cnt[msd_bucket_of(key, tab, tab + MSD_N1, m)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  __syncthreads();
  uint32_t* dst = reinterpret_cast<uint32_t*>(counts + (seg * gridDim.x + blockIdx.x) * static_cast<int64_t>(nbs));
  for (int b2 = tid; b2 < nbs / 2; b2 += TPB) dst[b2] = cnt[2 * b2] | (cnt[2 * b2 + 1] << 16);
}

// offs[(outcome * n_tiles + tile) * nbs + bucket] = keys of the bucket in the tiles before; totals[outcome * nbs + bucket].
// 64 buckets per workgroup, the tiles in 16 chunks of one wave each: chunk sums, their prefix through LDS, then the chunk's own walk
// (one thread per bucket walking all ~528 tiles was 32 workgroups of dependent loads: 34 us per group of 8 outcomes)
__global__ __launch_bounds__(1024) void msd_scan_kernel(const uint16_t* __restrict__ counts, uint32_t* __restrict__ offs, uint32_t* __restrict__ totals,
                                                       int n_tiles, int nbs) {
  __shared__ uint32_t csum[16][64];
  const int lane = threadIdx.x & 63, ch = threadIdx.x >> 6;
  const int b = blockIdx.x * 64 + lane;
  const int64_t seg = blockIdx.y;
  const int per = (n_tiles + 15) / 16, t0 = ch * per, t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
  const uint16_t* c = counts + seg * n_tiles * static_cast<int64_t>(nbs) + b;
  uint32_t* o = offs + seg * n_tiles * static_cast<int64_t>(nbs) + b;
  uint32_t sum = 0;
  int t = t0;
  for (; t + 8 <= t1; t += 8) {
    uint32_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = c[static_cast<int64_t>(t + e) * nbs];
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += v[e];
  }
  for (; t < t1; ++t) sum += c[static_cast<int64_t>(t) * nbs];
  csum[ch][lane] = sum;
  __syncthreads();
  uint32_t run = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const uint32_t v = csum[w][lane];
    run += w < ch ? v : 0u;
    tot += v;
  }
  t = t0;
  for (; t + 8 <= t1; t += 8) {
    uint32_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = c[static_cast<int64_t>(t + e) * nbs];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      o[static_cast<int64_t>(t + e) * nbs] = run;
      run += v[e];
    }
  }
  for (; t < t1; ++t) {
    o[static_cast<int64_t>(t) * nbs] = run;
    run += c[static_cast<int64_t>(t) * nbs];
  }
  if (ch == 0) totals[seg * nbs + b] = tot;
}

// base[outcome * nbs + bucket] = keys in the buckets before.  A bucket beyond the bucket sort's LDS room goes on the outcome's list for
// msd_big_bucket (big[outcome * MSD_BIG_WORDS] = count, then the bucket ids); more than MSD_BIG_MAX of them, a bucket of 65 536
// keys or more (a point mass of equal keys) or a total that is not M raises the outcome's flag
__global__ __launch_bounds__(1024) void msd_base_kernel(const uint32_t* __restrict__ totals, uint32_t* __restrict__ base, uint32_t* __restrict__ flags,
                                                       uint32_t* __restrict__ big, int nbs, int64_t M) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t nbig;
  constexpr int BPT = MSD_NB_MAX / 1024;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t seg = blockIdx.x;
  if (tid == 0) nbig = 0;
  uint32_t c[BPT], tot = 0;
#pragma unroll
  for (int e = 0; e < BPT; ++e) {
    const int b = BPT * tid + e;
    c[e] = b < nbs ? totals[seg * nbs + b] : 0u;
    tot += c[e];
  }
  const uint32_t inc = wave_inclusive(tot, lane);
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t run = inc - tot;
  for (int w = 0; w < wave; ++w) run += wsum[w];
  bool over = false;
#pragma unroll
  for (int e = 0; e < BPT; ++e) {
    const int b = BPT * tid + e;
    if (b < nbs) base[seg * nbs + b] = run;
    run += c[e];
    if (c[e] > static_cast<uint32_t>(MSD_CAP)) {
      const uint32_t slot = c[e] < 65536u ? atomicAdd(&nbig, 1u) : static_cast<uint32_t>(MSD_BIG_MAX);
      if (slot < static_cast<uint32_t>(MSD_BIG_MAX)) big[seg * MSD_BIG_WORDS + 1 + slot] = static_cast<uint32_t>(b);
      else over = true;
    }
  }
  if (over) atomicOr(&flags[seg], MSD_F_BUCKET);
  if (tid == 1023 && static_cast<int64_t>(run) != M) atomicOr(&flags[seg], MSD_F_TOTAL);
  __syncthreads();
  if (tid == 0) {
    big[seg * MSD_BIG_WORDS] = nbig < static_cast<uint32_t>(MSD_BIG_MAX) ? nbig : static_cast<uint32_t>(MSD_BIG_MAX);
    big[seg * MSD_BIG_WORDS + MSD_BIG_QUEUE] = 0u;
  }
}

// -DMDG_RANK_STAMPS (MDG_EXTRA_HIPCC_FLAGS): shader-clock stamps at the phase boundaries of the two persistent kernels, summed per phase
// over a workgroup's tiles / buckets and left in the (otherwise idle) big-bucket scratch; scripts/rank_msd_debug.py reads them
#ifdef MDG_RANK_STAMPS
#define MDG_ST_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0
#define MDG_ST(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if ((i) > 0) st_acc[i] += t_ - st_prev; st_prev = t_; } while (0)
#define MDG_ST_OUT(ptr) do { if (threadIdx.x == 0) for (int e_ = 0; e_ < 8; ++e_) (ptr)[e_] = st_acc[e_]; } while (0)
#else
#define MDG_ST_DECL
#define MDG_ST(i)
#define MDG_ST_OUT(ptr)
#endif

// Persistent: workgroup x of outcome y takes tiles x, x + gridDim.x, ...; the next tile's scores are in flight (registers) while
// this one is bucketed.  part[outcome * M + base[b] + offs[tile][b] + .]
__global__ __launch_bounds__(1024) void msd_partition_kernel(const float* __restrict__ scores, int64_t lds, const uint32_t* __restrict__ tables,
                                                            const uint32_t* __restrict__ offs, const uint32_t* __restrict__ bases,
                                                            u32x2* __restrict__ part, const uint32_t* __restrict__ flags, int N, int64_t M, int nbs,
                                                            int n_tiles, int src_is_keys, unsigned long long* stamps) {
  constexpr int TPB = 1024, ITEMS = 16, BPT = MSD_NB_MAX / TPB;
  MDG_ST_DECL;
  extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];       // spair[MSD_TILE] (u32x2) | tab[MSD_N1 + MSD_NC] | bcnt[MSD_NB_MAX]
  __shared__ uint32_t wsum[16];
  const int64_t seg = blockIdx.y;
  if (flags[seg]) return;
  u32x2* spair = reinterpret_cast<u32x2*>(dyn);
  uint32_t* tab = dyn + 2 * MSD_TILE;
  uint32_t* bcnt = tab + MSD_N1 + MSD_NC;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* sc = scores + seg * static_cast<int64_t>(N) * lds;
  const uint32_t* t_hdr = tables + seg * MSD_TABLE_WORDS;
  for (int c = tid; c < (MSD_N1 + MSD_NC) / 4; c += TPB) reinterpret_cast<u32x4*>(tab)[c] = reinterpret_cast<const u32x4*>(t_hdr + MSD_HDR)[c];
  const MsdMap m = msd_load_map(t_hdr);
  u32x2* dst = part + seg * M;
  float raw[ITEMS];
  MsdTileGeom gnext;
  const auto load_tile = [&](int t) {
    gnext.of(t);
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      int i, j;
      raw[k] = gnext.cell(k, tid, N, i, j) ? sc[static_cast<int64_t>(i) * lds + j] : 0.f;
    }
  };
  // where the tile's run of bucket b starts: thread tid owns buckets BPT * tid .. (loaded a tile ahead, like the scores)
  uint32_t gpos[BPT];
  const auto load_offsets = [&](int t) {
#pragma unroll
    for (int e = 0; e < BPT; ++e) {
      const int b = BPT * tid + e;
      gpos[e] = b < nbs ? bases[seg * nbs + b] + offs[(seg * n_tiles + t) * static_cast<int64_t>(nbs) + b] : 0u;
    }
  };
  int t = blockIdx.x;
  if (t < n_tiles) { load_tile(t); load_offsets(t); }
  for (; t < n_tiles; t += gridDim.x) {
    const MsdTileGeom g = gnext;
    MDG_ST(0);
    for (int b = tid; b < MSD_NB_MAX; b += TPB) bcnt[b] = 0;
    uint32_t key[ITEMS], sb[ITEMS];                       // sb = slot in the tile's run | bucket << 16
    uint32_t gcur[BPT];
    uint32_t okm = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      int i, j;
      okm |= g.cell(k, tid, N, i, j) ? 1u << k : 0u;
      key[k] = src_is_keys ? __builtin_bit_cast(uint32_t, raw[k]) : mdg_order_key(raw[k]);
    }
#pragma unroll
    for (int e = 0; e < BPT; ++e) gcur[e] = gpos[e];
    if (t + static_cast<int>(gridDim.x) < n_tiles) { load_tile(t + gridDim.x); load_offsets(t + gridDim.x); }
    __syncthreads();                                       // bcnt zeroed, tab loaded (and the previous tile's copy-out done with them)
    MDG_ST(1);
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      sb[k] = MSD_SKIP;
      if ((okm >> k) & 1u) {
        const uint32_t b = msd_bucket_of(key[k], tab, tab + MSD_N1, m);
        sb[k] = atomicAdd(&bcnt[b], 1u) | (b << 16);
      }
    }
    MDG_ST(2);
    __syncthreads();
    MDG_ST(3);
    uint32_t c4[BPT], st[BPT];                             // exclusive scan of the bucket counts, BPT per thread
    {
      uint32_t tot = 0;
#pragma unroll
      for (int e = 0; e < BPT; ++e) { c4[e] = bcnt[BPT * tid + e]; tot += c4[e]; }
      const uint32_t inc = wave_inclusive(tot, lane);
      if (lane == 63) wsum[wave] = inc;
      __syncthreads();
      uint32_t run = inc - tot;
      for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
      for (int e = 0; e < BPT; ++e) {
        st[e] = run;
        bcnt[BPT * tid + e] = run;
        run += c4[e];
      }
    }
    __syncthreads();
    MDG_ST(4);
    // in LDS a pair is (key, bucket << 14 | cell of the tile): the copy-out neither re-derives the bucket from the key (two table reads
    // and ~25 instructions per key) nor did the position have to stay in registers
#pragma unroll
    for (int k = 0; k < ITEMS; ++k)
      if (sb[k] != MSD_SKIP) spair[bcnt[sb[k] >> 16] + (sb[k] & 0xFFFFu)] = u32x2{key[k], ((sb[k] >> 16) << 14) | static_cast<uint32_t>(k * TPB + tid)};
    __syncthreads();
#pragma unroll
    for (int e = 0; e < BPT; ++e) bcnt[BPT * tid + e] = gcur[e] - st[e];       // (position in the bucket) - (position in LDS)
    __syncthreads();
    MDG_ST(5);
    // (copy-out by sixteen lanes per run from the bucket tables was measured: 177 against 175 us per outcome -- the idle lanes of short
    // runs cost more than they save)
    const int n_valid = g.keys(N);
    const uint32_t qbase = (static_cast<uint32_t>(g.r0) << 16) | static_cast<uint32_t>(g.c0);
#pragma unroll 4
    for (int k = 0; k < ITEMS; ++k) {
      const int idx = k * TPB + tid;
      if (idx >= n_valid) break;
      const u32x2 v = spair[idx];
      const uint32_t cellw = v[1] & 16383u;
      dst[bcnt[v[1] >> 14] + static_cast<uint32_t>(idx)] = u32x2{v[0], qbase + ((cellw >> 7) << 16) + (cellw & 127u)};
    }
    MDG_ST(6);
    __syncthreads();                                       // spair / bcnt are read: the next tile may overwrite them
    MDG_ST(7);
  }
  MDG_ST_OUT(stamps + (blockIdx.y * gridDim.x + blockIdx.x) * 8);
}

// The same partition with the copy-out of tile t inside the slot phase of tile t + 1 (per item: one returning LDS atomic of the new tile,
// one LDS pair + one global store of the old one): the 256 workgroups of a launch run in lock step, so with the phases in a row every CU
// stored in the same sixth of a tile's time (HBM saturated) and every CU sorted in LDS in the rest (HBM idle).  Needs the run offsets of
// the old tile (gofs) beside the counters of the new one, and the counters double-buffered (zeroed a tile ahead: four barriers per tile
// instead of six): 12 x nbs bytes of LDS beside the 128-KB tile and the 18-KB table -- nbs <= 1024, i.e. N <= 4096; the kernel above
// takes the rest.
__global__ __launch_bounds__(1024) void msd_partition_pipe_kernel(const float* __restrict__ scores, int64_t lds, const uint32_t* __restrict__ tables,
                                                                 const uint32_t* __restrict__ offs, const uint32_t* __restrict__ bases,
                                                                 u32x2* __restrict__ part, const uint32_t* __restrict__ flags, int N, int64_t M, int nbs,
                                                                 int n_tiles, int src_is_keys, unsigned long long* stamps) {
  constexpr int TPB = 1024, ITEMS = 16;
  MDG_ST_DECL;
  extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];       // spair[MSD_TILE] (u32x2) | tab[MSD_N1 + MSD_NC] | bcnt[2][nbs] | gofs[nbs]
  __shared__ uint32_t wsum[16];
  const int64_t seg = blockIdx.y;
  if (flags[seg]) return;
  u32x2* spair = reinterpret_cast<u32x2*>(dyn);
  uint32_t* tab = dyn + 2 * MSD_TILE;
  uint32_t* bcnt2 = tab + MSD_N1 + MSD_NC;
  uint32_t* gofs = bcnt2 + 2 * nbs;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const auto add_op = [](uint32_t x, uint32_t y) { return x + y; };
  const float* sc = scores + seg * static_cast<int64_t>(N) * lds;
  const uint32_t* t_hdr = tables + seg * MSD_TABLE_WORDS;
  for (int c = tid; c < (MSD_N1 + MSD_NC) / 4; c += TPB) reinterpret_cast<u32x4*>(tab)[c] = reinterpret_cast<const u32x4*>(t_hdr + MSD_HDR)[c];
  for (int b = tid; b < 2 * nbs; b += TPB) bcnt2[b] = 0;
  const MsdMap m = msd_load_map(t_hdr);
  u32x2* dst = part + seg * M;
  float raw[ITEMS];
  MsdTileGeom gnext;
  // thread tid owns bucket tid (nbs <= 1024): its count, its start in the tile, its run's place in the pair buffer (loaded a tile ahead,
  // like the scores; the two words are added where they are used -- added at the load, the sum waited for the scores as well)
  uint32_t gbase = 0, goff = 0;
  const auto load_tile = [&](int t) {
    gnext.of(t);
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      int i, j;
      raw[k] = gnext.cell(k, tid, N, i, j) ? sc[static_cast<int64_t>(i) * lds + j] : 0.f;
    }
    if (tid < nbs) {
      gbase = bases[seg * nbs + tid];
      goff = offs[(seg * n_tiles + t) * static_cast<int64_t>(nbs) + tid];
    }
  };
  int p_valid = 0;                                         // the tile whose pairs wait in spair: its size and the position word of its first cell
  uint32_t p_qbase = 0;
  const auto copy_out_item = [&](int k) {
    const int idx = k * TPB + tid;
    if (idx < p_valid) {
      const u32x2 v = spair[idx];
      const uint32_t cellw = v[1] & 16383u;
      dst[gofs[v[1] >> 14] + static_cast<uint32_t>(idx)] = u32x2{v[0], p_qbase + ((cellw >> 7) << 16) + (cellw & 127u)};
    }
  };
  int t = blockIdx.x, par = 0;
  if (t < n_tiles) load_tile(t);
  __syncthreads();                                         // table and zeroed counters
  for (; t < n_tiles; t += gridDim.x, par ^= 1) {
    const MsdTileGeom g = gnext;
    uint32_t* bcnt = bcnt2 + par * nbs;
    MDG_ST(0);
    uint32_t key[ITEMS], sb[ITEMS];                       // sb = slot in the tile's run | bucket << 16
    uint32_t okm = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      int i, j;
      okm |= g.cell(k, tid, N, i, j) ? 1u << k : 0u;
      key[k] = src_is_keys ? __builtin_bit_cast(uint32_t, raw[k]) : mdg_order_key(raw[k]);
    }
    const uint32_t gcur = gbase + goff;
    if (t + static_cast<int>(gridDim.x) < n_tiles) load_tile(t + gridDim.x);
    MDG_ST(1);
    // slots of this tile | pairs of the previous tile out
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      sb[k] = MSD_SKIP;
      if ((okm >> k) & 1u) {
        const uint32_t b = msd_bucket_of(key[k], tab, tab + MSD_N1, m);
        sb[k] = atomicAdd(&bcnt[b], 1u) | (b << 16);
      }
      copy_out_item(k);
    }
    MDG_ST(2);
    __syncthreads();                                       // counts complete; spair and gofs are read
    MDG_ST(3);
    uint32_t st = 0;                                       // exclusive scan of the bucket counts, one per thread
    {
      const uint32_t c = tid < nbs ? bcnt[tid] : 0u;
      const uint32_t inc = wave_scan_dpp(c, 0u, add_op);
      if (lane == 63) wsum[wave] = inc;
      __syncthreads();
      uint32_t run = inc - c;
      for (int w = 0; w < wave; ++w) run += wsum[w];
      st = run;
      if (tid < nbs) bcnt[tid] = run;
    }
    __syncthreads();
    MDG_ST(4);
    // in LDS a pair is (key, bucket << 14 | cell of the tile): the copy-out neither re-derives the bucket from the key (two table reads
    // and ~25 instructions per key) nor did the position have to stay in registers
#pragma unroll
    for (int k = 0; k < ITEMS; ++k)
      if (sb[k] != MSD_SKIP) spair[bcnt[sb[k] >> 16] + (sb[k] & 0xFFFFu)] = u32x2{key[k], ((sb[k] >> 16) << 14) | static_cast<uint32_t>(k * TPB + tid)};
    if (tid < nbs) {
      gofs[tid] = gcur - st;                               // (position in the bucket) - (position in LDS)
      bcnt2[(par ^ 1) * nbs + tid] = 0u;                   // the next tile's counters (last read a tile ago)
    }
    p_valid = g.keys(N);
    p_qbase = (static_cast<uint32_t>(g.r0) << 16) | static_cast<uint32_t>(g.c0);
    MDG_ST(5);
    __syncthreads();                                       // the tile's pairs, their offsets and the zeroed counters
    MDG_ST(6);
  }
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) copy_out_item(k);
  MDG_ST_OUT(stamps + (blockIdx.y * gridDim.x + blockIdx.x) * 8);
}

// Fine bins of the bucket sorts: on the COMPOSITE (u(key) - u(lo), position).  u is the key itself -- linear in the score inside a binade,
// logarithmic across binades, which suits the tail buckets of heavy-tailed scores -- EXCEPT in a bucket that reaches towards zero (see init);
// the one bucket of an outcome that holds scores of both signs: it spans 2^31 keys (every binade down to the denormals, twice) with its scores at the two ends, and bins linear
// in the key put 1 700 of its 4 000 keys in one bin.  There u is the score as a fixed-point number, round(score * 2^e) with e from the
// bucket's largest magnitude (|u| <= 2^30): a narrow quantile slice around zero is flat in the score.  Either u is monotone in the key
// (exact power-of-two scaling, rounding and saturating conversion are monotone; NaN keys are clamped to the largest finite score first),
// which the ranks need; distinct keys may share a u (scores far below the bucket's largest magnitude), which only costs bin occupancy.
// The position bits below the key part spread a tie group over several bins in position order (fp32 scores around 6 are 2^-21 apart:
// 8.4 million of them tie in groups of 2..12), so a bin holds about half a key whatever the tie structure.
struct MsdFine {
  int e, sh;
  uint32_t umin;
  bool fixed;
  bool wide;                 // sh >= 26: the position bits are shifted out, the bin is (u - umin) >> (sh - 26)
  __device__ __forceinline__ static float score_of(uint32_t key) {
    const uint32_t kc = key < 0x00800000u ? 0x00800000u : (key > 0xFF7FFFFFu ? 0xFF7FFFFFu : key);      // -FLT_MAX .. FLT_MAX: infinities and NaNs saturate
    return __builtin_bit_cast(float, (kc & 0x80000000u) ? (kc & 0x7FFFFFFFu) : ~kc);
  }
  __device__ __forceinline__ uint32_t u_of(uint32_t key) const {
    return fixed ? static_cast<uint32_t>(__float2int_rn(ldexpf(score_of(key), e))) + 0x80000000u : key;
  }
  // mean = the bucket's mean key.  Fixed point when the bucket holds both signs, or holds one sign with its keys massed at the
  // large-magnitude end of its key range (a slice that reaches down towards zero through many binades: flat in the score, not in its
  // logarithm); the key itself otherwise (within a binade the two agree; a heavy tail's bucket is massed at its small-magnitude end)
  __device__ __forceinline__ void init(uint32_t kmin, uint32_t kmax, float mean, int lgnf) {
    const float pos = (mean - static_cast<float>(kmin)) / (static_cast<float>(kmax - kmin) + 1.f);
    fixed = (kmin < 0x80000000u && kmax >= 0x80000000u) || (kmin >= 0x80000000u && pos > 0.6f) || (kmax < 0x80000000u && pos < 0.4f);
    const float a = fabsf(score_of(kmin)), b = fabsf(score_of(kmax)), m = a > b ? a : b, mn = a > b ? b : a;
    e = m > 0.f ? 29 - ilogbf(m) : 0;
    // one sign: no finer than the smallest score's own spacing (u then steps by one between neighbouring fp32 values instead of by
    // 64: the bins' resolution goes to the position bits, which is what spreads tie groups)
    if (!(kmin < 0x80000000u && kmax >= 0x80000000u) && mn > 0.f && 23 - ilogbf(mn) < e) e = 23 - ilogbf(mn);
    umin = u_of(kmin);
    const uint64_t span = (static_cast<uint64_t>(u_of(kmax) - umin) << 26) | 0x3FFFFFFull;
    const int bits = 64 - __builtin_clzll(span);
    sh = bits > lgnf ? bits - lgnf : 0;                    // (span >> sh) < number of fine bins
    // the same in every lane: into scalar registers (the bin function's branches are then scalar ones)
    e = __builtin_amdgcn_readfirstlane(e);
    sh = __builtin_amdgcn_readfirstlane(sh);
    umin = __builtin_amdgcn_readfirstlane(umin);
    fixed = __builtin_amdgcn_readfirstlane(fixed ? 1 : 0) != 0;
    wide = sh >= 26;
  }
  // ((u - umin) << 26 | position) >> sh in 32-bit pieces (sh < 26 only when u - umin has fewer than 14 bits: the shifted difference fits)
  __device__ __forceinline__ uint32_t operator()(uint32_t key, uint32_t qq) const {
    const uint32_t d = u_of(key) - umin;
    if (wide) return d >> (sh - 26);
    return (d << (26 - sh)) | ((((qq >> 16) << 13) | (qq & 0x1FFFu)) >> sh);
  }
};

// One bucket at a time per workgroup (persistent: workgroup x of outcome y takes buckets x, x + gridDim.x, ... with the next bucket's
// pairs in flight in registers): the rank of every key inside the bucket, written at the key's own place in the bucket's range as one
// word -- (rank in the bucket) << 14 | (row in the output block) << 7 | (column in the output block).  The bucket's range is already
// in runs by output block (see MsdTileGeom), so nothing is regrouped and there are no global atomics.
// One returning LDS atomic per key (its slot in its fine bin); items behind the bucket's last key take none (thousands of lanes on one
// dummy counter serialise).
// A bucket beyond the LDS room of the bucket sort proper (fewer than 65 536 keys; the bucket function's sub-ranges assume a density
// that is flat inside a level-2 bin, which a sparse region at the edge of a score distribution is not): the same counting sort and the
// same output words, with the pairs streamed through global memory instead of held in registers and LDS -- sorted by fine bin into
// the bucket's slot of `tmp` (65 536 pairs per listed bucket), then every key's bin-mates read back from there.  Called by the
// workgroups of msd_bucket_kernel before they start on the queue of ordinary buckets (a launch of its own ran with a handful of
// workgroups on an otherwise idle chip: 79 us per group of 8 outcomes of the bench's score tensor).  lds: 2 x (NF / 2 + 4) words.
template <int NF>
__device__ __forceinline__ void msd_big_bucket(const u32x2* __restrict__ in, u32x2* __restrict__ tmp, uint32_t* __restrict__ dst, int n, uint32_t* lds,
                                               uint32_t* wsum_f, uint32_t* kst, bool& too_many) {
  constexpr int TPB = 1024, LGNF = 31 - __builtin_clz(NF), WPT = NF / 2 / TPB, U = 4;
  uint32_t* fc = lds;                                      // fine-bin counters, then starts (two u16 per word)
  uint32_t* cur = lds + NF / 2 + 4;                        // cursors: start + keys placed so far
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const auto umin_op = [](uint32_t x, uint32_t y) { return x < y ? x : y; };
  const auto umax_op = [](uint32_t x, uint32_t y) { return x > y ? x : y; };
  const auto add_op = [](uint32_t x, uint32_t y) { return x + y; };
  for (int i = tid; i < NF / 2 + 4; i += TPB) fc[i] = 0;
  if (tid == 0) { kst[0] = 0xFFFFFFFFu; kst[1] = 0u; kst[2] = 0u; }
  uint32_t kmin = 0xFFFFFFFFu, kmax = 0u, ksum = 0u;       // (sum of key >> 16: below 2^32 for 65 535 keys)
  for (int i0 = tid; i0 < n; i0 += U * TPB) {
    uint32_t kk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) kk[u] = i0 + u * TPB < n ? in[i0 + u * TPB][0] : 0xFFFFFFFFu;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = i0 + u * TPB < n;
      kmin = kmin < kk[u] ? kmin : kk[u];
      kmax = (ok && kk[u] > kmax) ? kk[u] : kmax;
      ksum += ok ? kk[u] >> 16 : 0u;
    }
  }
  kmin = wave_reduce_dpp(kmin, 0xFFFFFFFFu, umin_op);
  kmax = wave_reduce_dpp(kmax, 0u, umax_op);
  ksum = wave_reduce_dpp(ksum, 0u, add_op);
  __syncthreads();
  if (lane == 0 && kmin <= kmax) { atomicMin(&kst[0], kmin); atomicMax(&kst[1], kmax); atomicAdd(&kst[2], ksum); }
  __syncthreads();
  MsdFine fine_of;
  fine_of.init(kst[0], kst[1], static_cast<float>(kst[2]) / static_cast<float>(n) * 65536.f, LGNF);
  for (int i0 = tid; i0 < n; i0 += U * TPB) {
    u32x2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i0 + u * TPB < n) v[u] = in[i0 + u * TPB];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i0 + u * TPB < n) {
        const uint32_t fi = fine_of(v[u][0], v[u][1]);
        atomicAdd(&fc[fi >> 1], 1u << (16u * (fi & 1u)));
      }
  }
  __syncthreads();
  {
    uint32_t w[WPT], tot = 0;
#pragma unroll
    for (int e = 0; e < WPT; ++e) { w[e] = fc[tid * WPT + e]; tot += (w[e] & 0xFFFFu) + (w[e] >> 16); }
    const uint32_t inc = wave_scan_dpp(tot, 0u, add_op);
    if (lane == 63) wsum_f[wave] = inc;
    __syncthreads();
    uint32_t run = inc - tot;
    for (int v = 0; v < wave; ++v) run += wsum_f[v];
#pragma unroll
    for (int e = 0; e < WPT; ++e) {
      const uint32_t c0 = w[e] & 0xFFFFu, c1 = w[e] >> 16;
      const uint32_t st = run | ((run + c0) << 16);        // (starts below 65 536: the bucket has fewer keys)
      fc[tid * WPT + e] = st;
      cur[tid * WPT + e] = st;
      run += c0 + c1;
    }
  }
  __syncthreads();
  // sorted by fine bin into `tmp`
  for (int i0 = tid; i0 < n; i0 += U * TPB) {
    u32x2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i0 + u * TPB < n) v[u] = in[i0 + u * TPB];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i0 + u * TPB < n) {
        const uint32_t fi = fine_of(v[u][0], v[u][1]), fh = 16u * (fi & 1u);
        const uint32_t pos = (atomicAdd(&cur[fi >> 1], 1u << fh) >> fh) & 0xFFFFu;
        tmp[pos] = v[u];
      }
  }
  __threadfence();
  __syncthreads();
  __threadfence();
  // ranks: a bin's keys ordered by (key, position) by counting (four keys' bin-mates in flight per thread)
  for (int i0 = tid; i0 < n; i0 += U * TPB) {
    u32x2 v[U];
    uint32_t s0[U], cnt[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      cnt[u] = 0u; s0[u] = 0u; r[u] = 0u;
      v[u] = u32x2{0u, 0u};
      if (i0 + u * TPB < n) v[u] = in[i0 + u * TPB];
    }
    uint32_t cmax = 0u;
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i0 + u * TPB < n) {
        const uint32_t fi = fine_of(v[u][0], v[u][1]);
        s0[u] = (fc[fi >> 1] >> (16u * (fi & 1u))) & 0xFFFFu;
        cnt[u] = ((cur[fi >> 1] >> (16u * (fi & 1u))) & 0xFFFFu) - s0[u];
        if (cnt[u] > static_cast<uint32_t>(MSD_BIG_TIES)) { too_many = true; cnt[u] = 0u; }
        cmax = cmax > cnt[u] ? cmax : cnt[u];
      }
    for (uint32_t m = 0; m < cmax; ++m) {
      u32x2 o[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (m < cnt[u]) o[u] = __builtin_nontemporal_load(&tmp[s0[u] + m]);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (m < cnt[u]) r[u] += (o[u][0] < v[u][0] || (o[u][0] == v[u][0] && o[u][1] < v[u][1])) ? 1u : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i0 + u * TPB < n) dst[i0 + u * TPB] = ((s0[u] + r[u]) << 14) | (((v[u][1] >> 16) & 127u) << 7) | (v[u][1] & 127u);
  }
  __syncthreads();                                         // the LDS words are the caller's again
}

template <int NF>
__global__ __launch_bounds__(1024, 4) void msd_bucket_kernel(const u32x2* __restrict__ part, const uint32_t* __restrict__ bases,
                                                           const uint32_t* __restrict__ totals, uint32_t* __restrict__ ranked,
                                                           u32x2* __restrict__ tmp_all, uint32_t* __restrict__ flags, uint32_t* __restrict__ big,
                                                           int64_t M, int nbs, int nbt, unsigned long long* stamps) {
  MDG_ST_DECL;
  constexpr int TPB = 1024, CAP = MSD_CAP, ITEMS = CAP / TPB, WPT = NF / 2 / TPB, WAVES = TPB / 64;     // NF fine bins, two u16 counters per word
  constexpr int LGNF = 31 - __builtin_clz(NF);
  static_assert(CAP % TPB == 0 && (NF & (NF - 1)) == 0 && NF % (2 * TPB) == 0 && CAP <= 16384 && NF <= 65536, "bucket sort shape");
  static_assert(WPT % 4 == 0, "16-byte accesses cover the fine counters");
  extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];       // sorted[CAP] (u32x2) | fc[NF / 2 + 32]
  __shared__ __attribute__((aligned(16))) uint32_t wsum_f[WAVES];
  // smallest key | largest key | sum of key >> 14 (below 2^32 for 12 288 keys: the mean key to 2^14, which is all MsdFine::init asks
  // of it) of a bucket, double-buffered: the NEXT bucket's are gathered at the end of this bucket's last phase (its pairs have
  // arrived by then) and are behind the closing barrier, so a bucket starts without a reduction and its two barriers
  __shared__ uint32_t kstat[2][4];
  __shared__ uint32_t q_next[2];
  const int64_t seg = blockIdx.y;
  if (flags[seg]) return;
  u32x2* sorted = reinterpret_cast<u32x2*>(dyn);
  uint32_t* fc = dyn + 2 * CAP;
  const uint16_t* fc16 = reinterpret_cast<const uint16_t*>(fc);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const auto umin_op = [](uint32_t x, uint32_t y) { return x < y ? x : y; };
  const auto umax_op = [](uint32_t x, uint32_t y) { return x > y ? x : y; };
  const auto add_op = [](uint32_t x, uint32_t y) { return x + y; };
  // the next bucket's pairs are in flight (registers) while this one is sorted
  uint32_t nkey[ITEMS], nq[ITEMS];
  int nn = 0;
  uint32_t nrb = 0;
  const auto fetch = [&](int bb) {
    nn = bb < nbt ? static_cast<int>(totals[seg * nbs + bb]) : 0;
    if (nn > CAP) nn = 0;                                  // msd_big_bucket's
    nrb = bb < nbt ? bases[seg * nbs + bb] : 0u;
    const u32x2* src = part + seg * M + nrb;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int idx = k * TPB + tid;
      u32x2 v = u32x2{0xFFFFFFFFu, 0u};
      if (idx < nn) v = src[idx];
      nkey[k] = v[0];
      nq[k] = v[1];
    }
  };
  const auto zero_counters = [&]() {
    uint32_t z0 = 0u;
    asm volatile("" : "+v"(z0));                           // (a zero made here: hoisted out of the bucket loop it was spilled and reloaded)
    const u32x4 zero4{z0, z0, z0, z0};
#pragma unroll
    for (int v4 = 0; v4 < WPT / 4; ++v4) reinterpret_cast<u32x4*>(fc)[tid * (WPT / 4) + v4] = zero4;
    if (tid < 8) reinterpret_cast<u32x4*>(fc + NF / 2)[tid] = zero4;
  };
  // extremes and key sum of the pairs in nkey (the absent items' keys are ~0), published to kstat[slot]
  const auto publish_stats = [&](int slot) {
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u, ksum = 0u;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const bool ok = k * TPB + tid < nn;
      kmin = kmin < nkey[k] ? kmin : nkey[k];
      kmax = (ok && nkey[k] > kmax) ? nkey[k] : kmax;
      ksum += ok ? nkey[k] >> 14 : 0u;
    }
    kmin = wave_reduce_dpp(kmin, 0xFFFFFFFFu, umin_op);
    kmax = wave_reduce_dpp(kmax, 0u, umax_op);
    ksum = wave_reduce_dpp(ksum, 0u, add_op);
    if (lane == 0 && kmin <= kmax) { atomicMin(&kstat[slot][0], kmin); atomicMax(&kstat[slot][1], kmax); atomicAdd(&kstat[slot][2], ksum); }
  };
  bool too_many = false;
  // ---- the outcome's big buckets first, one per workgroup (the workgroups that take one join the queue below late: it evens out)
  {
    const uint32_t nbig = big[seg * MSD_BIG_WORDS];
    static_assert(2 * (MSD_BIG_NF / 2 + 4) <= 2 * CAP + NF / 2 + 32, "the big-bucket sort works in the bucket sort's LDS");
    for (uint32_t sl = blockIdx.x; sl < nbig; sl += gridDim.x) {
      const int bb = static_cast<int>(big[seg * MSD_BIG_WORDS + 1 + sl]);
      const uint32_t rbb = bases[seg * nbs + bb];
      msd_big_bucket<MSD_BIG_NF>(part + seg * M + rbb, tmp_all + (seg * MSD_BIG_MAX + sl) * int64_t{65536}, ranked + seg * M + rbb,
                                 static_cast<int>(totals[seg * nbs + bb]), dyn, wsum_f, kstat[0], too_many);
    }
  }
  // ---- then the ordinary buckets, handed out by a counter (big[.. + MSD_BIG_QUEUE], zeroed by msd_base_kernel): a workgroup that got
  // the bucket with the long tie walks, or started late, takes fewer.  The bucket after next is drawn at the top of a bucket and read
  // behind its closing barrier, so the draw's latency is off the path.
  uint32_t* qctr = big + seg * MSD_BIG_WORDS + MSD_BIG_QUEUE;
  if (tid == 0) {
    q_next[0] = atomicAdd(qctr, 1u);
    q_next[1] = atomicAdd(qctr, 1u);
  }
  __syncthreads();
  int b = __builtin_amdgcn_readfirstlane(static_cast<int>(q_next[0])), bn = __builtin_amdgcn_readfirstlane(static_cast<int>(q_next[1]));
  fetch(b);
  zero_counters();
  if (tid < 2) { kstat[tid][0] = 0xFFFFFFFFu; kstat[tid][1] = 0u; kstat[tid][2] = 0u; }
  __syncthreads();
  publish_stats(0);
  __syncthreads();
  int par = 0;
  for (; b < nbt; par ^= 1) {
    const int n = nn;
    const uint32_t rb = nrb;
    // item slots in use (the same in every lane: slots behind it are skipped by scalar branches -- a bucket fills 8 or 9 of its 12, and
    // the kernel is bound by instruction issue)
    const int used = __builtin_amdgcn_readfirstlane((n + TPB - 1) / TPB);
    uint32_t key[ITEMS], q[ITEMS], ss[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) { key[k] = nkey[k]; q[k] = nq[k]; }
    fetch(bn);
    uint32_t drawn = 0u;
    if (tid == 0) drawn = atomicAdd(qctr, 1u);
    MDG_ST(0);
    MsdFine fine_of;
    {
      const uint32_t lo = kstat[par][0], hi = kstat[par][1], sm = kstat[par][2];
      fine_of.init(lo <= hi ? lo : 0u, lo <= hi ? hi : 0u, n > 0 ? static_cast<float>(sm) / static_cast<float>(n) * 16384.f : 0.f, LGNF);
    }
    if (tid == 0) { kstat[par ^ 1][0] = 0xFFFFFFFFu; kstat[par ^ 1][1] = 0u; kstat[par ^ 1][2] = 0u; }      // (last read a bucket ago; published into four barriers from here)
    MDG_ST(1);
    // ---- the key's slot in its fine bin (the counters were zeroed in the previous bucket's last phase)
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      ss[k] = 0u;
      if (k < used && k * TPB + tid < n) {
        const uint32_t fi = fine_of(key[k], q[k]);
        const uint32_t fh = 16u * (fi & 1u);
        ss[k] = ((atomicAdd(&fc[fi >> 1], 1u << fh) >> fh) & 0xFFFFu) | (fi << 16);      // slot in the bin | bin
      }
      if (k % 4 == 3) __builtin_amdgcn_sched_barrier(0);    // four items' atomics in flight at a time (all of them at once spilled registers)
    }
    MDG_ST(2);
    __syncthreads();
    MDG_ST(3);
    {   // exclusive scan of the fine bins in place (four words per thread)
      uint32_t w[WPT];
#pragma unroll
      for (int v4 = 0; v4 < WPT / 4; ++v4) {
        const u32x4 t4 = reinterpret_cast<const u32x4*>(fc)[tid * (WPT / 4) + v4];
        w[4 * v4] = t4[0]; w[4 * v4 + 1] = t4[1]; w[4 * v4 + 2] = t4[2]; w[4 * v4 + 3] = t4[3];
      }
      uint32_t tot = 0;
#pragma unroll
      for (int e = 0; e < WPT; ++e) tot += (w[e] & 0xFFFFu) + (w[e] >> 16);
      const uint32_t inc = wave_scan_dpp(tot, 0u, add_op);
      if (lane == 63) wsum_f[wave] = inc;
      __syncthreads();
      uint32_t run = inc - tot;
#pragma unroll
      for (int v4 = 0; v4 < WAVES / 4; ++v4) {              // every wave total, four per LDS read (the same addresses in every lane: broadcast)
        const u32x4 f4 = reinterpret_cast<const u32x4*>(wsum_f)[v4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * v4 + e < wave) run += f4[e];
      }
#pragma unroll
      for (int e = 0; e < WPT; ++e) {
        const uint32_t c0 = w[e] & 0xFFFFu, c1 = w[e] >> 16;
        w[e] = run | ((run + c0) << 16);
        run += c0 + c1;
      }
#pragma unroll
      for (int v4 = 0; v4 < WPT / 4; ++v4) reinterpret_cast<u32x4*>(fc)[tid * (WPT / 4) + v4] = u32x4{w[4 * v4], w[4 * v4 + 1], w[4 * v4 + 2], w[4 * v4 + 3]};
      if (tid == 0) fc[NF / 2] = static_cast<uint32_t>(n);  // fstart(NF) = the bucket's size
    }
    __syncthreads();
    MDG_ST(4);
    // Where the key's fine bin starts and how many keys it holds (the bins' starts are u16: two 16-bit reads).  A
    // key ALONE in its bin -- six of ten -- is in place by that alone (rank = bin start): it is neither written to `sorted` nor probed.
    uint32_t sc[ITEMS];                                      // bin start | keys in the bin << 16
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      sc[k] = 0u;
      if (k < used && k * TPB + tid < n) {
        const uint32_t fi = ss[k] >> 16;
        const uint32_t s0 = fc16[fi], s1 = fc16[fi + 1];      // (two u16 reads: the halves need no selecting)
        sc[k] = s0 | ((s1 - s0) << 16);
        if (s1 - s0 > 1u) sorted[s0 + (ss[k] & 0xFFFFu)] = u32x2{q[k], key[k]};
      }
    }
    MDG_ST(5);
    __syncthreads();
    zero_counters();                                       // for the next bucket: the counters were last read above, `sorted` is all that is read from here
    // keys that share a fine bin: their order is (key, position) -- one 64-bit compare per bin-mate (`sorted` holds position | key << 32).
    // Four bin-mates are probed with predicated, independent LDS reads (a bin holds ~half a key: two keys in a thousand have more and walk on)
    constexpr int PROBES = 4, GRP = 3;
    static_assert(ITEMS % GRP == 0, "tie-fix groups");
    const unsigned long long* sorted64 = reinterpret_cast<const unsigned long long*>(sorted);
    uint32_t* dst = ranked + seg * M + rb;
#pragma unroll
    for (int k0 = 0; k0 < ITEMS; k0 += GRP) {                // three items at a time: 12 probes in flight
      if (k0 < used) {
        unsigned long long o[GRP][PROBES];
#pragma unroll
        for (int e = 0; e < GRP; ++e) {
          const uint32_t s0 = sc[k0 + e] & 0xFFFFu, c = sc[k0 + e] >> 16, c1 = c > 1u ? c : 0u;
#pragma unroll
          for (int mth = 0; mth < PROBES; ++mth) {
            o[e][mth] = ~0ull;
            if (static_cast<uint32_t>(mth) < c1) o[e][mth] = sorted64[s0 + mth];
          }
        }
#pragma unroll
        for (int e = 0; e < GRP; ++e) {
          const int k = k0 + e;
          const uint32_t s0 = sc[k] & 0xFFFFu, c = sc[k] >> 16;
          const unsigned long long me = (static_cast<unsigned long long>(key[k]) << 32) | q[k];
          uint32_t r = 0;
#pragma unroll
          for (int mth = 0; mth < PROBES; ++mth) r += o[e][mth] < me ? 1u : 0u;
          if (c > static_cast<uint32_t>(PROBES)) {
            if (c > static_cast<uint32_t>(MSD_TIE_LIMIT)) too_many = true;
            else
              for (uint32_t mth = PROBES; mth < c; ++mth) r += sorted64[s0 + mth] < me ? 1u : 0u;
          }
          const int idx = k * TPB + tid;
          if (idx < n) dst[idx] = ((s0 + r) << 14) | (((q[k] >> 16) & 127u) << 7) | (q[k] & 127u);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    MDG_ST(6);
    publish_stats(par ^ 1);                                // (waits for the next bucket's pairs: they have had this bucket's time to arrive)
    if (tid == 0) q_next[0] = drawn;
    __syncthreads();                                       // `sorted` is read, the counters are zero, the next bucket's statistics are in: it may start
    b = bn;
    bn = __builtin_amdgcn_readfirstlane(static_cast<int>(q_next[0]));
    MDG_ST(7);
  }
  if (too_many) atomicOr(&flags[seg], MSD_F_TIES);
  MDG_ST_OUT(stamps + 8 * 4096 + (blockIdx.y * gridDim.x + blockIdx.x) * 8);
}

// one 128 x 128 block of the lower triangle, gathered from every bucket's run of it (8 lanes per run; the runs' starts and lengths are
// the block's row of the layout tables, staged in LDS first): ranks into an LDS tile, then whole rows of out[i, j] and of the mirrored block
template <bool VEC>
__global__ __launch_bounds__(512) void msd_block_gather_kernel(const uint32_t* __restrict__ ranked, const uint32_t* __restrict__ offs,
                                                               const uint16_t* __restrict__ counts, const uint32_t* __restrict__ bases,
                                                               float* __restrict__ out, int64_t ldo, int N, int64_t M,
                                                               int nbt, int nbs, int n_blocks, double denom, const uint32_t* __restrict__ flags) {
  constexpr int TPB = 512;
  extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];       // tile[BB][BB + 1] (float) | st[nbs] | bs[nbs] | ln[nbs] (u16)
  float (*tile)[BB + 1] = reinterpret_cast<float (*)[BB + 1]>(dyn);
  uint32_t (*tile_u)[BB + 1] = reinterpret_cast<uint32_t (*)[BB + 1]>(dyn);
  uint32_t* st = dyn + BB * (BB + 1);
  uint32_t* bs = st + nbs;
  uint16_t* ln = reinterpret_cast<uint16_t*>(bs + nbs);
  const int64_t seg = blockIdx.y;
  if (flags[seg]) return;
  // Workgroups go to the 8 XCDs round robin (gridDim.x is a multiple of 8): XCD x takes the blocks [x * per, (x + 1) * per) in order,
  // so that blocks t and t + 1 -- whose runs are neighbours in every bucket and share 128-byte lines -- are resident together on one
  // XCD and the shared line is fetched into that L2 once (blockIdx order: every 64-byte run cost a whole line, 2.6 TB/s of pairs)
  const int per = gridDim.x >> 3, t = static_cast<int>(blockIdx.x & 7u) * per + static_cast<int>(blockIdx.x >> 3), tid = threadIdx.x;
  if (t >= n_blocks) return;
  MsdTileGeom g;
  g.of(t);
  const int r0 = g.r0, c0 = g.c0;
  const int rcount = N - r0 < BB ? N - r0 : BB;
  const bool diag = r0 == c0;
  {
    const uint32_t* o_row = offs + (seg * n_blocks + t) * static_cast<int64_t>(nbs);
    const uint16_t* l_row = counts + (seg * n_blocks + t) * static_cast<int64_t>(nbs);
    const uint32_t* b_row = bases + seg * nbs;
    for (int r = tid; r < nbt; r += TPB) {
      const uint32_t bb = b_row[r];
      bs[r] = bb + 1u;                                     // ranks count from 1
      st[r] = bb + o_row[r];
      ln[r] = l_row[r];
    }
  }
  if (diag)
    for (int e = tid; e < BB; e += TPB) tile_u[e][e] = 0u;
  __syncthreads();
  const uint32_t* src = ranked + seg * M;
  const int grp = tid >> 3, l8 = tid & 7;
  // rank / M: both are integers below 2^24 on this path (N <= 5793), exact in fp32, and the correctly rounded fp32 quotient equals numpy's
  // float64 quotient rounded to fp32 (|r 2^e - k M| >= 1 keeps r / M away from every fp32 rounding boundary by more than a float64 ulp)
  // (the tile takes the integer ranks; they are divided once per element where the rows leave, all lanes busy, not here per gathered
  // word with half the lanes idle)
  const float denom_f = static_cast<float>(denom);
  const auto put = [&](uint32_t v, uint32_t base1) {
    const int r = (v >> 7) & 127, c = v & 127;
    const uint32_t rank = base1 + (v >> 14);
    tile_u[r][c] = rank;
    if (diag) tile_u[c][r] = rank;
  };
  // eight lanes per run, two words per lane (a run is ~16 words: few are longer), eight runs per lane in flight: the word
  // buffer of a launch group is larger than the Infinity Cache, so every load is an HBM miss and their number in flight is the speed
  constexpr int RU = 8;
  for (int rbase = 0; rbase < nbt; rbase += RU * (TPB / 8)) {
    uint32_t s[RU], b1[RU];
    int len[RU];
    uint32_t v[RU][3];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int r = rbase + u * (TPB / 8) + grp;
      len[u] = r < nbt ? static_cast<int>(ln[r]) : 0;
      s[u] = r < nbt ? st[r] : 0u;
      b1[u] = r < nbt ? bs[r] : 0u;
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
#pragma unroll
      for (int h = 0; h < 3; ++h)
        if (l8 + 8 * h < len[u]) v[u][h] = src[s[u] + l8 + 8 * h];
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
#pragma unroll
      for (int h = 0; h < 3; ++h)
        if (l8 + 8 * h < len[u]) put(v[u][h], b1[u]);
      for (int l = l8 + 24; l < len[u]; l += 8) put(src[s[u] + l], b1[u]);
    }
  }
  __syncthreads();
  float* o = out + seg * static_cast<int64_t>(N) * ldo;
  const int qd = tid & 31, rr = tid >> 5;                  // 32 lanes x 4 columns cover a 128-wide row; 16 rows per sweep
  const int ccount = diag ? rcount : BB;
  for (int r = rr; r < rcount; r += TPB / 32) {
    float* row = o + static_cast<int64_t>(r0 + r) * ldo + c0;
    f32x4 val;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      val[e] = static_cast<float>(tile_u[r][4 * qd + e]) / denom_f;
      tile[r][4 * qd + e] = val[e];                        // (the same thread's own four cells: the mirrored rows below read floats)
    }
    if (VEC && 4 * qd + 3 < ccount) *reinterpret_cast<f32x4*>(row + 4 * qd) = val;
    else
      for (int e = 0; e < 4; ++e)
        if (4 * qd + e < ccount) row[4 * qd + e] = val[e];
  }
  for (int r = rcount + rr; r < BB; r += TPB / 32)          // rows beyond a ragged N: never gathered, but the mirrored loop below reads the columns' cells
    for (int e = 0; e < 4; ++e) tile[r][4 * qd + e] = 0.f;
  if (diag) return;
  __syncthreads();
  for (int c = rr; c < BB; c += TPB / 32) {
    float* row = o + static_cast<int64_t>(c0 + c) * ldo + r0;
    if (VEC && 4 * qd + 3 < rcount) *reinterpret_cast<f32x4*>(row + 4 * qd) = f32x4{tile[4 * qd][c], tile[4 * qd + 1][c], tile[4 * qd + 2][c], tile[4 * qd + 3][c]};
    else
      for (int e = 0; e < 4; ++e)
        if (4 * qd + e < rcount) row[4 * qd + e] = tile[4 * qd + e][c];
  }
}

template <bool LIST, bool VEC>
__global__ __launch_bounds__(512) void rank_block_write_kernel(const u32x2* __restrict__ pairs, float* __restrict__ out, int64_t ldo, int N,
                                                               int64_t M, int n_blocks, double denom, const uint32_t* __restrict__ only) {
  constexpr int TPB = 512;
  __shared__ float tile[BB][BB + 1];
  const int t = blockIdx.x, tid = threadIdx.x;
  MDG_SEG_BEGIN(LIST, only, blockIdx.y, gridDim.y)
  int bi = static_cast<int>((sqrtf(8.0f * static_cast<float>(t) + 1.0f) - 1.0f) * 0.5f);
  while (bi * (bi + 1) / 2 > t) --bi;
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  const int bj = t - bi * (bi + 1) / 2;
  const int r0 = bi * BB, c0 = bj * BB;
  const int rcount = N - r0 < BB ? N - r0 : BB;
  const bool diag = bi == bj;
  const int count = diag ? rcount * (rcount - 1) / 2 : rcount * BB;
  if (diag) {
    for (int e = tid; e < BB; e += TPB) tile[e][e] = 0.f;
  }
  const u32x2* src = pairs + seg * M + block_base(bi, bj, N);
  for (int e = tid; e < count; e += TPB) {
    const u32x2 v = src[e];
    const float val = static_cast<float>(static_cast<double>(v[0] + 1u) / denom);
    const int r = v[1] >> 7, c = v[1] & 127;
    tile[r][c] = val;
    if (diag) tile[c][r] = val;
  }
  __syncthreads();
  float* o = out + seg * static_cast<int64_t>(N) * ldo;
  const int q = tid & 31, rr = tid >> 5;                   // 32 lanes x 4 columns cover a 128-wide row; 16 rows per sweep
  const int ccount = diag ? rcount : BB;
  // rows of out[i, j]
  for (int r = rr; r < rcount; r += TPB / 32) {
    float* row = o + static_cast<int64_t>(r0 + r) * ldo + c0;
    if (VEC && 4 * q + 3 < ccount) *reinterpret_cast<f32x4*>(row + 4 * q) = f32x4{tile[r][4 * q], tile[r][4 * q + 1], tile[r][4 * q + 2], tile[r][4 * q + 3]};
    else
      for (int e = 0; e < 4; ++e)
        if (4 * q + e < ccount) row[4 * q + e] = tile[r][4 * q + e];
  }
  // rows of the mirrored block out[j, i]
  if (!diag)
    for (int c = rr; c < BB; c += TPB / 32) {
      float* row = o + static_cast<int64_t>(c0 + c) * ldo + r0;
      if (VEC && 4 * q + 3 < rcount) *reinterpret_cast<f32x4*>(row + 4 * q) = f32x4{tile[4 * q][c], tile[4 * q + 1][c], tile[4 * q + 2][c], tile[4 * q + 3][c]};
      else
        for (int e = 0; e < 4; ++e)
          if (4 * q + e < rcount) row[4 * q + e] = tile[4 * q + e][c];
    }
  MDG_SEG_END(LIST)
}

// list[0] = number of flagged outcomes, list[1..] = which (ascending)
__global__ __launch_bounds__(1024) void msd_flag_list_kernel(const uint32_t* __restrict__ flags, uint32_t* __restrict__ list, int L) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int l0 = 0; l0 < L; l0 += 1024) {
    const int l = l0 + tid;
    const uint32_t f = (l < L && flags[l]) ? 1u : 0u;
    const uint32_t inc = wave_inclusive(f, lane);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t run = carry + inc - f;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    if (f) list[1 + run] = static_cast<uint32_t>(l);
    __syncthreads();
    if (tid == 1023) carry = run + f;
    __syncthreads();
  }
  if (tid == 0) list[0] = carry;
}

__global__ __launch_bounds__(256) void zero_diag_kernel(float* __restrict__ out, int64_t ldo, int N) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < N) out[(static_cast<int64_t>(blockIdx.y) * N + i) * ldo + i] = 0.f;
}

inline size_t a256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

// Geometric mean over K <= 8 tensors, elementwise, fp32 like scipy.stats.mstats.gmean on float32 input:
// exp(mean_k(log x_k)).  Entries where any x_k <= 0 (the zero diagonal of normalised ranks; masked by scipy)
// give 0.  (generate_embeddings.ipynb, the 5-seed ensembling cell.)
struct GmeanArgs { const float* in[8]; int K; };

__global__ __launch_bounds__(256) void gmean_kernel(const GmeanArgs a, float* __restrict__ out, int64_t n4, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i == 0) {                                            // the n % 4 last elements (an odd N^2 in a contiguous tensor)
    for (int64_t j = 4 * n4; j < n; ++j) {
      float acc = 0.f;
      bool pos = true;
      for (int k = 0; k < a.K; ++k) {
        const float v = a.in[k][j];
        pos = pos && v > 0.f;
        acc += logf(v > 0.f ? v : 1.f);
      }
      out[j] = pos ? expf(acc / static_cast<float>(a.K)) : 0.f;
    }
  }
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

static int64_t rank_blocks_of(int64_t N) { const int64_t nb = mdg_cdiv(N, BB); return nb * (nb + 1) / 2; }
// 16384-key tiles for the first three LSD passes (8192-key tiles everywhere measured 0.31 against 0.28 ms per 4096^2 outcome)
static bool rank_use_big(int64_t N) {
  (void)N;
  return true;
}

// ---- MSD path: eligibility, workspace ------------------------------------------------------------------------------------------
struct MsdPlan {
  bool on;
  int group;                 // outcomes per launch group (the group's buffers are reused by the next group: Infinity-Cache resident)
  int nbt, nbs, n_blocks;    // buckets per outcome; counter stride (a multiple of 256); output blocks = count / partition tiles
  size_t part_bytes, word_bytes, count_bytes, offs_bytes, tot_bytes, bigtmp_bytes;      // per outcome
  size_t group_bytes(int g) const {
    return a256(g * part_bytes) + a256(g * word_bytes) + a256(g * count_bytes) + a256(g * offs_bytes) + 2 * a256(g * tot_bytes) + a256(g * bigtmp_bytes) +
           a256(static_cast<size_t>(g) * MSD_BIG_WORDS * 4);
  }
  // per call (all outcomes): sample extremes | sample histograms (level 1, level 2) | tables
  size_t call_bytes(int64_t L) const { return a256(static_cast<size_t>(L) * 8) + a256(static_cast<size_t>(L) * (MSD_N1 + MSD_NC) * 4) + a256(static_cast<size_t>(L) * MSD_TABLE_WORDS * 4); }
};

static MsdPlan msd_plan(int64_t n_outcomes, int64_t N) {
  static MdgEnvInt msd_sw{"MDG_RANKS_MSD", 1};             // 0: the four-pass LSD sort only (what larger N and handed-back outcomes take)
  static MdgEnvInt group_sw{"MDG_RANKS_GROUP", 8};         // outcomes per launch group
  MsdPlan pl{};
  const int64_t M = N * (N - 1) / 2;
  const int64_t nbt = mdg_cdiv(M, int64_t{1} << MSD_QLG);
  pl.on = msd_sw.get() != 0 && N >= 2 && nbt <= MSD_NB_MAX;
  if (!pl.on) return pl;
  int g = group_sw.get();
  g = g < 1 ? 1 : g;
  pl.group = static_cast<int>(g < n_outcomes ? g : n_outcomes);
  pl.nbt = static_cast<int>(nbt);
  pl.nbs = (pl.nbt + 255) & ~255;
  pl.n_blocks = static_cast<int>(rank_blocks_of(N));
  pl.part_bytes = static_cast<size_t>(M) * 8;
  pl.word_bytes = static_cast<size_t>(M) * 4;
  pl.count_bytes = static_cast<size_t>(pl.n_blocks) * pl.nbs * 2;
  pl.offs_bytes = static_cast<size_t>(pl.n_blocks) * pl.nbs * 4;
  pl.tot_bytes = static_cast<size_t>(pl.nbs) * 4;
  pl.bigtmp_bytes = static_cast<size_t>(MSD_BIG_MAX) * 65536 * 8;
  return pl;
}

template <class C>
static size_t lsd_workspace_bytes(int64_t n_outcomes, int64_t N) {
  const size_t M = static_cast<size_t>(N) * (N - 1) / 2;
  const size_t nblk = (M + CfgStd::TILE - 1) / CfgStd::TILE;          // the last pass always runs on 8192-key tiles (the finer table)
  // keys / payloads x 2 | per-tile digit table | block fill counters
  return 4 * a256(static_cast<size_t>(n_outcomes) * M * 4) + a256(static_cast<size_t>(n_outcomes) * 256 * nblk * 4) +
         a256(static_cast<size_t>(n_outcomes) * static_cast<size_t>(rank_blocks_of(N)) * 4 * FILL_STRIDE);
}

// flags (one per outcome, live from the MSD path to the fallback) | MSD per-call tables | max(LSD scratch of all outcomes, MSD scratch of one group)
template <class C>
static size_t rank_workspace_bytes(int64_t n_outcomes, int64_t N) {
  const size_t lsd = lsd_workspace_bytes<C>(n_outcomes, N);
  const MsdPlan pl = msd_plan(n_outcomes, N);
  if (!pl.on) return lsd;
  const size_t fast = pl.group_bytes(pl.group);
  return 2 * a256(static_cast<size_t>(n_outcomes + 1) * 4) + pl.call_bytes(n_outcomes) + (lsd > fast ? lsd : fast);      // flags | list | ...
}

extern "C" size_t mdg_rank_normalize_workspace_bytes(int64_t n_outcomes, int64_t N) {
  if (n_outcomes <= 0 || N < 2) return 0;
  return rank_use_big(N) ? rank_workspace_bytes<CfgBig>(n_outcomes, N) : rank_workspace_bytes<CfgStd>(n_outcomes, N);
}

extern "C" int mdg_rank_normalize_fast_path(int64_t n_outcomes, int64_t N) {
  return (n_outcomes > 0 && N >= 2 && msd_plan(n_outcomes, N).on) ? 1 : 0;
}

extern "C" int mdg_rank_normalize(const float* scores, float* out, int64_t n_outcomes, int64_t N, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  return mdg_rank_normalize_ld(scores, N, out, N, n_outcomes, N, workspace, workspace_bytes, stream);
}

// the MSD path over all outcomes of the call, `group` at a time; raises flags[outcome] for what it leaves to the LSD kernels
static void msd_run(const MsdPlan& pl, const float* scores, int64_t lds, float* out, int64_t ldo, int64_t n_outcomes, int64_t N, char* call_ws, char* ws,
                    uint32_t* flags, hipStream_t st, int src_is_keys) {
  const int64_t M = N * (N - 1) / 2;
  const int n_blocks = pl.n_blocks;
  const double denom = static_cast<double>(N) * static_cast<double>(N - 1) / 2.0;
  const int G = pl.group;
  const unsigned L = static_cast<unsigned>(n_outcomes);
  // ---- bucket function of every outcome of the call: sample extremes -> sample histogram -> table
  char* c = call_ws;
  uint32_t* mm = reinterpret_cast<uint32_t*>(c); c += a256(static_cast<size_t>(L) * 8);
  uint32_t* hist1 = reinterpret_cast<uint32_t*>(c);
  uint32_t* hist2 = hist1 + static_cast<size_t>(L) * MSD_N1; c += a256(static_cast<size_t>(L) * (MSD_N1 + MSD_NC) * 4);
  uint32_t* tables = reinterpret_cast<uint32_t*>(c);
  hipLaunchKernelGGL(msd_init_minmax_kernel, dim3(static_cast<unsigned>(mdg_cdiv(L, 256))), dim3(256), 0, st, mm, static_cast<int>(L));
  (void)hipMemsetAsync(hist1, 0, static_cast<size_t>(L) * (MSD_N1 + MSD_NC) * 4, st);
  hipLaunchKernelGGL(msd_minmax_kernel, dim3(MSD_SAMPLE_WGS, L), dim3(256), 0, st, scores, lds, mm, static_cast<int>(N), M, src_is_keys);
  hipLaunchKernelGGL(msd_hist1_kernel, dim3(MSD_SAMPLE_WGS, L), dim3(256), 0, st, scores, lds, mm, hist1, static_cast<int>(N), M, src_is_keys);
  hipLaunchKernelGGL(msd_level1_kernel, dim3(L), dim3(MSD_N1), 0, st, hist1, mm, tables);
  hipLaunchKernelGGL(msd_hist2_kernel, dim3(MSD_SAMPLE_WGS, L), dim3(256), 0, st, scores, lds, tables, hist2, static_cast<int>(N), M, src_is_keys);
  hipLaunchKernelGGL(msd_table_kernel, dim3(L), dim3(1024), 0, st, hist2, tables, pl.nbt);
  // ---- group buffers
  char* p = ws;
  u32x2* part = reinterpret_cast<u32x2*>(p); p += a256(G * pl.part_bytes);
  uint32_t* ranked = reinterpret_cast<uint32_t*>(p); p += a256(G * pl.word_bytes);
  uint16_t* counts = reinterpret_cast<uint16_t*>(p); p += a256(G * pl.count_bytes);
  uint32_t* offs = reinterpret_cast<uint32_t*>(p); p += a256(G * pl.offs_bytes);
  uint32_t* totals = reinterpret_cast<uint32_t*>(p); p += a256(G * pl.tot_bytes);
  uint32_t* bases = reinterpret_cast<uint32_t*>(p); p += a256(G * pl.tot_bytes);
  u32x2* bigtmp = reinterpret_cast<u32x2*>(p); p += a256(G * pl.bigtmp_bytes);
  uint32_t* big = reinterpret_cast<uint32_t*>(p); p += a256(static_cast<size_t>(G) * MSD_BIG_WORDS * 4);
  const size_t part_lds = static_cast<size_t>(2 * MSD_TILE + MSD_N1 + MSD_NC + MSD_NB_MAX) * 4;
  const size_t pipe_lds = static_cast<size_t>(2 * MSD_TILE + MSD_N1 + MSD_NC + 3 * pl.nbs) * 4;
  const size_t bucket_lds = static_cast<size_t>(2 * MSD_CAP + MSD_NF / 2 + 32) * 4;
  const size_t gather_lds = static_cast<size_t>(BB * (BB + 1) + 2 * pl.nbs + pl.nbs / 2) * 4;
  const bool vec = ldo % 4 == 0 && mdg_aligned16(out);
  // (-DMDG_RANK_STAMPS: the phase stamps go to the tail of the big-bucket scratch, its least used slot)
  unsigned long long* stamp_buf = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(bigtmp) + G * pl.bigtmp_bytes) - 2 * 8 * 4096;
  constexpr int part_wgs = 256;                            // persistent partition workgroups of a launch (all outcomes of the group): one per CU
  for (int64_t s0 = 0; s0 < n_outcomes; s0 += G) {
    const unsigned g = static_cast<unsigned>(n_outcomes - s0 < G ? n_outcomes - s0 : G);
    const float* sc = scores + s0 * N * lds;
    float* o = out + s0 * N * ldo;
    uint32_t* fl = flags + s0;
    const uint32_t* tb = tables + s0 * MSD_TABLE_WORDS;
    unsigned pw = static_cast<unsigned>(mdg_cdiv(part_wgs, g));
    pw = pw < 1u ? 1u : (pw > static_cast<unsigned>(n_blocks) ? static_cast<unsigned>(n_blocks) : pw);
    hipLaunchKernelGGL(msd_count_kernel, dim3(static_cast<unsigned>(n_blocks), g), dim3(1024), 0, st, sc, lds, tb, counts, static_cast<int>(N), pl.nbs, src_is_keys);
    hipLaunchKernelGGL(msd_scan_kernel, dim3(static_cast<unsigned>(pl.nbs / 64), g), dim3(1024), 0, st, counts, offs, totals, n_blocks, pl.nbs);
    hipLaunchKernelGGL(msd_base_kernel, dim3(g), dim3(1024), 0, st, totals, bases, fl, big, pl.nbs, M);
    if (pipe_lds <= size_t{160} * 1024 - 256 && pl.nbs <= 1024)
      hipLaunchKernelGGL(msd_partition_pipe_kernel, dim3(pw, g), dim3(1024), pipe_lds, st, sc, lds, tb, offs, bases, part, fl, static_cast<int>(N), M, pl.nbs, n_blocks,
                         src_is_keys, stamp_buf);
    else
      hipLaunchKernelGGL(msd_partition_kernel, dim3(pw, g), dim3(1024), part_lds, st, sc, lds, tb, offs, bases, part, fl, static_cast<int>(N), M, pl.nbs, n_blocks,
                         src_is_keys, stamp_buf);
    unsigned bw = static_cast<unsigned>(mdg_cdiv(part_wgs, g));       // persistent, like the partition
    bw = bw < 1u ? 1u : (bw > static_cast<unsigned>(pl.nbt) ? static_cast<unsigned>(pl.nbt) : bw);
    hipLaunchKernelGGL((msd_bucket_kernel<MSD_NF>), dim3(bw, g), dim3(1024), bucket_lds, st, part, bases, totals, ranked, bigtmp, fl, big, M, pl.nbs, pl.nbt, stamp_buf);
    const dim3 bgrid(static_cast<unsigned>(8 * mdg_cdiv(n_blocks, 8)), g);
    if (vec) hipLaunchKernelGGL(msd_block_gather_kernel<true>, bgrid, dim3(512), gather_lds, st, ranked, offs, counts, bases, o, ldo, static_cast<int>(N), M, pl.nbt,
                                pl.nbs, n_blocks, denom, fl);
    else hipLaunchKernelGGL(msd_block_gather_kernel<false>, bgrid, dim3(512), gather_lds, st, ranked, offs, counts, bases, o, ldo, static_cast<int>(N), M, pl.nbt,
                            pl.nbs, n_blocks, denom, fl);
  }
}

#define MDG_UNPAREN(...) __VA_ARGS__
#define MDG_LSD_LAUNCH(K, TARGS, ...)                                          \
  do {                                                                         \
    if (only) hipLaunchKernelGGL((K<true, MDG_UNPAREN TARGS>), __VA_ARGS__);   \
    else hipLaunchKernelGGL((K<false, MDG_UNPAREN TARGS>), __VA_ARGS__);       \
  } while (0)

template <class C>
static int rank_normalize_impl(const float* scores, int64_t lds, float* out, int64_t ldo, int64_t n_outcomes, int64_t N, void* workspace,
                               size_t workspace_bytes, hipStream_t st, int src_is_keys) {
  constexpr int TPB = C::TPB, TILE = C::TILE;
  const unsigned L = static_cast<unsigned>(n_outcomes);
  const int64_t M = N * (N - 1) / 2;
  const int nblk = static_cast<int>(mdg_cdiv(M, TILE));
  const size_t need = rank_workspace_bytes<C>(n_outcomes, N);
  if (!workspace || workspace_bytes < need || !mdg_aligned16(workspace)) {
    mdg_set_error("mdg_rank_normalize: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return MDG_EWORKSPACE;
  }
  char* ws = static_cast<char*>(workspace);
  // MSD path first; `only` = its per-outcome flags: the LSD kernels below then touch the flagged outcomes alone
  const uint32_t* only = nullptr;
  const MsdPlan pl = msd_plan(n_outcomes, N);
  if (pl.on) {
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msd_partition_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msd_partition_pipe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msd_bucket_kernel<MSD_NF>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msd_block_gather_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msd_block_gather_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      attr_done = true;
    }
    uint32_t* flags = reinterpret_cast<uint32_t*>(ws);
    ws += a256(static_cast<size_t>(n_outcomes + 1) * 4);
    uint32_t* list = reinterpret_cast<uint32_t*>(ws);
    ws += a256(static_cast<size_t>(n_outcomes + 1) * 4);
    char* call_ws = ws;
    ws += pl.call_bytes(n_outcomes);
    (void)hipMemsetAsync(flags, 0, static_cast<size_t>(n_outcomes) * 4, st);
    msd_run(pl, scores, lds, out, ldo, n_outcomes, N, call_ws, ws, flags, st, src_is_keys);
    hipLaunchKernelGGL(msd_flag_list_kernel, dim3(1), dim3(1024), 0, st, flags, list, static_cast<int>(n_outcomes));
    only = list;
  }
  const size_t kb = a256(static_cast<size_t>(n_outcomes) * M * 4);
  uint32_t* k0 = reinterpret_cast<uint32_t*>(ws);          // k0 | p0 adjacent: together they hold the last pass's (rank, position) pairs
  uint32_t* p0 = reinterpret_cast<uint32_t*>(ws + kb);
  uint32_t* k1 = reinterpret_cast<uint32_t*>(ws + 2 * kb);
  uint32_t* p1 = reinterpret_cast<uint32_t*>(ws + 3 * kb);
  uint32_t* hist = reinterpret_cast<uint32_t*>(ws + 4 * kb);
  const int nblk3 = static_cast<int>(mdg_cdiv(M, CfgStd::TILE));
  const size_t hb = a256(static_cast<size_t>(n_outcomes) * 256 * nblk3 * 4);       // the per-tile digit table
  uint32_t* fill = reinterpret_cast<uint32_t*>(ws + 4 * kb + hb);
  const int64_t n_blocks = rank_blocks_of(N);
  static MdgEnvInt direct_sw{"MDG_RANKS_DIRECT", 0};        // test hook: 1 = the last pass stores the ranks one by one (what N > 16256 takes) at any N
  // the blocked last pass on 8192-key tiles whatever the other passes use: two workgroups per CU there beat one of 16384 keys
  // (48 against 63 us per 4096^2 outcome); its histogram and scan use the same tiling
  const size_t blocks_lds = static_cast<size_t>(2 * CfgStd::TILE + 2 * n_blocks) * 4;
  const bool blocked = n_blocks <= MAX_BLOCKS && !direct_sw.get();
  // behind the MSD path: two rows of workgroups walk the list of flagged outcomes (usually empty)
  const unsigned Ly = only ? (L < 2u ? L : 2u) : L;
  const dim3 grid3(static_cast<unsigned>(nblk3), Ly);
  const double denom = static_cast<double>(N) * static_cast<double>(N - 1) / 2.0;
  const dim3 grid(static_cast<unsigned>(nblk), Ly);
  MDG_LSD_LAUNCH(extract_keys_kernel, (C), grid, dim3(TPB), 0, st, scores, lds, k0, hist, static_cast<int>(N), M, nblk, src_is_keys, only);
  for (int pass = 0; pass < 4; ++pass) {
    uint32_t* kin = (pass & 1) ? k1 : k0;
    uint32_t* kout = (pass & 1) ? k0 : k1;
    uint32_t* pin = (pass & 1) ? p1 : p0;
    uint32_t* pout = (pass & 1) ? p0 : p1;
    const bool std3 = pass == 3 && blocked;
    // keys narrow as the passes go: pass 0 and 1 read all 32 bits, pass 1 leaves the upper 16, pass 2 the upper 8 (the bits below
    // the current digit are sorted already): 15 of 96 bytes per key less to move
    typedef uint16_t u16;
    typedef uint8_t u8;
    if (std3) MDG_LSD_LAUNCH(histogram_kernel, (CfgStd, u8), grid3, dim3(CfgStd::TPB), 0, st, reinterpret_cast<const u8*>(kin), hist, M, nblk3, 0, only);
    else if (pass == 3) MDG_LSD_LAUNCH(histogram_kernel, (C, u8), grid, dim3(TPB), 0, st, reinterpret_cast<const u8*>(kin), hist, M, nblk, 0, only);
    else if (pass == 2) MDG_LSD_LAUNCH(histogram_kernel, (C, u16), grid, dim3(TPB), 0, st, reinterpret_cast<const u16*>(kin), hist, M, nblk, 0, only);
    else if (pass == 1) MDG_LSD_LAUNCH(histogram_kernel, (C, uint32_t), grid, dim3(TPB), 0, st, kin, hist, M, nblk, 8, only);
    if (only) hipLaunchKernelGGL(scan_kernel<true>, dim3(Ly), dim3(1024), 0, st, hist, std3 ? nblk3 : nblk, only);
    else hipLaunchKernelGGL(scan_kernel<false>, dim3(Ly), dim3(1024), 0, st, hist, std3 ? nblk3 : nblk, only);
    if (pass == 0)
      MDG_LSD_LAUNCH(scatter_kernel, (C, true, false), grid, dim3(TPB), 0, st, kin, pin, kout, pout, hist, out, ldo, static_cast<int>(N), M, nblk, 0, denom, only);
    else if (pass == 1)
      MDG_LSD_LAUNCH(scatter_kernel, (C, false, false, uint32_t, u16), grid, dim3(TPB), 0, st, kin, pin, reinterpret_cast<u16*>(kout), pout, hist, out, ldo,
                         static_cast<int>(N), M, nblk, 8, denom, only);
    else if (pass == 2)
      MDG_LSD_LAUNCH(scatter_kernel, (C, false, false, u16, u8), grid, dim3(TPB), 0, st, reinterpret_cast<const u16*>(kin), pin, reinterpret_cast<u8*>(kout), pout,
                         hist, out, ldo, static_cast<int>(N), M, nblk, 0, denom, only);
    else if (blocked) {
      (void)hipMemsetAsync(fill, 0, static_cast<size_t>(n_outcomes) * n_blocks * 4 * FILL_STRIDE, st);
      u32x2* pairs = reinterpret_cast<u32x2*>(k0);            // pass 3 reads k1 / p1
      MDG_LSD_LAUNCH(rank_blocks_kernel, (CfgStd, u8), grid3, dim3(CfgStd::TPB), blocks_lds, st, reinterpret_cast<const u8*>(kin), pin, hist, pairs, fill,
                         static_cast<int>(N), M, nblk3, static_cast<int>(n_blocks), only);
      const dim3 bgrid(static_cast<unsigned>(n_blocks), Ly);
      if (ldo % 4 == 0 && mdg_aligned16(out))
        MDG_LSD_LAUNCH(rank_block_write_kernel, (true), bgrid, dim3(512), 0, st, pairs, out, ldo, static_cast<int>(N), M, static_cast<int>(n_blocks), denom, only);
      else
        MDG_LSD_LAUNCH(rank_block_write_kernel, (false), bgrid, dim3(512), 0, st, pairs, out, ldo, static_cast<int>(N), M, static_cast<int>(n_blocks), denom, only);
    } else {
      MDG_LSD_LAUNCH(scatter_kernel, (C, false, true, u8, u8), grid, dim3(TPB), 0, st, reinterpret_cast<const u8*>(kin), pin, reinterpret_cast<u8*>(kout), pout, hist,
                         out, ldo, static_cast<int>(N), M, nblk, 0, denom, only);
    }
  }
  MDG_CHECK_LAUNCH("mdg_rank_normalize");
  return MDG_OK;
}

static int rank_normalize_entry(const float* src, int64_t lds, float* out, int64_t ldo, int64_t n_outcomes, int64_t N, void* workspace,
                                size_t workspace_bytes, void* stream, int src_is_keys) {
  MDG_CHECK_ARG(lds >= N && ldo >= N, "mdg_rank_normalize: row pitches must be >= N");
  MDG_CHECK_ARG(n_outcomes >= 0 && N >= 0 && N <= 65535 && n_outcomes <= 65535, "mdg_rank_normalize: bad sizes (outcomes per call and N <= 65535)");
  if (n_outcomes == 0 || N == 0) return MDG_OK;
  MDG_CHECK_ARG(src && out, "mdg_rank_normalize: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(zero_diag_kernel, dim3(static_cast<unsigned>(mdg_cdiv(N, 256)), static_cast<unsigned>(n_outcomes)), dim3(256), 0, st, out, ldo, static_cast<int>(N));
  if (N < 2) { MDG_CHECK_LAUNCH("mdg_rank_normalize"); return MDG_OK; }
  return rank_use_big(N) ? rank_normalize_impl<CfgBig>(src, lds, out, ldo, n_outcomes, N, workspace, workspace_bytes, st, src_is_keys)
                         : rank_normalize_impl<CfgStd>(src, lds, out, ldo, n_outcomes, N, workspace, workspace_bytes, st, src_is_keys);
}

extern "C" int mdg_rank_normalize_ld(const float* scores, int64_t lds, float* out, int64_t ldo, int64_t n_outcomes, int64_t N, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  return rank_normalize_entry(scores, lds, out, ldo, n_outcomes, N, workspace, workspace_bytes, stream, 0);
}

// the strict lower triangle of `keys` [n_outcomes, N, ldk] holds the order keys of the scores (mdg_bilinear_allpairs_ld with
// MDG_EPI_TRIKEYS); the rest of the tensor is never read.  out may be the same memory (the keys leave in the first kernel).
extern "C" int mdg_rank_normalize_keys_ld(const uint32_t* keys, int64_t ldk, float* out, int64_t ldo, int64_t n_outcomes, int64_t N, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  return rank_normalize_entry(reinterpret_cast<const float*>(keys), ldk, out, ldo, n_outcomes, N, workspace, workspace_bytes, stream, 1);
}

extern "C" int mdg_gmean(const float* const* inputs_host, int K, float* out, int64_t n, void* stream) {
  MDG_CHECK_ARG(inputs_host && out && K >= 1 && K <= 8 && n >= 0, "mdg_gmean: 1 <= K <= 8 tensors of n floats");
  if (n == 0) return MDG_OK;
  GmeanArgs a{};
  a.K = K;
  for (int k = 0; k < K; ++k) {
    MDG_CHECK_ARG(inputs_host[k] && mdg_aligned16(inputs_host[k]), "mdg_gmean: input %d null or not 16-byte aligned", k);
    a.in[k] = inputs_host[k];
  }
  MDG_CHECK_ARG(mdg_aligned16(out), "mdg_gmean: out not 16-byte aligned");
  hipLaunchKernelGGL(gmean_kernel, dim3(static_cast<unsigned>(mdg_cdiv(n / 4 > 0 ? n / 4 : 1, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), a, out, n / 4, n);
  MDG_CHECK_LAUNCH("mdg_gmean");
  return MDG_OK;
}
