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

// keys of the strict lower triangle in p order + the tile histogram of the lowest digit
template <class C>
__global__ __launch_bounds__(C::TPB) void extract_keys_kernel(const float* __restrict__ scores, int64_t lds, uint32_t* __restrict__ keys,
                                                           uint32_t* __restrict__ hist, uint32_t* __restrict__ ghist, int N, int64_t M, int nblk,
                                                           int src_is_keys) {
  MDG_RANK_USING(C);
  __shared__ uint32_t cnt[WAVES][256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < WAVES * 256; i += TPB) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const int64_t seg = blockIdx.y, base = static_cast<int64_t>(blockIdx.x) * TILE;
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
  if (ghist) {
    // look-back passes: the outcome's digit totals of ALL FOUR passes (a digit histogram does not depend on the order of the keys):
    // wave-private counters, one atomic per tile, pass and digit.  ghist[(pass * outcomes + seg) * 256 + d]
    const int64_t n_seg = gridDim.y;
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
      wave_digit_counts(key, 8 * pass, cnt[wave]);
      __syncthreads();
      if (tid < 256) {
        uint32_t c = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { c += cnt[w][tid]; cnt[w][tid] = 0; }
        if (tid == 255 && base + TILE > M) c -= static_cast<uint32_t>(base + TILE - M);
        if (c) atomicAdd(&ghist[(pass * n_seg + seg) * 256 + tid], c);
      }
      __syncthreads();
    }
  } else {
    wave_digit_counts(key, 0, cnt[wave]);
    __syncthreads();
    store_tile_histogram<C>(cnt, hist, seg, nblk, base, M);
  }
}

// KT: what the previous pass left of the key -- the bits below the current digit are sorted already and are not carried along:
// pass 1 writes the upper 16 bits (uint16_t), pass 2 the upper 8 (uint8_t); `shift` = position of the current digit inside a KT
template <class C, class KT>
__global__ __launch_bounds__(C::TPB) void histogram_kernel(const KT* __restrict__ keys, uint32_t* __restrict__ hist, int64_t M,
                                                           int nblk, int shift) {
  MDG_RANK_USING(C);
  __shared__ uint32_t cnt[WAVES][256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < WAVES * 256; i += TPB) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const int64_t seg = blockIdx.y, base = static_cast<int64_t>(blockIdx.x) * TILE;
  uint32_t key[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int64_t p = base + wave * WSPAN + k * 64 + lane;
    key[k] = p < M ? static_cast<uint32_t>(keys[seg * M + p]) : 0xFFFFFFFFu;
  }
  wave_digit_counts(key, shift, cnt[wave]);
  __syncthreads();
  store_tile_histogram<C>(cnt, hist, seg, nblk, base, M);
}

// ---- tile offsets without the histogram / scan launches: decoupled look-back ------------------------------------------------
// A scatter needs, per digit d, the number of keys with digit d in the tiles before it.  Instead of a histogram kernel + a scan
// kernel per pass (a fourth of the bytes of a pass, two launches), every tile publishes its own digit counts in status[tile][d]
// (flag AGGREGATE), walks back over its predecessors adding their counts until it meets one that already knows its inclusive
// prefix (flag INCLUSIVE), then publishes its own inclusive prefix.  One 32-bit word carries flag and value, written and read
// with relaxed agent-scope atomics (sc1: L2-coherent, no fence needed for a self-contained word: MI355X_MICROARCH.md, granules).
// A tile waits only for tiles with a LOWER index of the same outcome, which were dispatched before it (blockIdx.x fastest) and wait
// only for still lower ones; polls are bounded (a lost word must not hang the card: the result is then wrong, loudly, in the tests).
// The digit totals of a pass (its exclusive scan over the 256 digits = where each digit's run starts) come from a global
// histogram the PREVIOUS pass (the key extraction for pass 0) accumulates with one atomic per tile and digit.
constexpr uint32_t LB_AGG = 1u << 30, LB_INC = 2u << 30, LB_VAL = (1u << 30) - 1;
constexpr unsigned LB_SPIN_LIMIT = 1u << 22;

// step 1, as early as the tile knows its digit counts (a cheap counting sweep in front of the ranking): successors can add them
__device__ __forceinline__ void lookback_publish(uint32_t* __restrict__ status_seg, int blk, int d, uint32_t local) {
  __hip_atomic_store(status_seg + static_cast<int64_t>(blk) * 256 + d, local | (blk == 0 ? LB_INC : LB_AGG), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// step 2, as late as the offsets are needed (after the ranking): by then the predecessors published long ago and most of them
// already hold their inclusive prefix, so the walk is one or two loads deep instead of as deep as the set of co-resident tiles
__device__ __forceinline__ uint32_t lookback_walk(uint32_t* __restrict__ status_seg, int blk, int d, uint32_t local) {
  if (blk == 0) return 0;
  uint32_t prefix = 0;
  for (int t = blk - 1; t >= 0; --t) {
    const uint32_t* theirs = status_seg + static_cast<int64_t>(t) * 256 + d;
    uint32_t v = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while ((v >> 30) == 0 && ++spins < LB_SPIN_LIMIT) {
      __builtin_amdgcn_s_sleep(2);
      v = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    prefix += v & LB_VAL;
    if ((v >> 30) == 2u) break;
  }
  __hip_atomic_store(status_seg + static_cast<int64_t>(blk) * 256 + d, (prefix + local) | LB_INC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return prefix;
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

// exclusive scan of the 256*nblk counters of one outcome, in place; one workgroup per outcome, coalesced: every wave owns a
// contiguous range and walks it 256 counters (16 bytes per lane) at a time, the next group's load issued before the scan of this one
__global__ __launch_bounds__(1024) void scan_kernel(uint32_t* __restrict__ hist, int nblk) {
  __shared__ uint32_t part[16];
  u32x4* h = reinterpret_cast<u32x4*>(hist + static_cast<int64_t>(blockIdx.x) * 256 * nblk);
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
}

// KIN / KOUT: key representation read / written (see histogram_kernel): the output drops the digit this pass sorts by when
// KOUT is narrower than KIN (`shift` must then be 0 within KIN ... the current digit is its low byte, or bits 8..15 for pass 1)
template <class C, bool FIRST, bool LAST, class KIN = uint32_t, class KOUT = uint32_t>
__global__ __launch_bounds__(C::TPB) void scatter_kernel(const KIN* __restrict__ keys_in, const uint32_t* __restrict__ pay_in,
                                                      KOUT* __restrict__ keys_out, uint32_t* __restrict__ pay_out,
                                                      const uint32_t* __restrict__ offsets, float* __restrict__ out, int64_t ldo, int N,
                                                      int64_t M, int nblk, int shift, double denom, uint32_t* __restrict__ status,
                                                      const uint32_t* __restrict__ ghist) {
  // offsets != null: tile offsets from the histogram + scan launches; else look-back (status, this pass's digit totals ghist)
  MDG_RANK_USING(C);
  __shared__ uint32_t cnt[WAVES][256];     // per-wave digit counts, then their exclusive prefix over the waves
  __shared__ uint32_t dstart[256];         // first slot of digit d in the sorted tile
  __shared__ uint32_t gofs[256];           // global position of slot 0 of digit d's run, minus dstart[d]
  __shared__ uint32_t wsum[4], wsum2[4];
  __shared__ uint32_t skey[TILE];
  __shared__ uint32_t spay[TILE];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < WAVES * 256; i += TPB) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const int64_t seg = blockIdx.y, base = static_cast<int64_t>(blockIdx.x) * TILE;
  uint32_t key[ITEMS], pay[ITEMS], rk[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int64_t p = base + wave * WSPAN + k * 64 + lane;
    const bool valid = p < M;
    key[k] = valid ? static_cast<uint32_t>(keys_in[seg * M + p]) : 0xFFFFFFFFu;   // padding: digit 255 in every pass, behind every real key of the tile
    pay[k] = valid ? (FIRST ? static_cast<uint32_t>(p) : pay_in[seg * M + p]) : NO_PAY;
  }
  uint32_t* const status_seg = offsets ? nullptr : status + seg * static_cast<int64_t>(nblk) * 256;
  if (!offsets) {                           // (uniform) look-back: count first, publish, count again with ranks
    wave_digit_counts(key, shift, cnt[wave]);
    __syncthreads();
    if (tid < 256) {
      uint32_t c = 0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) c += cnt[w][tid];
      if (tid == 255 && base + TILE > M) c -= static_cast<uint32_t>(base + TILE - M);
      lookback_publish(status_seg, static_cast<int>(blockIdx.x), tid, c);
    }
    __syncthreads();
    for (int i = tid; i < WAVES * 256; i += TPB) (&cnt[0][0])[i] = 0;
    __syncthreads();
  }
  wave_digit_ranks<true, ITEMS>(key, shift, cnt[wave], lane, rk);
  __syncthreads();
  uint32_t run = 0, gh = 0, gh_inc = 0;    // thread d < 256: the tile's count of digit d; the outcome's count of digit d
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
    if (!offsets) {
      gh = ghist[seg * 256 + tid];
      gh_inc = wave_inclusive(gh, lane);
      if (lane == 63) wsum2[wave] = gh_inc;
    }
  }
  __syncthreads();
  if (tid < 256) {
    uint32_t add = 0, add2 = 0;
    for (int w = 0; w < wave; ++w) { add += wsum[w]; add2 += wsum2[w]; }
    const uint32_t ds = dstart[tid] + add;
    dstart[tid] = ds;
    if (offsets) {
      gofs[tid] = offsets[(seg * 256 + tid) * nblk + blockIdx.x] - ds;
    } else {
      const uint32_t pad = (tid == 255 && base + TILE > M) ? static_cast<uint32_t>(base + TILE - M) : 0u;
      const uint32_t before = lookback_walk(status_seg, static_cast<int>(blockIdx.x), tid, run - pad);
      gofs[tid] = (gh_inc - gh + add2) + before - ds;
    }
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
      int i, j;
      tri_decode(pp, i, j);
      const float v = static_cast<float>(static_cast<double>(g + 1u) / denom);
      o[static_cast<int64_t>(i) * ldo + j] = v;
      o[static_cast<int64_t>(j) * ldo + i] = v;
    } else {
      keys_out[seg * M + g] = static_cast<KOUT>(kk >> (8 * (static_cast<int>(sizeof(KIN)) - static_cast<int>(sizeof(KOUT)))));
      pay_out[seg * M + g] = pp;
    }
  }
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
constexpr int MAX_BLOCKS = 8192;           // LDS budget of the sorting kernel: N <= 16256; larger N take the direct scatter

// first pair slot of block (bi, bj), bj <= bi: all rows above block row bi, then the blocks left of it
__device__ __forceinline__ uint32_t block_base(int bi, int bj, int N) {
  const int64_t r0 = static_cast<int64_t>(bi) * BB;
  const int rcount = N - r0 < BB ? static_cast<int>(N - r0) : BB;
  return static_cast<uint32_t>(r0 * (r0 - 1) / 2 + static_cast<int64_t>(bj) * rcount * BB);
}

template <class C, class KIN>
__global__ __launch_bounds__(C::TPB) void rank_blocks_kernel(const KIN* __restrict__ keys_in, const uint32_t* __restrict__ pay_in,
                                                          const uint32_t* __restrict__ offsets, u32x2* __restrict__ pairs,
                                                          uint32_t* __restrict__ fill, int N, int64_t M, int nblk, int n_blocks,
                                                          uint32_t* __restrict__ status, const uint32_t* __restrict__ ghist) {
  MDG_RANK_USING(C);
  extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];            // [TILE] pairs (u32x2) | bcnt[n_blocks] | bdst[n_blocks]
  __shared__ uint32_t cnt[WAVES][256];
  __shared__ uint32_t gbase[256];
  __shared__ uint32_t wsum[WAVES], wsum2[4];
  u32x2* spair = reinterpret_cast<u32x2*>(dyn);
  uint32_t* bcnt = dyn + 2 * TILE;
  uint32_t* bdst = bcnt + n_blocks;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < WAVES * 256; i += TPB) (&cnt[0][0])[i] = 0;
  for (int i = tid; i < n_blocks; i += TPB) bcnt[i] = 0;
  __syncthreads();
  const int64_t seg = blockIdx.y, base = static_cast<int64_t>(blockIdx.x) * TILE;
  uint32_t key[ITEMS], pay[ITEMS], rk[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int64_t p = base + wave * WSPAN + k * 64 + lane;
    const bool valid = p < M;
    key[k] = valid ? static_cast<uint32_t>(keys_in[seg * M + p]) : 0xFFFFFFFFu;
    pay[k] = valid ? pay_in[seg * M + p] : NO_PAY;
  }
  constexpr int SH = 8 * (static_cast<int>(sizeof(KIN)) - 1);          // the last digit = the top byte of what is left of the key
  uint32_t* const status_seg = offsets ? nullptr : status + seg * static_cast<int64_t>(nblk) * 256;
  if (!offsets) {                           // (uniform) look-back: see scatter_kernel
    wave_digit_counts(key, SH, cnt[wave]);
    __syncthreads();
    if (tid < 256) {
      uint32_t c = 0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) c += cnt[w][tid];
      if (tid == 255 && base + TILE > M) c -= static_cast<uint32_t>(base + TILE - M);
      lookback_publish(status_seg, static_cast<int>(blockIdx.x), tid, c);
    }
    __syncthreads();
    for (int i = tid; i < WAVES * 256; i += TPB) (&cnt[0][0])[i] = 0;
    __syncthreads();
  }
  wave_digit_ranks<true, ITEMS>(key, SH, cnt[wave], lane, rk);
  __syncthreads();
  {
    uint32_t run = 0, gh = 0, gh_inc = 0;
    if (tid < 256) {
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        const uint32_t c = cnt[w][tid];
        cnt[w][tid] = run;
        run += c;
      }
      if (offsets) {
        gbase[tid] = offsets[(seg * 256 + tid) * nblk + blockIdx.x];
      } else {
        gh = ghist[seg * 256 + tid];
        gh_inc = wave_inclusive(gh, lane);
        if (lane == 63) wsum2[wave] = gh_inc;
      }
    }
    if (!offsets) {                                        // (uniform) look-back: see scatter_kernel
      __syncthreads();
      if (tid < 256) {
        uint32_t add2 = 0;
        for (int w = 0; w < wave; ++w) add2 += wsum2[w];
        const uint32_t pad = (tid == 255 && base + TILE > M) ? static_cast<uint32_t>(base + TILE - M) : 0u;
        gbase[tid] = (gh_inc - gh + add2) + lookback_walk(status_seg, static_cast<int>(blockIdx.x), tid, run - pad);
      }
    }
  }
  __syncthreads();
  uint32_t blk[ITEMS], slot[ITEMS];                       // key[] is reused for g, pay[] for the position inside the block
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    blk[k] = 0xFFFFFFFFu;
    if (pay[k] == NO_PAY) continue;
    const uint32_t dg = (key[k] >> SH) & 255u;
    key[k] = gbase[dg] + cnt[wave][dg] + rk[k];
    int i, j;
    tri_decode(pay[k], i, j);
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
        bdst[t] = block_base(bi, bj, N) + atomicAdd(&fill[seg * n_blocks + t], c) - run;
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
}

template <bool VEC>
__global__ __launch_bounds__(512) void rank_block_write_kernel(const u32x2* __restrict__ pairs, float* __restrict__ out, int64_t ldo, int N,
                                                               int64_t M, int n_blocks, double denom) {
  constexpr int TPB = 512;
  __shared__ float tile[BB][BB + 1];
  const int t = blockIdx.x, tid = threadIdx.x;
  const int64_t seg = blockIdx.y;
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
  if (diag) return;
  // rows of the mirrored block out[j, i]
  for (int c = rr; c < BB; c += TPB / 32) {
    float* row = o + static_cast<int64_t>(c0 + c) * ldo + r0;
    if (VEC && 4 * q + 3 < rcount) *reinterpret_cast<f32x4*>(row + 4 * q) = f32x4{tile[4 * q][c], tile[4 * q + 1][c], tile[4 * q + 2][c], tile[4 * q + 3][c]};
    else
      for (int e = 0; e < 4; ++e)
        if (4 * q + e < rcount) row[4 * q + e] = tile[4 * q + e][c];
  }
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
// 16384-key tiles for the first three passes (MDG_RANKS_TILE=8192: the smaller shape everywhere)
static bool rank_use_big(int64_t N) {
  static MdgEnvInt tile_sw{"MDG_RANKS_TILE", 0};           // 8192 / 16384: force a tile shape (diagnostics)
  (void)N;
  return tile_sw.get() != 8192;
}

template <class C>
static size_t rank_workspace_bytes(int64_t n_outcomes, int64_t N) {
  const size_t M = static_cast<size_t>(N) * (N - 1) / 2;
  const size_t nblk = (M + CfgStd::TILE - 1) / CfgStd::TILE;          // the last pass always runs on 8192-key tiles (the finer table)
  // keys / payloads x 2 | per-tile digit table (histogram path: one; look-back: one status table per pass) | block fill counters | digit totals
  return 4 * a256(static_cast<size_t>(n_outcomes) * M * 4) + 4 * a256(static_cast<size_t>(n_outcomes) * 256 * nblk * 4) +
         a256(static_cast<size_t>(n_outcomes) * static_cast<size_t>(rank_blocks_of(N)) * 4) + a256(static_cast<size_t>(n_outcomes) * 4 * 256 * 4);
}

extern "C" size_t mdg_rank_normalize_workspace_bytes(int64_t n_outcomes, int64_t N) {
  if (n_outcomes <= 0 || N < 2) return 0;
  return rank_use_big(N) ? rank_workspace_bytes<CfgBig>(n_outcomes, N) : rank_workspace_bytes<CfgStd>(n_outcomes, N);
}

extern "C" int mdg_rank_normalize(const float* scores, float* out, int64_t n_outcomes, int64_t N, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  return mdg_rank_normalize_ld(scores, N, out, N, n_outcomes, N, workspace, workspace_bytes, stream);
}

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
  const size_t kb = a256(static_cast<size_t>(n_outcomes) * M * 4);
  uint32_t* k0 = reinterpret_cast<uint32_t*>(ws);          // k0 | p0 adjacent: together they hold the last pass's (rank, position) pairs
  uint32_t* p0 = reinterpret_cast<uint32_t*>(ws + kb);
  uint32_t* k1 = reinterpret_cast<uint32_t*>(ws + 2 * kb);
  uint32_t* p1 = reinterpret_cast<uint32_t*>(ws + 3 * kb);
  uint32_t* hist = reinterpret_cast<uint32_t*>(ws + 4 * kb);
  const int nblk3 = static_cast<int>(mdg_cdiv(M, CfgStd::TILE));
  const size_t hb = a256(static_cast<size_t>(n_outcomes) * 256 * nblk3 * 4);       // one per-tile digit table
  uint32_t* fill = reinterpret_cast<uint32_t*>(ws + 4 * kb + 4 * hb);
  const size_t fb = a256(static_cast<size_t>(n_outcomes) * static_cast<size_t>(rank_blocks_of(N)) * 4);
  uint32_t* ghist = reinterpret_cast<uint32_t*>(ws + 4 * kb + 4 * hb + fb);        // [4 passes][outcomes][256]
  const int64_t n_blocks = rank_blocks_of(N);
  static MdgEnvInt direct_sw{"MDG_RANKS_DIRECT", 0};        // 1: the last pass stores the ranks one by one (the large-N path) at any N
  // Tile offsets from histogram + scan launches (default) or by decoupled look-back (1).  Measured on 4096^2 outcomes: the
  // look-back saves the 37 us of histogram / scan launches per outcome and gives 35 us back inside the scatters (every walk step is
  // a cross-XCD sc1 load, ~2 us, in front of the tile's write phase; 4 digit histograms in the extraction): 0.26 ms either way.
  static MdgEnvInt lb_sw{"MDG_RANKS_LOOKBACK", 0};
  const bool lb = lb_sw.get() != 0;
  // the blocked last pass on 8192-key tiles whatever the other passes use: two workgroups per CU there beat one of 16384 keys
  // (48 against 63 us per 4096^2 outcome); its histogram and scan use the same tiling
  const size_t blocks_lds = static_cast<size_t>(2 * CfgStd::TILE + 2 * n_blocks) * 4;
  const bool blocked = n_blocks <= MAX_BLOCKS && !direct_sw.get();
  const dim3 grid3(static_cast<unsigned>(nblk3), L);
  const double denom = static_cast<double>(N) * static_cast<double>(N - 1) / 2.0;
  const dim3 grid(static_cast<unsigned>(nblk), L);
  const auto status_of = [&](int pass) { return reinterpret_cast<uint32_t*>(ws + 4 * kb + pass * hb); };
  const auto ghist_of = [&](int pass) { return ghist + static_cast<size_t>(pass) * n_outcomes * 256; };
  if (lb) (void)hipMemsetAsync(ws + 4 * kb, 0, 4 * hb + fb + a256(static_cast<size_t>(n_outcomes) * 4 * 256 * 4), st);   // status tables, block fill counters, digit totals
  hipLaunchKernelGGL(extract_keys_kernel<C>, grid, dim3(TPB), 0, st, scores, lds, k0, hist, lb ? ghist : nullptr, static_cast<int>(N), M, nblk, src_is_keys);
  for (int pass = 0; pass < 4; ++pass) {
    uint32_t* kin = (pass & 1) ? k1 : k0;
    uint32_t* kout = (pass & 1) ? k0 : k1;
    uint32_t* pin = (pass & 1) ? p1 : p0;
    uint32_t* pout = (pass & 1) ? p0 : p1;
    const bool std3 = pass == 3 && blocked;
    const uint32_t* offs = lb ? nullptr : hist;
    uint32_t* stat = lb ? status_of(pass) : nullptr;
    const uint32_t* gcur = lb ? ghist_of(pass) : nullptr;
    // keys narrow as the passes go: pass 0 and 1 read all 32 bits, pass 1 leaves the upper 16, pass 2 the upper 8 (the bits below
    // the current digit are sorted already): 15 of 96 bytes per key less to move
    typedef uint16_t u16;
    typedef uint8_t u8;
    if (!lb) {
      if (std3) hipLaunchKernelGGL((histogram_kernel<CfgStd, u8>), grid3, dim3(CfgStd::TPB), 0, st, reinterpret_cast<const u8*>(kin), hist, M, nblk3, 0);
      else if (pass == 3) hipLaunchKernelGGL((histogram_kernel<C, u8>), grid, dim3(TPB), 0, st, reinterpret_cast<const u8*>(kin), hist, M, nblk, 0);
      else if (pass == 2) hipLaunchKernelGGL((histogram_kernel<C, u16>), grid, dim3(TPB), 0, st, reinterpret_cast<const u16*>(kin), hist, M, nblk, 0);
      else if (pass == 1) hipLaunchKernelGGL((histogram_kernel<C, uint32_t>), grid, dim3(TPB), 0, st, kin, hist, M, nblk, 8);
      hipLaunchKernelGGL(scan_kernel, dim3(L), dim3(1024), 0, st, hist, std3 ? nblk3 : nblk);
    }
    if (pass == 0)
      hipLaunchKernelGGL((scatter_kernel<C, true, false>), grid, dim3(TPB), 0, st, kin, pin, kout, pout, offs, out, ldo, static_cast<int>(N), M, nblk, 0, denom, stat, gcur);
    else if (pass == 1)
      hipLaunchKernelGGL((scatter_kernel<C, false, false, uint32_t, u16>), grid, dim3(TPB), 0, st, kin, pin, reinterpret_cast<u16*>(kout), pout, offs, out, ldo,
                         static_cast<int>(N), M, nblk, 8, denom, stat, gcur);
    else if (pass == 2)
      hipLaunchKernelGGL((scatter_kernel<C, false, false, u16, u8>), grid, dim3(TPB), 0, st, reinterpret_cast<const u16*>(kin), pin, reinterpret_cast<u8*>(kout), pout,
                         offs, out, ldo, static_cast<int>(N), M, nblk, 0, denom, stat, gcur);
    else if (blocked) {
      if (!lb) (void)hipMemsetAsync(fill, 0, static_cast<size_t>(n_outcomes) * n_blocks * 4, st);
      u32x2* pairs = reinterpret_cast<u32x2*>(k0);            // pass 3 reads k1 / p1
      hipLaunchKernelGGL((rank_blocks_kernel<CfgStd, u8>), grid3, dim3(CfgStd::TPB), blocks_lds, st, reinterpret_cast<const u8*>(kin), pin, offs, pairs, fill,
                         static_cast<int>(N), M, nblk3, static_cast<int>(n_blocks), stat, gcur);
      const dim3 bgrid(static_cast<unsigned>(n_blocks), L);
      if (ldo % 4 == 0 && mdg_aligned16(out))
        hipLaunchKernelGGL(rank_block_write_kernel<true>, bgrid, dim3(512), 0, st, pairs, out, ldo, static_cast<int>(N), M, static_cast<int>(n_blocks), denom);
      else
        hipLaunchKernelGGL(rank_block_write_kernel<false>, bgrid, dim3(512), 0, st, pairs, out, ldo, static_cast<int>(N), M, static_cast<int>(n_blocks), denom);
    } else {
      hipLaunchKernelGGL((scatter_kernel<C, false, true, u8, u8>), grid, dim3(TPB), 0, st, reinterpret_cast<const u8*>(kin), pin, reinterpret_cast<u8*>(kout), pout, offs,
                         out, ldo, static_cast<int>(N), M, nblk, 0, denom, stat, gcur);
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
