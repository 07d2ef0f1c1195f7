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

// keys of the strict lower triangle in p order + the tile histogram of the lowest digit
template <class C>
__global__ __launch_bounds__(C::TPB) void extract_keys_kernel(const float* __restrict__ scores, int64_t lds, uint32_t* __restrict__ keys,
                                                           uint32_t* __restrict__ hist, uint32_t* __restrict__ ghist, int N, int64_t M, int nblk,
                                                           int src_is_keys, const uint32_t* __restrict__ only) {
  MDG_RANK_USING(C);
  __shared__ uint32_t cnt[WAVES][256];
  if (only && !only[blockIdx.y]) return;      // LSD fallback behind the MSD fast path: flagged outcomes only
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
                                                           int nblk, int shift, const uint32_t* __restrict__ only) {
  MDG_RANK_USING(C);
  __shared__ uint32_t cnt[WAVES][256];
  if (only && !only[blockIdx.y]) return;
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
__global__ __launch_bounds__(1024) void scan_kernel(uint32_t* __restrict__ hist, int nblk, const uint32_t* __restrict__ only) {
  __shared__ uint32_t part[16];
  if (only && !only[blockIdx.x]) return;
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
                                                      const uint32_t* __restrict__ ghist, const uint32_t* __restrict__ only) {
  // offsets != null: tile offsets from the histogram + scan launches; else look-back (status, this pass's digit totals ghist)
  MDG_RANK_USING(C);
  if (only && !only[blockIdx.y]) return;
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
      const int i = static_cast<int>(pp >> 16), j = static_cast<int>(pp & 0xFFFFu);
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
                                                          uint32_t* __restrict__ status, const uint32_t* __restrict__ ghist,
                                                          const uint32_t* __restrict__ only) {
  MDG_RANK_USING(C);
  if (only && !only[blockIdx.y]) return;
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

// ================================================================================================================================
// OPT-IN path (round 4, MDG_RANKS_MSD=1; OFF by default): ONE adaptive MSD partition + an in-LDS counting sort per bucket, instead of
// four global LSD passes.  Measured on one MI355X (profiles/r04_rank_msd_experiment.txt): 184 us per 4096^2 outcome on
// i.i.d. scores against 243 us for the LSD sort -- and SLOWER than it on the bench's own score tensor, because nearly every outcome is
// handed back: a drug's scores sit around the drug's own level, so (i) they fill two or three of the fixed top-bit coarse bins
// unevenly and (ii) a bucket draws its keys from the tiles of a few dozen rows, which overflows the per-tile-shard segments that
// keep the global atomics uncontended.  Capacity-based layouts do not survive row-structured scores; the exact (count + scan)
// layout of the LSD passes does.  Kept as a tested, bit-exact path for smooth i.i.d.-like score tensors.
//
//   msd_hist_kernel      read the keys once: histogram of their top 14 bits per outcome (16 384 coarse bins: sign, exponent, 5
//                        mantissa bits -- a bin spans 1/32 of a binade, over which any smooth score density is flat)
//   msd_table_kernel     per outcome: prefix of the coarse counts; a coarse bin with more than Q/8 keys is cut into 2^lg equal
//                        sub-ranges of its low bits.  bucket(key) = (prefix[c] + sub(key) * (n_c >> lg)) >> log2(Q): monotone in the
//                        key, ~Q keys per bucket whatever the shape of the distribution (bell-shaped scores put 0.4 % of all keys
//                        in one coarse bin), with room for 1.5 Q (CAP) keys reserved per bucket
//   msd_partition_kernel a 16 384-key tile leaves as one run per bucket (sorted by bucket in LDS: slots handed out by LDS atomics,
//                        no digit matching; run positions inside the bucket's segment by one global atomic per tile and bucket)
//   msd_offsets_kernel   exclusive scan of the buckets' actual sizes = the rank of each bucket's first key
//   msd_bucket_kernel    one bucket (<= CAP keys) per workgroup: counting sort on 2 CAP fine bins of the bucket's own key range
//                        (one returning LDS atomic per key, ~0.5 keys per bin), keys that share a fine bin ordered by (key,
//                        position) explicitly -- this is what makes ties stable although neither partition nor counting preserves
//                        arrival order -- then the (rank, position) pairs binned by 128 x 128 output block exactly as the last LSD
//                        pass did, and rank_block_write_kernel writes whole rows.
//
// Bytes per key: 4 (histogram) + 4 + 8 (partition) + 8 + 8 (bucket sort) + 8 + 8 (block write) = 48, against ~80 for the four LSD
// passes, and the kernels run on a FEW outcomes at a time (MDG_RANKS_GROUP) in a workspace that is reused, so that the 67 MB a pass
// writes per outcome are still in the 256 MB Infinity Cache when the next pass reads them.
//
// The payload is q = (i << 16) | j instead of the triangle index p: the same order as p, and no square root to get back (i, j).
//
// What the fast path cannot sort it hands back: a bucket that overflows its segment (a point mass of equal keys, a score
// distribution that is not smooth inside a coarse bin) or a fine bin with more than MSD_TIE_LIMIT keys raises the outcome's flag;
// flagged outcomes are skipped by the bucket / block-write kernels and sorted by the LSD kernels (launched for every chunk, each
// workgroup leaving at once unless its outcome is flagged).  Either way the ranks are the same bits.
constexpr int MSD_CB = 14, MSD_NC = 1 << MSD_CB, MSD_LB = 32 - MSD_CB;
constexpr int MSD_NB_MAX = 2048;          // buckets per outcome: a 16 384-key tile then leaves as runs of >= 8 pairs = 64 B on average (measured,
                                          // scripts/micro/run_scatter_bw.hip: runs of 64 B and longer store at 4.7-6.4 TB/s, runs of 32 B at 1.6-2.3)
constexpr int MSD_QLG = 12, MSD_NF = 8192;   // ~4096 keys per bucket; fine bins of the bucket sort
// Every counter that many workgroups add to is SHARDED: returning global atomics of all tiles onto one outcome's 2048 bucket counters
// (8 KB) or 528 block counters (2 KB) ran at the contended rate of the guide's "every workgroup into ONE row" case and WERE the
// kernels' duration (first version: partition 60 us, bucket sort 111 us per outcome).  A bucket's segment is 8 shards (by tile
// index), an output block's pair region 32 shards (by bucket index), the coarse histogram 8 copies (by workgroup index).
constexpr int MSD_SH = 8, MSD_SCAP = 768, MSD_CAP = MSD_SH * MSD_SCAP;     // pairs per shard of a bucket: 8 x 768 = 6144 = 1.5 Q
constexpr int MSD_BSH = 32, MSD_BREGION = 20480;   // shards of a block's pair region (at most; a power of two with >= 8 buckets per shard) and its
                                                   // room: 1.25 x the 16384 pairs of a full 128 x 128 block, cut evenly (32 shards: 640 each, 512 +- 22 used)
constexpr int MSD_TILE = 16384;           // keys per partition tile (1024 threads x 16)
constexpr int MSD_TIE_LIMIT = 128;        // keys per fine bin ordered in place; more: LSD fallback
constexpr int MSD_MAX_BLOCKS = 1536;      // output blocks per outcome on the fast path (3 per thread of the bucket sort)
constexpr uint32_t MSD_F_SEG = 1u, MSD_F_TOTAL = 2u, MSD_F_TIES = 4u, MSD_F_BLOCK = 8u;      // why an outcome was handed back
constexpr uint32_t MSD_SKIP = 0xFFFFFFFFu;


// hist[(outcome * MSD_SH + workgroup % MSD_SH) * MSD_NC + top 14 bits of the key]
__global__ __launch_bounds__(1024) void msd_hist_kernel(const float* __restrict__ scores, int64_t lds, uint32_t* __restrict__ hist, int N, int64_t M,
                                                       int64_t span, int src_is_keys) {
  __shared__ uint32_t cnt[MSD_NC];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int c = tid; c < MSD_NC; c += 1024) cnt[c] = 0;
  __syncthreads();
  const int64_t seg = blockIdx.y, wspan = span / 16, base = static_cast<int64_t>(blockIdx.x) * span + wave * wspan;
  const float* sc = scores + seg * static_cast<int64_t>(N) * lds;
  if (base < M) {
    TriWalk w;
    w.start(base, M);
    const int steps = static_cast<int>(wspan / 64);
#pragma unroll 8
    for (int k = 0; k < steps; ++k) {
      const int64_t p = base + k * 64 + lane;
      int i, j;
      w.lane_pos(lane, i, j);
      if (p < M) {
        const float v = sc[static_cast<int64_t>(i) * lds + j];
        const uint32_t key = src_is_keys ? __builtin_bit_cast(uint32_t, v) : mdg_order_key(v);
        __hip_atomic_fetch_add(&cnt[key >> MSD_LB], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      w.step();
    }
  }
  __syncthreads();
  uint32_t* h = hist + (seg * MSD_SH + (blockIdx.x % MSD_SH)) * MSD_NC;
  for (int c = tid; c < MSD_NC; c += 1024) {
    const uint32_t v = cnt[c];
    if (v) atomicAdd(&h[c], v);
  }
}

// table[c] = {keys in the coarse bins before c, (n_c >> lg) << 5 | lg}
__global__ __launch_bounds__(1024) void msd_table_kernel(const uint32_t* __restrict__ hist, u32x2* __restrict__ table, int qlg) {
  __shared__ uint32_t wsum[16];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t seg = blockIdx.x;
  uint32_t n[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) n[e] = 0;
  for (int sh = 0; sh < MSD_SH; ++sh) {
    const u32x4* h = reinterpret_cast<const u32x4*>(hist + (seg * MSD_SH + sh) * MSD_NC) + tid * 4;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const u32x4 x = h[v];
      n[4 * v] += x[0]; n[4 * v + 1] += x[1]; n[4 * v + 2] += x[2]; n[4 * v + 3] += x[3];
    }
  }
  uint32_t tot = 0;
#pragma unroll
  for (int e = 0; e < 16; ++e) tot += n[e];
  const uint32_t inc = wave_inclusive(tot, lane);
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t run = inc - tot;
  for (int w = 0; w < wave; ++w) run += wsum[w];
  const uint32_t unit = 1u << (qlg - 3);                 // sub-ranges of at most Q / 8 keys
  u32x2* t = table + seg * MSD_NC + tid * 16;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    uint32_t lg = 0;
    if (n[e] > unit) {
      const uint32_t parts = (n[e] + unit - 1) >> (qlg - 3);
      lg = 32 - __builtin_clz(parts - 1);              // ceil(log2(parts)), parts >= 2
      if (lg > MSD_LB) lg = MSD_LB;
    }
    t[e] = u32x2{run, ((n[e] >> lg) << 5) | lg};
    run += n[e];
  }
}

__device__ __forceinline__ uint32_t msd_bucket_of(uint32_t key, const u32x2* __restrict__ table, int qlg) {
  const u32x2 ent = table[key >> MSD_LB];
  const uint32_t lg = ent[1] & 31u;
  const uint32_t sub = lg ? ((key << MSD_CB) >> (32 - lg)) : 0u;     // the lg bits below the coarse prefix
  return (ent[0] + sub * (ent[1] >> 5)) >> qlg;
}

// Persistent: workgroup x of outcome y takes tiles x, x + gridDim.x, ...; the next tile's scores are in flight (registers) while
// this one is bucketed.  fill[(outcome * MSD_SH + tile % MSD_SH) * nbs + bucket]; segs[((outcome * nb + bucket) * MSD_SH + shard) * MSD_SCAP + .]
__global__ __launch_bounds__(1024) void msd_partition_kernel(const float* __restrict__ scores, int64_t lds, const u32x2* __restrict__ table,
                                                            uint32_t* __restrict__ fill, u32x2* __restrict__ segs, uint32_t* __restrict__ flags,
                                                            int N, int64_t M, int nb, int nbs, int qlg, int n_tiles, int src_is_keys) {
  constexpr int TPB = 1024, ITEMS = 16, BPT = MSD_NB_MAX / TPB;
  extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];       // spair[MSD_TILE] (u32x2) | bcnt[MSD_NB_MAX]
  __shared__ uint32_t wsum[16];
  u32x2* spair = reinterpret_cast<u32x2*>(dyn);
  uint32_t* bcnt = dyn + 2 * MSD_TILE;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t seg = blockIdx.y;
  const float* sc = scores + seg * static_cast<int64_t>(N) * lds;
  const u32x2* tab = table + seg * MSD_NC;
  u32x2* dst = segs + seg * static_cast<int64_t>(nb) * MSD_CAP;
  float raw[ITEMS];
  const auto load_tile = [&](int t) {
    const int64_t base = static_cast<int64_t>(t) * MSD_TILE + wave * (MSD_TILE / 16);
    TriWalk w;
    w.start(base, M);
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      int i, j;
      w.lane_pos(lane, i, j);
      raw[k] = (base + k * 64 + lane < M) ? sc[static_cast<int64_t>(i) * lds + j] : 0.f;
      w.step();
    }
  };
  int t = blockIdx.x;
  if (t < n_tiles) load_tile(t);
  for (; t < n_tiles; t += gridDim.x) {
    const int64_t base = static_cast<int64_t>(t) * MSD_TILE;
    const int shard = t % MSD_SH;
    for (int b = tid; b < MSD_NB_MAX; b += TPB) bcnt[b] = 0;
    uint32_t key[ITEMS], q[ITEMS], sb[ITEMS];             // sb = slot in the tile's run | bucket << 16
    {
      TriWalk w;                                           // the walk again, for the positions (the loads took it one tile ahead)
      w.start(base + wave * (MSD_TILE / 16), M);
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) {
        int i, j;
        w.lane_pos(lane, i, j);
        key[k] = src_is_keys ? __builtin_bit_cast(uint32_t, raw[k]) : mdg_order_key(raw[k]);
        q[k] = (static_cast<uint32_t>(i) << 16) | static_cast<uint32_t>(j);
        w.step();
      }
    }
    if (t + static_cast<int>(gridDim.x) < n_tiles) load_tile(t + gridDim.x);
    __syncthreads();                                       // bcnt zeroed (and the previous tile's copy-out done with it)
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int64_t p = base + wave * (MSD_TILE / 16) + k * 64 + lane;
      sb[k] = MSD_SKIP;
      if (p < M) {
        const uint32_t b = msd_bucket_of(key[k], tab, qlg);
        sb[k] = atomicAdd(&bcnt[b], 1u) | (b << 16);
      }
    }
    __syncthreads();
    // exclusive scan of the bucket counts, BPT per thread; room for this tile's run in every non-empty bucket's shard: the atomics
    // are issued here and their results are needed only after the placement
    uint32_t c4[BPT], st[BPT], old[BPT];
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
        old[e] = c4[e] ? atomicAdd(&fill[(seg * MSD_SH + shard) * nbs + BPT * tid + e], c4[e]) : 0u;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; ++k)
      if (sb[k] != MSD_SKIP) spair[bcnt[sb[k] >> 16] + (sb[k] & 0xFFFFu)] = u32x2{key[k], q[k]};
    __syncthreads();
    // bcnt[b] becomes (position in the bucket's shard) - (position in LDS)
#pragma unroll
    for (int e = 0; e < BPT; ++e)
      if (c4[e]) {
        if (old[e] + c4[e] > static_cast<uint32_t>(MSD_SCAP)) atomicOr(&flags[seg], MSD_F_SEG);
        bcnt[BPT * tid + e] = old[e] - st[e];
      }
    __syncthreads();
    const int n_valid = static_cast<int>(M - base < MSD_TILE ? M - base : MSD_TILE);
#pragma unroll 4
    for (int k = 0; k < ITEMS; ++k) {
      const int idx = k * TPB + tid;
      if (idx >= n_valid) break;
      const u32x2 v = spair[idx];
      const uint32_t b = msd_bucket_of(v[0], tab, qlg);
      const uint32_t pos = bcnt[b] + static_cast<uint32_t>(idx);
      if (pos < static_cast<uint32_t>(MSD_SCAP)) dst[(static_cast<int64_t>(b) * MSD_SH + shard) * MSD_SCAP + pos] = v;
    }
    __syncthreads();                                       // spair / bcnt are read: the next tile may overwrite them
  }
}

// rbase[b] = keys in the buckets before b; a shard beyond its room or a total that is not M raises the flag
__global__ __launch_bounds__(1024) void msd_offsets_kernel(const uint32_t* __restrict__ fill, uint32_t* __restrict__ rbase, uint32_t* __restrict__ flags,
                                                          int nb, int nbs, int64_t M) {
  __shared__ uint32_t wsum[16];
  constexpr int BPT = MSD_NB_MAX / 1024;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t seg = blockIdx.x;
  uint32_t c4[BPT], tot = 0;
  bool over = false;
#pragma unroll
  for (int e = 0; e < BPT; ++e) {
    const int b = BPT * tid + e;
    c4[e] = 0;
    if (b < nb)
      for (int sh = 0; sh < MSD_SH; ++sh) {
        const uint32_t c = fill[(seg * MSD_SH + sh) * nbs + b];
        over = over || c > static_cast<uint32_t>(MSD_SCAP);
        c4[e] += c;
      }
    tot += c4[e];
  }
  const uint32_t inc = wave_inclusive(tot, lane);
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t run = inc - tot;
  for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
  for (int e = 0; e < BPT; ++e) {
    const int b = BPT * tid + e;
    if (b < nb) rbase[seg * nbs + b] = run;
    run += c4[e];
  }
  if (over) atomicOr(&flags[seg], MSD_F_SEG);
  if (tid == 1023 && static_cast<int64_t>(run) != M) atomicOr(&flags[seg], MSD_F_TOTAL);
}

// Persistent: workgroup x of outcome y takes buckets x, x + gridDim.x, ...; the next bucket's pairs are in flight (registers) while
// this one is sorted, its shard sizes were fetched one bucket earlier still.
// pairs[((outcome * n_blocks + block) * MSD_BSH + bucket % MSD_BSH) * MSD_BCAP + .]; blockfill[(outcome * MSD_BSH + shard) * n_blocks + block]
template <int NF, int TPB, bool PF>
__global__ __launch_bounds__(TPB, TPB / 128) void msd_bucket_kernel(const u32x2* __restrict__ segs, const uint32_t* __restrict__ fill, const uint32_t* __restrict__ rbase,
                                                         u32x2* __restrict__ pairs, uint32_t* __restrict__ blockfill, uint32_t* __restrict__ flags,
                                                         int N, int nb, int nbs, int n_blocks, int bsh) {
  constexpr int CAP = MSD_CAP, ITEMS = CAP / TPB, WPT = NF / 2 / TPB, WAVES = TPB / 64;     // NF fine bins, two u16 counters per word
  constexpr int LGNF = 31 - __builtin_clz(NF), BPT = (MSD_MAX_BLOCKS + TPB - 1) / TPB;
  static_assert(CAP % TPB == 0 && (NF & (NF - 1)) == 0 && NF % (2 * TPB) == 0 && CAP < 65536, "bucket sort shape");
  extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];       // sorted[CAP] (u32x2) | fc[NF / 2] | bcnt[n_blocks] | bdst[n_blocks]
  __shared__ uint32_t wsum[WAVES], krange[2];
  u32x2* sorted = reinterpret_cast<u32x2*>(dyn);
  uint32_t* fc = dyn + 2 * CAP;
  uint32_t* bcnt = fc + NF / 2;
  uint32_t* bdst = bcnt + n_blocks;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t seg = blockIdx.y;
  const int stride = gridDim.x;
  // shard sizes of a bucket as exclusive prefix sums ps[0..SH] (ps[SH] = the bucket's size, clamped shard by shard)
  const auto shard_sizes = [&](int b, uint32_t (&ps)[MSD_SH + 1]) {
    ps[0] = 0;
#pragma unroll
    for (int sh = 0; sh < MSD_SH; ++sh) {
      uint32_t c = b < nb ? fill[(seg * MSD_SH + sh) * nbs + b] : 0u;
      c = c < static_cast<uint32_t>(MSD_SCAP) ? c : static_cast<uint32_t>(MSD_SCAP);
      ps[sh + 1] = ps[sh] + c;
    }
  };
  const auto load_bucket = [&](int b, const uint32_t (&ps)[MSD_SH + 1], u32x2 (&v)[ITEMS]) {
    const u32x2* src = segs + (seg * nb + b) * static_cast<int64_t>(CAP);
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const uint32_t idx = static_cast<uint32_t>(k * TPB + tid);
      v[k] = u32x2{0u, 0u};
      if (idx < ps[MSD_SH]) {
        int sh = 0;
#pragma unroll
        for (int e = 1; e < MSD_SH; ++e) sh += idx >= ps[e] ? 1 : 0;
        v[k] = src[sh * MSD_SCAP + (idx - ps[sh])];
      }
    }
  };
  uint32_t ps_cur[MSD_SH + 1], ps_nxt[MSD_SH + 1];
  u32x2 cur[ITEMS];
  int b = blockIdx.x;
  if (b >= nb) return;
  if constexpr (PF) {
    shard_sizes(b, ps_cur);
    load_bucket(b, ps_cur, cur);
    shard_sizes(b + stride, ps_nxt);
  }
  for (; b < nb; b += stride) {
    if constexpr (!PF) {
      shard_sizes(b, ps_cur);
      load_bucket(b, ps_cur, cur);
    }
    const int n = static_cast<int>(ps_cur[MSD_SH]);
    uint32_t key[ITEMS], q[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) { key[k] = cur[k][0]; q[k] = cur[k][1]; }
    const uint32_t rb = rbase[seg * nbs + b];
    for (int i = tid; i < NF / 2; i += TPB) fc[i] = 0;
    for (int i = tid; i < n_blocks; i += TPB) bcnt[i] = 0;
    if (tid == 0) { krange[0] = 0xFFFFFFFFu; krange[1] = 0u; }
    __syncthreads();
    // ---- output blocks first: their room in the pair buffer comes from global atomics whose results are needed at the very end
    // per key: its slot in the bucket's run of its output block (low half) and, later, its slot in its fine bin (high half); the
    // block and the fine bin themselves are recomputed from q / key where they are needed (registers: two workgroups per CU)
    uint32_t ss[ITEMS];
    const auto block_of = [](uint32_t qq) -> uint32_t {
      const uint32_t bi = qq >> 23, bj = (qq & 0xFFFFu) >> 7;
      return bi * (bi + 1u) / 2u + bj;
    };
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int idx = k * TPB + tid;
      ss[k] = 0u;
      if (idx < n) {
        ss[k] = atomicAdd(&bcnt[block_of(q[k])], 1u);
        kmin = kmin < key[k] ? kmin : key[k];
        kmax = kmax > key[k] ? kmax : key[k];
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t a = __shfl_xor(kmin, o, 64), c = __shfl_xor(kmax, o, 64);
      kmin = kmin < a ? kmin : a;
      kmax = kmax > c ? kmax : c;
    }
    if (lane == 0 && n > 0) { atomicMin(&krange[0], kmin); atomicMax(&krange[1], kmax); }
    __syncthreads();
    uint32_t g_old[BPT], g_run[BPT], g_cnt[BPT];
    {
      const int a = tid * BPT;
      uint32_t s = 0;
#pragma unroll
      for (int e = 0; e < BPT; ++e) { g_cnt[e] = a + e < n_blocks ? bcnt[a + e] : 0u; s += g_cnt[e]; }
      const uint32_t inc = wave_inclusive(s, lane);
      if (lane == 63) wsum[wave] = inc;
      __syncthreads();
      uint32_t run = inc - s;
      for (int v = 0; v < wave; ++v) run += wsum[v];
#pragma unroll
      for (int e = 0; e < BPT; ++e) {
        g_run[e] = run;
        if (a + e < n_blocks) bcnt[a + e] = run;
        g_old[e] = g_cnt[e] ? atomicAdd(&blockfill[(seg * MSD_BSH + (b & (bsh - 1))) * n_blocks + a + e], g_cnt[e]) : 0u;
        run += g_cnt[e];
      }
    }
    // ---- counting sort on the fine bins of the bucket's own key range
    const uint32_t lo = krange[0], range = krange[1] - lo;
    const int sh = (n > 0 && (range >> LGNF)) ? (32 - __builtin_clz(range) - LGNF) : 0;     // (range >> sh) < NF
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int idx = k * TPB + tid;
      if (idx < n) {
        const uint32_t fi = (key[k] - lo) >> sh;
        const int hs = 16 * (fi & 1u);
        ss[k] |= ((atomicAdd(&fc[fi >> 1], 1u << hs) >> hs) & 0xFFFFu) << 16;
      }
    }
    __syncthreads();
    {   // exclusive scan of the NF u16 counters in place (a contiguous run of WPT words per thread): fc holds each bin's first slot
      uint32_t w[WPT], tot = 0;
#pragma unroll
      for (int e = 0; e < WPT; ++e) { w[e] = fc[tid * WPT + e]; tot += (w[e] & 0xFFFFu) + (w[e] >> 16); }
      const uint32_t inc = wave_inclusive(tot, lane);
      if (lane == 63) wsum[wave] = inc;                    // (the block scan's readers of wsum are behind the barrier above)
      __syncthreads();
      uint32_t run = inc - tot;
      for (int v = 0; v < wave; ++v) run += wsum[v];
#pragma unroll
      for (int e = 0; e < WPT; ++e) {
        const uint32_t c0 = w[e] & 0xFFFFu, c1 = w[e] >> 16;
        fc[tid * WPT + e] = run | ((run + c0) << 16);
        run += c0 + c1;
      }
    }
    __syncthreads();
    const auto fstart = [&](uint32_t f) -> uint32_t { return f >= static_cast<uint32_t>(NF) ? static_cast<uint32_t>(n) : (fc[f >> 1] >> (16 * (f & 1u))) & 0xFFFFu; };
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int idx = k * TPB + tid;
      if (idx < n) sorted[fstart((key[k] - lo) >> sh) + (ss[k] >> 16)] = u32x2{key[k], q[k]};
    }
    __syncthreads();
    // keys that share a fine bin: their order is (key, position); everything else is in place already
    bool too_many = false;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int idx = k * TPB + tid;
      if (idx < n) {
        const uint32_t fi = (key[k] - lo) >> sh, s0 = fstart(fi), c = fstart(fi + 1u) - s0;
        uint32_t r = 0;
        if (c > 1u) {
          if (c > static_cast<uint32_t>(MSD_TIE_LIMIT)) too_many = true;
          else
            for (uint32_t m = 0; m < c; ++m) {
              const u32x2 o = sorted[s0 + m];
              r += (o[0] < key[k] || (o[0] == key[k] && o[1] < q[k])) ? 1u : 0u;
            }
        }
        key[k] = rb + s0 + r;                              // rank - 1
      }
    }
    if (too_many) atomicOr(&flags[seg], MSD_F_TIES);
    // ---- (rank, position in block) pairs, block by block, through `sorted` (free now) into the blocks' shards
    {
      const int a = tid * BPT;
#pragma unroll
      for (int e = 0; e < BPT; ++e)
        if (g_cnt[e]) {
          const uint32_t bcap = static_cast<uint32_t>(MSD_BREGION / bsh);
          const bool fits = g_old[e] + g_cnt[e] <= bcap;
          if (!fits) atomicOr(&flags[seg], MSD_F_BLOCK);
          bdst[a + e] = fits ? static_cast<uint32_t>(a + e) * MSD_BREGION + static_cast<uint32_t>(b & (bsh - 1)) * bcap + g_old[e] - g_run[e] : MSD_SKIP;
        }
    }
    __syncthreads();                                       // fix-up reads of `sorted` done; bdst written
#pragma unroll
    for (int k = 0; k < ITEMS; ++k)
      if (k * TPB + tid < n) {
        const uint32_t i = q[k] >> 16, j = q[k] & 0xFFFFu, blk = block_of(q[k]);
        sorted[bcnt[blk] + (ss[k] & 0xFFFFu)] = u32x2{key[k], (((i & 127u) << 7) | (j & 127u)) | (blk << 14)};
      }
    // the per-key registers are dead: the next bucket's pairs leave now (in flight across the copy-out and the next bucket's
    // block counting), the sizes of the bucket after it too
    if constexpr (PF) {
      if (b + stride < nb) load_bucket(b + stride, ps_nxt, cur);
#pragma unroll
      for (int e = 0; e <= MSD_SH; ++e) ps_cur[e] = ps_nxt[e];
      shard_sizes(b + 2 * stride, ps_nxt);
    }
    __syncthreads();
    u32x2* dst = pairs + seg * static_cast<int64_t>(n_blocks) * MSD_BREGION;
#pragma unroll 4
    for (int k = 0; k < ITEMS; ++k) {
      const int idx = k * TPB + tid;
      if (idx >= n) break;
      const u32x2 v = sorted[idx];
      const uint32_t d = bdst[v[1] >> 14];
      if (d != MSD_SKIP) dst[d + static_cast<uint32_t>(idx)] = u32x2{v[0], v[1] & 16383u};
    }
    __syncthreads();                                       // sorted / bcnt / bdst are read: the next bucket may overwrite them
  }
}

// one 128 x 128 block of the lower triangle from the MSD_BSH shards of its pair region: ranks into an LDS tile, then whole rows of
// out[i, j] and of the mirrored block (as rank_block_write_kernel)
template <bool VEC>
__global__ __launch_bounds__(512) void msd_block_write_kernel(const u32x2* __restrict__ pairs, const uint32_t* __restrict__ blockfill, float* __restrict__ out,
                                                              int64_t ldo, int N, int n_blocks, double denom, const uint32_t* __restrict__ flags, int bsh) {
  constexpr int TPB = 512;
  __shared__ float tile[BB][BB + 1];
  const int64_t seg = blockIdx.y;
  if (flags[seg]) return;                                  // handed to the LSD kernels (no writer of the flags runs beside this kernel)
  const int t = blockIdx.x, tid = threadIdx.x;
  int bi = static_cast<int>((sqrtf(8.0f * static_cast<float>(t) + 1.0f) - 1.0f) * 0.5f);
  while (bi * (bi + 1) / 2 > t) --bi;
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  const int bj = t - bi * (bi + 1) / 2;
  const int r0 = bi * BB, c0 = bj * BB;
  const int rcount = N - r0 < BB ? N - r0 : BB;
  const bool diag = bi == bj;
  if (diag)
    for (int e = tid; e < BB; e += TPB) tile[e][e] = 0.f;
  const u32x2* src = pairs + (seg * n_blocks + t) * static_cast<int64_t>(MSD_BREGION);
  const int bcap = MSD_BREGION / bsh, parts = (bcap + TPB - 1) / TPB;
  __shared__ uint32_t scnt[MSD_BSH];
  if (tid < MSD_BSH) scnt[tid] = tid < bsh ? blockfill[(seg * MSD_BSH + tid) * n_blocks + t] : 0u;
  __syncthreads();
  // the whole region in one flat sweep, two slots (16 B) per load and ten loads in flight per thread: a launch covers few outcomes
  // (~2 workgroups per CU), so a thread's own loads are what hides the memory latency.  bcap is even: a load never straddles shards.
  const u32x4* src4 = reinterpret_cast<const u32x4*>(src);
  (void)parts;
#pragma unroll 10
  for (int x = tid; x < MSD_BREGION / 2; x += TPB) {
    const int sh = (2 * x) / bcap, off = 2 * x - sh * bcap, c = static_cast<int>(scnt[sh]);
    if (off < c) {
      const u32x4 v = src4[x];
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (off + h < c) {
          const float val = static_cast<float>(static_cast<double>(v[2 * h] + 1u) / denom);
          const int r = v[2 * h + 1] >> 7, cc = v[2 * h + 1] & 127;
          tile[r][cc] = val;
          if (diag) tile[cc][r] = val;
        }
    }
  }
  __syncthreads();
  float* o = out + seg * static_cast<int64_t>(N) * ldo;
  const int q = tid & 31, rr = tid >> 5;
  const int ccount = diag ? rcount : BB;
  for (int r = rr; r < rcount; r += TPB / 32) {
    float* row = o + static_cast<int64_t>(r0 + r) * ldo + c0;
    if (VEC && 4 * q + 3 < ccount) *reinterpret_cast<f32x4*>(row + 4 * q) = f32x4{tile[r][4 * q], tile[r][4 * q + 1], tile[r][4 * q + 2], tile[r][4 * q + 3]};
    else
      for (int e = 0; e < 4; ++e)
        if (4 * q + e < ccount) row[4 * q + e] = tile[r][4 * q + e];
  }
  if (diag) return;
  for (int c = rr; c < BB; c += TPB / 32) {
    float* row = o + static_cast<int64_t>(c0 + c) * ldo + r0;
    if (VEC && 4 * q + 3 < rcount) *reinterpret_cast<f32x4*>(row + 4 * q) = f32x4{tile[4 * q][c], tile[4 * q + 1][c], tile[4 * q + 2][c], tile[4 * q + 3][c]};
    else
      for (int e = 0; e < 4; ++e)
        if (4 * q + e < rcount) row[4 * q + e] = tile[4 * q + e][c];
  }
}

template <bool VEC>
__global__ __launch_bounds__(512) void rank_block_write_kernel(const u32x2* __restrict__ pairs, float* __restrict__ out, int64_t ldo, int N,
                                                               int64_t M, int n_blocks, double denom, const uint32_t* __restrict__ flags, int want) {
  constexpr int TPB = 512;
  __shared__ float tile[BB][BB + 1];
  if (flags && (flags[blockIdx.y] != 0u) != (want != 0)) return;      // MSD fast path: unflagged outcomes; LSD fallback behind it: flagged ones
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

// ---- MSD fast path: eligibility, workspace ------------------------------------------------------------------------------------
struct MsdPlan {
  bool on;
  int group;                 // outcomes per launch group (the group's buffers are reused by the next group: Infinity-Cache resident)
  int nb, nbs, n_blocks, bsh;   // buckets per outcome; counter stride; output blocks; shards of a block's pair region
  size_t seg_bytes, pair_bytes, hist_bytes, table_bytes, fill_bytes, rbase_bytes, blockfill_bytes;      // per outcome
  size_t group_bytes(int g) const { return a256(g * seg_bytes) + a256(g * pair_bytes) + a256(g * table_bytes) + a256(g * rbase_bytes) + a256(g * hist_bytes) +
                                           a256(g * fill_bytes) + a256(g * blockfill_bytes); }
};

static MsdPlan msd_plan(int64_t n_outcomes, int64_t N) {
  static MdgEnvInt msd_sw{"MDG_RANKS_MSD", 0};             // 1: the adaptive MSD path first, the LSD sort for what it hands back (see the header above)
  static MdgEnvInt group_sw{"MDG_RANKS_GROUP", 8};         // measured at 4096^2: 1: 249 us per outcome, 2: 216, 8: 184, 16: 188, 32: 198 (LSD: 239)
  MsdPlan pl{};
  const int64_t M = N * (N - 1) / 2;
  const int64_t nb = (M >> MSD_QLG) + 1;
  pl.on = msd_sw.get() != 0 && N >= 2 && nb <= MSD_NB_MAX && rank_blocks_of(N) <= MSD_MAX_BLOCKS;
  if (!pl.on) return pl;
  int g = group_sw.get();
  g = g < 1 ? 1 : g;
  pl.group = static_cast<int>(g < n_outcomes ? g : n_outcomes);
  pl.nb = static_cast<int>(nb);
  pl.nbs = (pl.nb + 63) & ~63;
  pl.n_blocks = static_cast<int>(rank_blocks_of(N));
  pl.seg_bytes = static_cast<size_t>(pl.nb) * MSD_CAP * 8;
  pl.pair_bytes = static_cast<size_t>(pl.n_blocks) * MSD_BREGION * 8;
  pl.bsh = 1;
  while (pl.bsh < MSD_BSH && pl.nb / (2 * pl.bsh) >= 8) pl.bsh *= 2;          // >= 8 buckets per shard of a block's pair region
  pl.hist_bytes = static_cast<size_t>(MSD_SH) * MSD_NC * 4;
  pl.table_bytes = static_cast<size_t>(MSD_NC) * 8;
  pl.fill_bytes = static_cast<size_t>(MSD_SH) * pl.nbs * 4;
  pl.rbase_bytes = static_cast<size_t>(pl.nbs) * 4;
  pl.blockfill_bytes = static_cast<size_t>(MSD_BSH) * pl.n_blocks * 4;
  return pl;
}

template <class C>
static size_t lsd_workspace_bytes(int64_t n_outcomes, int64_t N) {
  const size_t M = static_cast<size_t>(N) * (N - 1) / 2;
  const size_t nblk = (M + CfgStd::TILE - 1) / CfgStd::TILE;          // the last pass always runs on 8192-key tiles (the finer table)
  // keys / payloads x 2 | per-tile digit table (histogram path: one; look-back: one status table per pass) | block fill counters | digit totals
  return 4 * a256(static_cast<size_t>(n_outcomes) * M * 4) + 4 * a256(static_cast<size_t>(n_outcomes) * 256 * nblk * 4) +
         a256(static_cast<size_t>(n_outcomes) * static_cast<size_t>(rank_blocks_of(N)) * 4) + a256(static_cast<size_t>(n_outcomes) * 4 * 256 * 4);
}

// flags (one per outcome, live from the fast path to the fallback) | max(LSD scratch of all outcomes, fast-path scratch of one group)
template <class C>
static size_t rank_workspace_bytes(int64_t n_outcomes, int64_t N) {
  const size_t lsd = lsd_workspace_bytes<C>(n_outcomes, N);
  const MsdPlan pl = msd_plan(n_outcomes, N);
  if (!pl.on) return lsd;
  const size_t fast = pl.group_bytes(pl.group);
  return a256(static_cast<size_t>(n_outcomes) * 4) + (lsd > fast ? lsd : fast);
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

// the fast path over all outcomes of the call, `group` at a time; raises flags[outcome] for what it leaves to the LSD kernels
static void msd_run(const MsdPlan& pl, const float* scores, int64_t lds, float* out, int64_t ldo, int64_t n_outcomes, int64_t N, char* ws,
                    uint32_t* flags, hipStream_t st, int src_is_keys) {
  const int64_t M = N * (N - 1) / 2;
  const int n_blocks = pl.n_blocks;
  const double denom = static_cast<double>(N) * static_cast<double>(N - 1) / 2.0;
  const int G = pl.group;
  char* p = ws;
  u32x2* segs = reinterpret_cast<u32x2*>(p); p += a256(G * pl.seg_bytes);
  u32x2* pairs = reinterpret_cast<u32x2*>(p); p += a256(G * pl.pair_bytes);
  u32x2* table = reinterpret_cast<u32x2*>(p); p += a256(G * pl.table_bytes);
  uint32_t* rbase = reinterpret_cast<uint32_t*>(p); p += a256(G * pl.rbase_bytes);
  char* zero0 = p;                                                                  // hist | fill | blockfill: zeroed per group
  uint32_t* hist = reinterpret_cast<uint32_t*>(p); p += a256(G * pl.hist_bytes);
  uint32_t* fill = reinterpret_cast<uint32_t*>(p); p += a256(G * pl.fill_bytes);
  uint32_t* blockfill = reinterpret_cast<uint32_t*>(p); p += a256(G * pl.blockfill_bytes);
  const size_t zero_bytes = static_cast<size_t>(p - zero0);
  const int n_tiles = static_cast<int>(mdg_cdiv(M, MSD_TILE));
  static MdgEnvInt hwg_sw{"MDG_RANKS_HIST_WGS", 512};
  // bucket sort: 2 = 512 threads, one bucket per workgroup (default: 184 us per 4096^2 outcome); 0 = 1024 threads, persistent, the next
  // bucket prefetched (201); 1 = 512 + prefetch; 3 = 1024, one bucket per workgroup
  static MdgEnvInt variant_sw{"MDG_RANKS_BUCKET_VARIANT", 2};
  const size_t part_lds = static_cast<size_t>(2 * MSD_TILE + MSD_NB_MAX) * 4;
  const size_t bucket_lds = static_cast<size_t>(2 * MSD_CAP + MSD_NF / 2 + 2 * n_blocks) * 4;
  const bool vec = ldo % 4 == 0 && mdg_aligned16(out);
  static MdgEnvInt pwg_sw{"MDG_RANKS_PART_WGS", 256}, bwg_sw{"MDG_RANKS_BUCKET_WGS", 512};   // persistent workgroups of a launch (all outcomes of the group)
  for (int64_t s0 = 0; s0 < n_outcomes; s0 += G) {
    const unsigned g = static_cast<unsigned>(n_outcomes - s0 < G ? n_outcomes - s0 : G);
    const float* sc = scores + s0 * N * lds;
    float* o = out + s0 * N * ldo;
    uint32_t* fl = flags + s0;
    unsigned pw = static_cast<unsigned>(mdg_cdiv(pwg_sw.get(), g)), bw = static_cast<unsigned>(mdg_cdiv(bwg_sw.get(), g));
    pw = pw < 1u ? 1u : (pw > static_cast<unsigned>(n_tiles) ? static_cast<unsigned>(n_tiles) : pw);
    bw = bw < 1u ? 1u : (bw > static_cast<unsigned>(pl.nb) ? static_cast<unsigned>(pl.nb) : bw);
    (void)hipMemsetAsync(zero0, 0, zero_bytes, st);
    // two histogram workgroups per CU over the whole group: a wave keeps 8 x 256 B of scores in flight, a CU then 64 KB
    int64_t hw = mdg_cdiv(hwg_sw.get(), g);
    hw = hw < 1 ? 1 : (hw > mdg_cdiv(M, 1024) ? mdg_cdiv(M, 1024) : hw);
    const int64_t span = (mdg_cdiv(M, hw) + 1023) & ~static_cast<int64_t>(1023);
    hipLaunchKernelGGL(msd_hist_kernel, dim3(static_cast<unsigned>(hw), g), dim3(1024), 0, st, sc, lds, hist, static_cast<int>(N), M, span, src_is_keys);
    hipLaunchKernelGGL(msd_table_kernel, dim3(g), dim3(1024), 0, st, hist, table, MSD_QLG);
    hipLaunchKernelGGL(msd_partition_kernel, dim3(pw, g), dim3(1024), part_lds, st, sc, lds, table, fill, segs, fl, static_cast<int>(N), M, pl.nb, pl.nbs,
                       MSD_QLG, n_tiles, src_is_keys);
    hipLaunchKernelGGL(msd_offsets_kernel, dim3(g), dim3(1024), 0, st, fill, rbase, fl, pl.nb, pl.nbs, M);
    switch (variant_sw.get()) {
#define MDG_BUCKET_LAUNCH(T, P, GRID)                                                                                                                       \
  hipLaunchKernelGGL((msd_bucket_kernel<MSD_NF, T, P>), dim3(GRID, g), dim3(T), bucket_lds, st, segs, fill, rbase, pairs, blockfill, fl, static_cast<int>(N), pl.nb, \
                     pl.nbs, n_blocks, pl.bsh)
      case 0: MDG_BUCKET_LAUNCH(1024, true, bw); break;
      case 1: MDG_BUCKET_LAUNCH(512, true, bw); break;
      case 3: MDG_BUCKET_LAUNCH(1024, false, static_cast<unsigned>(pl.nb)); break;
      default: MDG_BUCKET_LAUNCH(512, false, static_cast<unsigned>(pl.nb)); break;
#undef MDG_BUCKET_LAUNCH
    }
    const dim3 bgrid(static_cast<unsigned>(n_blocks), g);
    if (vec) hipLaunchKernelGGL(msd_block_write_kernel<true>, bgrid, dim3(512), 0, st, pairs, blockfill, o, ldo, static_cast<int>(N), n_blocks, denom, fl, pl.bsh);
    else hipLaunchKernelGGL(msd_block_write_kernel<false>, bgrid, dim3(512), 0, st, pairs, blockfill, o, ldo, static_cast<int>(N), n_blocks, denom, fl, pl.bsh);
  }
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
  // MSD fast path first (round 4); `only` = its per-outcome flags: the LSD kernels below then touch the flagged outcomes alone
  const uint32_t* only = nullptr;
  const MsdPlan pl = msd_plan(n_outcomes, N);
  if (pl.on) {
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msd_partition_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msd_bucket_kernel<MSD_NF, 1024, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msd_bucket_kernel<MSD_NF, 512, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msd_bucket_kernel<MSD_NF, 1024, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msd_bucket_kernel<MSD_NF, 512, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      attr_done = true;
    }
    uint32_t* flags = reinterpret_cast<uint32_t*>(ws);
    ws += a256(static_cast<size_t>(n_outcomes) * 4);
    (void)hipMemsetAsync(flags, 0, static_cast<size_t>(n_outcomes) * 4, st);
    msd_run(pl, scores, lds, out, ldo, n_outcomes, N, ws, flags, st, src_is_keys);
    only = flags;
    static MdgEnvInt nofb_sw{"MDG_RANKS_NO_FALLBACK", 0};   // diagnostics (timing the fast path alone): flagged outcomes are then left unranked
    if (nofb_sw.get()) { MDG_CHECK_LAUNCH("mdg_rank_normalize"); return MDG_OK; }
  }
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
  hipLaunchKernelGGL(extract_keys_kernel<C>, grid, dim3(TPB), 0, st, scores, lds, k0, hist, lb ? ghist : nullptr, static_cast<int>(N), M, nblk, src_is_keys, only);
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
      if (std3) hipLaunchKernelGGL((histogram_kernel<CfgStd, u8>), grid3, dim3(CfgStd::TPB), 0, st, reinterpret_cast<const u8*>(kin), hist, M, nblk3, 0, only);
      else if (pass == 3) hipLaunchKernelGGL((histogram_kernel<C, u8>), grid, dim3(TPB), 0, st, reinterpret_cast<const u8*>(kin), hist, M, nblk, 0, only);
      else if (pass == 2) hipLaunchKernelGGL((histogram_kernel<C, u16>), grid, dim3(TPB), 0, st, reinterpret_cast<const u16*>(kin), hist, M, nblk, 0, only);
      else if (pass == 1) hipLaunchKernelGGL((histogram_kernel<C, uint32_t>), grid, dim3(TPB), 0, st, kin, hist, M, nblk, 8, only);
      hipLaunchKernelGGL(scan_kernel, dim3(L), dim3(1024), 0, st, hist, std3 ? nblk3 : nblk, only);
    }
    if (pass == 0)
      hipLaunchKernelGGL((scatter_kernel<C, true, false>), grid, dim3(TPB), 0, st, kin, pin, kout, pout, offs, out, ldo, static_cast<int>(N), M, nblk, 0, denom, stat, gcur, only);
    else if (pass == 1)
      hipLaunchKernelGGL((scatter_kernel<C, false, false, uint32_t, u16>), grid, dim3(TPB), 0, st, kin, pin, reinterpret_cast<u16*>(kout), pout, offs, out, ldo,
                         static_cast<int>(N), M, nblk, 8, denom, stat, gcur, only);
    else if (pass == 2)
      hipLaunchKernelGGL((scatter_kernel<C, false, false, u16, u8>), grid, dim3(TPB), 0, st, reinterpret_cast<const u16*>(kin), pin, reinterpret_cast<u8*>(kout), pout,
                         offs, out, ldo, static_cast<int>(N), M, nblk, 0, denom, stat, gcur, only);
    else if (blocked) {
      if (!lb) (void)hipMemsetAsync(fill, 0, static_cast<size_t>(n_outcomes) * n_blocks * 4, st);
      u32x2* pairs = reinterpret_cast<u32x2*>(k0);            // pass 3 reads k1 / p1
      hipLaunchKernelGGL((rank_blocks_kernel<CfgStd, u8>), grid3, dim3(CfgStd::TPB), blocks_lds, st, reinterpret_cast<const u8*>(kin), pin, offs, pairs, fill,
                         static_cast<int>(N), M, nblk3, static_cast<int>(n_blocks), stat, gcur, only);
      const dim3 bgrid(static_cast<unsigned>(n_blocks), L);
      if (ldo % 4 == 0 && mdg_aligned16(out))
        hipLaunchKernelGGL(rank_block_write_kernel<true>, bgrid, dim3(512), 0, st, pairs, out, ldo, static_cast<int>(N), M, static_cast<int>(n_blocks), denom, only, 1);
      else
        hipLaunchKernelGGL(rank_block_write_kernel<false>, bgrid, dim3(512), 0, st, pairs, out, ldo, static_cast<int>(N), M, static_cast<int>(n_blocks), denom, only, 1);
    } else {
      hipLaunchKernelGGL((scatter_kernel<C, false, true, u8, u8>), grid, dim3(TPB), 0, st, reinterpret_cast<const u8*>(kin), pin, reinterpret_cast<u8*>(kout), pout, offs,
                         out, ldo, static_cast<int>(N), M, nblk, 0, denom, stat, gcur, only);
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
