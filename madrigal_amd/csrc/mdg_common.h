// Shared device/host helpers for libmadrigal_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/madrigal_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define MDG_WAVE 64

// ---- host-side error plumbing ------------------------------------------------------------
void mdg_set_error(const char* fmt, ...);

#define MDG_CHECK_ARG(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      mdg_set_error(__VA_ARGS__);         \
      return MDG_EINVAL;                  \
    }                                     \
  } while (0)

#define MDG_CHECK_LAUNCH(what)                                                   \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      mdg_set_error("%s: launch failed: %s", what, hipGetErrorString(e__));      \
      return MDG_ELAUNCH;                                                        \
    }                                                                            \
  } while (0)

static inline bool mdg_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int64_t mdg_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- tuning switches: an MDG_* environment variable read once (see api.hip: mdg_tuning_reload) ----
#include <atomic>
extern std::atomic<int> g_mdg_env_generation;
struct MdgEnvInt {
  const char* name;
  int dflt;
  int gen = -1;
  int val = 0;
  int get();
};

// ---- device helpers ----------------------------------------------------------------------
// hi/lo bf16 split of an fp32 value: x ~= hi + lo with |x - hi - lo| <= 2^-18 |x|.
__device__ __forceinline__ void mdg_split_bf16(float x, __bf16& hi, __bf16& lo) {
  hi = (__bf16)x;
  lo = (__bf16)(x - (float)hi);
}

__device__ __forceinline__ float mdg_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float mdg_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// order-preserving 32-bit key of an fp32 score (ascending unsigned order = ascending float order; -0 < +0): the rank
// normalisation sorts these, and the all-pairs head can write them instead of the scores (MDG_EPI_TRIKEYS)
__device__ __forceinline__ uint32_t mdg_order_key(float f) {
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// counter-based random bits (splitmix64 finaliser): dropout masks are a pure function of (seed, index), so the backward pass
// regenerates them instead of storing them.  One 64-bit hash serves FOUR consecutive elements (16 bits each: the drop probability is
// resolved to 1/65536): element `index` is kept iff field (index & 3) of the hash of (seed, index >> 2) reaches the threshold.  A thread
// that owns an aligned group of four (16-byte elementwise passes, the dense block's epilogue) hashes once: mdg_keep_word + mdg_keep_field.
__device__ __forceinline__ uint64_t mdg_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint32_t mdg_drop_threshold(float p) { return static_cast<uint32_t>(static_cast<double>(p) * 65536.0); }
__device__ __forceinline__ uint64_t mdg_keep_word(uint64_t seed, uint64_t group) { return mdg_mix64(seed * 0x100000001B3ull + group); }
__device__ __forceinline__ bool mdg_keep_field(uint64_t word, int e, uint32_t thr) { return ((word >> (16 * e)) & 0xFFFFull) >= thr; }
__device__ __forceinline__ bool mdg_keep(uint64_t seed, uint64_t index, uint32_t thr) {
  return mdg_keep_field(mdg_keep_word(seed, index >> 2), static_cast<int>(index & 3), thr);
}

// wave-wide reductions over all 64 lanes
__device__ __forceinline__ float mdg_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float mdg_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
