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

// ---- device helpers ----------------------------------------------------------------------
// hi/lo bf16 split of an fp32 value: x ~= hi + lo with |x - hi - lo| <= 2^-18 |x|.
__device__ __forceinline__ void mdg_split_bf16(float x, __bf16& hi, __bf16& lo) {
  hi = (__bf16)x;
  lo = (__bf16)(x - (float)hi);
}

__device__ __forceinline__ float mdg_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float mdg_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

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
