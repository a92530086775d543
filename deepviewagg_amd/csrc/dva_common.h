// Internal helpers shared by the gfx950 kernels of libdva_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "../../include/dva.h"

#define DVA_WAVE 64

#define DVA_CHECK_LAUNCH()                           \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) return DVA_ERR_LAUNCH;    \
  } while (0)

namespace dva {

// experiment switches of the kernels' launch configuration (environment, read once by the caller)
static inline int tune_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

typedef uint16_t bf16_t;  // raw bfloat16 bits

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// float -> bfloat16, round-to-nearest-even (same rounding as torch): gfx950 has the packed hardware
// conversion v_cvt_pk_bf16_f32 (one instruction per pair, NaN stays NaN).
typedef __bf16 dva_bf16x2 __attribute__((ext_vector_type(2)));
typedef float dva_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const dva_f32x2 v = {lo, hi};
  const dva_bf16x2 b = __builtin_convertvector(v, dva_bf16x2);
  return __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

template <typename T>
struct Elt;
template <>
struct Elt<float> {
  static __device__ __forceinline__ float ld(const float* p, int64_t i) { return p[i]; }
  static __device__ __forceinline__ void st(float* p, int64_t i, float v) { p[i] = v; }
};
template <>
struct Elt<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p, int64_t i) { return bf2f(p[i]); }
  static __device__ __forceinline__ void st(bf16_t* p, int64_t i, float v) { p[i] = f2bf(v); }
};

// expand_group_feat (pooling.py:737-755): the first (C mod G) groups own floor(C/G)+1 channels,
// the others floor(C/G).  Returns the group of channel c.
__host__ __device__ __forceinline__ int group_of_channel(int c, int C, int G) {
  if (G <= 1) return 0;
  if (G >= C) return c;
  const int base = C / G, rem = C % G;
  const int split = rem * (base + 1);
  return c < split ? c / (base + 1) : rem + (c - split) / base;
}
// first channel of group g
__host__ __device__ __forceinline__ int group_begin(int g, int C, int G) {
  if (g >= G) return C;
  if (G <= 1) return 0;
  if (G >= C) return g;
  const int base = C / G, rem = C % G;
  return g < rem ? g * (base + 1) : rem * (base + 1) + (g - rem) * base;
}

// 8-byte packed gather index of one atom: {int32 image, int16 x, int16 y}
struct __attribute__((aligned(8))) PackedIdx {
  int32_t img;
  int16_t x;
  int16_t y;
};

static inline int blocks_for(int64_t n, int threads) { return (int)((n + threads - 1) / threads); }

}  // namespace dva
