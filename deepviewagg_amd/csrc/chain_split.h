// Three-term bf16 split: fp32-class products on the bf16 matrix cores (chain_set.hip, chain_f32.hip).
//   x = hi + lo (bf16 each, 16 mantissa bits together),  W x ~ W_hi x_hi + W_lo x_hi + W_hi x_lo
// (the dropped lo . lo term is < 2^-16 relative).  Operand tables hold a 2-block matrix as hi (2 blocks) | lo (2 blocks).
#pragma once
#include "chain_common.h"

namespace dva {
namespace chain {

struct Split {
  bf16x8 hi[2], lo[2];
};
__device__ __forceinline__ Split split16(const float (&x)[16], uint32_t keep) {
  Split s;
  float r[16];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    s.hi[m] = mask8(pack8(&x[8 * m]), keep);
    const u32x4 hv = __builtin_bit_cast(u32x4, s.hi[m]);
    const uint32_t hw[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      r[8 * m + 2 * i] = x[8 * m + 2 * i] - __uint_as_float(hw[i] << 16);
      r[8 * m + 2 * i + 1] = x[8 * m + 2 * i + 1] - __uint_as_float(hw[i] & 0xffff0000u);
    }
    s.lo[m] = mask8(pack8(&r[8 * m]), keep);
  }
  return s;
}
// three-term product with the operands of matrix `so` (hi at so, so + 1; lo at so + 2, so + 3) from the LDS table
__device__ __forceinline__ f32x16 mm3(const uint4* s_ops, int so, int lane, const Split& x, f32x16 c) {
  asm volatile("" ::: "memory");
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const bf16x8 wh = lds_op(s_ops, so + m, lane), wl = lds_op(s_ops, so + 2 + m, lane);
    c = CH_MFMA(wh, x.hi[m], c);
    c = CH_MFMA(wl, x.hi[m], c);
    c = CH_MFMA(wh, x.lo[m], c);
  }
  return c;
}
// BatchNorm + LeakyReLU in fp32 (no packing): a = leaky(z G + B)
__device__ __forceinline__ void act16(const f32x16& z, const float* tab, int h, float (&a)[16]) {
  asm volatile("" ::: "memory");
  float g[16], b[16];
  tab16(tab, T_G6, h, g);
  tab16(tab, T_B6, h, b);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float t = __builtin_fmaf(z[r], g[r], b[r]);
    a[r] = __builtin_fmaf(__builtin_fabsf(t), 0.6666667f, t);
  }
}
template <typename A16>
__device__ __forceinline__ void load_rows16(__amdgpu_buffer_rsrc_t R, bool ok, uint32_t row, int h, A16& x) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = as_f4(ld128(R, ok ? row * 128u + (8u * q + 4u * h) * 4u : OOB));
    x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
  }
}
template <typename A16>
__device__ __forceinline__ void store_rows16(__amdgpu_buffer_rsrc_t R, bool ok, uint32_t row, int h, const A16& x) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    st128(R, ok ? row * 128u + (8u * q + 4u * h) * 4u : OOB, as_u4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]));
}
__device__ __forceinline__ void tileT_put_f32(bf16_t* tile, int v, int h, const float (&x)[16], bool ok) {
#pragma unroll
  for (int r = 0; r < 16; r += 2) tileT_put(tile, chan(r, h), chan(r + 1, h), v, ok ? x[r] : 0.f, ok ? x[r + 1] : 0.f);
}


}  // namespace chain
}  // namespace dva
