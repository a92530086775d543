// Concatenation fusion of the pooled modality features into the 3D features (reference: modules/multimodal/fusion.py
// 'concatenation': torch.cat((x_main, x_mod), dim=-1)) with the dtype promotion of torch.cat folded in: x_main fp32
// [N][Ca], x_mod fp32 or bf16 [N][Cb] -> out fp32 [N][Ca + Cb] in one pass; the backward splits and casts in one pass.
// One thread per 4-column group: 16-byte stores.
#include "dva_common.h"

namespace dva {

// Flat mapping (round 4): thread t of the grid owns the 4-column groups t, t + T, t + 2 T, ... of the [N][C / 4] group
// matrix; (row, group) advance by the constant (T / cpg, T % cpg) per step, so there is no division in the loop and no
// idle lane (the 2-D form of rounds 1-3 gave a row a power-of-two number of threads: 17 of 32 active at C = 68, 129 of
// 256 at C = 516: 0.25 ms per direction at N = 2^20, C = 68, against 0.1 ms for the bytes).  bf16 groups move as 8 bytes.
template <typename T>
__device__ __forceinline__ float4 ld4(const T* p);
template <>
__device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <>
__device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                     __uint_as_float(v.y & 0xffff0000u));
}
template <typename T>
__device__ __forceinline__ void st4(T* p, const float4& g);
template <>
__device__ __forceinline__ void st4<float>(float* p, const float4& g) { *reinterpret_cast<float4*>(p) = g; }
template <>
__device__ __forceinline__ void st4<bf16_t>(bf16_t* p, const float4& g) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(g.x, g.y), pack_bf16x2(g.z, g.w));
}

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void concat_cast_kernel(float* __restrict__ a, T* __restrict__ b,
                                                          float* __restrict__ cat, int64_t N, int Ca, int Cb) {
  const int C = Ca + Cb, cpg = C >> 2, cga = Ca >> 2;
  const int64_t T_ = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t row = t0 / cpg;
  int cg = (int)(t0 - row * cpg);
  const int64_t d_row = T_ / cpg;
  const int d_cg = (int)(T_ - d_row * cpg);
  while (row < N) {
    float* o = cat + row * C + cg * 4;
    if (cg < cga) {
      float* src = a + row * Ca + cg * 4;
      if (!BWD) *reinterpret_cast<float4*>(o) = *reinterpret_cast<const float4*>(src);
      else *reinterpret_cast<float4*>(src) = *reinterpret_cast<const float4*>(o);
    } else {
      T* src = b + row * Cb + (cg - cga) * 4;
      if (!BWD) *reinterpret_cast<float4*>(o) = ld4<T>(src);
      else st4<T>(src, *reinterpret_cast<const float4*>(o));
    }
    row += d_row;
    cg += d_cg;
    if (cg >= cpg) {
      cg -= cpg;
      ++row;
    }
  }
}

template <bool BWD>
static int concat_cast(float* a, void* b, float* cat, int64_t N, int Ca, int Cb, int dtype, hipStream_t s) {
  if (N < 0 || Ca < 0 || Cb < 0 || (Ca & 3) || (Cb & 3) || Ca + Cb > 1024) return DVA_ERR_INVALID;
  if (N == 0 || Ca + Cb == 0) return DVA_OK;
  if ((Ca && !a) || (Cb && !b) || !cat) return DVA_ERR_INVALID;
  if (dtype == DVA_BF16 && ((uintptr_t)b & 7)) return DVA_ERR_UNSUPPORTED;     // 8-byte moves of the bf16 groups
  const int64_t groups = N * ((Ca + Cb) >> 2);
  int64_t blocks = (groups + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  const dim3 block(256), grid((int)blocks);
  if (dtype == DVA_F32)
    hipLaunchKernelGGL((concat_cast_kernel<float, BWD>), grid, block, 0, s, a, (float*)b, cat, N, Ca, Cb);
  else if (dtype == DVA_BF16)
    hipLaunchKernelGGL((concat_cast_kernel<bf16_t, BWD>), grid, block, 0, s, a, (bf16_t*)b, cat, N, Ca, Cb);
  else
    return DVA_ERR_INVALID;
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // namespace dva

extern "C" {

int dva_concat_cast_fwd(const float* x_main, const void* x_mod, float* out, int64_t N, int32_t C_main, int32_t C_mod,
                        int32_t mod_dtype, void* stream) {
  return dva::concat_cast<false>((float*)x_main, (void*)x_mod, out, N, C_main, C_mod, mod_dtype, (hipStream_t)stream);
}

int dva_concat_cast_bwd(const float* grad_out, float* grad_main, void* grad_mod, int64_t N, int32_t C_main,
                        int32_t C_mod, int32_t mod_dtype, void* stream) {
  return dva::concat_cast<true>(grad_main, grad_mod, (float*)grad_out, N, C_main, C_mod, mod_dtype, (hipStream_t)stream);
}

}  // extern "C"
