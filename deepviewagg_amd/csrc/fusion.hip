// Concatenation fusion of the pooled modality features into the 3D features (reference: modules/multimodal/fusion.py
// 'concatenation': torch.cat((x_main, x_mod), dim=-1)) with the dtype promotion of torch.cat folded in: x_main fp32
// [N][Ca], x_mod fp32 or bf16 [N][Cb] -> out fp32 [N][Ca + Cb] in one pass; the backward splits and casts in one pass.
// One thread per (row, 4-column group): 16-byte stores, no index division (2-D blocks).
#include "dva_common.h"

namespace dva {

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void concat_cast_kernel(float* __restrict__ a, T* __restrict__ b,
                                                          float* __restrict__ cat, int64_t N, int Ca, int Cb) {
  const int cg = threadIdx.x;                       // column group of 4
  const int C = Ca + Cb;
  if (cg * 4 >= C) return;
  const int64_t rows_per_grid = (int64_t)gridDim.x * blockDim.y;
  for (int64_t row = (int64_t)blockIdx.x * blockDim.y + threadIdx.y; row < N; row += rows_per_grid) {
    float* o = cat + row * C + cg * 4;
    const int c0 = cg * 4;
    if (c0 < Ca) {
      float* src = a + row * Ca + c0;
      if (!BWD) *reinterpret_cast<float4*>(o) = *reinterpret_cast<const float4*>(src);
      else *reinterpret_cast<float4*>(src) = *reinterpret_cast<const float4*>(o);
    } else {
      T* src = b + row * Cb + (c0 - Ca);
      if (!BWD) {
        float f[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = Elt<T>::ld(src, e);
        *reinterpret_cast<float4*>(o) = make_float4(f[0], f[1], f[2], f[3]);
      } else {
        const float4 g = *reinterpret_cast<const float4*>(o);
        Elt<T>::st(src, 0, g.x);
        Elt<T>::st(src, 1, g.y);
        Elt<T>::st(src, 2, g.z);
        Elt<T>::st(src, 3, g.w);
      }
    }
  }
}

template <bool BWD>
static int concat_cast(float* a, void* b, float* cat, int64_t N, int Ca, int Cb, int dtype, hipStream_t s) {
  if (N < 0 || Ca < 0 || Cb < 0 || (Ca & 3) || (Cb & 3) || Ca + Cb > 1024) return DVA_ERR_INVALID;
  if (N == 0 || Ca + Cb == 0) return DVA_OK;
  if ((Ca && !a) || (Cb && !b) || !cat) return DVA_ERR_INVALID;
  int bx = 1;
  while (bx * 4 < Ca + Cb) bx <<= 1;               // threads across a row (power of two)
  const int by = 256 / bx > 0 ? 256 / bx : 1;
  int64_t blocks = (N + by - 1) / by;
  if (blocks > 16384) blocks = 16384;
  const dim3 block(bx, by), grid((int)blocks);
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
