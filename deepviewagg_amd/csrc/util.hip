// Measurement helpers of the C ABI (not on the product path).
#include "dva_common.h"

namespace dva {
// float4 grid-stride copy: the practical HBM ceiling (read + write) the roofline fractions are quoted beside
// (MI355X_MICROARCH.md: 6.29 TB/s measured for this pattern, 8 TB/s spec).
__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // four 16-byte loads in flight per thread before the first store
  for (; i + 3 * stride < n; i += 4 * stride) {
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a;
    dst[i + stride] = b;
    dst[i + 2 * stride] = c;
    dst[i + 3 * stride] = d;
  }
  for (; i < n; i += stride) dst[i] = src[i];
}
}  // namespace dva

extern "C" int dva_copy_ceiling(const void* src, void* dst, int64_t nbytes, void* stream) {
  if (nbytes < 0 || (nbytes & 15)) return DVA_ERR_INVALID;
  if (nbytes == 0) return DVA_OK;
  if (!src || !dst) return DVA_ERR_INVALID;
  const int64_t n = nbytes / 16;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dva::copy_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)src,
                     (uint4*)dst, n);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}
