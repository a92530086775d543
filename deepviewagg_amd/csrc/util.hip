// Measurement helpers of the C ABI (not on the product path).
#include "dva_common.h"

namespace dva {
// Device copy (read + write): the practical HBM ceiling the roofline fractions are quoted beside.  One contiguous chunk of
// 4 x 256 x 16 bytes per block, blocks sweeping the buffer in launch order, 4 non-temporal loads in flight per thread
// before the first (non-temporal) store: 6.35 TB/s on MI355X (MI355X_MICROARCH.md: 6.29 for this pattern; 8 TB/s spec).
// Measured alternatives: the same without the non-temporal hint 5.9 TB/s; a persistent grid of 4096 blocks striding
// through the buffer (what this entry did until round 3) 4.9 TB/s -- 16 wavefronts per CU x 4 KiB do not keep enough
// bytes in flight; hipMemcpyAsync device-to-device 4.8 TB/s.
typedef unsigned int copy_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_kernel(const copy_u4* __restrict__ src, copy_u4* __restrict__ dst,
                                                    int64_t n) {
  const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  copy_u4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t i = base + k * 256;
    if (i < n) v[k] = __builtin_nontemporal_load(src + i);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t i = base + k * 256;
    if (i < n) __builtin_nontemporal_store(v[k], dst + i);
  }
}
}  // namespace dva

extern "C" int dva_copy_ceiling(const void* src, void* dst, int64_t nbytes, void* stream) {
  if (nbytes < 0 || (nbytes & 15)) return DVA_ERR_INVALID;
  if (nbytes == 0) return DVA_OK;
  if (!src || !dst) return DVA_ERR_INVALID;
  const int64_t n = nbytes / 16;
  const int64_t blocks = (n + 1023) / 1024;
  if (blocks > 0x7fffffffll) return DVA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(dva::copy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const dva::copy_u4*)src, (dva::copy_u4*)dst, n);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}
