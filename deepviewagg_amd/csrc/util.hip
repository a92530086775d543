// Small helpers of the C ABI: the copy-ceiling measurement and the zero rows of points without views.
#include "dva_common.h"

namespace dva {
// Pooled rows of points WITHOUT views are exact zeros (torch_scatter's empty-segment convention, reference
// modules/multimodal/pooling.py:870): the view kernels only write points that have views, so the rows of the others are
// cleared here -- 16-byte stores to those rows only (a tenth of the points on ragged scenes, none on the headline scene)
// instead of a fill of the whole [N, C] tensor in front of the view kernel (round 5).
__global__ __launch_bounds__(256) void zero_unseen_rows_kernel(const int64_t* __restrict__ ptr,
                                                               uint4* __restrict__ out, int64_t N, int chunks) {
  // a wavefront looks at 64 points (one pointer pair per lane), then clears the rows of those without views with all its
  // lanes: `chunks` 16-byte pieces per row, coalesced
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t p0 = wave * 64; p0 < N; p0 += n_waves * 64) {
    const int64_t p = p0 + lane;
    unsigned long long m = __ballot(p < N && ptr[p + 1] == ptr[p]);
    while (m) {
      const int b = __ffsll((long long)m) - 1;
      m &= m - 1;
      uint4* row = out + (p0 + b) * chunks;
      for (int c = lane; c < chunks; c += 64) row[c] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
}
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

extern "C" int dva_zero_unseen_rows(const int64_t* ptr, void* out, int64_t n_points, int64_t row_bytes, void* stream) {
  if (n_points < 0 || row_bytes <= 0 || (row_bytes & 15) || row_bytes > (1 << 20)) return DVA_ERR_INVALID;
  if (n_points == 0) return DVA_OK;
  if (!ptr || !out || ((uintptr_t)out & 15)) return DVA_ERR_INVALID;
  const int chunks = (int)(row_bytes / 16);
  int64_t blocks = (n_points + 255) / 256;        // 64 points per wavefront and iteration
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dva::zero_unseen_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ptr,
                     (uint4*)out, n_points, chunks);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}
