// Lexicographic integer-key primitives for gfx950 (reference: utils/multimodal.py:36-94 lexargsort /
// lexargunique on a composite int64 key, :253-323 lex ops).
//
// The composite key itself (sum_i a_i * prod_{j>i}(max_j+1), multimodal.py:141-156) is built by the
// caller; here: stable LSD radix sort of (key, original index) pairs with rocPRIM, then a
// head-of-run flag pass + stream compaction for "unique with first index".  Stability makes the
// result deterministic and equal to numpy's np.unique(return_index=True): the first element of a
// run of equal keys carries the smallest original index.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

#include "dva_common.h"

namespace dva {

__global__ __launch_bounds__(256) void iota_kernel(int64_t* __restrict__ out, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = i;
}

__global__ __launch_bounds__(256) void head_flags_kernel(const int64_t* __restrict__ sorted,
                                                          uint8_t* __restrict__ flags, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    flags[i] = (i == 0 || sorted[i] != sorted[i - 1]) ? 1 : 0;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct LexLayout {
  size_t off_iota, off_keys, off_vals, off_flags, off_temp, temp_bytes, total;
};

static int lex_layout(int64_t n, LexLayout* L) {
  size_t sort_tmp = 0, sel_tmp = 0;
  int64_t* nul = nullptr;
  uint8_t* nulf = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, sort_tmp, nul, nul, nul, nul, (size_t)n, 0, 64,
                                (hipStream_t)0) != hipSuccess)
    return DVA_ERR_LAUNCH;
  if (rocprim::select(nullptr, sel_tmp, nul, nulf, nul, nul, (size_t)n, (hipStream_t)0) != hipSuccess)
    return DVA_ERR_LAUNCH;
  size_t off = 0;
  L->off_iota = off;  off += align256((size_t)n * 8);
  L->off_keys = off;  off += align256((size_t)n * 8);
  L->off_vals = off;  off += align256((size_t)n * 8);
  L->off_flags = off; off += align256((size_t)n);
  L->off_temp = off;
  L->temp_bytes = sort_tmp > sel_tmp ? sort_tmp : sel_tmp;
  off += align256(L->temp_bytes);
  L->total = off;
  return DVA_OK;
}

static inline int grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace dva

using namespace dva;

extern "C" {

int64_t dva_lex_workspace_bytes(int64_t n) {
  if (n < 0) return DVA_ERR_INVALID;
  if (n == 0) return 256;
  LexLayout L;
  int rc = lex_layout(n, &L);
  if (rc) return rc;
  return (int64_t)L.total;
}

int dva_argsort_i64(const int64_t* keys, int64_t n, int64_t* order, int64_t* keys_sorted,
                    void* workspace, int64_t workspace_bytes, void* stream) {
  if (n < 0) return DVA_ERR_INVALID;
  if (n == 0) return DVA_OK;
  if (!keys || !order || !workspace) return DVA_ERR_INVALID;
  LexLayout L;
  int rc = lex_layout(n, &L);
  if (rc) return rc;
  if ((int64_t)L.total > workspace_bytes) return DVA_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  int64_t* iota = (int64_t*)(ws + L.off_iota);
  int64_t* kout = keys_sorted ? keys_sorted : (int64_t*)(ws + L.off_keys);
  hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n)), dim3(256), 0, s, iota, n);
  size_t tmp = L.temp_bytes;
  if (rocprim::radix_sort_pairs(ws + L.off_temp, tmp, keys, kout, iota, order, (size_t)n, 0, 64, s) !=
      hipSuccess)
    return DVA_ERR_LAUNCH;
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_argunique_i64(const int64_t* keys, int64_t n, int64_t* first, int64_t* n_unique_dev,
                      void* workspace, int64_t workspace_bytes, void* stream) {
  if (n < 0 || !n_unique_dev) return DVA_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    if (hipMemsetAsync(n_unique_dev, 0, sizeof(int64_t), s) != hipSuccess) return DVA_ERR_LAUNCH;
    return DVA_OK;
  }
  if (!keys || !first || !workspace) return DVA_ERR_INVALID;
  LexLayout L;
  int rc = lex_layout(n, &L);
  if (rc) return rc;
  if ((int64_t)L.total > workspace_bytes) return DVA_ERR_INVALID;
  char* ws = (char*)workspace;
  int64_t* iota = (int64_t*)(ws + L.off_iota);
  int64_t* kout = (int64_t*)(ws + L.off_keys);
  int64_t* vout = (int64_t*)(ws + L.off_vals);
  uint8_t* flags = (uint8_t*)(ws + L.off_flags);
  hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n)), dim3(256), 0, s, iota, n);
  size_t tmp = L.temp_bytes;
  if (rocprim::radix_sort_pairs(ws + L.off_temp, tmp, keys, kout, iota, vout, (size_t)n, 0, 64, s) !=
      hipSuccess)
    return DVA_ERR_LAUNCH;
  hipLaunchKernelGGL(head_flags_kernel, dim3(grid_for(n)), dim3(256), 0, s, kout, flags, n);
  tmp = L.temp_bytes;
  if (rocprim::select(ws + L.off_temp, tmp, vout, flags, first, n_unique_dev, (size_t)n, s) !=
      hipSuccess)
    return DVA_ERR_LAUNCH;
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // extern "C"
