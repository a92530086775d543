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
#include <rocprim/iterator/counting_iterator.hpp>

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


__global__ __launch_bounds__(256) void iota32_kernel(int32_t* __restrict__ out, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (int32_t)i;
}

// Sorted row keys -> CSR pointers over ALL rows (rows nobody maps to get empty segments):
// row_ptr[r] = first position whose key is >= r.  One extra thread (i == n) closes the tail.
__global__ __launch_bounds__(256) void row_ptr_kernel(const uint32_t* __restrict__ sorted, int64_t n,
                                                       int64_t n_rows, int32_t* __restrict__ row_ptr) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i <= n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t lo = i == 0 ? 0 : (int64_t)sorted[i - 1] + 1;
    int64_t hi = i == n ? n_rows : (int64_t)sorted[i];
    if (hi > n_rows) hi = n_rows;  // contract: keys < n_rows
    for (int64_t r = lo; r <= hi; ++r) row_ptr[r] = (int32_t)i;
  }
}

__global__ __launch_bounds__(256) void ptr_diff_kernel(const int32_t* __restrict__ row_ptr,
                                                        int64_t n_rows, int32_t* __restrict__ counts) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rows;
       r += (int64_t)gridDim.x * blockDim.x)
    counts[r] = row_ptr[r + 1] - row_ptr[r];
}

struct PlanLayout {
  size_t off_iota, off_keys, off_temp, temp_bytes, total;
};

static int key_bits(int64_t n_rows) {
  int b = 1;
  while (b < 32 && ((int64_t)1 << b) < n_rows) ++b;
  return b;
}

// Row keys of 17 / 18 bits (e.g. 32 feature maps of 64 x 128): two onesweep passes of 9 bits instead of the library's
// 8 + 8 + 2 (one pass over the 33 M (row, view) pairs less).
// Below 2^20 keys the library switches to a merge sort (block sort + 2 kernels per doubling: 21 launches of 6-13 us =
// 0.15 ms for the 8.8e5 views of the reference's own S3DIS batch, profiles/r04x_s3dis_kernel_stats.csv): the plan sorts
// carry their own limit (MERGE_LIMIT) and, under 4 M keys, smaller onesweep blocks (4096 keys instead of 8192: twice
// the blocks for the same keys -- a plan of 1 M keys is only 128 of the large ones on 256 CUs) (round 5).
constexpr size_t MERGE_LIMIT = 1 << 15;
template <int BITS, int BLOCK>
using PlanSortCfg =
    rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                               rocprim::radix_sort_onesweep_config<rocprim::kernel_config<BLOCK, 8>,
                                                                   rocprim::kernel_config<BLOCK, 8>, BITS,
                                                                   rocprim::block_radix_rank_algorithm::match>,
                               MERGE_LIMIT>;
typedef PlanSortCfg<9, 1024> PlanSort9;
typedef PlanSortCfg<9, 512> PlanSort9s;
static inline bool plan_wide_digits(int bits) { return bits == 17 || bits == 18; }
// 19 / 20 bits (the anchor plan of the bilinear backward: 32 x 65 x 129 padded cells): two passes of 10 bits
typedef PlanSortCfg<10, 1024> PlanSort10;
typedef PlanSortCfg<10, 512> PlanSort10s;
static inline bool plan_wider_digits(int bits) { return bits == 19 || bits == 20; }
// everything else: the library's own (architecture-tuned) onesweep configuration with the plan's merge limit
typedef rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, MERGE_LIMIT>
    PlanSortLib;
constexpr size_t SMALL_PLAN = (size_t)1 << 22;

// values in = the view numbers 0 .. n-1 as a counting iterator (no iota array is written or read)
static hipError_t plan_sort(void* temp, size_t& tmp, const uint32_t* kin, uint32_t* kout, int32_t* vout, size_t n,
                            int bits, hipStream_t s) {
  rocprim::counting_iterator<int32_t> vin(0);
  const bool small = n < SMALL_PLAN;
  if (plan_wide_digits(bits))
    return small ? rocprim::radix_sort_pairs<PlanSort9s>(temp, tmp, kin, kout, vin, vout, n, 0, bits, s)
                 : rocprim::radix_sort_pairs<PlanSort9>(temp, tmp, kin, kout, vin, vout, n, 0, bits, s);
  if (plan_wider_digits(bits))
    return small ? rocprim::radix_sort_pairs<PlanSort10s>(temp, tmp, kin, kout, vin, vout, n, 0, bits, s)
                 : rocprim::radix_sort_pairs<PlanSort10>(temp, tmp, kin, kout, vin, vout, n, 0, bits, s);
  return rocprim::radix_sort_pairs<PlanSortLib>(temp, tmp, kin, kout, vin, vout, n, 0, bits, s);
}

static int plan_layout(int64_t n, int bits, PlanLayout* L) {
  size_t tmp = 0;
  uint32_t* nk = nullptr;
  int32_t* nv = nullptr;
  if (plan_sort(nullptr, tmp, nk, nk, nv, (size_t)n, bits, (hipStream_t)0) != hipSuccess)
    return DVA_ERR_LAUNCH;
  size_t off = 0;
  L->off_iota = off;                                       // (no view-number array: a counting iterator feeds the sort)
  L->off_keys = off;  off += align256((size_t)n * 4);
  L->off_temp = off;
  L->temp_bytes = tmp;
  off += align256(tmp);
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

int64_t dva_row_plan_workspace_bytes(int64_t n_views, int64_t n_rows) {
  if (n_views < 0 || n_rows < 0) return DVA_ERR_INVALID;
  if (n_views > 0x7fffffffLL || n_rows > 0x7fffffffLL) return DVA_ERR_UNSUPPORTED;
  if (n_views == 0) return 256;
  PlanLayout L;
  int rc = plan_layout(n_views, key_bits(n_rows), &L);
  if (rc) return rc;
  return (int64_t)L.total;
}

__global__ __launch_bounds__(256) void plan_inverse_kernel(const int32_t* __restrict__ perm, int32_t* __restrict__ inv,
                                                           int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    inv[perm[i]] = (int32_t)i;
}

int dva_plan_inverse(const int32_t* perm, int32_t* inv, int64_t n_views, void* stream) {
  if (n_views < 0) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!perm || !inv) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(plan_inverse_kernel, dim3(grid_for(n_views)), dim3(256), 0, (hipStream_t)stream, perm, inv, n_views);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_row_plan(const int32_t* row_idx, int64_t n_views, int64_t n_rows, int32_t* perm,
                 int32_t* row_ptr, int32_t* counts, void* workspace, int64_t workspace_bytes,
                 void* stream) {
  if (n_views < 0 || n_rows < 0 || !row_ptr) return DVA_ERR_INVALID;
  if (n_views > 0x7fffffffLL || n_rows > 0x7fffffffLL) return DVA_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (n_views == 0) {
    if (hipMemsetAsync(row_ptr, 0, (size_t)(n_rows + 1) * 4, s) != hipSuccess) return DVA_ERR_LAUNCH;
    if (counts && n_rows && hipMemsetAsync(counts, 0, (size_t)n_rows * 4, s) != hipSuccess)
      return DVA_ERR_LAUNCH;
    return DVA_OK;
  }
  if (!row_idx || !perm || !workspace) return DVA_ERR_INVALID;
  PlanLayout L;
  const int bits = key_bits(n_rows);
  int rc = plan_layout(n_views, bits, &L);
  if (rc) return rc;
  if ((int64_t)L.total > workspace_bytes) return DVA_ERR_INVALID;
  char* ws = (char*)workspace;
  uint32_t* kout = (uint32_t*)(ws + L.off_keys);
  size_t tmp = L.temp_bytes;
  if (plan_sort(ws + L.off_temp, tmp, (const uint32_t*)row_idx, kout, perm, (size_t)n_views, bits, s) != hipSuccess)
    return DVA_ERR_LAUNCH;
  hipLaunchKernelGGL(row_ptr_kernel, dim3(grid_for(n_views + 1)), dim3(256), 0, s, kout, n_views,
                     n_rows, row_ptr);
  if (counts)
    hipLaunchKernelGGL(ptr_diff_kernel, dim3(grid_for(n_rows)), dim3(256), 0, s, row_ptr, n_rows,
                       counts);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // extern "C"
