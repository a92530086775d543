// Voxel parent index after a strided sparse 3D convolution (reference:
// modules/multimodal/modules.py:176-198, torchsparse 1.1 `sphashquery(sphash(in), sphash(out))`): for every
// input voxel, the index of the output voxel that holds its coordinates floored to the output stride.
// The mappings (point -> views -> pixels) of the input voxels are then merged onto their parents
// (`select_points(idx, mode='merge')`).
//
// Exact integer work: an open-addressing table of int32 row ids over the OUTPUT voxels, keyed by the full
// 4 x int32 coordinate row (the table stores the row id and compares the coordinates themselves, so there
// is no packing limit and no false match); capacity = next power of two >= 2 n_out, linear probing.
// Output coordinate rows are unique by construction (a sparse tensor); should duplicates occur, the
// smallest row id wins (atomicMin), which keeps the result deterministic.
#include "dva_common.h"

namespace dva {

__device__ __forceinline__ uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ uint32_t hash_coords(int4 c) {
  uint32_t h = mix32((uint32_t)c.x + 0x9e3779b9u);
  h = mix32(h ^ ((uint32_t)c.y + 0x7f4a7c15u));
  h = mix32(h ^ ((uint32_t)c.z + 0x94d049bbu));
  return mix32(h ^ ((uint32_t)c.w + 0x2545f491u));
}
__device__ __forceinline__ bool same(int4 a, int4 b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }

// floor(a / s) * s for s > 0 (the reference computes ((x.float() / s).floor() * s).int(), identical for
// |x| < 2^24)
__device__ __forceinline__ int floor_to(int a, int s) {
  int q = a / s;
  if ((a % s != 0) && (a < 0)) --q;
  return q * s;
}

__global__ __launch_bounds__(256) void voxel_insert_kernel(const int4* __restrict__ coords, int64_t n,
                                                            int32_t* __restrict__ table, uint32_t mask) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int4 c = coords[i];
    uint32_t h = hash_coords(c) & mask;
    for (;;) {
      const int32_t old = atomicCAS(&table[h], -1, (int32_t)i);
      if (old == -1) break;
      if (same(coords[old], c)) {   // duplicate coordinate row: keep the smallest id
        atomicMin(&table[h], (int32_t)i);
        break;
      }
      h = (h + 1) & mask;
    }
  }
}

__global__ __launch_bounds__(256) void voxel_query_kernel(const int4* __restrict__ in_coords, int64_t n_in,
                                                           const int4* __restrict__ out_coords,
                                                           const int32_t* __restrict__ table, uint32_t mask,
                                                           int stride, int batch_col,
                                                           int64_t* __restrict__ idx) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_in;
       i += (int64_t)gridDim.x * blockDim.x) {
    int4 c = in_coords[i];
    if (batch_col != 0) c.x = floor_to(c.x, stride);
    if (batch_col != 1) c.y = floor_to(c.y, stride);
    if (batch_col != 2) c.z = floor_to(c.z, stride);
    if (batch_col != 3) c.w = floor_to(c.w, stride);
    uint32_t h = hash_coords(c) & mask;
    int64_t found = -1;
    for (;;) {
      const int32_t j = table[h];
      if (j == -1) break;
      if (same(out_coords[j], c)) {
        found = j;
        break;
      }
      h = (h + 1) & mask;
    }
    idx[i] = found;
  }
}


// Kernel map of a sparse convolution (torchsparse 1.1 `sphash(dst, offsets)` + `sphashquery`): for every
// destination voxel j and kernel offset k, the source voxel whose coordinates are dst[j] + offsets[k]
// (batch column untouched), or -1.  One thread per (k, j); the table is over the SOURCE voxels.
__global__ __launch_bounds__(256) void voxel_kernel_map_kernel(const int4* __restrict__ src_coords,
                                                                const int4* __restrict__ dst_coords,
                                                                int64_t n_dst, const int32_t* __restrict__ offsets,
                                                                int K, const int32_t* __restrict__ table,
                                                                uint32_t mask, int32_t* __restrict__ nbr) {
  const int64_t total = n_dst * (int64_t)K;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(t / n_dst);
    const int64_t j = t - (int64_t)k * n_dst;
    int4 c = dst_coords[j];
    c.x += offsets[3 * k];
    c.y += offsets[3 * k + 1];
    c.z += offsets[3 * k + 2];
    uint32_t h = hash_coords(c) & mask;
    int32_t found = -1;
    for (;;) {
      const int32_t i = table[h];
      if (i == -1) break;
      if (same(src_coords[i], c)) {
        found = i;
        break;
      }
      h = (h + 1) & mask;
    }
    nbr[t] = found;
  }
}

static inline uint64_t table_capacity(int64_t n_out) {
  uint64_t cap = 64;
  while (cap < 2 * (uint64_t)n_out) cap <<= 1;
  return cap;
}
static inline int grid_of(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace dva

using namespace dva;

extern "C" {

int64_t dva_voxel_parent_workspace_bytes(int64_t n_out) {
  if (n_out < 0) return DVA_ERR_INVALID;
  if (n_out > 0x3fffffffLL) return DVA_ERR_UNSUPPORTED;
  return (int64_t)(table_capacity(n_out) * sizeof(int32_t));
}

int dva_voxel_parent_index(const int32_t* in_coords, int64_t n_in, const int32_t* out_coords, int64_t n_out,
                           int32_t stride_out, int32_t batch_col, int64_t* idx, void* workspace,
                           int64_t workspace_bytes, void* stream) {
  if (n_in < 0 || n_out < 0 || stride_out <= 0 || batch_col < -1 || batch_col > 3) return DVA_ERR_INVALID;
  if (n_out > 0x3fffffffLL) return DVA_ERR_UNSUPPORTED;
  if (n_in == 0) return DVA_OK;
  if (!in_coords || !idx || !workspace) return DVA_ERR_INVALID;
  if (n_out > 0 && !out_coords) return DVA_ERR_INVALID;
  if (((uintptr_t)in_coords | (uintptr_t)out_coords) & 15) return DVA_ERR_INVALID;  // rows are read as int4
  const uint64_t cap = table_capacity(n_out);
  if ((int64_t)(cap * sizeof(int32_t)) > workspace_bytes) return DVA_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  int32_t* table = (int32_t*)workspace;
  if (hipMemsetAsync(table, 0xff, cap * sizeof(int32_t), s) != hipSuccess) return DVA_ERR_LAUNCH;
  if (n_out > 0)
    hipLaunchKernelGGL(voxel_insert_kernel, dim3(grid_of(n_out)), dim3(256), 0, s, (const int4*)out_coords,
                       n_out, table, (uint32_t)(cap - 1));
  hipLaunchKernelGGL(voxel_query_kernel, dim3(grid_of(n_in)), dim3(256), 0, s, (const int4*)in_coords, n_in,
                     (const int4*)out_coords, table, (uint32_t)(cap - 1), stride_out, batch_col, idx);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_voxel_kernel_map(const int32_t* src_coords, int64_t n_src, const int32_t* dst_coords, int64_t n_dst,
                         const int32_t* offsets, int32_t K, int32_t* nbr, void* workspace,
                         int64_t workspace_bytes, void* stream) {
  if (n_src < 0 || n_dst < 0 || K <= 0) return DVA_ERR_INVALID;
  if (n_src > 0x3fffffffLL || n_dst > 0x3fffffffLL) return DVA_ERR_UNSUPPORTED;
  if (n_dst == 0) return DVA_OK;
  if (!dst_coords || !offsets || !nbr || !workspace) return DVA_ERR_INVALID;
  if (n_src > 0 && !src_coords) return DVA_ERR_INVALID;
  if (((uintptr_t)src_coords | (uintptr_t)dst_coords) & 15) return DVA_ERR_INVALID;
  const uint64_t cap = table_capacity(n_src);
  if ((int64_t)(cap * sizeof(int32_t)) > workspace_bytes) return DVA_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  int32_t* table = (int32_t*)workspace;
  if (hipMemsetAsync(table, 0xff, cap * sizeof(int32_t), s) != hipSuccess) return DVA_ERR_LAUNCH;
  if (n_src > 0)
    hipLaunchKernelGGL(voxel_insert_kernel, dim3(grid_of(n_src)), dim3(256), 0, s, (const int4*)src_coords,
                       n_src, table, (uint32_t)(cap - 1));
  hipLaunchKernelGGL(voxel_kernel_map_kernel, dim3(grid_of(n_dst * K)), dim3(256), 0, s,
                     (const int4*)src_coords, (const int4*)dst_coords, n_dst, offsets, (int)K, table,
                     (uint32_t)(cap - 1), nbr);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // extern "C"
