// Exact k-nearest-neighbour search on a uniform hash grid + per-view occlusion counts: the device side of
// NeighborhoodBasedMappingFeatures (reference: core/data_transform/multimodal/image.py:431-612, which uses
// KeOps `argKmin` over all pairs, or FAISS).  Mapping features 7-8 of the `in_map: 8` configurations:
//   density   ~ (k+1) / (pi d_k^2) with d_k the distance to the k-th neighbour (the point itself included),
//   occlusion = (1 + #neighbours seen by the same image) / (k+1) for every (point, image) view.
//
// Contract (what `argKmin(k)` of the fp32 squared distances ((dx*dx + dy*dy) + dz*dz) returns when ties are
// broken by the lower index): neighbours sorted ascending by (d2, index); the query point itself is its own
// first neighbour.  Exactness: cells are visited in Chebyshev shells around the query's cell; after shell R
// every unvisited point is farther than R*cell - slack, so the search stops as soon as the current k-th
// distance is below that bound (slack absorbs the float rounding of the cell assignment).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "dva_common.h"

namespace dva {

constexpr int KNN_TPB = 64;      // one wavefront per block: the candidate lists live in LDS, [slot][thread]
constexpr int KNN_KMAX = 128;     // k <= 64: 32 KB of candidate lists per block; k <= 128 (BiasuttiVisibility, k = 75): 64 KB
constexpr int CELL_BITS = 21;
constexpr int64_t CELL_BIAS = 1 << 20;

__device__ __forceinline__ uint64_t cell_key(int64_t cx, int64_t cy, int64_t cz) {
  return ((uint64_t)(cx + CELL_BIAS) << (2 * CELL_BITS)) | ((uint64_t)(cy + CELL_BIAS) << CELL_BITS) |
         (uint64_t)(cz + CELL_BIAS);
}
__device__ __forceinline__ uint32_t hash64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return (uint32_t)k;
}
__device__ __forceinline__ int cell_of(float p, float origin, float inv_cell) {
  return (int)floorf((p - origin) * inv_cell);
}

// bbox = min xyz | max xyz (device memory, written by the caller's stream)
__global__ __launch_bounds__(256) void knn_keys_kernel(const float* __restrict__ xyz, int64_t n,
                                                        const float* __restrict__ bbox, float inv_cell,
                                                        uint64_t* __restrict__ keys, int32_t* __restrict__ ids) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    keys[i] = cell_key(cell_of(xyz[3 * i], bbox[0], inv_cell), cell_of(xyz[3 * i + 1], bbox[1], inv_cell),
                       cell_of(xyz[3 * i + 2], bbox[2], inv_cell));
    ids[i] = (int32_t)i;
  }
}

// sorted copy of the points (x, y, z, original index) + cell table: key -> first sorted position
__global__ __launch_bounds__(256) void knn_cells_kernel(const float* __restrict__ xyz,
                                                         const uint64_t* __restrict__ keys_sorted,
                                                         const int32_t* __restrict__ perm, int64_t n,
                                                         float4* __restrict__ pts, uint64_t* __restrict__ tkeys,
                                                         int32_t* __restrict__ tvals, uint32_t mask) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t o = perm[i];
    pts[i] = make_float4(xyz[3 * (int64_t)o], xyz[3 * (int64_t)o + 1], xyz[3 * (int64_t)o + 2], __int_as_float(o));
    const uint64_t k = keys_sorted[i];
    if (i == 0 || keys_sorted[i - 1] != k) {
      uint32_t h = hash64(k) & mask;
      for (;;) {
        const unsigned long long old = atomicCAS((unsigned long long*)&tkeys[h], ~0ULL, (unsigned long long)k);
        if (old == ~0ULL || old == k) {
          tvals[h] = (int32_t)i;
          break;
        }
        h = (h + 1) & mask;
      }
    }
  }
}

__device__ __forceinline__ int cell_start(const uint64_t* __restrict__ tkeys, const int32_t* __restrict__ tvals,
                                          uint32_t mask, uint64_t k) {
  uint32_t h = hash64(k) & mask;
  for (;;) {
    const uint64_t t = tkeys[h];
    if (t == k) return tvals[h];
    if (t == ~0ULL) return -1;
    h = (h + 1) & mask;
  }
}

template <int KCAP>
__global__ __launch_bounds__(KNN_TPB) void knn_query_kernel(
    const float4* __restrict__ pts, const uint64_t* __restrict__ keys_sorted, int64_t n,
    const float* __restrict__ bbox, float cell, const uint64_t* __restrict__ tkeys,
    const int32_t* __restrict__ tvals, uint32_t mask, int k, int max_shell, uint8_t* __restrict__ done,
    int32_t* __restrict__ nbr, float* __restrict__ d2_out) {
  __shared__ float s_d[KCAP][KNN_TPB];
  __shared__ int32_t s_i[KCAP][KNN_TPB];
  const int t = threadIdx.x;
  const float inv_cell = 1.f / cell;
  const float ext = fmaxf(fmaxf(bbox[3] - bbox[0], bbox[4] - bbox[1]), bbox[5] - bbox[2]);
  const float slack = 1e-5f * (ext + cell);
  const int r_all = (int)(ext * inv_cell) + 2;          // shells that cover the whole cloud
  const int r_max = max_shell < r_all ? max_shell : r_all;
  // queries in sorted (cell) order: the threads of a wavefront walk the same few cells
  for (int64_t qs = blockIdx.x * (int64_t)KNN_TPB + t; qs < n; qs += (int64_t)gridDim.x * KNN_TPB) {
    const float4 q = pts[qs];
    const int qi = __float_as_int(q.w);
    if (done && done[qi]) continue;      // finished on a finer grid
    const int cx = cell_of(q.x, bbox[0], inv_cell), cy = cell_of(q.y, bbox[1], inv_cell),
              cz = cell_of(q.z, bbox[2], inv_cell);
    int cnt = 0, wslot = 0;
    float wd = -1.f;
    int wi = -1;
    bool finished = false;
    for (int R = 0; R <= r_max; ++R) {
      for (int dx = -R; dx <= R; ++dx)
        for (int dy = -R; dy <= R; ++dy) {
          const bool face = (dx == -R || dx == R || dy == -R || dy == R);
          const int step = face ? 1 : 2 * R;          // interior columns: only the two end caps
          for (int dz = -R; dz <= R; dz += (step > 0 ? step : 1)) {
            const uint64_t key = cell_key(cx + dx, cy + dy, cz + dz);
            int s = cell_start(tkeys, tvals, mask, key);
            if (s < 0) continue;
            for (; s < n && keys_sorted[s] == key; ++s) {
              const float4 p = pts[s];
              const float ex = q.x - p.x, ey = q.y - p.y, ez = q.z - p.z;
              const float d = (ex * ex + ey * ey) + ez * ez;
              const int j = __float_as_int(p.w);
              if (cnt < k) {
                s_d[cnt][t] = d;
                s_i[cnt][t] = j;
                if (d > wd || (d == wd && j > wi)) { wd = d; wi = j; wslot = cnt; }
                ++cnt;
              } else if (d < wd || (d == wd && j < wi)) {
                s_d[wslot][t] = d;
                s_i[wslot][t] = j;
                wd = -1.f; wi = -1;
                for (int m = 0; m < k; ++m) {
                  const float dm = s_d[m][t];
                  const int im = s_i[m][t];
                  if (dm > wd || (dm == wd && im > wi)) { wd = dm; wi = im; wslot = m; }
                }
              }
            }
            if (R == 0) break;
          }
        }
      if (cnt == k) {
        const float bound = (float)R * cell - slack;
        if (bound > 0.f && wd < bound * bound) { finished = true; break; }
      }
    }
    // not provably complete within max_shell shells (sparse region): left to a coarser grid, unless these
    // shells already covered the whole cloud
    if (!finished && r_max < r_all) continue;
    if (done) done[qi] = 1;
    // ascending (d2, index): insertion sort of the thread's column
    for (int a = 1; a < cnt; ++a) {
      const float da = s_d[a][t];
      const int ia = s_i[a][t];
      int b = a - 1;
      while (b >= 0 && (s_d[b][t] > da || (s_d[b][t] == da && s_i[b][t] > ia))) {
        s_d[b + 1][t] = s_d[b][t];
        s_i[b + 1][t] = s_i[b][t];
        --b;
      }
      s_d[b + 1][t] = da;
      s_i[b + 1][t] = ia;
    }
    for (int m = 0; m < k; ++m) {
      nbr[(int64_t)qi * k + m] = m < cnt ? s_i[m][t] : -1;
      if (d2_out) d2_out[(int64_t)qi * k + m] = m < cnt ? s_d[m][t] : INFINITY;
    }
  }
}

// bit (image) of word (image / 64) of point p: the views of the mapping as bit sets
__global__ __launch_bounds__(256) void view_bits_kernel(const int32_t* __restrict__ view_point,
                                                         const int64_t* __restrict__ images, int64_t V,
                                                         int n_words, unsigned long long* __restrict__ bits) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
    const int64_t img = images[v];
    atomicOr(&bits[(int64_t)view_point[v] * n_words + (img >> 6)], 1ULL << (img & 63));
  }
}

// out[v][c] = (1 + #{i < k_list[c] : neighbour i of the view's point is seen by the view's image}) / (k_c + 1)
__global__ __launch_bounds__(256) void occlusion_kernel(const int32_t* __restrict__ view_point,
                                                         const int64_t* __restrict__ images, int64_t V,
                                                         const int32_t* __restrict__ nbr, int k,
                                                         const unsigned long long* __restrict__ bits, int n_words,
                                                         const int32_t* __restrict__ k_list, int n_k,
                                                         float* __restrict__ out) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = view_point[v], img = images[v];
    const int w = (int)(img >> 6);
    const unsigned long long bit = 1ULL << (img & 63);
    int seen = 0, c = 0;
    for (int i = 0; i < k && c < n_k; ++i) {
      const int32_t q = nbr[p * k + i];
      if (q >= 0 && (bits[(int64_t)q * n_words + w] & bit)) ++seen;
      while (c < n_k && i + 1 == k_list[c]) {
        out[v * n_k + c] = (float)(1 + seen) / (float)(k_list[c] + 1);
        ++c;
      }
    }
  }
}

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
struct KnnLayout {
  size_t keys, ids, keys_sorted, perm, pts, tkeys, tvals, temp, temp_bytes, total;
  uint64_t cap;
};
static int knn_layout(int64_t n, KnnLayout* L) {
  size_t tmp = 0;
  uint64_t* nk = nullptr;
  int32_t* nv = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, tmp, nk, nk, nv, nv, (size_t)n, 0, 3 * CELL_BITS, (hipStream_t)0) !=
      hipSuccess)
    return DVA_ERR_LAUNCH;
  uint64_t cap = 64;
  while (cap < 2 * (uint64_t)n) cap <<= 1;
  L->cap = cap;
  size_t off = 0;
  L->keys = off;        off += al256((size_t)n * 8);
  L->ids = off;         off += al256((size_t)n * 4);
  L->keys_sorted = off; off += al256((size_t)n * 8);
  L->perm = off;        off += al256((size_t)n * 4);
  L->pts = off;         off += al256((size_t)n * 16);
  L->tkeys = off;       off += al256(cap * 8);
  L->tvals = off;       off += al256(cap * 4);
  L->temp = off;        L->temp_bytes = tmp; off += al256(tmp);
  L->total = off;
  return DVA_OK;
}
static inline int grid256(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace dva

using namespace dva;

extern "C" {

int64_t dva_knn_workspace_bytes(int64_t n) {
  if (n < 0) return DVA_ERR_INVALID;
  if (n > 0x3fffffffLL) return DVA_ERR_UNSUPPORTED;
  if (n == 0) return 256;
  KnnLayout L;
  int rc = knn_layout(n, &L);
  if (rc) return rc;
  return (int64_t)L.total;
}

int dva_knn(const float* xyz, int64_t n, const float* bbox, float cell, int32_t k, int32_t max_shell,
            uint8_t* done, int32_t* neighbors, float* dist2, void* workspace, int64_t workspace_bytes,
            void* stream) {
  if (n < 0 || k <= 0 || k > KNN_KMAX || !(cell > 0.f) || max_shell < 1) return DVA_ERR_INVALID;
  if (n > 0x3fffffffLL) return DVA_ERR_UNSUPPORTED;
  if (n == 0) return DVA_OK;
  if (!xyz || !bbox || !neighbors || !workspace) return DVA_ERR_INVALID;
  KnnLayout L;
  int rc = knn_layout(n, &L);
  if (rc) return rc;
  if ((int64_t)L.total > workspace_bytes) return DVA_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  uint64_t* keys = (uint64_t*)(ws + L.keys);
  int32_t* ids = (int32_t*)(ws + L.ids);
  uint64_t* keys_sorted = (uint64_t*)(ws + L.keys_sorted);
  int32_t* perm = (int32_t*)(ws + L.perm);
  float4* pts = (float4*)(ws + L.pts);
  uint64_t* tkeys = (uint64_t*)(ws + L.tkeys);
  int32_t* tvals = (int32_t*)(ws + L.tvals);
  hipLaunchKernelGGL(knn_keys_kernel, dim3(grid256(n)), dim3(256), 0, s, xyz, n, bbox, 1.f / cell, keys, ids);
  size_t tmp = L.temp_bytes;
  if (rocprim::radix_sort_pairs(ws + L.temp, tmp, keys, keys_sorted, ids, perm, (size_t)n, 0, 3 * CELL_BITS, s) !=
      hipSuccess)
    return DVA_ERR_LAUNCH;
  if (hipMemsetAsync(tkeys, 0xff, L.cap * 8, s) != hipSuccess) return DVA_ERR_LAUNCH;
  hipLaunchKernelGGL(knn_cells_kernel, dim3(grid256(n)), dim3(256), 0, s, xyz, keys_sorted, perm, n, pts, tkeys,
                     tvals, (uint32_t)(L.cap - 1));
  int64_t qb = (n + KNN_TPB - 1) / KNN_TPB;
  if (qb > 256 * 32) qb = 256 * 32;
  if (k <= 64)
    hipLaunchKernelGGL(knn_query_kernel<64>, dim3((int)qb), dim3(KNN_TPB), 0, s, pts, keys_sorted, n, bbox, cell, tkeys,
                       tvals, (uint32_t)(L.cap - 1), k, max_shell, done, neighbors, dist2);
  else
    hipLaunchKernelGGL(knn_query_kernel<128>, dim3((int)qb), dim3(KNN_TPB), 0, s, pts, keys_sorted, n, bbox, cell, tkeys,
                       tvals, (uint32_t)(L.cap - 1), k, max_shell, done, neighbors, dist2);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_view_occlusion(const int32_t* view_point, const int64_t* images, int64_t n_views, int64_t n_points,
                       int32_t n_images, const int32_t* neighbors, int32_t k, const int32_t* k_list,
                       int32_t n_k, uint64_t* bits, float* out, void* stream) {
  if (n_views < 0 || n_points < 0 || n_images <= 0 || k <= 0 || n_k <= 0) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!view_point || !images || !neighbors || !k_list || !bits || !out) return DVA_ERR_INVALID;
  const int n_words = (n_images + 63) / 64;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(bits, 0, (size_t)n_points * n_words * 8, s) != hipSuccess) return DVA_ERR_LAUNCH;
  hipLaunchKernelGGL(view_bits_kernel, dim3(grid256(n_views)), dim3(256), 0, s, view_point, images, n_views,
                     n_words, (unsigned long long*)bits);
  hipLaunchKernelGGL(occlusion_kernel, dim3(grid256(n_views)), dim3(256), 0, s, view_point, images, n_views,
                     neighbors, k, (const unsigned long long*)bits, n_words, k_list, n_k, out);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // extern "C"
