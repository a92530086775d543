// Compatibilities of QKVBimodalCSRPool from the recompute chain's key rows (reference modules/multimodal/pooling.py:520-531:
// compatibilities = (keys.view(V, G, nc_qk) * queries.view(V, G, nc_qk)).sum(2) / sqrt(nc_qk), queries = the point's query
// row expanded to its views).
//
// keys K' bf16 [V][32] come from dva_chain_keys in ACCUMULATOR order: position i = 16 h + r holds key channel
// chan(r, h) = (r & 3) + 8 (r >> 2) + 4 h, so the 32 bytes a lane of the chain kernels holds are contiguous -- the layout of
// the rows the chain's backward passes hand to each other, and the layout in which d keys goes back into
// dva_chain_score_stats / dva_chain_bwd_layer(6) (G = 32).  The group of position i is chan(i & 15, i >> 4) / nc_qk
// (G nc_qk = 32).  Q' fp32 [N][32] = the queries in the same position order (host: Q[:, channel_of_position]).
//   dva_qkv_compat      compat[v][g] = scale sum_{i in g} K'[v][i] Q'[p(v)][i]
//   dva_qkv_compat_bwd  dK'[v][i] = scale dcompat[v][g(i)] Q'[p][i] (bf16 row; optional: the chain backward builds it itself);
//                       dQ'[p][i] = scale sum_{v in p} dcompat[v][g(i)] K'[v][i]
#include "dva_common.h"

namespace dva {
namespace qkv {

static inline int grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 256 * 32) b = 256 * 32;
  if (b < 1) b = 1;
  return (int)b;
}

__host__ __device__ __forceinline__ int chan_of(int i) {
  const int r = i & 15, h = i >> 4;
  return (r & 3) + 8 * (r >> 2) + 4 * h;
}

// one thread per view
__global__ __launch_bounds__(256) void compat_kernel(const bf16_t* __restrict__ keys, const float* __restrict__ Qp,
                                                     const int32_t* __restrict__ vp, float* __restrict__ compat,
                                                     int64_t V, int G, float scale) {
  const int nc = 32 / G;
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
    const uint4* kr = reinterpret_cast<const uint4*>(keys + v * 32);
    const float4* qr = reinterpret_cast<const float4*>(Qp + (int64_t)vp[v] * 32);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const uint4 k8 = kr[c4];
      const float4 q0 = qr[2 * c4], q1 = qr[2 * c4 + 1];
      const uint32_t kw[4] = {k8.x, k8.y, k8.z, k8.w};
      const float qq[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i0 = 8 * c4 + 2 * e;
        const float k0 = __uint_as_float(kw[e] << 16), k1 = __uint_as_float(kw[e] & 0xffff0000u);
        const int g0 = chan_of(i0) / nc, g1 = chan_of(i0 + 1) / nc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (g == g0) acc[g] = fmaf(k0, qq[2 * e], acc[g]);
          if (g == g1) acc[g] = fmaf(k1, qq[2 * e + 1], acc[g]);
        }
      }
    }
    for (int g = 0; g < G; ++g) compat[v * G + g] = acc[g] * scale;
  }
}

// one thread per view: the bf16 row of d keys
__global__ __launch_bounds__(256) void dkeys_kernel(const float* __restrict__ dcompat, const float* __restrict__ Qp,
                                                    const int32_t* __restrict__ vp, bf16_t* __restrict__ dkeys,
                                                    int64_t V, int G, float scale) {
  const int nc = 32 / G;
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
    float dc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < G; ++g) dc[g] = dcompat[v * G + g] * scale;
    const float4* qr = reinterpret_cast<const float4*>(Qp + (int64_t)vp[v] * 32);
    uint4* out = reinterpret_cast<uint4*>(dkeys + v * 32);
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const float4 q0 = qr[2 * c4], q1 = qr[2 * c4 + 1];
      const float qq[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
      uint32_t w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i0 = 8 * c4 + 2 * e;
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (g == chan_of(i0) / nc) d0 = dc[g];
          if (g == chan_of(i0 + 1) / nc) d1 = dc[g];
        }
        w[e] = pack_bf16x2(d0 * qq[2 * e], d1 * qq[2 * e + 1]);
      }
      out[c4] = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

// one half-wave per point: lane i <-> position i, loop over the views of the point
__global__ __launch_bounds__(256) void dquery_kernel(const float* __restrict__ dcompat, const bf16_t* __restrict__ keys,
                                                     const int64_t* __restrict__ ptr, float* __restrict__ dQp, int64_t N,
                                                     int G, float scale, int ldc) {
  const int i = threadIdx.x & 31;
  const int g = chan_of(i) / (32 / G);
  const int64_t hw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5, n_hw = ((int64_t)gridDim.x * blockDim.x) >> 5;
  for (int64_t p = hw; p < N; p += n_hw) {
    const int64_t v0 = ptr[p], v1 = ptr[p + 1];
    float acc = 0.f;
    int64_t v = v0;
    for (; v + 4 <= v1; v += 4) {
      float k[4], d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        k[u] = bf2f(keys[(v + u) * 32 + i]);
        d[u] = dcompat[(v + u) * ldc + g];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = fmaf(d[u], k[u], acc);
    }
    for (; v < v1; ++v) acc = fmaf(dcompat[v * ldc + g], bf2f(keys[v * 32 + i]), acc);
    dQp[p * 32 + i] = acc * scale;
  }
}

}  // namespace qkv
}  // namespace dva

using namespace dva;

extern "C" {

int dva_qkv_compat(const void* keys, const float* queries, const int32_t* view_point, float* compat, int64_t n_views,
                   int32_t G, float scale, void* stream) {
  if (n_views < 0 || (G != 1 && G != 2 && G != 4)) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!keys || !queries || !view_point || !compat || ((uintptr_t)keys & 15) || ((uintptr_t)queries & 15))
    return DVA_ERR_INVALID;
  hipLaunchKernelGGL(qkv::compat_kernel, dim3(qkv::grid_for(n_views)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)keys,
                     queries, view_point, compat, n_views, (int)G, scale);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_qkv_compat_bwd(const float* grad_compat, const void* keys, const float* queries, const int32_t* view_point,
                       const int64_t* ptr, void* grad_keys, float* grad_queries, int64_t n_points, int64_t n_views,
                       int32_t G, float scale, void* stream) {
  if (n_views < 0 || n_points < 0 || (G != 1 && G != 2 && G != 4)) return DVA_ERR_INVALID;
  if (n_points == 0) return DVA_OK;
  if (!ptr || !grad_queries) return DVA_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  if (n_views > 0) {
    if (!grad_compat || !keys) return DVA_ERR_INVALID;
    // grad_keys == NULL (round 4, the chain path): the chain backward builds dK' in registers from grad_compat and the
    // query rows (dva_chain_score_stats_keys / dva_chain_bwd_layer6_keys); only dQ' is produced here
    if (grad_keys) {
      if (!queries || !view_point || ((uintptr_t)grad_keys & 15) || ((uintptr_t)queries & 15)) return DVA_ERR_INVALID;
      hipLaunchKernelGGL(qkv::dkeys_kernel, dim3(qkv::grid_for(n_views)), dim3(256), 0, s, grad_compat, queries, view_point,
                         (bf16_t*)grad_keys, n_views, (int)G, scale);
    }
  }
  hipLaunchKernelGGL(qkv::dquery_kernel, dim3(qkv::grid_for(n_points * 32)), dim3(256), 0, s, grad_compat, (const bf16_t*)keys,
                     ptr, grad_queries, n_points, (int)G, scale, (int)G);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

// dQ' alone, grad_compat with a leading dimension (the chain path's [V][4] layout: ld = 4)
int dva_qkv_dquery(const float* grad_compat, int32_t ld, const void* keys, const int64_t* ptr, float* grad_queries,
                   int64_t n_points, int64_t n_views, int32_t G, float scale, void* stream) {
  if (n_views < 0 || n_points < 0 || (G != 1 && G != 2 && G != 4) || ld < G) return DVA_ERR_INVALID;
  if (n_points == 0) return DVA_OK;
  if (!ptr || !grad_queries || (n_views > 0 && (!grad_compat || !keys))) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(qkv::dquery_kernel, dim3(qkv::grid_for(n_points * 32)), dim3(256), 0, (hipStream_t)stream, grad_compat,
                     (const bf16_t*)keys, ptr, grad_queries, n_points, (int)G, scale, (int)ld);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // extern "C"
