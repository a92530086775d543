// CSR segment primitives for gfx950 (see include/dva.h for the reference call sites).
//
// Layout: rows are row-major [rows][C]; one thread owns one (group, channel) pair and walks the
// group's rows in order, so a wavefront reads 64 consecutive channels of a row per step
// (coalesced) and every reduction is evaluated in a fixed order => bit-deterministic, like
// torch_scatter's CSR kernels (pooling.py:885-887).  HBM-bound: each source element is read once.
#include "dva_common.h"

namespace dva {

template <typename T, int REDUCE>
__global__ __launch_bounds__(256) void segment_csr_fwd_kernel(const T* __restrict__ src,
                                                               const int64_t* __restrict__ ptr,
                                                               T* __restrict__ out,
                                                               int32_t* __restrict__ arg,
                                                               int64_t n_groups, int C) {
  const int64_t total = n_groups * (int64_t)C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t g = t / C;
    const int c = (int)(t - g * C);
    const int64_t beg = ptr[g], end = ptr[g + 1];
    float acc;
    int64_t best = -1;
    if (REDUCE == DVA_SUM || REDUCE == DVA_MEAN) {
      acc = 0.f;
      for (int64_t r = beg; r < end; ++r) acc += Elt<T>::ld(src, r * C + c);
      if (REDUCE == DVA_MEAN && end > beg) acc /= (float)(end - beg);
    } else {
      acc = 0.f;
      for (int64_t r = beg; r < end; ++r) {
        const float v = Elt<T>::ld(src, r * C + c);
        const bool better = (r == beg) || (REDUCE == DVA_MAX ? v > acc : v < acc);
        if (better) {
          acc = v;
          best = r;
        }
      }
      if (arg) arg[t] = (int32_t)best;
    }
    Elt<T>::st(out, t, acc);
  }
}

template <typename T, int REDUCE>
__global__ __launch_bounds__(256) void segment_csr_bwd_kernel(const T* __restrict__ gout,
                                                               const int64_t* __restrict__ ptr,
                                                               const int32_t* __restrict__ arg,
                                                               T* __restrict__ gsrc,
                                                               int64_t n_groups, int C) {
  const int64_t total = n_groups * (int64_t)C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t g = t / C;
    const int c = (int)(t - g * C);
    const int64_t beg = ptr[g], end = ptr[g + 1];
    if (end <= beg) continue;
    float go = Elt<T>::ld(gout, t);
    if (REDUCE == DVA_MEAN) go /= (float)(end - beg);
    if (REDUCE == DVA_SUM || REDUCE == DVA_MEAN) {
      for (int64_t r = beg; r < end; ++r) Elt<T>::st(gsrc, r * C + c, go);
    } else {
      const int64_t a = arg[t];
      for (int64_t r = beg; r < end; ++r) Elt<T>::st(gsrc, r * C + c, r == a ? go : 0.f);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_csr_kernel(const T* __restrict__ src,
                                                          const int64_t* __restrict__ ptr,
                                                          T* __restrict__ out, int64_t n_groups,
                                                          int C) {
  const int64_t total = n_groups * (int64_t)C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t g = t / C;
    const int c = (int)(t - g * C);
    const int64_t beg = ptr[g], end = ptr[g + 1];
    const T v = src[t];
    for (int64_t r = beg; r < end; ++r) out[r * C + c] = v;
  }
}

// One thread per (group, score column).
__global__ __launch_bounds__(256) void segment_softmax_fwd_kernel(const float* __restrict__ src,
                                                                   const int64_t* __restrict__ ptr,
                                                                   float* __restrict__ out,
                                                                   int64_t n_groups, int G,
                                                                   int scaling, float eps) {
  const int64_t total = n_groups * (int64_t)G;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = t / G;
    const int g = (int)(t - p * G);
    const int64_t beg = ptr[p], end = ptr[p + 1];
    if (end <= beg) continue;
    float m = src[beg * G + g];
    for (int64_t r = beg + 1; r < end; ++r) m = fmaxf(m, src[r * G + g]);
    const float d = scaling ? sqrtf((float)(end - beg)) : 1.f;
    float s = 0.f;
    for (int64_t r = beg; r < end; ++r) {
      const float e = expf((src[r * G + g] - m) / d);
      out[r * G + g] = e;
      s += e;
    }
    s += eps;
    for (int64_t r = beg; r < end; ++r) out[r * G + g] = out[r * G + g] / s;
  }
}

__global__ __launch_bounds__(256) void segment_softmax_bwd_kernel(const float* __restrict__ gout,
                                                                   const float* __restrict__ out,
                                                                   const int64_t* __restrict__ ptr,
                                                                   float* __restrict__ gsrc,
                                                                   int64_t n_groups, int G,
                                                                   int scaling) {
  const int64_t total = n_groups * (int64_t)G;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = t / G;
    const int g = (int)(t - p * G);
    const int64_t beg = ptr[p], end = ptr[p + 1];
    if (end <= beg) continue;
    float dot = 0.f;
    for (int64_t r = beg; r < end; ++r) dot += out[r * G + g] * gout[r * G + g];
    const float d = scaling ? sqrtf((float)(end - beg)) : 1.f;
    for (int64_t r = beg; r < end; ++r) gsrc[r * G + g] = out[r * G + g] * (gout[r * G + g] - dot) / d;
  }
}


// ------------------------------------------------------------------------------------------------
// 16-byte variants (C a multiple of 8 bf16 / 4 fp32 channels, 16-byte aligned rows): one thread owns VEC
// consecutive channels of a group, so a row costs one load / store instruction per thread instead of VEC.
// Same arithmetic and tie rule (first row) as the scalar kernels.
// ------------------------------------------------------------------------------------------------
template <typename T> struct SVec;
template <> struct SVec<float> {
  static constexpr int N = 4;
  typedef float4 raw;
  static __device__ __forceinline__ void unpack(const raw& r, float* f) { f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w; }
  static __device__ __forceinline__ raw pack(const float* f) { return make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct SVec<bf16_t> {
  static constexpr int N = 8;
  typedef uint4 raw;
  static __device__ __forceinline__ void unpack(const raw& r, float* f) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[2 * e] = __uint_as_float(w[e] << 16);
      f[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ raw pack(const float* f) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]),
                      pack_bf16x2(f[6], f[7]));
  }
};

template <typename T, int REDUCE>
__global__ __launch_bounds__(256) void segment_csr_fwd_vec_kernel(const T* __restrict__ src,
                                                                   const int64_t* __restrict__ ptr,
                                                                   T* __restrict__ out,
                                                                   int32_t* __restrict__ arg,
                                                                   int64_t n_groups, int C) {
  constexpr int VEC = SVec<T>::N;
  typedef typename SVec<T>::raw raw_t;
  const int cpr = C / VEC;
  const int64_t total = n_groups * (int64_t)cpr;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t g = t / cpr;
    const int c0 = (int)(t - g * cpr) * VEC;
    const int64_t beg = ptr[g], end = ptr[g + 1];
    float acc[VEC];
    int32_t best[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      acc[k] = 0.f;
      best[k] = -1;
    }
    for (int64_t r = beg; r < end; ++r) {
      float f[VEC];
      SVec<T>::unpack(*reinterpret_cast<const raw_t*>(src + r * C + c0), f);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if (REDUCE == DVA_SUM || REDUCE == DVA_MEAN) {
          acc[k] += f[k];
        } else if (r == beg || (REDUCE == DVA_MAX ? f[k] > acc[k] : f[k] < acc[k])) {
          acc[k] = f[k];
          best[k] = (int32_t)r;
        }
      }
    }
    if (REDUCE == DVA_MEAN && end > beg) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] /= (float)(end - beg);
    }
    *reinterpret_cast<raw_t*>(out + g * C + c0) = SVec<T>::pack(acc);
    if ((REDUCE == DVA_MAX || REDUCE == DVA_MIN) && arg) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) arg[g * C + c0 + k] = best[k];
    }
  }
}

template <typename T, int REDUCE>
__global__ __launch_bounds__(256) void segment_csr_bwd_vec_kernel(const T* __restrict__ gout,
                                                                   const int64_t* __restrict__ ptr,
                                                                   const int32_t* __restrict__ arg,
                                                                   T* __restrict__ gsrc,
                                                                   int64_t n_groups, int C) {
  constexpr int VEC = SVec<T>::N;
  typedef typename SVec<T>::raw raw_t;
  const int cpr = C / VEC;
  const int64_t total = n_groups * (int64_t)cpr;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t g = t / cpr;
    const int c0 = (int)(t - g * cpr) * VEC;
    const int64_t beg = ptr[g], end = ptr[g + 1];
    if (end <= beg) continue;
    float go[VEC];
    SVec<T>::unpack(*reinterpret_cast<const raw_t*>(gout + g * C + c0), go);
    if (REDUCE == DVA_MEAN) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) go[k] /= (float)(end - beg);
    }
    int32_t a[VEC];
    if (REDUCE == DVA_MAX || REDUCE == DVA_MIN) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) a[k] = arg[g * C + c0 + k];
    }
    for (int64_t r = beg; r < end; ++r) {
      float f[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k)
        f[k] = (REDUCE == DVA_SUM || REDUCE == DVA_MEAN) ? go[k] : ((int64_t)a[k] == r ? go[k] : 0.f);
      *reinterpret_cast<raw_t*>(gsrc + r * C + c0) = SVec<T>::pack(f);
    }
  }
}

template <typename T>
static inline bool vec_ok(const void* a, const void* b, int C) {
  return C % SVec<T>::N == 0 && ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0);
}

static inline int grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  const int64_t cap = 256 * 32;  // 256 CUs x 32 blocks, grid-stride beyond
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

template <typename T>
static int launch_segment_fwd(const void* src, const int64_t* ptr, void* out, int32_t* arg,
                              int64_t n, int C, int reduce, hipStream_t s) {
  const bool vec = vec_ok<T>(src, out, C);
  const int grid = grid_for(n * (int64_t)(vec ? C / SVec<T>::N : C));
#define DVA_L(R)                                                                                        \
  do {                                                                                                  \
    if (vec)                                                                                            \
      hipLaunchKernelGGL((segment_csr_fwd_vec_kernel<T, R>), dim3(grid), dim3(256), 0, s, (const T*)src, \
                         ptr, (T*)out, arg, n, C);                                                      \
    else                                                                                                \
      hipLaunchKernelGGL((segment_csr_fwd_kernel<T, R>), dim3(grid), dim3(256), 0, s, (const T*)src,    \
                         ptr, (T*)out, arg, n, C);                                                      \
  } while (0)
  switch (reduce) {
    case DVA_SUM: DVA_L(DVA_SUM); break;
    case DVA_MEAN: DVA_L(DVA_MEAN); break;
    case DVA_MAX: DVA_L(DVA_MAX); break;
    case DVA_MIN: DVA_L(DVA_MIN); break;
    default: return DVA_ERR_INVALID;
  }
#undef DVA_L
  return DVA_OK;
}

template <typename T>
static int launch_segment_bwd(const void* gout, const int64_t* ptr, const int32_t* arg, void* gsrc,
                              int64_t n, int C, int reduce, hipStream_t s) {
  const bool vec = vec_ok<T>(gout, gsrc, C);
  const int grid = grid_for(n * (int64_t)(vec ? C / SVec<T>::N : C));
#define DVA_L(R)                                                                                         \
  do {                                                                                                   \
    if (vec)                                                                                             \
      hipLaunchKernelGGL((segment_csr_bwd_vec_kernel<T, R>), dim3(grid), dim3(256), 0, s, (const T*)gout, \
                         ptr, arg, (T*)gsrc, n, C);                                                      \
    else                                                                                                 \
      hipLaunchKernelGGL((segment_csr_bwd_kernel<T, R>), dim3(grid), dim3(256), 0, s, (const T*)gout,    \
                         ptr, arg, (T*)gsrc, n, C);                                                      \
  } while (0)
  switch (reduce) {
    case DVA_SUM: DVA_L(DVA_SUM); break;
    case DVA_MEAN: DVA_L(DVA_MEAN); break;
    case DVA_MAX: DVA_L(DVA_MAX); break;
    case DVA_MIN: DVA_L(DVA_MIN); break;
    default: return DVA_ERR_INVALID;
  }
#undef DVA_L
  return DVA_OK;
}

}  // namespace dva

using namespace dva;

extern "C" {

int dva_version(void) { return 306; }

int dva_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e == hipErrorNoDevice) return 0;
  if (e != hipSuccess) return DVA_ERR_LAUNCH;
  return n;
}

int dva_segment_csr_fwd(const void* src, const int64_t* ptr, void* out, int32_t* arg,
                        int64_t n_groups, int32_t C, int32_t dtype, int32_t reduce, void* stream) {
  if (n_groups < 0 || C < 0 || !ptr) return DVA_ERR_INVALID;
  if (reduce < DVA_SUM || reduce > DVA_MIN) return DVA_ERR_INVALID;
  if (n_groups == 0 || C == 0) return DVA_OK;
  if (!out) return DVA_ERR_INVALID;
  int rc;
  if (dtype == DVA_F32)
    rc = launch_segment_fwd<float>(src, ptr, out, arg, n_groups, C, reduce, (hipStream_t)stream);
  else if (dtype == DVA_BF16)
    rc = launch_segment_fwd<bf16_t>(src, ptr, out, arg, n_groups, C, reduce, (hipStream_t)stream);
  else
    return DVA_ERR_INVALID;
  if (rc) return rc;
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_segment_csr_bwd(const void* grad_out, const int64_t* ptr, const int32_t* arg,
                        void* grad_src, int64_t n_groups, int32_t C, int32_t dtype,
                        int32_t reduce, void* stream) {
  if (n_groups < 0 || C < 0 || !ptr) return DVA_ERR_INVALID;
  if (reduce < DVA_SUM || reduce > DVA_MIN) return DVA_ERR_INVALID;
  if ((reduce == DVA_MAX || reduce == DVA_MIN) && !arg && n_groups > 0 && C > 0)
    return DVA_ERR_INVALID;
  if (n_groups == 0 || C == 0) return DVA_OK;
  int rc;
  if (dtype == DVA_F32)
    rc = launch_segment_bwd<float>(grad_out, ptr, arg, grad_src, n_groups, C, reduce,
                                   (hipStream_t)stream);
  else if (dtype == DVA_BF16)
    rc = launch_segment_bwd<bf16_t>(grad_out, ptr, arg, grad_src, n_groups, C, reduce,
                                    (hipStream_t)stream);
  else
    return DVA_ERR_INVALID;
  if (rc) return rc;
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_gather_csr(const void* src, const int64_t* ptr, void* out, int64_t n_groups, int32_t C,
                   int32_t dtype, void* stream) {
  if (n_groups < 0 || C < 0 || !ptr) return DVA_ERR_INVALID;
  if (n_groups == 0 || C == 0) return DVA_OK;
  // pure byte movement: with 16-byte aligned rows every thread copies 16 bytes of a group's row
  const int64_t row_bytes = (int64_t)C * (dtype == DVA_F32 ? 4 : 2);
  if (dtype != DVA_F32 && dtype != DVA_BF16) return DVA_ERR_INVALID;
  if (row_bytes % 16 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0)) {
    const int u = (int)(row_bytes / 16);
    hipLaunchKernelGGL((gather_csr_kernel<uint4>), dim3(grid_for(n_groups * (int64_t)u)), dim3(256), 0,
                       (hipStream_t)stream, (const uint4*)src, ptr, (uint4*)out, n_groups, u);
    DVA_CHECK_LAUNCH();
    return DVA_OK;
  }
  const int grid = grid_for(n_groups * (int64_t)C);
  if (dtype == DVA_F32)
    hipLaunchKernelGGL((gather_csr_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const float*)src, ptr, (float*)out, n_groups, C);
  else
    hipLaunchKernelGGL((gather_csr_kernel<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, ptr, (bf16_t*)out, n_groups, C);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_segment_softmax_csr_fwd(const float* src, const int64_t* ptr, float* out,
                                int64_t n_groups, int32_t G, int32_t scaling, float eps,
                                void* stream) {
  if (n_groups < 0 || G < 0 || !ptr) return DVA_ERR_INVALID;
  if (n_groups == 0 || G == 0) return DVA_OK;
  const int grid = grid_for(n_groups * (int64_t)G);
  hipLaunchKernelGGL(segment_softmax_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src,
                     ptr, out, n_groups, G, scaling, eps);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_segment_softmax_csr_bwd(const float* grad_out, const float* out, const int64_t* ptr,
                                float* grad_src, int64_t n_groups, int32_t G, int32_t scaling,
                                void* stream) {
  if (n_groups < 0 || G < 0 || !ptr) return DVA_ERR_INVALID;
  if (n_groups == 0 || G == 0) return DVA_OK;
  const int grid = grid_for(n_groups * (int64_t)G);
  hipLaunchKernelGGL(segment_softmax_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     grad_out, out, ptr, grad_src, n_groups, G, scaling);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // extern "C"
