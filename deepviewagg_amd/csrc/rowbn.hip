// Weighted BatchNorm + LeakyReLU over the R rows of a feature map ("E_mod hoisting").
//
// The reference applies E_mod = [Linear -> BatchNorm1d -> LeakyReLU] x 2 to the V gathered view features
// (modules/multimodal/pooling.py:245,275; MLP blocks core/common_modules/base_modules.py:38-48).  With an
// exact nearest mapping those V rows are copies of the R feature-map rows, row r appearing counts[r]
// times, so the train-mode batch statistics over the views are the statistics over the map rows weighted
// by counts, and the whole block runs on R << V rows:
//     mean = sum_r w_r y_r / n,  var = sum_r w_r (y_r - mean)^2 / n,  n = sum_r w_r = V
//     out_r = leaky(gamma (y_r - mean) / sqrt(var + eps) + beta)
// backward (gout_r already holds the sum of the gradients of the views of row r):
//     dz_r = gout_r leaky'(z_r);  S1 = sum_r dz_r;  S2 = sum_r dz_r a_r   (a = normalised y)
//     dy_r = gamma invstd (dz_r - w_r S1 / n - w_r a_r S2 / n);  dgamma = S2;  dbeta = S1
// Four row-streaming passes (statistics / apply, forward and backward): lanes over channels (coalesced
// rows), fp32 partial sums per thread, LDS per block, fp64 atomics per block.
#include "dva_common.h"

namespace dva {

constexpr int RB_ROWS = 4;  // rows per block iteration (blockDim = 64 x 4)

// sums[0..C) += sum_r w_r f1, sums[C..2C) += sum_r w_r f2 with (f1, f2) produced per element by `F`
template <typename T, typename F>
__device__ __forceinline__ void row_sums(int64_t R, int C, double* __restrict__ sums, double* s_red, F&& f) {
  // fp64 partial sums: the variance is E[y^2] - mean^2, and with fp32 partials the cancellation showed up as
  // 1e-4 relative errors downstream when |mean| >> std (the passes are memory bound, fp64 adds are free)
  for (int i = threadIdx.y * 64 + threadIdx.x; i < 2 * C; i += 256) s_red[i] = 0.0;
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 64) {
    double a0 = 0.0, a1 = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * RB_ROWS + threadIdx.y; r < R; r += (int64_t)gridDim.x * RB_ROWS) {
      float f1, f2;
      f(r, c, f1, f2);
      a0 += (double)f1;
      a1 += (double)f2;
    }
    atomicAdd(&s_red[c], a0);
    atomicAdd(&s_red[C + c], a1);
  }
  __syncthreads();
  for (int i = threadIdx.y * 64 + threadIdx.x; i < 2 * C; i += 256) atomicAdd(&sums[i], s_red[i]);
}

template <typename T>
__global__ __launch_bounds__(256) void rowbn_stats_kernel(const T* __restrict__ y,
                                                           const int32_t* __restrict__ counts,
                                                           double* __restrict__ sums, int64_t R, int C) {
  extern __shared__ double s_red[];
  row_sums<T>(R, C, sums, s_red, [&](int64_t r, int c, float& f1, float& f2) {
    const float w = counts ? (float)counts[r] : 1.f, v = Elt<T>::ld(y, r * C + c);
    f1 = w * v;
    f2 = w * v * v;
  });
}

// bn = [4][C] fp32: mean | invstd | gamma | beta
template <typename T>
__global__ __launch_bounds__(256) void rowbn_apply_kernel(const T* __restrict__ y,
                                                           const float* __restrict__ bn,
                                                           T* __restrict__ out, int64_t R, int C,
                                                           float slope) {
  const int64_t total = R * (int64_t)C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(t % C);
    const float a = (Elt<T>::ld(y, t) - bn[c]) * bn[C + c];
    const float z = a * bn[2 * C + c] + bn[3 * C + c];
    Elt<T>::st(out, t, z > 0.f ? z : slope * z);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void rowbn_bwd_stats_kernel(const T* __restrict__ gout,
                                                               const T* __restrict__ y,
                                                               const float* __restrict__ bn,
                                                               double* __restrict__ sums, int64_t R,
                                                               int C, float slope) {
  extern __shared__ double s_red[];
  row_sums<T>(R, C, sums, s_red, [&](int64_t r, int c, float& f1, float& f2) {
    const float a = (Elt<T>::ld(y, r * C + c) - bn[c]) * bn[C + c];
    const float z = a * bn[2 * C + c] + bn[3 * C + c];
    const float dz = Elt<T>::ld(gout, r * C + c) * (z > 0.f ? 1.f : slope);
    f1 = dz;
    f2 = dz * a;
  });
}

// sm = [2][C] fp32: S1/n | S2/n (zeros when the statistics are not batch statistics)
template <typename T>
__global__ __launch_bounds__(256) void rowbn_bwd_apply_kernel(
    const T* __restrict__ gout, const T* __restrict__ y, const int32_t* __restrict__ counts,
    const float* __restrict__ bn, const float* __restrict__ sm, T* __restrict__ dy, int64_t R, int C,
    float slope) {
  const int64_t total = R * (int64_t)C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / C;
    const int c = (int)(t - r * C);
    const float a = (Elt<T>::ld(y, t) - bn[c]) * bn[C + c];
    const float z = a * bn[2 * C + c] + bn[3 * C + c];
    const float dz = Elt<T>::ld(gout, t) * (z > 0.f ? 1.f : slope);
    const float w = counts ? (float)counts[r] : 1.f;
    Elt<T>::st(dy, t, bn[2 * C + c] * bn[C + c] * (dz - w * sm[c] - w * a * sm[C + c]));
  }
}


// ------------------------------------------------------------------------------------------------
// 16-byte variants (C a multiple of 8 bf16 / 4 fp32 channels with C / VEC a power of two <= 256, aligned
// rows): a thread owns VEC consecutive channels, one load / store instruction per 16 bytes.  Used for the
// V-sized calls (E_mod on materialised views), where the scalar kernels ran at 2-3 TB/s.
// ------------------------------------------------------------------------------------------------
template <typename T> struct RVec;
template <> struct RVec<float> {
  static constexpr int N = 4;
  typedef float4 raw;
  static __device__ __forceinline__ void unpack(const raw& r, float* f) { f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w; }
  static __device__ __forceinline__ raw pack(const float* f) { return make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct RVec<bf16_t> {
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

// MODE 0: forward statistics (w y | w y^2); MODE 1: backward statistics (dz | dz a)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void rowbn_sums_vec_kernel(const T* __restrict__ y, const T* __restrict__ gout,
                                                              const int32_t* __restrict__ counts,
                                                              const float* __restrict__ bn,
                                                              double* __restrict__ sums, int64_t R, int C,
                                                              float slope) {
  constexpr int VEC = RVec<T>::N;
  typedef typename RVec<T>::raw raw_t;
  extern __shared__ double s_red[];
  for (int i = threadIdx.x; i < 2 * C; i += 256) s_red[i] = 0.0;
  __syncthreads();
  const int cpr = C / VEC, ci = threadIdx.x % cpr, slot = threadIdx.x / cpr, rpb = 256 / cpr, c0 = ci * VEC;
  double a0[VEC], a1[VEC];
  float mu[VEC], is[VEC], ga[VEC], be[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    a0[k] = a1[k] = 0.0;
    mu[k] = MODE == 1 ? bn[c0 + k] : 0.f;
    is[k] = MODE == 1 ? bn[C + c0 + k] : 0.f;
    ga[k] = MODE == 1 ? bn[2 * C + c0 + k] : 0.f;
    be[k] = MODE == 1 ? bn[3 * C + c0 + k] : 0.f;
  }
  // four rows per thread in flight (round 5): with one 16-byte load per iteration the pass was latency-bound
  // (134 MB in 54 us at C = 512: 2.5 TB/s); fp32 partial sums over the four rows, fp64 across iterations
  constexpr int U = 4;
  const int64_t stride = (int64_t)gridDim.x * rpb;
  for (int64_t r0 = (int64_t)blockIdx.x * rpb + slot; r0 < R; r0 += stride * U) {
    raw_t yr[U], gr[U];
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u * stride;
      const bool ok = r < R;
      yr[u] = *reinterpret_cast<const raw_t*>(y + (ok ? r : r0) * C + c0);
      if (MODE == 1) gr[u] = *reinterpret_cast<const raw_t*>(gout + (ok ? r : r0) * C + c0);
      w[u] = ok ? ((MODE == 0 && counts) ? (float)counts[r] : 1.f) : 0.f;
    }
    float p0[VEC], p1[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) p0[k] = p1[k] = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float v[VEC];
      RVec<T>::unpack(yr[u], v);
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const float wv = w[u] * v[k];
          p0[k] += wv;
          p1[k] = fmaf(wv, v[k], p1[k]);
        }
      } else {
        float g[VEC];
        RVec<T>::unpack(gr[u], g);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const float a = (v[k] - mu[k]) * is[k];
          const float z = a * ga[k] + be[k];
          const float dz = w[u] * g[k] * (z > 0.f ? 1.f : slope);
          p0[k] += dz;
          p1[k] = fmaf(dz, a, p1[k]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      a0[k] += (double)p0[k];
      a1[k] += (double)p1[k];
    }
  }
  // the threads of a wavefront that own the same channels (lanes ci, ci + cpr, ...) reduce with shuffles first:
  // one LDS fp64 atomic per (wavefront, channel) instead of one per thread (fp64 LDS atomics are compare-and-swap
  // loops), then one global atomic per (block, channel)
  if (cpr <= 32 && (cpr & (cpr - 1)) == 0) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      for (int off = cpr; off < 64; off <<= 1) {
        a0[k] += __shfl_xor(a0[k], off);
        a1[k] += __shfl_xor(a1[k], off);
      }
    }
    if ((threadIdx.x & 63) < cpr) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        atomicAdd(&s_red[c0 + k], a0[k]);
        atomicAdd(&s_red[C + c0 + k], a1[k]);
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      atomicAdd(&s_red[c0 + k], a0[k]);
      atomicAdd(&s_red[C + c0 + k], a1[k]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) atomicAdd(&sums[i], s_red[i]);
}

// MODE 0: out = leaky(BN(y)); MODE 1: grad_y (backward apply)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void rowbn_apply_vec_kernel(const T* __restrict__ y, const T* __restrict__ gout,
                                                               const int32_t* __restrict__ counts,
                                                               const float* __restrict__ bn,
                                                               const float* __restrict__ sm, T* __restrict__ out,
                                                               int64_t R, int C, float slope) {
  constexpr int VEC = RVec<T>::N;
  typedef typename RVec<T>::raw raw_t;
  // the grid stride is a multiple of C / VEC (a power of two <= 256): a thread keeps its channels, so the
  // per-channel constants are loaded once
  const int cpr = C / VEC, sh = __ffs(cpr) - 1, c0 = (threadIdx.x & (cpr - 1)) * VEC;
  const int64_t total = R * (int64_t)cpr;
  float mu[VEC], is[VEC], ga[VEC], be[VEC], s0[VEC], s1[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    mu[k] = bn[c0 + k];
    is[k] = bn[C + c0 + k];
    ga[k] = bn[2 * C + c0 + k];
    be[k] = bn[3 * C + c0 + k];
    s0[k] = MODE == 1 ? sm[c0 + k] : 0.f;
    s1[k] = MODE == 1 ? sm[C + c0 + k] : 0.f;
  }
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t >> sh;
    float v[VEC], o[VEC];
    RVec<T>::unpack(*reinterpret_cast<const raw_t*>(y + r * C + c0), v);
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float z = (v[k] - mu[k]) * is[k] * ga[k] + be[k];
        o[k] = z > 0.f ? z : slope * z;
      }
    } else {
      float g[VEC];
      RVec<T>::unpack(*reinterpret_cast<const raw_t*>(gout + r * C + c0), g);
      const float w = counts ? (float)counts[r] : 1.f;
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float a = (v[k] - mu[k]) * is[k];
        const float z = a * ga[k] + be[k];
        const float dz = g[k] * (z > 0.f ? 1.f : slope);
        o[k] = ga[k] * is[k] * (dz - w * s0[k] - w * a * s1[k]);
      }
    }
    *reinterpret_cast<raw_t*>(out + r * C + c0) = RVec<T>::pack(o);
  }
}

template <typename T>
static inline bool rv_ok(int C, const void* a, const void* b, const void* c) {
  const int cpr = C / RVec<T>::N;
  return C % RVec<T>::N == 0 && cpr >= 1 && cpr <= 256 && (cpr & (cpr - 1)) == 0 &&
         (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) % 16 == 0);
}

static inline int rows_grid(int64_t R) {
  int64_t b = (R + RB_ROWS - 1) / RB_ROWS;
  if (b > 256 * 8) b = 256 * 8;
  if (b < 1) b = 1;
  return (int)b;
}
static inline int elems_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace dva

using namespace dva;

extern "C" {

int dva_rowbn_stats(const void* y, const int32_t* counts, double* sums, int64_t R, int32_t C,
                    int32_t dtype, void* stream) {
  if (R < 0 || C <= 0 || C > 4096 || !sums) return DVA_ERR_INVALID;
  if (dtype != DVA_F32 && dtype != DVA_BF16) return DVA_ERR_INVALID;
  if (R == 0) return DVA_OK;
  if (!y) return DVA_ERR_INVALID;
  const dim3 block(64, 4);
  const size_t lds = 2 * (size_t)C * sizeof(double);
  if (dtype == DVA_F32 ? rv_ok<float>(C, y, y, y) : rv_ok<bf16_t>(C, y, y, y)) {
    const int rpb = 256 / (C / (dtype == DVA_F32 ? 4 : 8));
    int64_t b = (R + 4 * rpb - 1) / (4 * rpb);      // four rows per thread and iteration
    if (b > 256 * 2) b = 256 * 2;      // every block ends with 2 C fp64 atomics: more blocks measured slower
    if (dtype == DVA_F32)
      hipLaunchKernelGGL((rowbn_sums_vec_kernel<float, 0>), dim3((int)b), dim3(256), lds, (hipStream_t)stream,
                         (const float*)y, (const float*)nullptr, counts, (const float*)nullptr, sums, R, C, 0.f);
    else
      hipLaunchKernelGGL((rowbn_sums_vec_kernel<bf16_t, 0>), dim3((int)b), dim3(256), lds, (hipStream_t)stream,
                         (const bf16_t*)y, (const bf16_t*)nullptr, counts, (const float*)nullptr, sums, R, C, 0.f);
    DVA_CHECK_LAUNCH();
    return DVA_OK;
  }
  if (dtype == DVA_F32)
    hipLaunchKernelGGL((rowbn_stats_kernel<float>), dim3(rows_grid(R)), block, lds, (hipStream_t)stream,
                       (const float*)y, counts, sums, R, C);
  else
    hipLaunchKernelGGL((rowbn_stats_kernel<bf16_t>), dim3(rows_grid(R)), block, lds, (hipStream_t)stream,
                       (const bf16_t*)y, counts, sums, R, C);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_rowbn_apply(const void* y, const float* bn, void* out, int64_t R, int32_t C, float slope,
                    int32_t dtype, void* stream) {
  if (R < 0 || C <= 0) return DVA_ERR_INVALID;
  if (dtype != DVA_F32 && dtype != DVA_BF16) return DVA_ERR_INVALID;
  if (R == 0) return DVA_OK;
  if (!y || !bn || !out) return DVA_ERR_INVALID;
  if (dtype == DVA_F32 ? rv_ok<float>(C, y, out, out) : rv_ok<bf16_t>(C, y, out, out)) {
    const dim3 vg(elems_grid(R * (C / (dtype == DVA_F32 ? 4 : 8))));
    if (dtype == DVA_F32)
      hipLaunchKernelGGL((rowbn_apply_vec_kernel<float, 0>), vg, dim3(256), 0, (hipStream_t)stream, (const float*)y,
                         (const float*)nullptr, (const int32_t*)nullptr, bn, (const float*)nullptr, (float*)out, R,
                         C, slope);
    else
      hipLaunchKernelGGL((rowbn_apply_vec_kernel<bf16_t, 0>), vg, dim3(256), 0, (hipStream_t)stream,
                         (const bf16_t*)y, (const bf16_t*)nullptr, (const int32_t*)nullptr, bn,
                         (const float*)nullptr, (bf16_t*)out, R, C, slope);
    DVA_CHECK_LAUNCH();
    return DVA_OK;
  }
  const dim3 grid(elems_grid(R * C));
  if (dtype == DVA_F32)
    hipLaunchKernelGGL((rowbn_apply_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream,
                       (const float*)y, bn, (float*)out, R, C, slope);
  else
    hipLaunchKernelGGL((rowbn_apply_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)y, bn, (bf16_t*)out, R, C, slope);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_rowbn_bwd_stats(const void* grad_out, const void* y, const float* bn, double* sums, int64_t R,
                        int32_t C, float slope, int32_t dtype, void* stream) {
  if (R < 0 || C <= 0 || C > 4096 || !sums) return DVA_ERR_INVALID;
  if (dtype != DVA_F32 && dtype != DVA_BF16) return DVA_ERR_INVALID;
  if (R == 0) return DVA_OK;
  if (!grad_out || !y || !bn) return DVA_ERR_INVALID;
  const dim3 block(64, 4);
  const size_t lds = 2 * (size_t)C * sizeof(double);
  if (dtype == DVA_F32 ? rv_ok<float>(C, y, grad_out, y) : rv_ok<bf16_t>(C, y, grad_out, y)) {
    const int rpb = 256 / (C / (dtype == DVA_F32 ? 4 : 8));
    int64_t b = (R + 4 * rpb - 1) / (4 * rpb);      // four rows per thread and iteration
    if (b > 256 * 2) b = 256 * 2;      // every block ends with 2 C fp64 atomics: more blocks measured slower
    if (dtype == DVA_F32)
      hipLaunchKernelGGL((rowbn_sums_vec_kernel<float, 1>), dim3((int)b), dim3(256), lds, (hipStream_t)stream,
                         (const float*)y, (const float*)grad_out, (const int32_t*)nullptr, bn, sums, R, C, slope);
    else
      hipLaunchKernelGGL((rowbn_sums_vec_kernel<bf16_t, 1>), dim3((int)b), dim3(256), lds, (hipStream_t)stream,
                         (const bf16_t*)y, (const bf16_t*)grad_out, (const int32_t*)nullptr, bn, sums, R, C, slope);
    DVA_CHECK_LAUNCH();
    return DVA_OK;
  }
  if (dtype == DVA_F32)
    hipLaunchKernelGGL((rowbn_bwd_stats_kernel<float>), dim3(rows_grid(R)), block, lds,
                       (hipStream_t)stream, (const float*)grad_out, (const float*)y, bn, sums, R, C, slope);
  else
    hipLaunchKernelGGL((rowbn_bwd_stats_kernel<bf16_t>), dim3(rows_grid(R)), block, lds,
                       (hipStream_t)stream, (const bf16_t*)grad_out, (const bf16_t*)y, bn, sums, R, C,
                       slope);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_rowbn_bwd_apply(const void* grad_out, const void* y, const int32_t* counts, const float* bn,
                        const float* sm, void* grad_y, int64_t R, int32_t C, float slope, int32_t dtype,
                        void* stream) {
  if (R < 0 || C <= 0) return DVA_ERR_INVALID;
  if (dtype != DVA_F32 && dtype != DVA_BF16) return DVA_ERR_INVALID;
  if (R == 0) return DVA_OK;
  if (!grad_out || !y || !bn || !sm || !grad_y) return DVA_ERR_INVALID;
  if (dtype == DVA_F32 ? rv_ok<float>(C, y, grad_out, grad_y) : rv_ok<bf16_t>(C, y, grad_out, grad_y)) {
    const dim3 vg(elems_grid(R * (C / (dtype == DVA_F32 ? 4 : 8))));
    if (dtype == DVA_F32)
      hipLaunchKernelGGL((rowbn_apply_vec_kernel<float, 1>), vg, dim3(256), 0, (hipStream_t)stream, (const float*)y,
                         (const float*)grad_out, counts, bn, sm, (float*)grad_y, R, C, slope);
    else
      hipLaunchKernelGGL((rowbn_apply_vec_kernel<bf16_t, 1>), vg, dim3(256), 0, (hipStream_t)stream,
                         (const bf16_t*)y, (const bf16_t*)grad_out, counts, bn, sm, (bf16_t*)grad_y, R, C, slope);
    DVA_CHECK_LAUNCH();
    return DVA_OK;
  }
  const dim3 grid(elems_grid(R * C));
  if (dtype == DVA_F32)
    hipLaunchKernelGGL((rowbn_bwd_apply_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream,
                       (const float*)grad_out, (const float*)y, counts, bn, sm, (float*)grad_y, R, C, slope);
  else
    hipLaunchKernelGGL((rowbn_bwd_apply_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)grad_out, (const bf16_t*)y, counts, bn, sm, (bf16_t*)grad_y, R, C,
                       slope);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// BatchNorm bookkeeping between two passes of the DeepSet / rowbn kernels, one launch instead of ~18
// tiny library kernels: batch statistics -> constants table, running statistics update
// (nn.BatchNorm1d semantics: biased variance for the normalisation, unbiased for running_var).
// ------------------------------------------------------------------------------------------------
namespace dva {

__global__ void bn_finalize_kernel(const double* __restrict__ sums, double m, float* __restrict__ rmean,
                                   float* __restrict__ rvar, int64_t* __restrict__ nbt,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float momentum, float eps, int training, int C,
                                   float* __restrict__ bn) {
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float mean, var;
    if (training) {
      const double mu = sums[c] / m;
      double v = sums[C + c] / m - mu * mu;
      if (v < 0.0) v = 0.0;
      mean = (float)mu;
      var = (float)v;
      if (rmean) {
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(v * (m / (m > 1.0 ? m - 1.0 : 1.0)));
      }
    } else {
      mean = rmean[c];
      var = rvar[c];
    }
    bn[c] = mean;
    bn[C + c] = rsqrtf(var + eps);
    bn[2 * C + c] = gamma ? gamma[c] : 1.f;
    bn[3 * C + c] = beta ? beta[c] : 0.f;
  }
  if (training && nbt && threadIdx.x == 0) *nbt += 1;
}

__global__ void scale_f64_kernel(const double* __restrict__ in, double scale, float* __restrict__ out, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = (float)(in[i] * scale);
}

}  // namespace dva

extern "C" {

int dva_bn_finalize(const double* sums, double m, float* running_mean, float* running_var,
                    int64_t* num_batches_tracked, const float* gamma, const float* beta, float momentum,
                    float eps, int32_t training, int32_t C, float* bn, void* stream) {
  if (C <= 0 || !bn) return DVA_ERR_INVALID;
  if (training ? (!sums || m <= 0.0) : (!running_mean || !running_var)) return DVA_ERR_INVALID;
  if ((running_mean == nullptr) != (running_var == nullptr)) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(dva::bn_finalize_kernel, dim3(1), dim3(C < 256 ? 64 * ((C + 63) / 64) : 256), 0,
                     (hipStream_t)stream, sums, m, running_mean, running_var, num_batches_tracked, gamma,
                     beta, momentum, eps, training, C, bn);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_scale_f64(const double* in, double scale, float* out, int32_t n, void* stream) {
  if (n < 0 || (n > 0 && (!in || !out))) return DVA_ERR_INVALID;
  if (n == 0) return DVA_OK;
  hipLaunchKernelGGL(dva::scale_f64_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, in, scale, out, n);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // extern "C"
