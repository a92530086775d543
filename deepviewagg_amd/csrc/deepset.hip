// Fused DeepSetFeat + E_score chain for gfx950: the per-view mapping-feature encoder of
// GroupBimodalCSRPool / QKVBimodalCSRPool (reference: modules/multimodal/pooling.py:604-673 DeepSetFeat
// with pool='max', fusion='concatenation'; :258,:282 E_score; MLP / FastBatchNorm1d of
// core/common_modules/base_modules.py:38-48,:131-156), forward AND backward, exact fp32.
//
// Why: the chain is six tiny Linear(<=64 -> 32) layers with train-mode BatchNorm between them over
// V = tens of millions of views.  Library GEMM + BN + activation kernels move ~4 KB/view and the
// skinny GEMMs run at <5 % of HBM speed; here every stage between two batch-statistics barriers is
// ONE pass that reads the previous pre-BN activation (128 B/view), applies BN + LeakyReLU on load,
// does the 32x32 product and writes the next pre-BN activation (128 B/view) while accumulating the
// next layer's statistics.
//
// Layout ("row streaming"): lane = channel (32 channels, two rows per wavefront step), a wavefront
// owns 32-row tiles.  Inputs of a row are broadcast to the 32 lanes through LDS (ds_read_b128 of a
// row all lanes share = hardware broadcast), weights of lane n live in its registers, so a 32x32
// layer is 32 v_fmac per row pair, column statistics / weight gradients are per-lane accumulators
// (no cross-lane reduction in the loop), and every global access is a contiguous 128-B row.
// VALU-bound part: V/2 * 32 fmac * 2 clk / 1024 SIMDs ~= 0.45 ms per layer pass at V = 33.5 M,
// below the HBM time of the pass (8.6 GB ~= 1.4 ms), so the passes are HBM-bound.
#include "dva_common.h"

namespace dva {

constexpr int D = 32;       // nc_inner of the fused path
constexpr int TILE = 32;    // rows per wavefront tile
constexpr float SLOPE = 0.2f;

struct BNc {  // batch-norm constants of one channel
  float mean, invstd, gamma, beta;
};
// bn arrays are [4][D] = mean | invstd | gamma | beta
__device__ __forceinline__ BNc load_bn(const float* __restrict__ bn, int c) {
  BNc b;
  b.mean = bn[c]; b.invstd = bn[D + c]; b.gamma = bn[2 * D + c]; b.beta = bn[3 * D + c];
  return b;
}
__device__ __forceinline__ float bn_hat(float a, const BNc& b) { return (a - b.mean) * b.invstd; }
__device__ __forceinline__ float bn_z(float a, const BNc& b) { return bn_hat(a, b) * b.gamma + b.beta; }
__device__ __forceinline__ float leaky(float z) { return z > 0.f ? z : SLOPE * z; }
__device__ __forceinline__ float dleaky(float z) { return z > 0.f ? 1.f : SLOPE; }

// Intra-wavefront LDS hand-off.  The DS instructions of one wavefront execute in program order, so
// a later ds_read sees an earlier ds_write of another lane of the same wavefront; only the compiler
// must be kept from reordering.  Deliberately NOT a memory fence: a release fence waits for
// vmcnt(0), i.e. for the tile's global STORES to be acknowledged before the next tile's loads can
// issue, which made the first version of these kernels latency-bound (17 us per tile per wave).
__device__ __forceinline__ void wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// sum_k row[k] * w[k], k ascending, row broadcast from LDS
template <int K>
__device__ __forceinline__ float dot_row(const float* __restrict__ row, const float (&w)[K]) {
  // two independent fma chains (even / odd float4 blocks) halve the dependent-latency exposure
  float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
  for (int k4 = 0; k4 < K / 4; k4 += 2) {
    const float4 x = *reinterpret_cast<const float4*>(row + 4 * k4);
    const float4 y = *reinterpret_cast<const float4*>(row + 4 * k4 + 4);
    acc0 = fmaf(x.x, w[4 * k4], acc0);
    acc1 = fmaf(y.x, w[4 * k4 + 4], acc1);
    acc0 = fmaf(x.y, w[4 * k4 + 1], acc0);
    acc1 = fmaf(y.y, w[4 * k4 + 5], acc1);
    acc0 = fmaf(x.z, w[4 * k4 + 2], acc0);
    acc1 = fmaf(y.z, w[4 * k4 + 6], acc1);
    acc0 = fmaf(x.w, w[4 * k4 + 3], acc0);
    acc1 = fmaf(y.w, w[4 * k4 + 7], acc1);
  }
  return acc0 + acc1;
}

// Add the two row-parity halves, then accumulate per-block partials into double accumulators.
// vals[j] (j < NV) are per-lane partial sums of channel n; out[j*D + n] += total.
template <int NV>
__device__ __forceinline__ void flush_channel_sums(float (&vals)[NV], double* __restrict__ out,
                                                   float* s_red /* [NV*D] */, int lane) {
  const int n = lane & 31;
  for (int i = threadIdx.x; i < NV * D; i += blockDim.x) s_red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    float v = vals[j] + __shfl_xor(vals[j], 32);
    if (lane < 32) atomicAdd(&s_red[j * D + n], v);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NV * D; i += blockDim.x) atomicAdd(&out[i], (double)s_red[i]);
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------

// x_map [V,F=8] -> a1 = x.Wa^T ; STATS_ONLY: statistics of a1.  Otherwise a1 -> BN1 -> leaky -> .Wb^T = a2,
// written, with statistics of a2.
template <bool STATS_ONLY>
__global__ __launch_bounds__(256) void dsf_fwd_first_kernel(const float* __restrict__ x_map,
                                                             const float* __restrict__ Wa,
                                                             const float* __restrict__ bn1,
                                                             const float* __restrict__ Wb,
                                                             float* __restrict__ a2,
                                                             double* __restrict__ stats, int64_t V) {
  __shared__ __attribute__((aligned(16))) float s_x[4][TILE * 8];
  __shared__ __attribute__((aligned(16))) float s_h[4][TILE * D];
  __shared__ float s_red[2 * D];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n = lane & 31, h = lane >> 5;
  float wa[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) wa[k] = Wa[n * 8 + k];
  float wb[D];
  BNc b1 = {0.f, 1.f, 1.f, 0.f};
  if (!STATS_ONLY) {
#pragma unroll
    for (int k = 0; k < D; ++k) wb[k] = Wb[n * D + k];
    b1 = load_bn(bn1, n);
  }
  float st[2] = {0.f, 0.f};
  const int64_t tiles = (V + TILE - 1) / TILE;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave; t < tiles; t += n_waves) {
    const int64_t row0 = t * TILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = i * 64 + lane;
      const int64_t g = row0 * 8 + e;
      s_x[wv][e] = g < V * 8 ? x_map[g] : 0.f;
    }
    wave_sync();
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const float a1 = dot_row<8>(&s_x[wv][(2 * i + h) * 8], wa);
      if (STATS_ONLY) {
        if (row0 + 2 * i + h < V) {
          st[0] += a1;
          st[1] += a1 * a1;
        }
      } else {
        s_h[wv][(2 * i + h) * D + n] = leaky(bn_z(a1, b1));
      }
    }
    if (!STATS_ONLY) {
      wave_sync();
#pragma unroll 2
      for (int i = 0; i < 16; ++i) {
        const int64_t r = row0 + 2 * i + h;
        const float acc = dot_row<D>(&s_h[wv][(2 * i + h) * D], wb);
        if (r < V) {
          a2[r * D + n] = acc;
          st[0] += acc;
          st[1] += acc * acc;
        }
      }
    }
    wave_sync();
  }
  flush_channel_sums<2>(st, stats, s_red, lane);
}

// pooled[p, c] = max over the point's views of leaky(BN(a[v, c])) (first row on ties), arg = that row.
// One half-wave per point: every lane reads 16 bytes (VEC channels) of a row, 32/LPR rows are in flight
// per half-wave and four row loads per lane are issued before the first is used; the row slots are then
// merged with xor-shuffles (value, row) -- larger value wins, smaller row on ties.
template <typename AT>
struct SegVec;
template <>
struct SegVec<float> {
  static constexpr int VEC = 4;
  typedef float4 raw;
  static __device__ __forceinline__ void unpack(const raw& r, float* f) { f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w; }
};
template <>
struct SegVec<bf16_t> {
  static constexpr int VEC = 8;
  typedef uint4 raw;
  static __device__ __forceinline__ void unpack(const raw& r, float* f) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[2 * e] = __uint_as_float(w[e] << 16);
      f[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
    }
  }
};

// LPP lanes per point (32, 16 or 8; at least LPR): the host picks it from the average segment length so that
// short ragged segments (a handful of views per point) do not leave most row slots idle.
template <typename AT, int LPP>
__global__ __launch_bounds__(256) void dsf_segmax_kernel(const AT* __restrict__ a,
                                                          const float* __restrict__ bn,
                                                          const int64_t* __restrict__ ptr,
                                                          float* __restrict__ pooled,
                                                          int32_t* __restrict__ arg, int64_t N) {
  constexpr int VEC = SegVec<AT>::VEC, LPR = D / VEC, SLOTS = LPP / LPR, U = 4, PPW = 64 / LPP;
  static_assert(LPP >= LPR && SLOTS >= 1, "a point needs at least one row of lanes");
  typedef typename SegVec<AT>::raw raw_t;
  const int lane = threadIdx.x & 63, hl = lane & (LPP - 1), h = lane / LPP;
  const int cl = hl % LPR, slot = hl / LPR, c0 = cl * VEC;
  BNc b[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) b[k] = load_bn(bn, c0 + k);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t p2 = wave * PPW; p2 < N; p2 += n_waves * PPW) {
    const int64_t p = p2 + h;
    const bool valid = p < N;
    const int64_t beg = valid ? ptr[p] : 0;
    const int64_t end = valid ? ptr[p + 1] : 0;
    float m[VEC];
    int am[VEC];            // row offset inside the segment (32-bit: halves the shuffle / select work)
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      m[k] = -INFINITY;
      am[k] = -1;
    }
    // the points of a wavefront iterate together (shuffles below need every lane)
    const int n_seg = (int)(end - beg);
    int n_max = n_seg;
#pragma unroll
    for (int off = LPP; off < 64; off <<= 1) n_max = max(n_max, __shfl_xor(n_max, off));
    for (int i0 = 0; i0 < n_max; i0 += SLOTS * U) {
      raw_t raw[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * SLOTS + slot;
        ok[u] = i < n_seg;
        const int64_t r = n_seg > 0 ? beg + (ok[u] ? i : 0) : 0;   // clamped: loads are unconditional
        raw[u] = *reinterpret_cast<const raw_t*>(a + r * D + c0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float f[VEC];
        SegVec<AT>::unpack(raw[u], f);
        const int i = i0 + u * SLOTS + slot;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const float v = leaky(bn_z(f[k], b[k]));
          if (ok[u] && v > m[k]) {   // rows ascend within a lane: strict > keeps the first row on ties
            m[k] = v;
            am[k] = i;
          }
        }
      }
    }
    // merge the row slots: lanes hl, hl ^ off hold the same channels for different rows
#pragma unroll
    for (int off = LPR; off < LPP; off <<= 1) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float ov = __shfl_xor(m[k], off);
        const int oa = __shfl_xor(am[k], off);
        if (oa >= 0 && (am[k] < 0 || ov > m[k] || (ov == m[k] && oa < am[k]))) {
          m[k] = ov;
          am[k] = oa;
        }
      }
    }
    if (valid && slot == 0) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        pooled[p * D + c0 + k] = am[k] >= 0 ? m[k] : 0.f;
        arg[p * D + c0 + k] = am[k] >= 0 ? (int32_t)(beg + am[k]) : -1;
      }
    }
  }
}

// a_out[v] = leaky(BN_in(a_in[v])) . W^T (+ addend[vp[v]]) ; statistics of a_out
template <bool HAS_ADD>
__global__ __launch_bounds__(256) void dsf_fwd_layer_kernel(
    const float* __restrict__ a_in, const float* __restrict__ bn_in, const float* __restrict__ W,
    const float* __restrict__ addend, const int32_t* __restrict__ vp, float* __restrict__ a_out,
    double* __restrict__ stats, int64_t V) {
  __shared__ __attribute__((aligned(16))) float s_in[4][TILE * D];
  __shared__ float s_add[HAS_ADD ? 4 : 1][HAS_ADD ? TILE * D : 1];
  __shared__ float s_red[2 * D];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n = lane & 31, h = lane >> 5;
  const BNc b = load_bn(bn_in, n);
  float w[D];
#pragma unroll
  for (int k = 0; k < D; ++k) w[k] = W[n * D + k];
  float st[2] = {0.f, 0.f};
  const int64_t tiles = (V + TILE - 1) / TILE;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave; t < tiles; t += n_waves) {
    const int64_t row0 = t * TILE;
    // load phase: every global load of the tile is issued before the first use (the per-point
    // addend is a dependent load through vp: two latencies per tile instead of one per row)
    float av[16];
    int32_t pv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t r = row0 + 2 * i + h;
      av[i] = r < V ? a_in[r * D + n] : 0.f;
      if (HAS_ADD) pv[i] = r < V ? vp[r] : 0;
    }
    if (HAS_ADD) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s_add[wv][(2 * i + h) * D + n] = addend[(int64_t)pv[i] * D + n];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t r = row0 + 2 * i + h;
      s_in[wv][(2 * i + h) * D + n] = r < V ? leaky(bn_z(av[i], b)) : 0.f;
    }
    wave_sync();
#pragma unroll 2
    for (int i = 0; i < 16; ++i) {
      const int64_t r = row0 + 2 * i + h;
      float acc = dot_row<D>(&s_in[wv][(2 * i + h) * D], w);
      if (r < V) {
        if (HAS_ADD) acc += s_add[wv][(2 * i + h) * D + n];
        a_out[r * D + n] = acc;
        st[0] += acc;
        st[1] += acc * acc;
      }
    }
    wave_sync();
  }
  flush_channel_sums<2>(st, stats, s_red, lane);
}

// compat[v, g] = leaky(BN(a[v])) . Ws[g] + bs[g],  G <= 32.  Lanes = (g, row-in-pass).
__global__ __launch_bounds__(256) void dsf_fwd_score_kernel(const float* __restrict__ a,
                                                             const float* __restrict__ bn,
                                                             const float* __restrict__ Ws,
                                                             const float* __restrict__ bs,
                                                             float* __restrict__ compat, int64_t V,
                                                             int G, int GP /* pow2 >= G */) {
  __shared__ __attribute__((aligned(16))) float s_in[4][TILE * D];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n = lane & 31, h = lane >> 5;
  const BNc b = load_bn(bn, n);
  const int g = lane & (GP - 1), rr = lane / GP, rpp = 64 / GP;
  float w[D];
#pragma unroll
  for (int k = 0; k < D; ++k) w[k] = g < G ? Ws[g * D + k] : 0.f;
  const float bias = g < G ? bs[g] : 0.f;
  const int64_t tiles = (V + TILE - 1) / TILE;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave; t < tiles; t += n_waves) {
    const int64_t row0 = t * TILE;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t r = row0 + 2 * i + h;
      s_in[wv][(2 * i + h) * D + n] = r < V ? leaky(bn_z(a[r * D + n], b)) : 0.f;
    }
    wave_sync();
    for (int row = rr; row < TILE; row += rpp) {
      const int64_t r = row0 + row;
      const float acc = dot_row<D>(&s_in[wv][row * D], w) + bias;
      if (r < V && g < G) compat[r * G + g] = acc;
    }
    wave_sync();
  }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------

// Score layer: dz[v,k] = (sum_g dc[v,g] Ws[g,k]) * leaky'(z[v,k]);  dWs, dbs;  S1 = sum dz, S2 = sum dz*a_hat
template <int GMAX>
__global__ __launch_bounds__(256) void dsf_bwd_score_kernel(
    const float* __restrict__ dcompat, const float* __restrict__ a, const float* __restrict__ bn,
    const float* __restrict__ Ws, float* __restrict__ dz, float* __restrict__ dWs,
    float* __restrict__ dbs, double* __restrict__ st, int64_t V, int G) {
  __shared__ float s_red[(2 + GMAX) * D];
  const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
  const BNc b = load_bn(bn, n);
  float ws[GMAX], acc[2 + GMAX], dbacc[GMAX];
#pragma unroll
  for (int g = 0; g < GMAX; ++g) {
    ws[g] = g < G ? Ws[g * D + n] : 0.f;
    dbacc[g] = 0.f;
  }
#pragma unroll
  for (int j = 0; j < 2 + GMAX; ++j) acc[j] = 0.f;
  constexpr int RP = GMAX <= 8 ? 8 : 1;  // row pairs per iteration (independent loads in flight)
  const int64_t chunks = (V + 2 * RP - 1) / (2 * RP);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t q = wave; q < chunks; q += n_waves) {
    float avs[RP], dcs[RP][GMAX];
#pragma unroll
    for (int j = 0; j < RP; ++j) {
      const int64_t r = (q * RP + j) * 2 + h;
      avs[j] = r < V ? a[r * D + n] : 0.f;
#pragma unroll
      for (int g = 0; g < GMAX; ++g) dcs[j][g] = (g < G && r < V) ? dcompat[r * G + g] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < RP; ++j) {
      const int64_t r = (q * RP + j) * 2 + h;
      if (r >= V) continue;
      const float ah = bn_hat(avs[j], b);
      const float z = ah * b.gamma + b.beta;
      const float x = leaky(z);
      float dx = 0.f;
#pragma unroll
      for (int g = 0; g < GMAX; ++g) {
        const float dc = dcs[j][g];
        dx = fmaf(dc, ws[g], dx);
        acc[2 + g] = fmaf(dc, x, acc[2 + g]);
        dbacc[g] += dc;
      }
      const float d = dx * dleaky(z);
      dz[r * D + n] = d;
      acc[0] += d;
      acc[1] = fmaf(d, ah, acc[1]);
    }
  }
  // S1, S2 (double) and dWs (float) through the block reduction
  const int nn = lane & 31;
  for (int i = threadIdx.x; i < (2 + GMAX) * D; i += blockDim.x) s_red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2 + GMAX; ++j) {
    const float v = acc[j] + __shfl_xor(acc[j], 32);
    if (lane < 32) atomicAdd(&s_red[j * D + nn], v);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += blockDim.x) atomicAdd(&st[i], (double)s_red[i]);
  for (int i = threadIdx.x; i < G * D; i += blockDim.x) atomicAdd(&dWs[i], s_red[2 * D + i]);
  // bias gradient: lanes n == 0 of both halves hold complete per-row sums; they meet in LDS, one global atomic per
  // block and group (atomic requests to one cache line are served one after the other)
  __syncthreads();
  if (threadIdx.x < GMAX) s_red[threadIdx.x] = 0.f;
  __syncthreads();
  if (nn == 0) {
#pragma unroll
    for (int g = 0; g < GMAX; ++g)
      if (g < G && dbacc[g] != 0.f) atomicAdd(&s_red[g], dbacc[g]);
  }
  __syncthreads();
  if ((int)threadIdx.x < G && s_red[threadIdx.x] != 0.f) atomicAdd(&dbs[threadIdx.x], s_red[threadIdx.x]);
}

// Generic layer backward.  Given dz_L (gradient w.r.t. the BN_L output), a_L, W_L and the layer input
// x_L = leaky(BN_prev(a_prev)):
//   da_L  = gamma_L*invstd_L*(dz_L - S1m_L - a_hat_L*S2m_L)          (S1m, S2m = S1/M, S2/M; 0 in eval)
//   dW_L += da_L^T x_L ;   dx = da_L . W_L
//   RAW_OUT: out = dx                       (concat layer: the max-pool path is added later)
//   else   : out = dz_prev = dx*leaky'(z_prev), with S1_prev, S2_prev accumulated
//   dt[vp[v]] += da_L[v]                    (gradient of the per-point addend, optional)
// PREV_XMAP: a_prev is recomputed as x_map . Wa^T (first hidden layer).
template <bool PREV_XMAP, bool RAW_OUT>
__global__ __launch_bounds__(256) void dsf_bwd_layer_kernel(
    const float* __restrict__ dz_L, const float* __restrict__ a_L, const float* __restrict__ bn_L,
    const float* __restrict__ sm_L /* [2][D] S1/M | S2/M */, const float* __restrict__ W_L,
    const float* __restrict__ a_prev /* [V,D] or x_map [V,8] */, const float* __restrict__ Wa,
    const float* __restrict__ bn_prev, float* __restrict__ out, float* __restrict__ dW,
    double* __restrict__ st_prev, float* __restrict__ dt, const int32_t* __restrict__ vp, int64_t V) {
  __shared__ __attribute__((aligned(16))) float s_da[4][TILE * D];
  __shared__ __attribute__((aligned(16))) float s_x[4][TILE * D];
  __shared__ __attribute__((aligned(16))) float s_xm[4][TILE * 8];
  __shared__ float s_ap[4][TILE * D];
  __shared__ float s_red[D * D];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n = lane & 31, h = lane >> 5;
  const BNc bL = load_bn(bn_L, n), bp = load_bn(bn_prev, n);
  const float s1m = sm_L[n], s2m = sm_L[D + n];
  const float gsc = bL.gamma * bL.invstd;
  float wt[D];  // column n of W_L: dx[k=n] = sum_n' da[n'] W_L[n'][n]
#pragma unroll
  for (int k = 0; k < D; ++k) wt[k] = W_L[k * D + n];
  float wa[8];
  if (PREV_XMAP) {
#pragma unroll
    for (int k = 0; k < 8; ++k) wa[k] = Wa[n * 8 + k];
  }
  float dWacc[D];
#pragma unroll
  for (int k = 0; k < D; ++k) dWacc[k] = 0.f;
  float st[2] = {0.f, 0.f};
  const int64_t tiles = (V + TILE - 1) / TILE;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave; t < tiles; t += n_waves) {
    const int64_t row0 = t * TILE;
    if (PREV_XMAP) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = i * 64 + lane;
        const int64_t g = row0 * 8 + e;
        s_xm[wv][e] = g < V * 8 ? a_prev[g] : 0.f;
      }
      wave_sync();
    }
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int row = 2 * i + h;
      const int64_t r = row0 + row;
      float da = 0.f, ap = 0.f, x = 0.f;
      if (r < V) {
        const float ah = bn_hat(a_L[r * D + n], bL);
        da = gsc * (dz_L[r * D + n] - s1m - ah * s2m);
        ap = PREV_XMAP ? dot_row<8>(&s_xm[wv][row * 8], wa) : a_prev[r * D + n];
        x = leaky(bn_z(ap, bp));
      }
      s_da[wv][row * D + n] = da;
      s_x[wv][row * D + n] = x;
      s_ap[wv][row * D + n] = ap;
    }
    wave_sync();
    int32_t cur_p = -1;
    float cur_s = 0.f;
#pragma unroll 2
    for (int i = 0; i < 16; ++i) {
      const int row = 2 * i + h;
      const int64_t r = row0 + row;
      // dW[n][k] += da[v][n] * x[v][k]
      const float da = s_da[wv][row * D + n];
#pragma unroll
      for (int k4 = 0; k4 < D / 4; ++k4) {
        const float4 x = *reinterpret_cast<const float4*>(&s_x[wv][row * D + 4 * k4]);
        dWacc[4 * k4] = fmaf(da, x.x, dWacc[4 * k4]);
        dWacc[4 * k4 + 1] = fmaf(da, x.y, dWacc[4 * k4 + 1]);
        dWacc[4 * k4 + 2] = fmaf(da, x.z, dWacc[4 * k4 + 2]);
        dWacc[4 * k4 + 3] = fmaf(da, x.w, dWacc[4 * k4 + 3]);
      }
      // dx[v][n] = sum_n' da[v][n'] * W_L[n'][n]
      const float dx = dot_row<D>(&s_da[wv][row * D], wt);
      if (r < V) {
        if (RAW_OUT) {
          out[r * D + n] = dx;
        } else {
          const float ah = bn_hat(s_ap[wv][row * D + n], bp);
          const float z = ah * bp.gamma + bp.beta;
          const float d = dx * dleaky(z);
          out[r * D + n] = d;
          st[0] += d;
          st[1] = fmaf(d, ah, st[1]);
        }
        if (dt) {
          const int32_t p = vp[r];
          if (p != cur_p) {
            if (cur_p >= 0) atomicAdd(&dt[(int64_t)cur_p * D + n], cur_s);
            cur_p = p;
            cur_s = 0.f;
          }
          cur_s += da;
        }
      }
    }
    if (dt && cur_p >= 0) atomicAdd(&dt[(int64_t)cur_p * D + n], cur_s);
    wave_sync();
  }
  // dW: add the two row-parity halves, reduce over the block in LDS, one atomic per element per block
  for (int i = threadIdx.x; i < D * D; i += blockDim.x) s_red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const float v = dWacc[k] + __shfl_xor(dWacc[k], 32);
    if (lane < 32) atomicAdd(&s_red[n * D + k], v);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < D * D; i += blockDim.x) atomicAdd(&dW[i], s_red[i]);
  if (!RAW_OUT) {
    __syncthreads();
    flush_channel_sums<2>(st, st_prev, s_red, lane);
  }
}

// dz2[v,c] = (dcat[v,c] + [arg[p,c]==v] dpooled[p,c]) * leaky'(z2[v,c]),  p = vp[v];  S1, S2 of BN2
__global__ __launch_bounds__(256) void dsf_bwd_max_kernel(
    const float* __restrict__ dcat, const float* __restrict__ a2, const float* __restrict__ bn2,
    const int32_t* __restrict__ arg, const float* __restrict__ dpooled, const int32_t* __restrict__ vp,
    float* __restrict__ dz2, double* __restrict__ st, int64_t V) {
  __shared__ float s_red[2 * D];
  const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
  const BNc b = load_bn(bn2, n);
  float acc[2] = {0.f, 0.f};
  const int64_t tiles = (V + TILE - 1) / TILE;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave; t < tiles; t += n_waves) {
    const int64_t row0 = t * TILE;
    float gv[16], av[16], dp[16];
    int32_t pv[16], ag[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t r = row0 + 2 * i + h;
      const bool ok = r < V;
      pv[i] = ok ? vp[r] : 0;
      gv[i] = ok ? dcat[r * D + n] : 0.f;
      av[i] = ok ? a2[r * D + n] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      ag[i] = arg[(int64_t)pv[i] * D + n];
      dp[i] = dpooled[(int64_t)pv[i] * D + n];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t r = row0 + 2 * i + h;
      if (r >= V) continue;
      float g = gv[i];
      if ((int64_t)ag[i] == r) g += dp[i];
      const float ah = bn_hat(av[i], b);
      const float z = ah * b.gamma + b.beta;
      const float d = g * dleaky(z);
      dz2[r * D + n] = d;
      acc[0] += d;
      acc[1] = fmaf(d, ah, acc[1]);
    }
  }
  flush_channel_sums<2>(acc, st, s_red, lane);
}

// First layer: a1 = x.Wa^T recomputed, da1 = BN-backward(dz1), dWa[n][j] += da1[v][n] x[v][j]
template <typename AT>
__global__ __launch_bounds__(256) void dsf_bwd_first_kernel(
    const AT* __restrict__ dz1, const float* __restrict__ x_map, const float* __restrict__ Wa,
    const float* __restrict__ bn1, const float* __restrict__ sm1, float* __restrict__ dWa, int64_t V) {
  __shared__ __attribute__((aligned(16))) float s_x[4][TILE * 8];
  __shared__ float s_red[D * 8];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n = lane & 31, h = lane >> 5;
  const BNc b = load_bn(bn1, n);
  const float s1m = sm1[n], s2m = sm1[D + n], gsc = b.gamma * b.invstd;
  float wa[8], acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    wa[k] = Wa[n * 8 + k];
    acc[k] = 0.f;
  }
  const int64_t tiles = (V + TILE - 1) / TILE;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave; t < tiles; t += n_waves) {
    const int64_t row0 = t * TILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = i * 64 + lane;
      const int64_t g = row0 * 8 + e;
      s_x[wv][e] = g < V * 8 ? x_map[g] : 0.f;
    }
    wave_sync();
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int row = 2 * i + h;
      const int64_t r = row0 + row;
      if (r < V) {
        const float a1 = dot_row<8>(&s_x[wv][row * 8], wa);
        const float da = gsc * (Elt<AT>::ld(dz1, r * D + n) - s1m - bn_hat(a1, b) * s2m);
        const float4 x0 = *reinterpret_cast<const float4*>(&s_x[wv][row * 8]);
        const float4 x1 = *reinterpret_cast<const float4*>(&s_x[wv][row * 8 + 4]);
        acc[0] = fmaf(da, x0.x, acc[0]); acc[1] = fmaf(da, x0.y, acc[1]);
        acc[2] = fmaf(da, x0.z, acc[2]); acc[3] = fmaf(da, x0.w, acc[3]);
        acc[4] = fmaf(da, x1.x, acc[4]); acc[5] = fmaf(da, x1.y, acc[5]);
        acc[6] = fmaf(da, x1.z, acc[6]); acc[7] = fmaf(da, x1.w, acc[7]);
      }
    }
    wave_sync();
  }
  for (int i = threadIdx.x; i < D * 8; i += blockDim.x) s_red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float v = acc[k] + __shfl_xor(acc[k], 32);
    if (lane < 32) atomicAdd(&s_red[n * 8 + k], v);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < D * 8; i += blockDim.x) atomicAdd(&dWa[i], s_red[i]);
}

// view -> point index (dense expansion of the CSR pointers): a wavefront takes 64 points, reads their pointers
// coalesced and writes the views of one point after the other with all lanes (coalesced stores; one thread per point
// wrote 4-byte words 128 bytes apart on the 32-views-per-point scenes)
__global__ __launch_bounds__(256) void csr_expand_kernel(const int64_t* __restrict__ ptr, int64_t N,
                                                          int32_t* __restrict__ vp) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t p0 = wave * 64; p0 < N; p0 += n_waves * 64) {
    const int64_t pl = p0 + lane < N ? p0 + lane : N - 1;
    const int64_t b = ptr[pl], e = ptr[pl + 1];
    const int n_here = (int)(N - p0 < 64 ? N - p0 : 64);
    const int64_t first = __shfl(b, 0), last = __shfl(e, n_here - 1);
    if (last - first <= 8 * n_here) {
      // short segments (ragged scenes): every lane writes the few views of its own point
      if (lane < n_here)
        for (int64_t r = b; r < e; ++r) vp[r] = (int32_t)(p0 + lane);
    } else {
      for (int i = 0; i < n_here; ++i) {
        const int64_t bi = __shfl(b, i), ei = __shfl(e, i);
        for (int64_t r = bi + lane; r < ei; r += 64) vp[r] = (int32_t)(p0 + i);
      }
    }
  }
}

static inline int grid_rows(int64_t V) {
  // one wavefront per 32-row tile, 4 wavefronts per block, capped at 8 blocks per CU
  int64_t b = ((V + TILE - 1) / TILE + 3) / 4;
  if (b > 256 * 8) b = 256 * 8;
  if (b < 1) b = 1;
  return (int)b;
}

// fp32-MFMA generation of the layer kernels (deepset_mfma.hip); bf = bf16 activation storage
int dsm_launch_fwd_first(const float*, const float*, const float*, const float*, void*, double*, int64_t,
                         int, int, hipStream_t);
int dsm_launch_fwd_layer(const void*, const float*, const float*, const float*, const int32_t*, void*,
                         double*, int64_t, int, hipStream_t);
int dsm_launch_fwd_score(const void*, const float*, const float*, const float*, float*, int64_t, int,
                         const float*, const float*, int, hipStream_t);
int dsm_launch_bwd_layer(const void*, const void*, const float*, const float*, const float*, const void*,
                         const float*, const float*, void*, float*, double*, float*, const int32_t*, float*,
                         const float*, int64_t, int, int, int, hipStream_t);
int dsm_launch_bwd_max(const void*, const void*, const float*, const int32_t*, const float*, const int32_t*,
                       void*, double*, int64_t, int, hipStream_t);
int dsm_launch_bwd_score(const float*, const void*, const float*, const float*, void*, float*, float*,
                         double*, int64_t, int, const float*, const float*, int, hipStream_t);

// activation storage code of the C ABI -> 0 (fp32) / 1 (bf16) / -1 (invalid)
static inline int act_bf(int32_t act_dtype) {
  return act_dtype == DVA_F32 ? 0 : act_dtype == DVA_BF16 ? 1 : -1;
}

}  // namespace dva

using namespace dva;

extern "C" {

int dva_csr_expand(const int64_t* ptr, int64_t n_groups, int32_t* group_of_row, void* stream) {
  if (n_groups < 0 || !ptr) return DVA_ERR_INVALID;
  if (n_groups == 0) return DVA_OK;
  if (!group_of_row) return DVA_ERR_INVALID;
  int64_t b = (n_groups + 255) / 256;            // 4 wavefronts x 64 points per block
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(csr_expand_kernel, dim3((int)b), dim3(256), 0, (hipStream_t)stream, ptr, n_groups,
                     group_of_row);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_deepset_fwd_first(const float* x_map, const float* Wa, const float* bn1, const float* Wb,
                          void* a2_, double* stats, int64_t V, int32_t F, int32_t stats_only,
                          int32_t algo, int32_t act_dtype, void* stream) {
  const int bf = act_bf(act_dtype);
  if (V < 0 || !stats || bf < 0) return DVA_ERR_INVALID;
  if (bf && algo == 1) return DVA_ERR_UNSUPPORTED;  // bf16 storage only in the MFMA generation
  float* a2 = (float*)a2_;
  if (F != 8) return DVA_ERR_UNSUPPORTED;
  if (V == 0) return DVA_OK;
  if (!x_map || !Wa) return DVA_ERR_INVALID;
  if (!stats_only && (!bn1 || !Wb || !a2)) return DVA_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  if (algo != 1) {
    dsm_launch_fwd_first(x_map, Wa, bn1, Wb, a2_, stats, V, stats_only, bf, s);
    DVA_CHECK_LAUNCH();
    return DVA_OK;
  }
  if (stats_only) {
    hipLaunchKernelGGL((dsf_fwd_first_kernel<true>), dim3(grid_rows(V)), dim3(256), 0, s, x_map, Wa,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr, stats, V);
  } else {
    if (!bn1 || !Wb || !a2) return DVA_ERR_INVALID;
    hipLaunchKernelGGL((dsf_fwd_first_kernel<false>), dim3(grid_rows(V)), dim3(256), 0, s, x_map, Wa,
                       bn1, Wb, a2, stats, V);
  }
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_deepset_segmax(const void* a, const float* bn, const int64_t* ptr, float* pooled,
                       int32_t* arg, int64_t N, int64_t n_views, int32_t act_dtype, void* stream) {
  const int bf = act_bf(act_dtype);
  if (N < 0 || bf < 0) return DVA_ERR_INVALID;
  if (N == 0) return DVA_OK;
  if (!a || !bn || !ptr || !pooled || !arg) return DVA_ERR_INVALID;
  // lanes per point from the average segment: a batch covers 4 * (lanes / lanes-per-row) views
  const double avg = n_views > 0 ? (double)n_views / (double)N : 32.0;
  const int lpr = bf ? 4 : 8;
  int lpp = 32;
  while (lpp > lpr && lpp > 8 && 4.0 * (lpp / 2 / lpr) >= 3.0 * avg) lpp >>= 1;
  const int ppb = 4 * (64 / lpp);   // points per block
  int64_t b = (N + ppb - 1) / ppb;
  if (b > 256 * 8) b = 256 * 8;
  hipStream_t s = (hipStream_t)stream;
#define DVA_SM(T, L) \
  hipLaunchKernelGGL((dsf_segmax_kernel<T, L>), dim3((int)b), dim3(256), 0, s, (const T*)a, bn, ptr, pooled, arg, N)
  if (bf) {
    if (lpp == 32) DVA_SM(bf16_t, 32);
    else if (lpp == 16) DVA_SM(bf16_t, 16);
    else DVA_SM(bf16_t, 8);
  } else {
    if (lpp == 32) DVA_SM(float, 32);
    else if (lpp == 16) DVA_SM(float, 16);
    else DVA_SM(float, 8);
  }
#undef DVA_SM
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_deepset_fwd_layer(const void* a_in_, const float* bn_in, const float* W, const float* addend,
                          const int32_t* group_of_row, void* a_out_, double* stats, int64_t V,
                          int32_t algo, int32_t act_dtype, void* stream) {
  const int bf = act_bf(act_dtype);
  if (V < 0 || !stats || bf < 0) return DVA_ERR_INVALID;
  if (bf && algo == 1) return DVA_ERR_UNSUPPORTED;
  const float* a_in = (const float*)a_in_;
  float* a_out = (float*)a_out_;
  if (V == 0) return DVA_OK;
  if (!a_in || !W) return DVA_ERR_INVALID;
  if (!a_out && algo == 1) return DVA_ERR_UNSUPPORTED;   // statistics-only pass: MFMA generation
  if (addend && !group_of_row) return DVA_ERR_INVALID;
  if (!bn_in && algo == 1) return DVA_ERR_UNSUPPORTED;  // raw-input layers only in the MFMA generation
  hipStream_t s = (hipStream_t)stream;
  if (algo != 1) {
    dsm_launch_fwd_layer(a_in_, bn_in, W, addend, group_of_row, a_out_, stats, V, bf, s);
    DVA_CHECK_LAUNCH();
    return DVA_OK;
  }
  if (addend)
    hipLaunchKernelGGL((dsf_fwd_layer_kernel<true>), dim3(grid_rows(V)), dim3(256), 0, s, a_in, bn_in, W,
                       addend, group_of_row, a_out, stats, V);
  else
    hipLaunchKernelGGL((dsf_fwd_layer_kernel<false>), dim3(grid_rows(V)), dim3(256), 0, s, a_in, bn_in,
                       W, (const float*)nullptr, (const int32_t*)nullptr, a_out, stats, V);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_deepset_fwd_score(const void* a_, const float* bn, const float* Ws, const float* bs,
                          float* compat, int64_t V, int32_t G, const float* bn_pre, const float* W_pre,
                          int32_t algo, int32_t act_dtype, void* stream) {
  const int bf = act_bf(act_dtype);
  if (V < 0 || G <= 0 || G > 32 || bf < 0) return DVA_ERR_INVALID;
  if (bf && algo == 1) return DVA_ERR_UNSUPPORTED;
  if ((bn_pre == nullptr) != (W_pre == nullptr)) return DVA_ERR_INVALID;
  if (W_pre && algo == 1) return DVA_ERR_UNSUPPORTED;
  const float* a = (const float*)a_;
  if (V == 0) return DVA_OK;
  if (!a || !bn || !Ws || !bs || !compat) return DVA_ERR_INVALID;
  if (algo != 1) {
    dsm_launch_fwd_score(a_, bn, Ws, bs, compat, V, G, bn_pre, W_pre, bf, (hipStream_t)stream);
    DVA_CHECK_LAUNCH();
    return DVA_OK;
  }
  int GP = 1;
  while (GP < G) GP <<= 1;
  hipLaunchKernelGGL(dsf_fwd_score_kernel, dim3(grid_rows(V)), dim3(256), 0, (hipStream_t)stream, a, bn,
                     Ws, bs, compat, V, G, GP);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_deepset_bwd_score(const float* dcompat, const void* a_, const float* bn, const float* Ws,
                          void* dz_, float* dWs, float* dbs, double* st, int64_t V, int32_t G,
                          const float* bn_pre, const float* W_pre, int32_t algo, int32_t act_dtype,
                          void* stream) {
  const int bf = act_bf(act_dtype);
  if (V < 0 || G <= 0 || bf < 0) return DVA_ERR_INVALID;
  if (bf && algo == 1) return DVA_ERR_UNSUPPORTED;
  if ((bn_pre == nullptr) != (W_pre == nullptr)) return DVA_ERR_INVALID;
  if (W_pre && (!bf || algo == 1)) return DVA_ERR_UNSUPPORTED;   // recompute: bf16 storage, MFMA generation
  const float* a = (const float*)a_;
  float* dz = (float*)dz_;
  if (G > 32) return DVA_ERR_UNSUPPORTED;
  if (V == 0) return DVA_OK;
  if (!dcompat || !a || !bn || !Ws || !dz || !dWs || !dbs || !st) return DVA_ERR_INVALID;
  if (algo != 1) {
    dsm_launch_bwd_score(dcompat, a_, bn, Ws, dz_, dWs, dbs, st, V, G, bn_pre, W_pre, bf, (hipStream_t)stream);
    DVA_CHECK_LAUNCH();
    return DVA_OK;
  }
  if (G <= 8)
    hipLaunchKernelGGL((dsf_bwd_score_kernel<8>), dim3(grid_rows(V)), dim3(256), 0, (hipStream_t)stream,
                       dcompat, a, bn, Ws, dz, dWs, dbs, st, V, G);
  else
    hipLaunchKernelGGL((dsf_bwd_score_kernel<32>), dim3(grid_rows(V)), dim3(256), 0,
                       (hipStream_t)stream, dcompat, a, bn, Ws, dz, dWs, dbs, st, V, G);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_deepset_bwd_layer(const void* dz_L_, const void* a_L_, const float* bn_L, const float* sm_L,
                          const float* W_L, const void* a_prev_, const float* Wa, const float* bn_prev,
                          void* out_, float* dW, double* st_prev, float* dt,
                          const int32_t* group_of_row, float* first_grad, const float* addend, int64_t V,
                          int32_t prev_is_xmap, int32_t raw_out, int32_t algo, int32_t act_dtype,
                          void* stream) {
  const int bf = act_bf(act_dtype);
  if (V < 0 || bf < 0) return DVA_ERR_INVALID;
  if (bf && algo == 1) return DVA_ERR_UNSUPPORTED;
  // fused first-layer gradient: bf16 storage, x_map input, batch-norm output, no per-point sum
  if (first_grad && (!bf || !prev_is_xmap || raw_out || dt)) return DVA_ERR_UNSUPPORTED;
  const float *dz_L = (const float*)dz_L_, *a_L = (const float*)a_L_, *a_prev = (const float*)a_prev_;
  float* out = (float*)out_;
  if (V == 0) return DVA_OK;
  if (!dz_L || !bn_L || !sm_L || !W_L || !a_prev || (!out && !first_grad) || !dW) return DVA_ERR_INVALID;
  // a_L == NULL: the layer output is recomputed from its input (bf16 storage, MFMA generation)
  if (!a_L && (!bf || algo == 1 || !bn_prev)) return DVA_ERR_UNSUPPORTED;
  if (addend && (a_L || !group_of_row)) return DVA_ERR_INVALID;
  if (!bn_prev && (!raw_out || algo == 1 || prev_is_xmap)) return DVA_ERR_UNSUPPORTED;
  if (!raw_out && !st_prev) return DVA_ERR_INVALID;
  if (prev_is_xmap && !Wa) return DVA_ERR_INVALID;
  if (dt && !group_of_row) return DVA_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  if (algo != 1) {
    if (dsm_launch_bwd_layer(dz_L_, a_L_, bn_L, sm_L, W_L, a_prev_, Wa, bn_prev, out_, dW, st_prev, dt,
                             group_of_row, first_grad, addend, V, prev_is_xmap, raw_out, bf, s))
      return DVA_ERR_UNSUPPORTED;
    DVA_CHECK_LAUNCH();
    return DVA_OK;
  }
  const dim3 grid(grid_rows(V)), block(256);
#define DVA_L(P, R)                                                                                \
  hipLaunchKernelGGL((dsf_bwd_layer_kernel<P, R>), grid, block, 0, s, dz_L, a_L, bn_L, sm_L, W_L,  \
                     a_prev, Wa, bn_prev, out, dW, st_prev, dt, group_of_row, V)
  if (prev_is_xmap && raw_out) DVA_L(true, true);
  else if (prev_is_xmap) DVA_L(true, false);
  else if (raw_out) DVA_L(false, true);
  else DVA_L(false, false);
#undef DVA_L
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_deepset_bwd_max(const void* dcat_, const void* a2_, const float* bn2, const int32_t* arg,
                        const float* dpooled, const int32_t* group_of_row, void* dz2_, double* st,
                        int64_t V, int32_t algo, int32_t act_dtype, void* stream) {
  const int bf = act_bf(act_dtype);
  if (V < 0 || bf < 0) return DVA_ERR_INVALID;
  if (bf && algo == 1) return DVA_ERR_UNSUPPORTED;
  const float *dcat = (const float*)dcat_, *a2 = (const float*)a2_;
  float* dz2 = (float*)dz2_;
  if (V == 0) return DVA_OK;
  if (!dcat || !a2 || !bn2 || !arg || !dpooled || !group_of_row || !dz2 || !st) return DVA_ERR_INVALID;
  if (algo != 1) {
    dsm_launch_bwd_max(dcat_, a2_, bn2, arg, dpooled, group_of_row, dz2_, st, V, bf, (hipStream_t)stream);
    DVA_CHECK_LAUNCH();
    return DVA_OK;
  }
  hipLaunchKernelGGL(dsf_bwd_max_kernel, dim3(grid_rows(V)), dim3(256), 0, (hipStream_t)stream, dcat, a2,
                     bn2, arg, dpooled, group_of_row, dz2, st, V);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_deepset_bwd_first(const void* dz1, const float* x_map, const float* Wa, const float* bn1,
                          const float* sm1, float* dWa, int64_t V, int32_t F, int32_t act_dtype,
                          void* stream) {
  const int bf = act_bf(act_dtype);
  if (V < 0 || bf < 0) return DVA_ERR_INVALID;
  if (F != 8) return DVA_ERR_UNSUPPORTED;
  if (V == 0) return DVA_OK;
  if (!dz1 || !x_map || !Wa || !bn1 || !sm1 || !dWa) return DVA_ERR_INVALID;
  if (bf)
    hipLaunchKernelGGL((dsf_bwd_first_kernel<bf16_t>), dim3(grid_rows(V)), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)dz1, x_map, Wa, bn1, sm1, dWa, V);
  else
    hipLaunchKernelGGL((dsf_bwd_first_kernel<float>), dim3(grid_rows(V)), dim3(256), 0,
                       (hipStream_t)stream, (const float*)dz1, x_map, Wa, bn1, sm1, dWa, V);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // extern "C"
