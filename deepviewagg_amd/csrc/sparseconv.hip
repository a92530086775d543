// Sparse 3D convolution on voxel tensors for gfx950 (the consumer of the fused
// features, reference modules/SparseConv3d/modules.py:103-220 over torchsparse 1.1.0 `Conv3d`, which is not in
// the reference tree).  torchsparse runs gather -> GEMM -> scatter per kernel offset (three passes over HBM
// and a scatter-add).  Here the convolution is OUTPUT-STATIONARY over the kernel map of voxel.hip
// (nbr[k][j] = source row of destination voxel j under offset k, or -1):
//
//   out[j, :] = bias + sum_k  x[nbr[k][j], :] @ W_k
//
// One wavefront owns 64 destination voxels and a 64-channel slice of the output; the 32x32 accumulator
// blocks stay in registers over all offsets and input channels, the gathered rows go straight from global
// memory into the B operand of v_mfma_f32_32x32x16_bf16 (a lane supplies 8 consecutive channels of one
// voxel = one 16-byte load), the weight tile of the current (offset, channel chunk) is staged in LDS once
// per workgroup (double buffered, one barrier per tile).  Offsets under which none of the wavefront's 64
// voxels has a neighbour are skipped.  No scatter, no atomics, every output row is written exactly once.
// The same kernel serves the forward pass, the input gradient (transposed kernel map, W_k^T) and the
// transposed convolution of the decoder.
//
// fp32 features keep fp32 accuracy with the split x = hi + lo (three bf16 products, as deepset_mfma.hip);
// bf16 features use the bf16-rounded weights (what autocast would feed a GEMM), fp32 accumulation.
//
// Weight gradient: dW[k][a][b] = sum_j x[nbr[k][j]][a] * dout[j][b], reduction over voxels.  A wavefront
// transposes 32-voxel tiles of the gathered rows and of dout through its private LDS tiles ([channel][voxel],
// bf16) so that the 8 voxels a lane feeds to the MFMA are one ds_read_b128, keeps the 64 x 64 block of dW in
// registers over its share of the voxels and adds it to dW with fp32 atomics at the end.
#include "dva_common.h"

namespace dva {

typedef float sc_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 sc_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t sc_u4 __attribute__((ext_vector_type(4)));
#define SC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int SC_TILE = 64;  // output channels per workgroup, input channels per staged chunk
constexpr int SC_WS = 72;    // LDS row stride of the weight tile in bf16 (144 B: 16-byte aligned, 8 lanes of
                             // a ds_read_b128 phase cover the 32 banks once)
constexpr int SC_TSB = 40;   // row stride of the transposed 32-voxel tiles of the weight gradient

__device__ __forceinline__ int sc_acc_chan(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ void sc_split8(const float* x, sc_bf16x8& hi, sc_bf16x8& lo) {
  uint32_t H[4], L[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    H[p] = pack_bf16x2(x[2 * p], x[2 * p + 1]);
    const float h0 = __uint_as_float(H[p] << 16), h1 = __uint_as_float(H[p] & 0xffff0000u);
    L[p] = pack_bf16x2(x[2 * p] - h0, x[2 * p + 1] - h1);
  }
  const sc_u4 hv = {H[0], H[1], H[2], H[3]}, lv = {L[0], L[1], L[2], L[3]};
  hi = __builtin_bit_cast(sc_bf16x8, hv);
  lo = __builtin_bit_cast(sc_bf16x8, lv);
}

// 8 consecutive channels of one voxel row as the B operand (hi, and lo for fp32 features); `ok` false -> 0
template <typename T> struct XFrag;
template <> struct XFrag<bf16_t> {
  uint4 raw;
  __device__ __forceinline__ void load(const bf16_t* p) { raw = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void get(bool ok, sc_bf16x8& hi, sc_bf16x8& lo) const {
    const sc_u4 v = {ok ? raw.x : 0u, ok ? raw.y : 0u, ok ? raw.z : 0u, ok ? raw.w : 0u};
    hi = __builtin_bit_cast(sc_bf16x8, v);
    lo = hi;  // unused
  }
};
template <> struct XFrag<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) {
    a = *reinterpret_cast<const float4*>(p);
    b = *reinterpret_cast<const float4*>(p + 4);
  }
  __device__ __forceinline__ void get(bool ok, sc_bf16x8& hi, sc_bf16x8& lo) const {
    const float x[8] = {ok ? a.x : 0.f, ok ? a.y : 0.f, ok ? a.z : 0.f, ok ? a.w : 0.f,
                        ok ? b.x : 0.f, ok ? b.y : 0.f, ok ? b.z : 0.f, ok ? b.w : 0.f};
    sc_split8(x, hi, lo);
  }
};

// Wt[k][n][c] (n = output channel, c = input channel), zero padded to multiples of SC_TILE, bf16 hi | lo.
// mode 0: W is [K, Cin, Cout] (forward: Wt[k][n][c] = W[k][c][n]); mode 1: W is [K, Cout, Cin] seen from this
// op (input gradient: Wt[k][n][c] = W[k][n][c]).
__global__ __launch_bounds__(256) void sconv_prep_weights_kernel(const float* __restrict__ W, int K, int Cin,
                                                                  int Cout, int Cin_p, int Cout_p, int mode,
                                                                  bf16_t* __restrict__ w_hi,
                                                                  bf16_t* __restrict__ w_lo) {
  const int64_t total = (int64_t)K * Cout_p * Cin_p;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(t % Cin_p);
    const int n = (int)((t / Cin_p) % Cout_p);
    const int k = (int)(t / ((int64_t)Cin_p * Cout_p));
    float w = 0.f;
    if (c < Cin && n < Cout)
      w = mode == 0 ? W[((int64_t)k * Cin + c) * Cout + n] : W[((int64_t)k * Cout + n) * Cin + c];
    const bf16_t h = f2bf(w);
    w_hi[t] = h;
    if (w_lo) w_lo[t] = f2bf(w - bf2f(h));
  }
}

template <typename T> struct OutStore;
template <> struct OutStore<float> {
  static __device__ __forceinline__ void st4(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
  }
};
template <> struct OutStore<bf16_t> {
  static __device__ __forceinline__ void st4(bf16_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
  }
};

// NT = 32-channel output blocks per wavefront (2: 64 output channels per workgroup, 4: 128 -- wide layers
// gather each input row once per 128 output channels instead of once per 64)
template <typename T, int NT>
__global__ __launch_bounds__(256, NT == 4 ? 2 : 1) void sconv_apply_kernel(const T* __restrict__ x, const int32_t* __restrict__ nbr,
                                                           const bf16_t* __restrict__ w_hi,
                                                           const bf16_t* __restrict__ w_lo,
                                                           const float* __restrict__ bias, T* __restrict__ out,
                                                           int64_t n_dst, int K, int Cin, int Cout, int Cin_p,
                                                           int Cout_p) {
  constexpr bool SPLIT = sizeof(T) == 4;
  constexpr int NB = SPLIT ? 2 : 1;
  constexpr int NROWS = 32 * NT;                                 // output channels of this workgroup
  extern __shared__ __attribute__((aligned(16))) bf16_t s_w[];  // [2 buffers][NB][NROWS][SC_WS]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int jl = lane & 31, h = lane >> 5;
  const int64_t j0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
  const int n0 = blockIdx.y * NROWS;
  const int chunks = Cin_p / SC_TILE;
  const int total = K * chunks;

  sc_f32x16 acc[NT][2];
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][t][r] = 0.f;

  auto stage = [&](int it, int buf) {
    const int k = it / chunks, cc = it - k * chunks;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int p = threadIdx.x + 256 * i;  // NROWS * 8 pieces of 16 bytes
      const int n = p >> 3, c8 = p & 7;
      const int64_t g = ((int64_t)k * Cout_p + n0 + n) * Cin_p + cc * SC_TILE + c8 * 8;
      bf16_t* dst = s_w + ((size_t)(buf * NB) * NROWS + n) * SC_WS + c8 * 8;
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(w_hi + g);  // Cout_p: multiple of NROWS
      if (SPLIT) *reinterpret_cast<uint4*>(dst + NROWS * SC_WS) = *reinterpret_cast<const uint4*>(w_lo + g);
    }
  };

  stage(0, 0);
  for (int it = 0; it < total; ++it) {
    __syncthreads();
    if (it + 1 < total) stage(it + 1, (it + 1) & 1);
    const int k = it / chunks, cc = it - k * chunks;
    int idx[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t j = j0 + 32 * t + jl;
      idx[t] = j < n_dst ? nbr[(int64_t)k * n_dst + j] : -1;
    }
    if (__ballot(idx[0] >= 0 || idx[1] >= 0) == 0ull) continue;  // no neighbour under this offset
    const int c0 = cc * SC_TILE;
    const int msteps = min(4, (Cin - c0) / 16);  // Cin is a multiple of 16
    XFrag<T> xf[2][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
      if (m < msteps) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          xf[t][m].load(x + (int64_t)max(idx[t], 0) * Cin + c0 + 16 * m + 8 * h);
      }
    const bf16_t* wb = s_w + (size_t)((it & 1) * NB) * NROWS * SC_WS;
#pragma unroll
    for (int m = 0; m < 4; ++m)
      if (m < msteps) {
        sc_bf16x8 bh[2], bl[2], ah[NT], al[NT];
#pragma unroll
        for (int t = 0; t < 2; ++t) xf[t][m].get(idx[t] >= 0, bh[t], bl[t]);
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          const bf16_t* wp = wb + (32 * u + jl) * SC_WS + 16 * m + 8 * h;
          ah[u] = *reinterpret_cast<const sc_bf16x8*>(wp);
          if (SPLIT) al[u] = *reinterpret_cast<const sc_bf16x8*>(wp + NROWS * SC_WS);
        }
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            acc[u][t] = SC_MFMA(ah[u], bh[t], acc[u][t]);
            if (SPLIT) {
              acc[u][t] = SC_MFMA(al[u], bh[t], acc[u][t]);
              acc[u][t] = SC_MFMA(ah[u], bl[t], acc[u][t]);
            }
          }
      }
  }

  // epilogue: lane (voxel jl of tile t, half h) holds channels n0 + 32 u + 8 q + 4 h + (0..3)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int64_t j = j0 + 32 * t + jl;
    if (j >= n_dst) continue;
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + 32 * u + 8 * q + 4 * h;
        if (n < Cout) {  // Cout is a multiple of 4
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[u][t][4 * q + e] + (bias ? bias[n + e] : 0.f);
          OutStore<T>::st4(out + j * Cout + n, v[0], v[1], v[2], v[3]);
        }
      }
  }
}

// ---- weight gradient ----------------------------------------------------------------------------
template <typename T> struct Row16;
template <> struct Row16<bf16_t> {
  uint4 q[2];
  __device__ __forceinline__ void load(const bf16_t* p) {
    q[0] = *reinterpret_cast<const uint4*>(p);
    q[1] = *reinterpret_cast<const uint4*>(p + 8);
  }
  __device__ __forceinline__ void get(bool ok, float (&x)[16]) const {
    const uint32_t w[8] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      x[2 * e] = ok ? __uint_as_float(w[e] << 16) : 0.f;
      x[2 * e + 1] = ok ? __uint_as_float(w[e] & 0xffff0000u) : 0.f;
    }
  }
};
template <> struct Row16<float> {
  float4 q[4];
  __device__ __forceinline__ void load(const float* p) {
#pragma unroll
    for (int e = 0; e < 4; ++e) q[e] = *reinterpret_cast<const float4*>(p + 4 * e);
  }
  __device__ __forceinline__ void get(bool ok, float (&x)[16]) const {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x[4 * e] = ok ? q[e].x : 0.f;
      x[4 * e + 1] = ok ? q[e].y : 0.f;
      x[4 * e + 2] = ok ? q[e].z : 0.f;
      x[4 * e + 3] = ok ? q[e].w : 0.f;
    }
  }
};

// [channel][voxel] bf16 tile of 32 x 32: lane (voxel v, half h) writes its 16 channels 16 h .. 16 h + 15
template <bool SPLIT>
__device__ __forceinline__ void sc_tileT_put(bf16_t* hi, bf16_t* lo, int v, int h, const float (&x)[16]) {
#pragma unroll
  for (int s = 0; s < 16; s += 2) {
    const uint32_t d = pack_bf16x2(x[s], x[s + 1]);
    hi[(16 * h + s) * SC_TSB + v] = (bf16_t)(d & 0xffffu);
    hi[(16 * h + s + 1) * SC_TSB + v] = (bf16_t)(d >> 16);
    if (SPLIT) {
      const uint32_t e = pack_bf16x2(x[s] - __uint_as_float(d << 16), x[s + 1] - __uint_as_float(d & 0xffff0000u));
      lo[(16 * h + s) * SC_TSB + v] = (bf16_t)(e & 0xffffu);
      lo[(16 * h + s + 1) * SC_TSB + v] = (bf16_t)(e >> 16);
    }
  }
}
__device__ __forceinline__ sc_bf16x8 sc_tileT_get(const bf16_t* tile, int c, int h, int m) {
  return *reinterpret_cast<const sc_bf16x8*>(tile + c * SC_TSB + 16 * m + 8 * h);
}
__device__ __forceinline__ void sc_wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

template <typename T>
__global__ __launch_bounds__(256) void sconv_wgrad_kernel(const T* __restrict__ x, const int32_t* __restrict__ nbr,
                                                           const T* __restrict__ dout, float* __restrict__ dW,
                                                           int64_t n_dst, int Cin, int Cout, int cout_tiles) {
  constexpr bool SPLIT = sizeof(T) == 4;
  constexpr int NB = SPLIT ? 2 : 1;
  constexpr int TILE_ELTS = 32 * SC_TSB;
  extern __shared__ __attribute__((aligned(16))) bf16_t s_t[];  // [4 waves][X: 2 blocks | dO: 2 blocks][NB][tile]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int jl = lane & 31, h = lane >> 5;
  const int k = blockIdx.y;
  const int a0 = (blockIdx.z / cout_tiles) * SC_TILE, b0 = (blockIdx.z % cout_tiles) * SC_TILE;
  bf16_t* my = s_t + (size_t)wave * 4 * NB * TILE_ELTS;
  auto xt = [&](int blk, int part) { return my + (size_t)((blk)*NB + part) * TILE_ELTS; };
  auto ot = [&](int blk, int part) { return my + (size_t)((2 + blk) * NB + part) * TILE_ELTS; };
  // channel blocks of 32 that exist (Cin, Cout multiples of 16: a block is full, half, or absent)
  const int xa[2] = {min(32, Cin - a0), min(32, Cin - a0 - 32)};
  const int ob[2] = {min(32, Cout - b0), min(32, Cout - b0 - 32)};

  sc_f32x16 acc[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][t][r] = 0.f;

  const int64_t tiles = (n_dst + 31) / 32;
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  for (int64_t tt = (int64_t)blockIdx.x * 4 + wave; tt < tiles; tt += n_waves) {
    const int64_t j = tt * 32 + jl;
    const int idx = j < n_dst ? nbr[(int64_t)k * n_dst + j] : -1;
    if (__ballot(idx >= 0) == 0ull) continue;
    const bool ok = idx >= 0;
    Row16<T> xr[2], dr[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (16 * h < xa[u]) xr[u].load(x + (int64_t)max(idx, 0) * Cin + a0 + 32 * u + 16 * h);
      if (16 * h < ob[u]) dr[u].load(dout + (ok ? j : 0) * Cout + b0 + 32 * u + 16 * h);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float f[16];
      xr[u].get(ok && 16 * h < xa[u], f);
      sc_tileT_put<SPLIT>(xt(u, 0), xt(u, NB - 1), jl, h, f);
      dr[u].get(ok && 16 * h < ob[u], f);
      sc_tileT_put<SPLIT>(ot(u, 0), ot(u, NB - 1), jl, h, f);
    }
    sc_wave_sync();
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      sc_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        ah[u] = sc_tileT_get(xt(u, 0), jl, h, m);
        bh[u] = sc_tileT_get(ot(u, 0), jl, h, m);
        if (SPLIT) {
          al[u] = sc_tileT_get(xt(u, 1), jl, h, m);
          bl[u] = sc_tileT_get(ot(u, 1), jl, h, m);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[u][t] = SC_MFMA(ah[u], bh[t], acc[u][t]);
          if (SPLIT) {
            acc[u][t] = SC_MFMA(al[u], bh[t], acc[u][t]);
            acc[u][t] = SC_MFMA(ah[u], bl[t], acc[u][t]);
          }
        }
    }
    sc_wave_sync();  // the tiles are rewritten by this wavefront's next iteration
  }
  // acc[u][t][r]: row a = a0 + 32 u + acc_chan(r, h), column b = b0 + 32 t + jl
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int b = b0 + 32 * t + jl;
      if (b >= Cout) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int a = a0 + 32 * u + sc_acc_chan(r, h);
        if (a < Cin && acc[u][t][r] != 0.f) atomicAdd(&dW[((int64_t)k * Cin + a) * Cout + b], acc[u][t][r]);
      }
    }
}

static inline int pad_tile(int c) { return (c + SC_TILE - 1) / SC_TILE * SC_TILE; }
// output channels of the re-packed weights: one 64-wide tile, or multiples of the 128-wide tile of wide layers
static inline int pad_out(int c) { return c <= SC_TILE ? SC_TILE : (c + 127) / 128 * 128; }

template <typename T>
static int sconv_apply_impl(const void* x, const int32_t* nbr, const float* W, const float* bias, void* out,
                            int64_t n_dst, int K, int Cin, int Cout, int mode, void* ws, hipStream_t s) {
  constexpr bool SPLIT = sizeof(T) == 4;
  const int Cin_p = pad_tile(Cin), Cout_p = pad_out(Cout);
  const int64_t elems = (int64_t)K * Cin_p * Cout_p;
  bf16_t* w_hi = (bf16_t*)ws;
  bf16_t* w_lo = SPLIT ? w_hi + elems : nullptr;
  int64_t pb = (elems + 255) / 256;
  if (pb > 4096) pb = 4096;
  hipLaunchKernelGGL(sconv_prep_weights_kernel, dim3((int)pb), dim3(256), 0, s, W, K, Cin, Cout, Cin_p, Cout_p,
                     mode, w_hi, w_lo);
  const int64_t gx = (n_dst + 255) / 256;
  if (Cout_p >= 128) {
    const size_t lds = (size_t)2 * (SPLIT ? 2 : 1) * 128 * SC_WS * sizeof(bf16_t);
    hipLaunchKernelGGL((sconv_apply_kernel<T, 4>), dim3((unsigned)gx, Cout_p / 128), dim3(256), lds, s,
                       (const T*)x, nbr, w_hi, w_lo, bias, (T*)out, n_dst, K, Cin, Cout, Cin_p, Cout_p);
  } else {
    const size_t lds = (size_t)2 * (SPLIT ? 2 : 1) * SC_TILE * SC_WS * sizeof(bf16_t);
    hipLaunchKernelGGL((sconv_apply_kernel<T, 2>), dim3((unsigned)gx, Cout_p / SC_TILE), dim3(256), lds, s,
                       (const T*)x, nbr, w_hi, w_lo, bias, (T*)out, n_dst, K, Cin, Cout, Cin_p, Cout_p);
  }
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // namespace dva

using namespace dva;

extern "C" {

int64_t dva_sparse_conv_workspace_bytes(int32_t K, int32_t Cin, int32_t Cout, int32_t dtype) {
  if (K <= 0 || Cin <= 0 || Cout <= 0 || (dtype != DVA_F32 && dtype != DVA_BF16)) return DVA_ERR_INVALID;
  return (int64_t)K * pad_tile(Cin) * pad_out(Cout) * 2 * (dtype == DVA_F32 ? 2 : 1);
}

int dva_sparse_conv_apply(const void* x, const int32_t* nbr, const float* W, const float* bias, void* out,
                          int64_t n_src, int64_t n_dst, int32_t K, int32_t Cin, int32_t Cout, int32_t mode,
                          int32_t dtype, void* workspace, int64_t workspace_bytes, void* stream) {
  if (n_src < 0 || n_dst < 0 || K <= 0 || Cin <= 0 || Cout <= 0 || (mode != 0 && mode != 1))
    return DVA_ERR_INVALID;
  if (dtype != DVA_F32 && dtype != DVA_BF16) return DVA_ERR_INVALID;
  if (Cin % 16 != 0 || Cout % 16 != 0) return DVA_ERR_UNSUPPORTED;  // the host pads the channel counts
  if (n_src > 0x7fffffffLL || n_dst > 0x7fffffffLL / K) return DVA_ERR_UNSUPPORTED;
  if (n_dst == 0) return DVA_OK;
  if (!nbr || !W || !out || !workspace || (n_src > 0 && !x)) return DVA_ERR_INVALID;
  if (workspace_bytes < dva_sparse_conv_workspace_bytes(K, Cin, Cout, dtype)) return DVA_ERR_INVALID;
  if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)workspace) & 15) return DVA_ERR_INVALID;
  if (n_src == 0) return DVA_ERR_INVALID;  // missing neighbours read the (masked) row 0: it must exist
  hipStream_t s = (hipStream_t)stream;
  return dtype == DVA_F32
             ? sconv_apply_impl<float>(x, nbr, W, bias, out, n_dst, K, Cin, Cout, mode, workspace, s)
             : sconv_apply_impl<bf16_t>(x, nbr, W, bias, out, n_dst, K, Cin, Cout, mode, workspace, s);
}

int dva_sparse_conv_wgrad(const void* x, const int32_t* nbr, const void* grad_out, float* grad_W, int64_t n_src,
                          int64_t n_dst, int32_t K, int32_t Cin, int32_t Cout, int32_t dtype, void* stream) {
  if (n_src < 0 || n_dst < 0 || K <= 0 || Cin <= 0 || Cout <= 0) return DVA_ERR_INVALID;
  if (dtype != DVA_F32 && dtype != DVA_BF16) return DVA_ERR_INVALID;
  if (Cin % 16 != 0 || Cout % 16 != 0) return DVA_ERR_UNSUPPORTED;
  if (n_src > 0x7fffffffLL || n_dst > 0x7fffffffLL / K) return DVA_ERR_UNSUPPORTED;
  if (!grad_W) return DVA_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(grad_W, 0, (size_t)K * Cin * Cout * sizeof(float), s) != hipSuccess) return DVA_ERR_LAUNCH;
  if (n_dst == 0 || n_src == 0) return DVA_OK;
  if (!x || !nbr || !grad_out) return DVA_ERR_INVALID;
  if (((uintptr_t)x | (uintptr_t)grad_out) & 15) return DVA_ERR_INVALID;
  const int cin_tiles = pad_tile(Cin) / SC_TILE, cout_tiles = pad_tile(Cout) / SC_TILE;
  const int64_t tiles = (n_dst + 31) / 32;
  int64_t gx = (tiles + 4 * 16 - 1) / (4 * 16);  // >= 16 tiles per wavefront before it pays its atomics
  if (gx > 64) gx = 64;
  if (gx < 1) gx = 1;
  const dim3 grid((unsigned)gx, (unsigned)K, (unsigned)(cin_tiles * cout_tiles));
  const size_t lds = (size_t)4 * 4 * (dtype == DVA_F32 ? 2 : 1) * 32 * SC_TSB * sizeof(bf16_t);
  if (dtype == DVA_F32)
    hipLaunchKernelGGL((sconv_wgrad_kernel<float>), grid, dim3(256), lds, s, (const float*)x, nbr,
                       (const float*)grad_out, grad_W, n_dst, Cin, Cout, cout_tiles);
  else
    hipLaunchKernelGGL((sconv_wgrad_kernel<bf16_t>), grid, dim3(256), lds, s, (const bf16_t*)x, nbr,
                       (const bf16_t*)grad_out, grad_W, n_dst, Cin, Cout, cout_tiles);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // extern "C"
