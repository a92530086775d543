// Recompute chain with a PER-VIEW E_mod: the fused path of the bilinear gather (`interpolate=True`: the published
// KITTI-360 configuration, conf/models/segmentation/multimodal/sparseconv3d.yaml:7269-7340).
//
// Reference dataflow (core/multimodal/image.py:105-170 sparse_interpolation, modules/multimodal/modules.py:400,
// modules/multimodal/pooling.py:263-315): x_interp[v] = sum of 4 bilinear taps of the feature map, then
// E_mod = [Linear_a, BatchNorm_a, LeakyReLU, Linear_b, BatchNorm_b, LeakyReLU] PER VIEW, then the view attention.  E_mod
// cannot be hoisted to the map rows as with the nearest gather (BatchNorm + LeakyReLU do not commute with the
// interpolation) -- but its first Linear can: interp(x) W_a^T = interp(x W_a^T).  So Y = x W_a^T is one GEMM on the
// R map rows (host side), and per view remains: 4 taps of Y (C_o channels, bf16) -> z_a -> BatchNorm_a -> LeakyReLU ->
// Linear_b (C_o x C_o on the matrix cores, in registers) -> BatchNorm_b -> LeakyReLU -> value of the view, consumed
// by the softmax-weighted sum in the same kernel.  Eval mode is that one kernel.  Train-mode BatchNorm adds one
// statistics pass per layer, the backward one pass per BatchNorm barrier, exactly as for the DeepSetFeat chain
// (chain_fwd.hip / chain_bwd.hip, whose kernels evaluate the scores here as there); the first of them keeps the
// interpolated row z_a as bf16 [V][C_o] (the rounding of Linear_a's output under autocast) and the five later passes
// of the step read it -- 2 C_o contiguous bytes per view instead of 4 gathered taps (8 C_o + 32 bytes): the tap gathers
// were what bounded every pass (C_o = 64: 33.8 -> 27.4 ms/step; 128 -> 32: 22.2 -> 19.2):
//   dva_emod_prep        weight operands of Linear_b (bf16, MFMA k-slot order)
//   dva_emod_stats       layer 1: taps of Y -> z_a (stored), sum z_a | sum z_a^2;  layer 2: sum z_b | sum z_b^2
//   dva_emod_attn_fwd    x_map + z_a (eval: taps of Y) -> pooled features (DeepSetFeat scores, softmax, E_mod, weighted sum, gate)
//   dva_emod_attn_bwd    attention + gate backward from the stored scores: score gradients, view records, S of BatchNorm_b
//   dva_emod_bwd         stage 2: dW_b, dy_a = leaky'(y_a) W_b^T dz_b handed over as bf16 [V, C_o], S of BatchNorm_a
//                        stage 1: dz_a in place -> the weighted scatter over the row plan (dva_gather_rows_sum) gives dY
// Data layout: Y bf16 [R][C_o] in POSITION order: position 32 b + 16 h + r holds channel 32 b + chan(r, h), so that the
// 16 channels lane (view, h) owns of a 32-channel block are 32 contiguous bytes and land in the registers in the
// accumulator / B-operand order of chain_common.h.  The handed-over gradient [V][C_o] uses the same order.
// Two orientations of Linear_b: "standard" D[out channel][view] (lane = view: chains into further products) and
// "flipped" D[view][out channel] (the same operands in swapped roles: lane = channel, registers = views), in which
// BatchNorm constants are per-lane scalars and a reduction over the views of a point is in-lane arithmetic.
#include <type_traits>

#include "chain_common.h"

namespace dva {
namespace emod {
using namespace dva::chain;

// view of accumulator register r in the lane half h of a flipped product
__device__ __forceinline__ int view_of(int r, int h) { return chan(r, h); }

// ---- operands of Linear_b --------------------------------------------------------------------------------------
// FWD(mb, b, m): lane (rho, hh), slot s = W_b[32 mb + rho][32 b + chan(8 m + s, hh)]
// BWD(b, mb, m): lane (rho, hh), slot s = W_b[32 mb + chan(8 m + s, hh)][32 b + rho]
template <int NB> __host__ __device__ constexpr int op_fwd(int mb, int b, int m) { return (mb * NB + b) * 2 + m; }
template <int NB> __host__ __device__ constexpr int op_bwd(int b, int mb, int m) { return NB * NB * 2 + (b * NB + mb) * 2 + m; }

__global__ __launch_bounds__(64) void emod_prep_kernel(const float* __restrict__ Wb, int CO, uint4* __restrict__ ops) {
  const int NB = CO / 32;
  const int blk = blockIdx.x, lane = threadIdx.x, rho = lane & 31, hh = lane >> 5;
  const bool bwd = blk >= NB * NB * 2;
  const int i = bwd ? blk - NB * NB * 2 : blk;
  const int m = i & 1, x1 = (i >> 1) % NB, x0 = (i >> 1) / NB;     // fwd: (mb, b) = (x0, x1); bwd: (b, mb) = (x0, x1)
  float w[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int c = chan(8 * m + s, hh);
    w[s] = bwd ? Wb[(32 * x1 + c) * CO + 32 * x0 + rho] : Wb[(32 * x0 + rho) * CO + 32 * x1 + c];
  }
  ops[blk * 64 + lane] = __builtin_bit_cast(uint4, pack8(w));
}

// ---- BatchNorm tables ---------------------------------------------------------------------------------------------
// bn fp32 [4][CO] = mean | invstd | gamma | beta (dva_bn_finalize), natural channel order; sm fp32 [2][CO] = S1/M | S2/M.
// Table of the 32-channel block c0 in the accumulator-permuted order of chain_common.h (index 16 h + r <-> channel
// c0 + chan(r, h)), rows as there (T_G .. T_B6).
__device__ __forceinline__ void stage_tab_c(float* tab, const float* __restrict__ bn, int CO, int c0,
                                            const float* __restrict__ sm) {
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    const int c = c0 + chan(i & 15, i >> 4);
    const float mean = bn[c], inv = bn[CO + c], gam = bn[2 * CO + c], bet = bn[3 * CO + c];
    const float g = gam * inv;
    tab[T_G * D + i] = g;
    tab[T_B * D + i] = bet - mean * g;
    tab[T_I * D + i] = inv;
    tab[T_M * D + i] = -mean * inv;
    const float s1 = sm ? sm[c] : 0.f, s2 = sm ? sm[CO + c] : 0.f;
    tab[T_K1 * D + i] = g * (s1 - mean * inv * s2);
    tab[T_K2 * D + i] = g * inv * s2;
    tab[T_G6 * D + i] = 0.6f * g;
    tab[T_B6 * D + i] = 0.6f * (bet - mean * g);
  }
}

// ---- taps ---------------------------------------------------------------------------------------------------------
struct TapRec {
  int4 rows;
  float4 w;
};
// the 16 channels of block b this lane owns, for the 4 taps of its view: 4 x 2 x 16 bytes
template <int CO>
__device__ __forceinline__ void load_taps(__amdgpu_buffer_rsrc_t Y, const TapRec& t, bool ok, int b, int h,
                                          u32x4 (&x)[4][2]) {
  const int rr[4] = {t.rows.x, t.rows.y, t.rows.z, t.rows.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t off = ok ? (uint32_t)rr[k] * (uint32_t)(CO * 2) + (uint32_t)(32 * b + 16 * h) * 2u : OOB;
    x[k][0] = ld128(Y, off);
    x[k][1] = ld128(Y, ok ? off + 16u : OOB);
  }
}
__device__ __forceinline__ void interp16(const u32x4 (&x)[4][2], const float4& w, f32x16& z) {
  const float ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t v[4] = {x[k][q].x, x[k][q].y, x[k][q].z, x[k][q].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        z[8 * q + 2 * i] = __builtin_fmaf(ww[k], __uint_as_float(v[i] << 16), z[8 * q + 2 * i]);
        z[8 * q + 2 * i + 1] = __builtin_fmaf(ww[k], __uint_as_float(v[i] & 0xffff0000u), z[8 * q + 2 * i + 1]);
      }
    }
  }
}
// z_a of every block of the view (taps of one block in flight at a time: 32 registers)
template <int CO>
__device__ __forceinline__ void eval_za(__amdgpu_buffer_rsrc_t Y, const TapRec& t, bool ok, int h,
                                        f32x16 (&za)[CO / 32]) {
#pragma unroll
  for (int b = 0; b < CO / 32; ++b) {
    u32x4 x[4][2];
    load_taps<CO>(Y, t, ok, b, h, x);
    interp16(x, t.w, za[b]);
  }
}
// ---- stored z_a (train mode) -----------------------------------------------------------------------------------
// The first statistics pass rounds z_a to bf16 (what the output of Linear_a is under autocast in the reference) and
// keeps it as [V][CO] in position order; every later pass of the step reads 2 CO bytes per view, contiguous, instead
// of 4 taps x 2 CO bytes gathered (+ the 32-byte tap record): the tap gathers were what bounded those passes
// (statistics pass of layer 1: 17 GB in 2.6 ms at C_o = 64).  The tensor reaches 4 GiB at the headline size: one
// descriptor per tile, as for the handed-over gradient.
template <int NB>
struct ZaRows {
  u32x4 q[NB][2];
};
template <int CO>
__device__ __forceinline__ ZaRows<CO / 32> load_za(const bf16_t* __restrict__ zst, const TileInfo& ti, int j, int h) {
  const bool ok = j < ti.nv;
  const __amdgpu_buffer_rsrc_t Z = make_rsrc(zst + (int64_t)ti.v0 * CO, (uint64_t)ti.nv * CO * 2);
  ZaRows<CO / 32> z;
#pragma unroll
  for (int b = 0; b < CO / 32; ++b) {
    const uint32_t off = ok ? (uint32_t)j * (uint32_t)(CO * 2) + (uint32_t)(32 * b + 16 * h) * 2u : OOB;
    z.q[b][0] = ld128(Z, off);
    z.q[b][1] = ld128(Z, ok ? off + 16u : OOB);
  }
  return z;
}
template <int NB>
__device__ __forceinline__ void unpack_za(const ZaRows<NB>& z, f32x16 (&za)[NB]) {
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t v[4] = {z.q[b][q].x, z.q[b][q].y, z.q[b][q].z, z.q[b][q].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        za[b][8 * q + 2 * i] = __uint_as_float(v[i] << 16);
        za[b][8 * q + 2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
      }
    }
  }
}
// za <- bf16(za), stored for the later passes (lanes without a view store nothing)
template <int CO>
__device__ __forceinline__ void round_store_za(bf16_t* __restrict__ zst, const TileInfo& ti, int j, int h,
                                               f32x16 (&za)[CO / 32]) {
  const bool ok = j < ti.nv;
  const __amdgpu_buffer_rsrc_t Z = make_rsrc(zst + (int64_t)ti.v0 * CO, (uint64_t)ti.nv * CO * 2);
  ZaRows<CO / 32> z;
#pragma unroll
  for (int b = 0; b < CO / 32; ++b) {
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = za[b][r];
    z.q[b][0] = __builtin_bit_cast(u32x4, pack8(&t[0]));
    z.q[b][1] = __builtin_bit_cast(u32x4, pack8(&t[8]));
    const uint32_t off = ok ? (uint32_t)j * (uint32_t)(CO * 2) + (uint32_t)(32 * b + 16 * h) * 2u : OOB;
    st128(Z, off, z.q[b][0]);          // (non-temporal stores measured slower: 4.6 vs 4.1 ms at C_o = 64)
    st128(Z, ok ? off + 16u : OOB, z.q[b][1]);
  }
  unpack_za<CO / 32>(z, za);
}
// Linear_b, flipped: zb[mb][r] = z_b[view_of(r, h)][channel 32 mb + (lane & 31)]
template <int NB>
__device__ __forceinline__ void linear_b_flipped(const uint4* s_eops, int lane, const bf16x8 (&a)[NB][2],
                                                 f32x16 (&zb)[NB]) {
  asm volatile("" ::: "memory");
#pragma unroll
  for (int mb = 0; mb < NB; ++mb) {
    f32x16 acc = {0};
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int m = 0; m < 2; ++m) acc = CH_MFMA(a[b][m], lds_op(s_eops, op_fwd<NB>(mb, b, m), lane), acc);
    }
    zb[mb] = acc;
  }
}
// Linear_b, standard: zb[mb][r] = z_b[channel 32 mb + chan(r, h)][view lane & 31]
template <int NB>
__device__ __forceinline__ void linear_b_std(const uint4* s_eops, int lane, const bf16x8 (&a)[NB][2], f32x16 (&zb)[NB]) {
  asm volatile("" ::: "memory");
#pragma unroll
  for (int mb = 0; mb < NB; ++mb) {
    f32x16 acc = {0};
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int m = 0; m < 2; ++m) acc = CH_MFMA(lds_op(s_eops, op_fwd<NB>(mb, b, m), lane), a[b][m], acc);
    }
    zb[mb] = acc;
  }
}
// ---- wide rows (C_o >= 128: NB >= 4 blocks) ---------------------------------------------------------------------------
// The register-resident forms above hold every block of a view at once (z_a, z_b: 2 x 16 NB fp32 registers); from four
// blocks on the passes go OUTPUT BLOCK BY OUTPUT BLOCK: the packed activation of all input blocks (8 NB registers) stays,
// one 32 x 32 product is live at a time and its epilogue runs before the next one starts.
template <int NB>
__device__ __forceinline__ f32x16 linear_b_flipped_blk(const uint4* s_eops, int lane, const bf16x8 (&a)[NB][2], int mb) {
  asm volatile("" ::: "memory");
  f32x16 acc = {0};
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int m = 0; m < 2; ++m) acc = CH_MFMA(a[b][m], lds_op(s_eops, op_fwd<NB>(mb, b, m), lane), acc);
  }
  return acc;
}
template <int NB>
__device__ __forceinline__ f32x16 linear_b_std_blk(const uint4* s_eops, int lane, const bf16x8 (&a)[NB][2], int mb) {
  asm volatile("" ::: "memory");
  f32x16 acc = {0};
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int m = 0; m < 2; ++m) acc = CH_MFMA(lds_op(s_eops, op_fwd<NB>(mb, b, m), lane), a[b][m], acc);
  }
  return acc;
}
template <int NB>
__device__ __forceinline__ void unpack_za_blk(const ZaRows<NB>& z, int b, f32x16& za) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const uint32_t v[4] = {z.q[b][q].x, z.q[b][q].y, z.q[b][q].z, z.q[b][q].w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      za[8 * q + 2 * i] = __uint_as_float(v[i] << 16);
      za[8 * q + 2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
  }
}
// BatchNorm_a + LeakyReLU + bf16 packing straight from the packed rows, one block at a time
template <int NB>
__device__ __forceinline__ void act_a_rows(const ZaRows<NB>& z, const float (*taba)[TAB_FLOATS], int h, uint32_t keep,
                                           bf16x8 (&a)[NB][2]) {
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    f32x16 za;
    unpack_za_blk<NB>(z, b, za);
    act_pack(za, taba[b], h, keep, a[b]);
  }
}
// Sums over the 32 views of a tile in the lane = view layout, without 32 accumulator registers per block: the wavefront
// writes the packed block as a natural [view][column] tile (tileN_put_packed: column 16 h + r = accumulator register r
// of half h, i.e. channel cperm(column)) and reads it back through the transpose read: lane (n, hh) receives the views
// 8 hh .. 8 hh + 7 and 16 + 8 hh .. of column n -- 16 in-lane additions; the two half-waves hold the two halves of the
// column sum (added at the flush).  x, y: values as stored (bf16).
// (round 5: the packed bf16 pairs go straight into v_dot2c_f32_bf16 -- four dot products per eight values and statistic,
//  fp32 accumulation, no unpacking: a third of the vector instructions of the unpack + add + fma form)
typedef __bf16 bf16x2e_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2bf(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2e_t, a), __builtin_bit_cast(bf16x2e_t, b), c, false);
}
__device__ __forceinline__ void col_sums(const bf16_t* tx, int lane, float& s, float& ss) {        // sum x | sum x^2
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const u32x4 x = __builtin_bit_cast(u32x4, tileN_get(tx, lane, m));
    const uint32_t xx[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s = dot2bf(xx[i], 0x3f803f80u, s);
      ss = dot2bf(xx[i], xx[i], ss);
    }
  }
}
__device__ __forceinline__ void col_sums2(const bf16_t* tx, const bf16_t* ty, int lane, float& sx, float& sxy) {   // sum x | sum x y
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const u32x4 x = __builtin_bit_cast(u32x4, tileN_get(tx, lane, m)), y = __builtin_bit_cast(u32x4, tileN_get(ty, lane, m));
    const uint32_t xx[4] = {x.x, x.y, x.z, x.w}, yy[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sx = dot2bf(xx[i], 0x3f803f80u, sx);
      sxy = dot2bf(xx[i], yy[i], sxy);
    }
  }
}

// per-lane BatchNorm constants of the flipped layer: channel 32 mb + (lane & 31)
template <int NB>
struct LaneBN {
  float g6[NB], b6[NB];      // 0.6 G | 0.6 (beta - mean G)
};
template <int NB>
__device__ __forceinline__ LaneBN<NB> lane_bn(const float* __restrict__ bn, int CO, int lane) {
  LaneBN<NB> k;
#pragma unroll
  for (int mb = 0; mb < NB; ++mb) {
    const int c = 32 * mb + (lane & 31);
    const float g = bn[2 * CO + c] * bn[CO + c];
    k.g6[mb] = 0.6f * g;
    k.b6[mb] = 0.6f * (bn[3 * CO + c] - bn[c] * g);
  }
  return k;
}
__device__ __forceinline__ float leaky06(float t) { return __builtin_fmaf(__builtin_fabsf(t), 0.6666667f, t); }

// all-reduce over groups of LANES lanes (8, 16 or 32, aligned) inside a half-wave
template <int LANES>
__device__ __forceinline__ float group_sum(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1, 0, 3, 2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2, 3, 0, 1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  if (LANES >= 16) v += dpp_mov<0x140>(v);   // row_mirror
  if (LANES >= 32) {
    const uint32_t x = __float_as_uint(v);
    const u32x2 r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    v = __uint_as_float(r.x) + __uint_as_float(r.y);
  }
  return v;
}
__device__ __forceinline__ float other_half(float v) {     // value of lane ^ 32
  uint32_t a = __float_as_uint(v), b = a;
  swap_halves(a, b);
  return __uint_as_float((threadIdx.x & 32) ? a : b);
}

// deterministic block reduction of per-lane partials (one LDS slot per wavefront, fixed order, fp64) -> fp64 atomics
// (v0, v1 of the lanes h = 0: channel base + (lane & 31)); s_red: 4 x 2 x 32 floats
// nat: lane n holds image column n of a natural tile = channel base + cperm(n)
__device__ __forceinline__ void flush_lane_stats(float v0, float v1, double* __restrict__ out0, double* __restrict__ out1,
                                                 int base, float* s_red, bool nat = false) {
  const int wv = threadIdx.x >> 6, n = threadIdx.x & 31;
  __syncthreads();
  if ((threadIdx.x & 32) == 0) {
    s_red[(wv * 2 + 0) * 32 + n] = v0;
    s_red[(wv * 2 + 1) * 32 + n] = v1;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int which = threadIdx.x >> 5;
    double acc = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) acc += (double)s_red[(w * 2 + which) * 32 + n];
    const int c = base + (nat ? cperm(n) : n);
    atomicAdd(which ? &out1[c] : &out0[c], acc);
  }
}

// per-lane fp32 partial sums of a standard-layout block (16 accumulator channels) -> stats[n * ld + c0 + chan(r, h)]
template <int NV>
__device__ __forceinline__ void flush_stats_c(float (&st)[NV][16], double* __restrict__ out, int ld, int c0, float* s_red) {
  const int lane = threadIdx.x & 63, h = lane >> 5, wv = threadIdx.x >> 6;
  __syncthreads();
#pragma unroll
  for (int n = 0; n < NV; ++n) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = st[n][r];
#pragma unroll
      for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off);
      if ((lane & 31) == 0) s_red[wv * (NV * D) + n * D + chan(r, h)] = v;
    }
  }
  __syncthreads();
  const int n_waves = blockDim.x >> 6;
  for (int i = threadIdx.x; i < NV * D; i += blockDim.x) {
    double acc = 0.0;
    for (int w = 0; w < n_waves; ++w) acc += (double)s_red[w * (NV * D) + i];
    atomicAdd(&out[(i / D) * ld + c0 + (i % D)], acc);
  }
}

// BatchNorm_a + LeakyReLU + bf16 packing of every block (operands of Linear_b)
template <int NB>
__device__ __forceinline__ void act_a(const f32x16 (&za)[NB], const float (*taba)[TAB_FLOATS], int h, uint32_t keep,
                                      bf16x8 (&a)[NB][2]) {
#pragma unroll
  for (int b = 0; b < NB; ++b) act_pack(za[b], taba[b], h, keep, a[b]);
}

// ------------------------------------------------------------------------------------------------
// statistics passes (train mode).  L = 1: z_a (interpolated Y);  L = 2: z_b = W_b leaky(BatchNorm_a(z_a))
// stats fp64 [2][CO] = sum | sum of squares, natural channel order
// ------------------------------------------------------------------------------------------------
template <int CO, int L>
__global__ __launch_bounds__(CO >= 256 && L == 2 ? 512 : 256, CO >= 256 ? 1 : 2) void emod_stats_kernel(
    const bf16_t* __restrict__ Yp, const int4* __restrict__ rows4, const float4* __restrict__ w4,
    const int2* __restrict__ tiles, const int32_t* __restrict__ n_tiles_dev, const uint4* __restrict__ eops,
    const float* __restrict__ bna, double* __restrict__ stats, bf16_t* __restrict__ zst, int64_t V, int64_t R) {
  constexpr int NB = CO / 32;
  constexpr bool WIDE = NB >= 4;          // block by block, sums over the views through a natural LDS tile
  __shared__ __attribute__((aligned(16))) float s_taba[NB][TAB_FLOATS];
  __shared__ __attribute__((aligned(16))) uint4 s_eops[L == 2 ? NB * NB * 2 * 64 : 1];
  __shared__ __attribute__((aligned(16))) bf16_t s_tile[WIDE && L == 1 ? 4 : 1][WIDE && L == 1 ? 32 * TSB : 8];
  __shared__ float s_red[STATS_RED_FLOATS > 8 * 64 ? STATS_RED_FLOATS : 8 * 64];      // (8 wavefronts at C_o = 256, L = 2)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  if (L == 2) {
    for (int i = threadIdx.x; i < NB * NB * 2 * 64; i += blockDim.x) s_eops[i] = eops[i];
#pragma unroll
    for (int b = 0; b < NB; ++b) stage_tab_c(s_taba[b], bna, CO, 32 * b, nullptr);
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t Y = make_rsrc(Yp, (uint64_t)R * CO * 2), R4 = make_rsrc(rows4, (uint64_t)V * 16),
                               W4 = make_rsrc(w4, (uint64_t)V * 16);
  float st[L == 1 && !WIDE ? NB : 1][2][16];
  float s1[NB], s2[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    s1[b] = s2[b] = 0.f;
    if (L == 1 && !WIDE) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[b][0][r] = st[b][1][r] = 0.f;
    }
  }
  const int n_tiles = n_tiles_dev[0];
  int ta, tb;
  wave_tile_range(tiles, n_tiles, ta, tb);
  struct Pre {
    TileInfo ti;
    TapRec t;            // L = 1
    ZaRows<NB> z;        // L = 2: the stored z_a
  };
  run_tiles<Pre>(tiles, ta, tb, [&](const TileInfo& ti, int t) {
    Pre p;
    p.ti = ti;
    if constexpr (L == 1) {
      const bool ok = j < p.ti.nv;
      const uint32_t off = ok ? (uint32_t)(p.ti.v0 + j) * 16u : OOB;
      p.t.rows = __builtin_bit_cast(int4, ld128(R4, off));
      p.t.w = as_f4(ld128(W4, off));
    } else {
      p.z = load_za<CO>(zst, ti, j, h);
    }
    return p;
  }, [&](const Pre& p) {
    const bool ok = j < p.ti.nv;
    const uint32_t keep = ok ? 0xffffffffu : 0u;
    if constexpr (L == 1 && WIDE) {
      // one block at a time: taps -> z_a block -> bf16 (stored) -> natural tile -> column sums of the stored values
      const __amdgpu_buffer_rsrc_t Z = make_rsrc(zst + (int64_t)p.ti.v0 * CO, zst ? (uint64_t)p.ti.nv * CO * 2 : 0);
      bf16_t* tile = s_tile[wv];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        u32x4 x[4][2];
        load_taps<CO>(Y, p.t, ok, b, h, x);       // lanes without a view: weights and taps read 0 -> z_a = 0
        f32x16 z;
        interp16(x, p.t.w, z);
        float t16[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) t16[r] = z[r];
        bf16x8 pk[2] = {pack8(&t16[0]), pack8(&t16[8])};
        const uint32_t off = ok ? (uint32_t)j * (uint32_t)(CO * 2) + (uint32_t)(32 * b + 16 * h) * 2u : OOB;
        st128(Z, off, __builtin_bit_cast(u32x4, pk[0]));
        st128(Z, ok ? off + 16u : OOB, __builtin_bit_cast(u32x4, pk[1]));
        tileN_put_packed(tile, j, h, pk);
        wave_sync();
        col_sums(tile, lane, s1[b], s2[b]);
        wave_sync();
      }
    } else if constexpr (L == 1) {
      f32x16 za[NB];
      eval_za<CO>(Y, p.t, ok, h, za);         // lanes without a view: weights and taps read 0 -> z_a = 0
      if (zst) round_store_za<CO>(zst, p.ti, j, h, za);     // the statistics are those of the stored values
#pragma unroll
      for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          st[b][0][r] += za[b][r];
          st[b][1][r] = __builtin_fmaf(za[b][r], za[b][r], st[b][1][r]);
        }
      }
    } else if constexpr (WIDE) {
      bf16x8 a[NB][2];
      act_a_rows<NB>(p.z, s_taba, h, keep, a);
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
        const f32x16 zb = linear_b_flipped_blk<NB>(s_eops, lane, a, mb);   // views without a lane: a = 0 -> z_b = 0
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s1[mb] += zb[r];
          s2[mb] = __builtin_fmaf(zb[r], zb[r], s2[mb]);
        }
      }
    } else {
      f32x16 za[NB];
      unpack_za<NB>(p.z, za);
      bf16x8 a[NB][2];
      act_a<NB>(za, s_taba, h, keep, a);
      f32x16 zb[NB];
      linear_b_flipped<NB>(s_eops, lane, a, zb);    // views without a lane: a = 0 -> z_b = 0: no mask needed below
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s1[mb] += zb[mb][r];
          s2[mb] = __builtin_fmaf(zb[mb][r], zb[mb][r], s2[mb]);
        }
      }
    }
  });
  if constexpr (L == 1 && !WIDE) {
#pragma unroll
    for (int b = 0; b < NB; ++b) flush_stats_c<2>(st[b], stats, CO, 32 * b, s_red);
  } else {
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) {
      const float a0 = s1[mb] + other_half(s1[mb]), a1 = s2[mb] + other_half(s2[mb]);
      flush_lane_stats(a0, a1, stats, stats + CO, 32 * mb, s_red, L == 1);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// layer-1 statistics pass in ANCHOR order (round 4).  The four taps of a view are the 2 x 2 block of map rows under its
// anchor, and the backward already sorts the views by anchor (the anchor plan of the transposed interpolation).  Walking
// the views in that order -- a tile = 32 consecutive plan entries, the view of a lane = perm[position] -- makes
// neighbouring lanes and consecutive tiles read the SAME four rows of Y: the tap gathers (8 C_o bytes per view out of a
// map that only fits the last-level cache) become L1 / L2 hits, and what is left per view is the plan entry (4 B), its tap
// record (2 x 16 B at a random address) and the z_a row it writes (2 C_o B, a whole row at a random address).
// No point structure is needed here: the pass only stores rows and sums columns.  One form for every width: block by
// block, column sums of the stored values through the natural LDS tile.
// ------------------------------------------------------------------------------------------------
template <int CO>
__global__ __launch_bounds__(256, CO >= 128 ? 3 : 4) void emod_stats1_plan_kernel(
    const bf16_t* __restrict__ Yp, const int4* __restrict__ rows4, const float4* __restrict__ w4,
    const int32_t* __restrict__ perm, double* __restrict__ stats, bf16_t* __restrict__ zst, int64_t V, int64_t R) {
  constexpr int NB = CO / 32;
  __shared__ __attribute__((aligned(16))) bf16_t s_tile[4][32 * TSB];
  __shared__ float s_red[4 * 2 * 32];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  const __amdgpu_buffer_rsrc_t Y = make_rsrc(Yp, (uint64_t)R * CO * 2), R4 = make_rsrc(rows4, (uint64_t)V * 16),
                               W4 = make_rsrc(w4, (uint64_t)V * 16), PM = make_rsrc(perm, (uint64_t)V * 4);
  float s1[NB], s2[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) s1[b] = s2[b] = 0.f;
  const int64_t n_tiles = (V + 31) / 32;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t ta = n_tiles * wave / n_waves, tb = n_tiles * (wave + 1) / n_waves;
  bf16_t* tile = s_tile[wv];
  struct Pre {
    int view;
    bool ok;
    TapRec t;
  };
  auto fetch = [&](int64_t t) {
    Pre p;
    const int64_t pos = 32 * t + j;
    p.ok = t < tb && pos < V;
    p.view = (int)ld32(PM, p.ok ? (uint32_t)pos * 4u : OOB);
    return p;
  };
  auto fetch_taps = [&](Pre& p) {
    const uint32_t off = p.ok ? (uint32_t)p.view * 16u : OOB;
    p.t.rows = __builtin_bit_cast(int4, ld128(R4, off));
    p.t.w = as_f4(ld128(W4, off));
  };
  if (ta >= tb) {
    // (still takes part in the block-level flush below)
  }
  Pre cur = fetch(ta);
  fetch_taps(cur);
  for (int64_t t = ta; t < tb; ++t) {
    Pre nxt = fetch(t + 1);          // the next tile's plan entries and tap records fly during this tile
    fetch_taps(nxt);
    const bool ok = cur.ok;
    bf16_t* zrow = zst + (int64_t)cur.view * CO + 16 * h;
    // the taps of block b + 1 are requested before block b is interpolated, stored and summed (the LDS round trip of the
    // column sums is a scheduling barrier: without the second register set every block exposed a memory latency)
    u32x4 xx[2][4][2];
    load_taps<CO>(Y, cur.t, ok, 0, h, xx[0]);       // lanes without a view: weights and taps read 0 -> z_a = 0
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b + 1 < NB) load_taps<CO>(Y, cur.t, ok, b + 1, h, xx[(b + 1) & 1]);
      const u32x4 (&x)[4][2] = xx[b & 1];
      f32x16 z;
      interp16(x, cur.t.w, z);
      float t16[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) t16[r] = z[r];
      bf16x8 pk[2] = {pack8(&t16[0]), pack8(&t16[8])};
      if (ok && zst) {      // rows at random addresses beyond 4 GiB: flat 64-bit stores
        *reinterpret_cast<u32x4*>(zrow + 32 * b) = __builtin_bit_cast(u32x4, pk[0]);
        *reinterpret_cast<u32x4*>(zrow + 32 * b + 8) = __builtin_bit_cast(u32x4, pk[1]);
      }
      tileN_put_packed(tile, j, h, pk);
      wave_sync();
      col_sums(tile, lane, s1[b], s2[b]);
      wave_sync();
    }
    cur = nxt;
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const float a0 = s1[b] + other_half(s1[b]), a1 = s2[b] + other_half(s2[b]);
    flush_lane_stats(a0, a1, stats, stats + CO, 32 * b, s_red, true);
  }
}

// ------------------------------------------------------------------------------------------------
// the fused view kernel of the bilinear path
// ------------------------------------------------------------------------------------------------
// C_o = 128 / 256 (eval mode only: 356 / 512 registers, one wavefront per SIMD; the backward kernels do not exist at
// those widths)
template <int CO, int G, int ZM>      // ZM = 0: z_a from the taps of Y (eval mode), 1: the stored z_a (train mode)
__global__ __launch_bounds__((CO >= 256 && ZM == 1) ? 512 : 256, CO > 64 ? (CO == 128 ? 2 : 1) : (CO == 32 && ZM == 1 ? (G == 1 ? 3 : 4) : 2)) void emod_attn_fwd_kernel(
    const float* __restrict__ x_map, const int32_t* __restrict__ vp, const float* __restrict__ u,
    const int2* __restrict__ tiles, const int32_t* __restrict__ n_tiles_dev, const uint4* __restrict__ ops,
    const float* __restrict__ bn1, const float* __restrict__ bn2, const float* __restrict__ bn5,
    const float* __restrict__ bn6, const float* __restrict__ bs, const bf16_t* __restrict__ Yp,
    const int4* __restrict__ rows4, const float4* __restrict__ w4, const uint4* __restrict__ eops,
    const float* __restrict__ bna, const float* __restrict__ bnb, const int64_t* __restrict__ ptr,
    const float* __restrict__ gw, const float* __restrict__ gb, bf16_t* __restrict__ out,
    float* __restrict__ scores_out, const bf16_t* __restrict__ zst, int scaling, float eps, int64_t V, int64_t N,
    int64_t R) {
  constexpr int NB = CO / 32, NE = G == 1 ? 1 : 2, GS = CO / G;
  static_assert(GS % 8 == 0, "whole 8-lane groups");
  __shared__ __attribute__((aligned(16))) float s_tab[4][2 * D];
  __shared__ __attribute__((aligned(16))) uint4 s_ops[OP_W6T * 64];
  __shared__ __attribute__((aligned(16))) uint4 s_eops[NB * NB * 2 * 64];
  __shared__ __attribute__((aligned(16))) float s_taba[NB][TAB_FLOATS];
  constexpr int NW = (CO >= 256 && ZM == 1) ? 8 : 4;     // wavefronts per block (C_o = 256 from the stored z_a: two per SIMD)
  __shared__ __attribute__((aligned(16))) float s_ev[NW][4 * 32];     // exp(.) per [group][view]
  __shared__ __attribute__((aligned(16))) float s_sc[NW][4 * 32];     // gate / (sum + eps) per [group][view]
  __shared__ __attribute__((aligned(16))) int s_pid[NW][32];
  __shared__ __attribute__((aligned(16))) float s_alpha[NW][4], s_scg[NW][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < OP_W6T * 64; i += blockDim.x) s_ops[i] = ops[i];
  for (int i = threadIdx.x; i < NB * NB * 2 * 64; i += blockDim.x) s_eops[i] = eops[i];
  stage_tab_fwd(s_tab[0], bn1);
  stage_tab_fwd(s_tab[1], bn2);
  stage_tab_fwd(s_tab[2], bn5);
  stage_tab_fwd(s_tab[3], bn6);
#pragma unroll
  for (int b = 0; b < NB; ++b) stage_tab_c(s_taba[b], bna, CO, 32 * b, nullptr);
  __syncthreads();
  fold_ops(s_ops, OP_W1, ops, OP_W1, 1, bn1);
  fold_ops(s_ops, OP_W2, ops, OP_W2, 2, bn2);
  fold_ops(s_ops, OP_W6, ops, OP_W6, 2, bn6);
  __syncthreads();
  const __amdgpu_buffer_rsrc_t X = make_rsrc(x_map, (uint64_t)V * 32), P = make_rsrc(vp, (uint64_t)V * 4),
                               U = make_rsrc(u, (uint64_t)N * 128), Y = make_rsrc(Yp, (uint64_t)R * CO * 2),
                               R4 = make_rsrc(rows4, (uint64_t)V * 16), W4 = make_rsrc(w4, (uint64_t)V * 16),
                               O = make_rsrc(out, (uint64_t)N * CO * 2),
                               SC = make_rsrc(scores_out, scores_out ? (uint64_t)V * 16 : 0);
  const bool s_active = G == 4 || h == 0;
  int gl[NE];
  float bias[NE], gwl[NE], gbl[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    gl[e] = (G == 4 ? 2 * h : 0) + e;
    if (gl[e] >= G) gl[e] = G - 1;
    bias[e] = bs[gl[e]];
    gwl[e] = gw ? gw[gl[e]] : 0.f;
    gbl[e] = gw ? gb[gl[e]] : 0.f;
  }
  const LaneBN<NB> kb = lane_bn<NB>(bnb, CO, lane);
  int gch[NB];                 // group of this lane's channel in block mb
#pragma unroll
  for (int mb = 0; mb < NB; ++mb) gch[mb] = (32 * mb + j) / GS;
  float* ev_t = s_ev[wv];
  float* sc_t = s_sc[wv];
  int* pid_t = s_pid[wv];
  float run_m[NE], run_s[NE], run_acc[NB];
#pragma unroll
  for (int e = 0; e < NE; ++e) { run_m[e] = -INFINITY; run_s[e] = 0.f; }
#pragma unroll
  for (int mb = 0; mb < NB; ++mb) run_acc[mb] = 0.f;

  const int n_tiles = n_tiles_dev[0];
  int ta, tb;
  wave_tile_range(tiles, n_tiles, ta, tb);
  struct Pre {
    TileInfo ti;
    float4 x;
    int vpj;
    TapRec t;            // ZM = 0
    ZaRows<NB> z;        // ZM = 1
  };
  // C_o = 32, train mode: without the prefetch register set the kernel fits four wavefronts per SIMD (G = 1: three):
  // 1.39 -> 1.12 ms on the KITTI pair
  auto loop = [&](auto&& ld, auto&& bd) {
    if constexpr ((CO == 32 || CO >= 256) && ZM == 1) run_tiles_single<Pre>(tiles, ta, tb, ld, bd);
    else run_tiles<Pre>(tiles, ta, tb, ld, bd);
  };
  loop([&](const TileInfo& ti, int t) {
    Pre p;
    p.ti = ti;
    const bool ok = j < p.ti.nv;
    const uint32_t view = (uint32_t)(p.ti.v0 + j);
    p.x = as_f4(ld128(X, ok ? view * 32u + 16u * h : OOB));
    p.vpj = (int)ld32(P, ok ? view * 4u : OOB);
    if constexpr (ZM == 0) {
      p.t.rows = __builtin_bit_cast(int4, ld128(R4, ok ? view * 16u : OOB));
      p.t.w = as_f4(ld128(W4, ok ? view * 16u : OOB));
    } else {
      p.z = load_za<CO>(zst, ti, j, h);
    }
    return p;
  }, [&](const Pre& p) {
    const int nv = p.ti.nv, frag = p.ti.frag;
    const bool ok = j < nv;
    const uint32_t keepv = ok ? 0xffffffffu : 0u;
    f32x16 uacc;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const float4 v = as_f4(ld128(U, ok ? (uint32_t)p.vpj * 128u + (8u * qq + 4u * h) * 4u : OOB));
      uacc[4 * qq] = v.x; uacc[4 * qq + 1] = v.y; uacc[4 * qq + 2] = v.z; uacc[4 * qq + 3] = v.w;
    }
    // ---- DeepSetFeat chain -> scores (as chain_fwd.hip attn_fwd_kernel)
    float c[NE];
    {
      const f32x16 zero = {0};
      const uint32_t keep = 0xffffffffu;
      bf16x8 a[2], a2[2];
      asm volatile("" ::: "memory");
      f32x16 z = CH_MFMA(lds_op(s_ops, OP_W1, lane), pack_x(p.x), bias_acc(s_tab[0], 1, h));
      act_fold(z, keep, a);
      z = mm32_lds(s_ops, OP_W2, lane, a, bias_acc(s_tab[1], 1, h));
      act_fold(z, keep, a2);
      z = mm32_lds(s_ops, OP_W5, lane, a2, uacc);
      act_pack(z, s_tab[2], h, keep, a, 0, 1);
      z = mm32_lds(s_ops, OP_W6, lane, a, bias_acc(s_tab[3], 1, h));
      act_fold(z, keep, a2);
      z = mm32_lds(s_ops, OP_WS, lane, a2, zero);
      if constexpr (G == 4) {
        uint32_t A0 = __float_as_uint(z[0]), A2 = __float_as_uint(z[2]);
        uint32_t A1 = __float_as_uint(z[1]), A3 = __float_as_uint(z[3]);
        swap_halves(A0, A2);
        swap_halves(A1, A3);
        c[0] = __uint_as_float(A0) + bias[0];
        c[1] = __uint_as_float(A1) + bias[1];
      } else {
#pragma unroll
        for (int e = 0; e < NE; ++e) c[e] = z[e] + bias[e];
      }
    }
    if (s_active) {
      const uint32_t so = ok ? (uint32_t)(p.ti.v0 + j) * 16u + (G == 4 ? 8u * h : 0u) : OOB;
      if (NE == 2) {
        const u32x2 cv = {__float_as_uint(c[0]), __float_as_uint(c[NE - 1])};
        __builtin_amdgcn_raw_buffer_store_b64(cv, SC, (int)so, 0, 0);
      } else {
        st32(SC, so, __float_as_uint(c[0]));
      }
    }
    const int vp0 = __builtin_amdgcn_readfirstlane(p.vpj);
    const bool single = frag != 0 || __ballot(ok && p.vpj != vp0) == 0;
    // ---- softmax weights of the tile -> LDS tables
    SegInfo sg;
    if (single) {
      int n_pt = nv;
      if (frag != 0) n_pt = (int)(ptr[vp0 + 1] - ptr[vp0]);
      const float isn = (scaling ? __builtin_amdgcn_rsqf((float)n_pt) : 1.f) * 1.44269504f;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        float m = half_max(ok ? c[e] : -INFINITY);
        float alpha = 0.f;
        if (frag != 0) {
          const float m_new = vmaxf(run_m[e], m);
          alpha = __builtin_amdgcn_exp2f((run_m[e] - m_new) * isn);
          m = m_new;
        }
        const float ev = ok ? __builtin_amdgcn_exp2f((c[e] - m) * isn) : 0.f;
        float s = half_sum(ev);
        if (frag != 0) {
          s = run_s[e] * alpha + s;
          run_s[e] = frag == 3 ? 0.f : s;
          run_m[e] = frag == 3 ? -INFINITY : m;
        }
        const float gt = gw ? tanh_pos(vmaxf(__builtin_fmaf(gwl[e], m, gbl[e]), 0.f)) : 1.f;
        if (s_active) {
          ev_t[gl[e] * 32 + j] = ev;
          if (j == 0) {
            s_scg[wv][gl[e]] = gt * __builtin_amdgcn_rcpf(s + eps);
            s_alpha[wv][gl[e]] = alpha;
          }
        }
      }
    } else {
      sg = seg_setup(p.vpj, j, lane, nv);
      const float isn = (scaling ? __builtin_amdgcn_rsqf((float)(sg.se - sg.ss + 1)) : 1.f) * 1.44269504f;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const float m = seg_total(seg_scan_max(ok ? c[e] : -INFINITY, sg, lane), sg, h);
        const float ev = ok ? __builtin_amdgcn_exp2f((c[e] - m) * isn) : 0.f;
        const float s = seg_total(seg_scan_sum(ev, sg, lane), sg, h);
        const float gt = gw ? tanh_pos(vmaxf(__builtin_fmaf(gwl[e], m, gbl[e]), 0.f)) : 1.f;
        if (s_active) {
          ev_t[gl[e] * 32 + j] = ev;
          sc_t[gl[e] * 32 + j] = gt * __builtin_amdgcn_rcpf(s + eps);
        }
      }
      if (h == 0) pid_t[j] = p.vpj;
    }
    // ---- E_mod of the 32 views: taps of Y -> z_a -> BatchNorm_a, LeakyReLU -> Linear_b (flipped) -> value
    // softmax-weighted sum over the views of a point: in-lane (the lane owns channel 32 mb + j, its registers
    // 16 of the 32 views; the other 16 sit in lane ^ 32)
    auto weighted = [&](const f32x16& zbm, int mb, int lo, int hi, bool masked) {   // sum over the views [lo, hi] of the tile
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 w = *reinterpret_cast<const float4*>(ev_t + gch[mb] * 32 + 8 * q + 4 * h);
        const float ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * q + i, v = 8 * q + 4 * h + i;       // = view_of(r, h)
          const float val = leaky06(__builtin_fmaf(zbm[r], kb.g6[mb], kb.b6[mb]));
          const float wv_ = (!masked || (v >= lo && v <= hi)) ? ww[i] : 0.f;
          acc = __builtin_fmaf(wv_, val, acc);
        }
      }
      return acc + other_half(acc);
    };
    // the epilogue of output block mb
    auto pool_block = [&](const f32x16& zbm, int mb) {
      if (single) {
        const bool done = frag == 0 || frag == 3;
        float acc = weighted(zbm, mb, 0, 31, false);        // views without a lane carry weight 0
        const float sc = s_scg[wv][gch[mb]], al = s_alpha[wv][gch[mb]];
        if (frag != 0) {
          acc = __builtin_fmaf(run_acc[mb], al, acc);
          run_acc[mb] = frag == 3 ? 0.f : acc;
        }
        if (done)
          __builtin_amdgcn_raw_buffer_store_b16((short)f2bf(acc * sc), O,
                                                h == 0 ? (int)((uint32_t)vp0 * (uint32_t)(CO * 2) + (uint32_t)(32 * mb + j) * 2u) : (int)OOB,
                                                0, 0);
      } else {
        // several points in the tile: one pass per point (uniform loop over the segments)
        uint32_t sm_ = sg.smask, em_ = sg.emask;
        while (sm_) {
          const int lo = __builtin_ctz(sm_), hi = __builtin_ctz(em_);
          sm_ &= sm_ - 1;
          em_ &= em_ - 1;
          const uint32_t pid = (uint32_t)pid_t[lo];
          const float acc = weighted(zbm, mb, lo, hi, true);
          const float sc = sc_t[gch[mb] * 32 + lo];
          __builtin_amdgcn_raw_buffer_store_b16((short)f2bf(acc * sc), O,
                                                h == 0 ? (int)(pid * (uint32_t)(CO * 2) + (uint32_t)(32 * mb + j) * 2u) : (int)OOB,
                                                0, 0);
        }
      }
    };
    if constexpr (NB >= 4 && !(NB == 8 && ZM == 0)) {
      // wide rows (C_o = 256 eval keeps the all-blocks form below: with one wavefront per SIMD the taps of all blocks in
      // flight matter more than the registers: 14.9 against 18.2 ms): the packed activation of every input block first (taps of one block in flight at a time), then one
      // output block at a time: product -> weighted sum -> store
      bf16x8 a[NB][2];
      if constexpr (ZM == 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          u32x4 x[4][2];
          load_taps<CO>(Y, p.t, ok, b, h, x);
          f32x16 za;
          interp16(x, p.t.w, za);
          act_pack(za, s_taba[b], h, keepv, a[b]);
        }
      } else {
        act_a_rows<NB>(p.z, s_taba, h, keepv, a);
      }
      wave_sync();
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
        const f32x16 zbm = linear_b_flipped_blk<NB>(s_eops, lane, a, mb);
        pool_block(zbm, mb);
      }
    } else {
      f32x16 zb[NB];
      {
        f32x16 za[NB];
        if constexpr (ZM == 0) eval_za<CO>(Y, p.t, ok, h, za);
        else unpack_za<NB>(p.z, za);
        bf16x8 a[NB][2];
        act_a<NB>(za, s_taba, h, keepv, a);
        linear_b_flipped<NB>(s_eops, lane, a, zb);
      }
      wave_sync();
      if (single) {
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) pool_block(zb[mb], mb);
      } else {
        // several points in the tile: one pass per point (uniform loop over the segments)
        uint32_t sm_ = sg.smask, em_ = sg.emask;
        while (sm_) {
          const int lo = __builtin_ctz(sm_), hi = __builtin_ctz(em_);
          sm_ &= sm_ - 1;
          em_ &= em_ - 1;
          const uint32_t pid = (uint32_t)pid_t[lo];
#pragma unroll
          for (int mb = 0; mb < NB; ++mb) {
            const float acc = weighted(zb[mb], mb, lo, hi, true);
            const float sc = sc_t[gch[mb] * 32 + lo];
            __builtin_amdgcn_raw_buffer_store_b16((short)f2bf(acc * sc), O,
                                                  h == 0 ? (int)(pid * (uint32_t)(CO * 2) + (uint32_t)(32 * mb + j) * 2u) : (int)OOB,
                                                  0, 0);
          }
        }
      }
    }
    wave_sync();
  });
}

// ------------------------------------------------------------------------------------------------
// attention backward of the bilinear path: scores (from the forward) + E_mod re-evaluated ->
//   dc [V, 4], view records {point | gate * attention (bf16 x 4) | pad}, statistics of the BatchNorm_b backward
//   (S1 = sum dy_b | sum dy_b z_b with dy_b = leaky'(y_b) gate attention grad_out)
// ------------------------------------------------------------------------------------------------
template <int CO, int G, int OCC = (CO == 32 ? 4 : (CO >= 128 ? 1 : 2))>
__global__ __launch_bounds__(CO >= 256 ? 512 : 256, OCC) void emod_attn_bwd_kernel(
    const float* __restrict__ compat, const int32_t* __restrict__ vp, const int2* __restrict__ tiles,
    const int32_t* __restrict__ n_tiles_dev, const bf16_t* __restrict__ Yp, const int4* __restrict__ rows4,
    const float4* __restrict__ w4, const uint4* __restrict__ eops, const float* __restrict__ bna,
    const float* __restrict__ bnb, const int64_t* __restrict__ ptr, const float* __restrict__ gw,
    const float* __restrict__ gb, const bf16_t* __restrict__ gout, const bf16_t* __restrict__ out,
    float* __restrict__ dc_out, uint32_t* __restrict__ rec, float* __restrict__ gwb, double* __restrict__ stats_b,
    const bf16_t* __restrict__ zst, int scaling, float eps, int64_t V, int64_t N, int64_t R) {
  constexpr int NB = CO / 32, NE = G == 1 ? 1 : 2, GS = CO / G;
  constexpr int GL = GS < 32 ? GS : 32;           // lanes of a group inside one block
  constexpr int BPG = GS >= 32 ? GS / 32 : 1;     // blocks of a channel group
  __shared__ __attribute__((aligned(16))) uint4 s_eops[NB * NB * 2 * 64];
  __shared__ __attribute__((aligned(16))) float s_taba[NB][TAB_FLOATS];
  constexpr int NW = CO >= 256 ? 8 : 4;            // wavefronts per block (C_o = 256: W_b fills LDS once per CU -> 8, two per SIMD)
  __shared__ __attribute__((aligned(16))) float s_q[NW][4 * 32], s_ga[NW][4 * 32];
  __shared__ __attribute__((aligned(16))) int s_pid[NW][32];
  __shared__ __attribute__((aligned(16))) float s_E[NW][4];
  __shared__ float s_red[NW * 2 * 32];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < NB * NB * 2 * 64; i += blockDim.x) s_eops[i] = eops[i];
#pragma unroll
  for (int b = 0; b < NB; ++b) stage_tab_c(s_taba[b], bna, CO, 32 * b, nullptr);
  __syncthreads();
  const __amdgpu_buffer_rsrc_t CP = make_rsrc(compat, (uint64_t)V * 16), P = make_rsrc(vp, (uint64_t)V * 4),
                               GO = make_rsrc(gout, (uint64_t)N * CO * 2),
                               OU = make_rsrc(out, (uint64_t)N * CO * 2), DC = make_rsrc(dc_out, (uint64_t)V * 16),
                               RC = make_rsrc(rec, (uint64_t)V * 16);
  const bool s_active = G == 4 || h == 0;
  const uint32_t coff = G == 4 ? 8u * h : 0u;
  int gl[NE];
  float gwl[NE], gbl[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    gl[e] = (G == 4 ? 2 * h : 0) + e;
    if (gl[e] >= G) gl[e] = G - 1;
    gwl[e] = gw ? gw[gl[e]] : 0.f;
    gbl[e] = gw ? gb[gl[e]] : 0.f;
  }
  // per-lane BatchNorm_b constants of the flipped layer: registers up to four blocks, an LDS table from eight on (16
  // registers less in a kernel that spills at that width)
  constexpr bool KB_LDS = NB >= 8;
  __shared__ float s_kb[KB_LDS ? 2 * CO : 1];
  if (KB_LDS) {
    for (int c = threadIdx.x; c < CO; c += blockDim.x) {
      const float g = bnb[2 * CO + c] * bnb[CO + c];
      s_kb[c] = 0.6f * g;
      s_kb[CO + c] = 0.6f * (bnb[3 * CO + c] - bnb[c] * g);
    }
    __syncthreads();
  }
  const LaneBN<KB_LDS ? 1 : NB> kb = lane_bn<KB_LDS ? 1 : NB>(bnb, CO, lane);
  int gch[NB];
#pragma unroll
  for (int mb = 0; mb < NB; ++mb) gch[mb] = (32 * mb + j) / GS;
  float* q_t = s_q[wv];
  float* ga_t = s_ga[wv];
  int* pid_t = s_pid[wv];
  float sb1[NB], sb2[NB];
#pragma unroll
  for (int mb = 0; mb < NB; ++mb) sb1[mb] = sb2[mb] = 0.f;
  float dwa[NE], dba[NE], glob_m[NE], glob_s[NE], glob_E[NE];
  bool seen[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) { dwa[e] = dba[e] = 0.f; glob_m[e] = glob_s[e] = glob_E[e] = 0.f; seen[e] = false; }

  // sum over the lanes of a channel group of per-lane values that live in block mb (GS = 64: both blocks, in-lane first)
  auto group_reduce = [&](float v) { return group_sum<GL>(v); };

  const int n_tiles = n_tiles_dev[0];
  int ta, tb;
  wave_tile_range(tiles, n_tiles, ta, tb);
  struct Pre {
    TileInfo ti;
    int t;
    u32x2 cc;
    int vpj;
    ZaRows<NB> z;        // the stored z_a
  };
  // C_o = 32: without the prefetch register set the kernel fits four wavefronts per SIMD (C_o >= 128 at one wavefront per
  // SIMD lives on the prefetch: 9.0 -> see DESIGN)
  auto loop = [&](auto&& ld, auto&& bd) {
    if constexpr (CO == 32 || (CO >= 128 && OCC == 2) || CO >= 256) run_tiles_single<Pre>(tiles, ta, tb, ld, bd);
    else run_tiles<Pre>(tiles, ta, tb, ld, bd);
  };
  loop([&](const TileInfo& ti, int t) {
    Pre p;
    p.ti = ti;
    p.t = t;
    const bool ok = j < p.ti.nv;
    const uint32_t view = (uint32_t)(p.ti.v0 + j);
    p.cc = ld64(CP, ok ? view * 16u + coff : OOB);
    p.vpj = (int)ld32(P, ok ? view * 4u : OOB);
    p.z = load_za<CO>(zst, ti, j, h);
    return p;
  }, [&](const Pre& p) {
    const int nv = p.ti.nv, frag = p.ti.frag;
    const bool ok = j < nv;
    const uint32_t keepv = ok ? 0xffffffffu : 0u;
    if (h == 0) pid_t[j] = ok ? p.vpj : 0;
    float c[NE];
    c[0] = __uint_as_float(p.cc.x);
    if (NE > 1) c[NE - 1] = __uint_as_float(p.cc.y);
    const int vp0 = __builtin_amdgcn_readfirstlane(p.vpj);
    const bool single = frag != 0 || __ballot(ok && p.vpj != vp0) == 0;
    SegInfo sg;
    if (single) {
      sg.ss = 0;
      sg.se = nv - 1;
    } else {
      sg = seg_setup(p.vpj, j, lane, nv);
    }
    int n_pt = sg.se - sg.ss + 1;
    if (frag != 0) n_pt = (int)(ptr[vp0 + 1] - ptr[vp0]);
    const float isn = scaling ? __builtin_amdgcn_rsqf((float)n_pt) : 1.f;
    const float isl = isn * 1.44269504f;
    auto red_max = [&](float v) { return single ? half_max(v) : seg_total(seg_scan_max(v, sg, lane), sg, h); };
    auto red_sum = [&](float v) { return single ? half_sum(v) : seg_total(seg_scan_sum(v, sg, lane), sg, h); };
    // grad_out of this lane's channels: one value per block for a single-point tile
    float go1[NB];
#pragma unroll
    for (int mb = 0; mb < NB; ++mb)
      go1[mb] = bf2f((bf16_t)__builtin_amdgcn_raw_buffer_load_b16(
          GO, (int)((uint32_t)vp0 * (uint32_t)(CO * 2) + (uint32_t)(32 * mb + j) * 2u), 0, 0));
    wave_sync();
    if (frag == 1) {
      float m_run[NE], s_run[NE];
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        m_run[e] = half_max(ok ? c[e] : -INFINITY);
        s_run[e] = half_sum(ok ? __builtin_amdgcn_exp2f((c[e] - m_run[e]) * isl) : 0.f);
        seen[e] = false;
      }
      for (int t2 = p.t + 1;; ++t2) {
        const TileInfo t2i = get_tile(tiles, t2);
        const bool ok2 = j < t2i.nv;
        const u32x2 cc2 = ld64(CP, ok2 ? (uint32_t)(t2i.v0 + j) * 16u + coff : OOB);
        float c2[NE];
        c2[0] = __uint_as_float(cc2.x);
        if (NE > 1) c2[NE - 1] = __uint_as_float(cc2.y);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const float m2 = vmaxf(m_run[e], half_max(ok2 ? c2[e] : -INFINITY));
          s_run[e] = s_run[e] * __builtin_amdgcn_exp2f((m_run[e] - m2) * isl) + half_sum(ok2 ? __builtin_amdgcn_exp2f((c2[e] - m2) * isl) : 0.f);
          m_run[e] = m2;
        }
        if (t2i.frag == 3) break;
      }
      // E_g = sum_{ch in g} grad_out out / gate from the saved forward output
      float dsum = 0.f;
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
        const float ou = bf2f((bf16_t)__builtin_amdgcn_raw_buffer_load_b16(
            OU, (int)((uint32_t)vp0 * (uint32_t)(CO * 2) + (uint32_t)(32 * mb + j) * 2u), 0, 0));
        const float d = go1[mb] * ou;
        if (GS >= 64) {
          dsum += d;
          if ((mb % BPG) == BPG - 1) {          // the last block of the group
            const float r = group_reduce(dsum);
            if (j == 0 && h == 0) s_E[wv][mb / BPG] = r;
            dsum = 0.f;
          }
        } else {
          const float r = group_reduce(d);
          if ((j % GL) == 0 && h == 0) s_E[wv][gch[mb]] = r;
        }
      }
      wave_sync();
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        glob_m[e] = m_run[e];
        glob_s[e] = s_run[e];
        const float gt0 = gw ? tanh_pos(vmaxf(__builtin_fmaf(gwl[e], m_run[e], gbl[e]), 0.f)) : 1.f;
        glob_E[e] = gt0 > 0.f ? s_E[wv][gl[e]] / gt0 : 0.f;
      }
    }
    // ---- softmax of the tile's views
    float m[NE], a[NE], gt[NE], pre[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if (frag == 0) {
        m[e] = red_max(ok ? c[e] : -INFINITY);
        const float ev = ok ? __builtin_amdgcn_exp2f((c[e] - m[e]) * isl) : 0.f;
        const float s = red_sum(ev);
        a[e] = ev * __builtin_amdgcn_rcpf(s + eps);
      } else {
        m[e] = glob_m[e];
        a[e] = ok ? __builtin_amdgcn_exp2f((c[e] - m[e]) * isl) * __builtin_amdgcn_rcpf(glob_s[e] + eps) : 0.f;
      }
      pre[e] = __builtin_fmaf(gwl[e], m[e], gbl[e]);
      gt[e] = gw ? tanh_pos(fmaxf(pre[e], 0.f)) : 1.f;
    }
    // grad_out value of (block mb, register r): the point of view view_of(r, h) (tag: the tile is one point)
    auto go_at = [&](auto tag, int mb, int r) {
      if constexpr (decltype(tag)::value) {
        return go1[mb];
      } else {
        const int v = view_of(r, h);
        const uint32_t pid = (uint32_t)pid_t[v];
        return bf2f((bf16_t)__builtin_amdgcn_raw_buffer_load_b16(
            GO, v < nv ? (int)(pid * (uint32_t)(CO * 2) + (uint32_t)(32 * mb + j) * 2u) : (int)OOB, 0, 0));
      }
    };
    // ---- E_mod of the views (flipped): raw z_b stays for the statistics
    f32x16 zb[NB < 4 ? NB : 1];
    if constexpr (NB >= 4) {
      // wide rows, one output block at a time: product -> its share of q[v][g] AND of the BatchNorm_b statistics (they
      // need gate * attention of the views, which does not depend on q: written to the LDS table before the products)
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        if (s_active) ga_t[gl[e] * 32 + j] = ok ? gt[e] * a[e] : 0.f;
      }
      bf16x8 aa[NB][2];
      act_a_rows<NB>(p.z, s_taba, h, keepv, aa);
      wave_sync();
      float dq[16];
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
        const f32x16 zbm = linear_b_flipped_blk<NB>(s_eops, lane, aa, mb);
        const float g6m = KB_LDS ? s_kb[32 * mb + j] : kb.g6[KB_LDS ? 0 : mb];
        const float b6m = KB_LDS ? s_kb[CO + 32 * mb + j] : kb.b6[KB_LDS ? 0 : mb];
        if ((mb % BPG) == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) dq[r] = 0.f;
        }
        // the epilogue of the block in two instances behind ONE wave-uniform branch: with the per-value select inside
        // (go_of) hipcc issues the 16 gathered grad_out loads of the several-points case for every tile and keeps them
        // in flight across the blocks
        auto epilogue = [&](auto single_tag) {
          constexpr bool SINGLE = decltype(single_tag)::value;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 w = *reinterpret_cast<const float4*>(ga_t + gch[mb] * 32 + 8 * q + 4 * h);
            const float ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int r = 4 * q + i;
              float go;
              if constexpr (SINGLE) {
                go = go1[mb];
              } else {
                const int v = view_of(r, h);
                const uint32_t pid = (uint32_t)pid_t[v];
                go = bf2f((bf16_t)__builtin_amdgcn_raw_buffer_load_b16(
                    GO, v < nv ? (int)(pid * (uint32_t)(CO * 2) + (uint32_t)(32 * mb + j) * 2u) : (int)OOB, 0, 0));
              }
              const float t = __builtin_fmaf(zbm[r], g6m, b6m);
              const float d = go * leaky06(t);
              if (GS >= 64) {
                dq[r] += d;
              } else {
                const float dr = group_reduce(d);
                if ((j % GL) == 0) q_t[gch[mb] * 32 + view_of(r, h)] = dr;
              }
              // the records carry gate * attention as bf16: the later passes see the rounded weight
              const float gar = bf2f(f2bf(ww[i]));
              const float dval = gar * go;
              const float dy = t > 0.f ? dval : SLOPE * dval;
              sb1[mb] += dy;
              sb2[mb] = __builtin_fmaf(dy, zbm[r], sb2[mb]);
            }
          }
        };
        if (single) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        if (GS >= 64 && (mb % BPG) == BPG - 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float dr = group_reduce(dq[r]);
            if (j == 0) q_t[(mb / BPG) * 32 + view_of(r, h)] = dr;
          }
        }
      }
    } else {
      {
        f32x16 za[NB];
        unpack_za<NB>(p.z, za);
        bf16x8 aa[NB][2];
        act_a<NB>(za, s_taba, h, keepv, aa);
        linear_b_flipped<NB>(s_eops, lane, aa, zb);
      }
      // (the two grad_out forms behind one wave-uniform branch, as in the wide path: round 4)
      auto qsec = [&](auto tag) {
      // ---- q[v][g] = sum_{ch in g} grad_out[p(v)][ch] value[v][ch]: reduce over the lanes of the group
      if (GS >= 64) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float d = 0.f;
#pragma unroll
          for (int mb = 0; mb < NB; ++mb)
            d = __builtin_fmaf(go_at(tag, mb, r), leaky06(__builtin_fmaf(zb[mb][r], kb.g6[mb], kb.b6[mb])), d);
          d = group_reduce(d);
          if (j == 0) q_t[view_of(r, h)] = d;
        }
      } else {
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float d = go_at(tag, mb, r) * leaky06(__builtin_fmaf(zb[mb][r], kb.g6[mb], kb.b6[mb]));
            d = group_reduce(d);
            if ((j % GL) == 0) q_t[gch[mb] * 32 + view_of(r, h)] = d;
          }
        }
      }
          };
      if (single) qsec(std::true_type{});
      else qsec(std::false_type{});
    }
    wave_sync();
    // ---- softmax + gate backward (as chain_bwd.hip attn_bwd_kernel)
    float dcv[NE], gav[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const float qv = q_t[gl[e] * 32 + j];
      float E;
      if (frag == 0) E = red_sum(a[e] * qv);
      else E = glob_E[e];
      const float dpre = (gw && pre[e] > 0.f) ? E * (1.f - gt[e] * gt[e]) : 0.f;
      const bool is_max = ok && c[e] == m[e];
      const uint64_t F = __ballot(is_max);
      const uint32_t Fh = (uint32_t)(F >> (32 * h));
      const uint32_t before = Fh & ((1u << j) - 1u) & ~((1u << sg.ss) - 1u);
      const bool first = is_max && before == 0u && !seen[e];
      if (frag != 0) {
        seen[e] = seen[e] || (Fh != 0u);
        if (frag == 3) seen[e] = false;
      }
      dcv[e] = gt[e] * a[e] * (qv - E) * isn + (first ? dpre * gwl[e] : 0.f);
      gav[e] = gt[e] * a[e];
      const bool last_view = ok && j == sg.se && (frag == 0 || frag == 3);
      if (last_view) {
        dwa[e] += dpre * m[e];
        dba[e] += dpre;
      }
      if (NB < 4 && s_active) ga_t[gl[e] * 32 + j] = ok ? gav[e] : 0.f;
    }
    float dc4[4] = {0.f, 0.f, 0.f, 0.f}, ga4[4] = {0.f, 0.f, 0.f, 0.f};
    if (G == 4) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        uint32_t x0 = __float_as_uint(dcv[e]), x1 = x0;
        swap_halves(x0, x1);
        dc4[e] = dcv[e];
        dc4[2 + e] = __uint_as_float(x1);
        uint32_t y0 = __float_as_uint(gav[e]), y1 = y0;
        swap_halves(y0, y1);
        ga4[e] = gav[e];
        ga4[2 + e] = __uint_as_float(y1);
      }
    } else {
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        dc4[e] = dcv[e];
        ga4[e] = gav[e];
      }
    }
    const bool wr = ok && h == 0;
    const uint32_t vg = (uint32_t)(p.ti.v0 + j);
    st128(DC, wr ? vg * 16u : OOB, as_u4(dc4[0], dc4[1], dc4[2], dc4[3]));
    {
      const u32x4 r = {(uint32_t)p.vpj, pack_bf16x2(ga4[0], ga4[1]), pack_bf16x2(ga4[2], ga4[3]), 0u};
      st128(RC, wr ? vg * 16u : OOB, r);
    }
    wave_sync();
    // ---- statistics of the BatchNorm_b backward: d value = gate attention grad_out, dy_b = leaky'(y_b) d value
    //      (wide rows: accumulated block by block above)
    auto ssec = [&](auto tag) {
#pragma unroll
    for (int mb = 0; mb < (NB < 4 ? NB : 0); ++mb) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 w = *reinterpret_cast<const float4*>(ga_t + gch[mb] * 32 + 8 * q + 4 * h);
        const float ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * q + i;
          // the records carry gate * attention as bf16: the later passes see the rounded weight
          const float gar = bf2f(f2bf(ww[i]));
          const float dval = gar * go_at(tag, mb, r);
          const float t = __builtin_fmaf(zb[mb][r], kb.g6[mb], kb.b6[mb]);
          const float dy = t > 0.f ? dval : SLOPE * dval;
          sb1[mb] += dy;
          sb2[mb] = __builtin_fmaf(dy, zb[mb][r], sb2[mb]);
        }
      }
    }
    };
    if (single) ssec(std::true_type{});
    else ssec(std::false_type{});
    wave_sync();
  });
#pragma unroll
  for (int mb = 0; mb < NB; ++mb) {
    const float a0 = sb1[mb] + other_half(sb1[mb]), a1 = sb2[mb] + other_half(sb2[mb]);
    flush_lane_stats(a0, a1, stats_b, stats_b + CO, 32 * mb, s_red);
  }
  if (gw) {      // one atomic per block and address (chain_common.h flush_red)
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const float dw = half_sum(dwa[e]), db = half_sum(dba[e]);
      if (j == 0 && s_active && ((G == 4 ? 2 * h : 0) + e) < G) {
        s_red[wv * 8 + gl[e]] = dw;
        s_red[wv * 8 + 4 + gl[e]] = db;
      }
    }
    __syncthreads();
    if (threadIdx.x < 2 * G) {
      const int which = threadIdx.x / G, g = threadIdx.x % G;
      float v = 0.f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) v += s_red[w * 8 + 4 * which + g];
      atomicAdd(&gwb[which * G + g], v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// E_mod backward.
// STAGE 2 (standard orientation): d value = record weight x grad_out -> dy_b -> dz_b (BatchNorm_b backward)
//   -> dW_b, da = W_b^T dz_b, dy_a = leaky'(y_a) da written as bf16 [V][CO] (position order), S of BatchNorm_a.
// STAGE 1: dz_a = G_a dy_a - K1 - K2 z_a in place: the gradient of the interpolated Y rows, scattered to the map by
//   the weighted segmented reduction over the row plan of the taps (dva_gather_rows_sum).
// ------------------------------------------------------------------------------------------------
template <int CO, int G, int STAGE>
__global__ __launch_bounds__(256, 2) void emod_bwd_kernel(
    const bf16_t* __restrict__ Yp, const int4* __restrict__ rows4, const float4* __restrict__ w4,
    const int2* __restrict__ tiles, const int32_t* __restrict__ n_tiles_dev, const uint4* __restrict__ eops,
    const float* __restrict__ bna, const float* __restrict__ bnb, const float* __restrict__ sma,
    const float* __restrict__ smb, const uint32_t* __restrict__ rec, const bf16_t* __restrict__ gout,
    bf16_t* __restrict__ da, float* __restrict__ dWb, double* __restrict__ stats_a, const bf16_t* __restrict__ zst,
    int64_t V, int64_t N, int64_t R) {
  constexpr int NB = CO / 32, GS = CO / G;
  constexpr int NT = STAGE == 2 ? NB : 1;
  __shared__ __attribute__((aligned(16))) float s_taba[NB][TAB_FLOATS];
  __shared__ __attribute__((aligned(16))) float s_tabb[STAGE == 2 ? NB : 1][TAB_FLOATS];
  __shared__ __attribute__((aligned(16))) uint4 s_eops[STAGE == 2 ? 2 * NB * NB * 2 * 64 : 1];
  __shared__ __attribute__((aligned(16))) bf16_t s_ta[4][NT][STAGE == 2 ? 32 * TSB : 8], s_tb[4][NT][STAGE == 2 ? 32 * TSB : 8];
  __shared__ float s_redx[STAGE == 2 ? 1 : STATS_RED_FLOATS];
  float* s_red = STAGE == 2 ? reinterpret_cast<float*>(&s_ta[0][0][0]) : s_redx;     // epilogue: D x D floats
  static_assert(STAGE != 2 || sizeof(bf16_t) * 4 * NT * 32 * TSB >= sizeof(float) * D * D, "epilogue buffer");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  if (STAGE == 2) {
    for (int i = threadIdx.x; i < 2 * NB * NB * 2 * 64; i += blockDim.x) s_eops[i] = eops[i];
#pragma unroll
    for (int b = 0; b < NB; ++b) stage_tab_c(s_tabb[b], bnb, CO, 32 * b, smb);
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) stage_tab_c(s_taba[b], bna, CO, 32 * b, STAGE == 1 ? sma : nullptr);
  __syncthreads();
  const __amdgpu_buffer_rsrc_t RC = make_rsrc(rec, (uint64_t)V * 16),
                               GO = make_rsrc(gout, (uint64_t)N * CO * 2);
  float st[STAGE == 2 ? NB : 1][2][16];
  f32x16 accW[STAGE == 2 ? NB : 1][STAGE == 2 ? NB : 1];
  if (STAGE == 2) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[b][0][r] = st[b][1][r] = 0.f;
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
        const f32x16 zero = {0};
        accW[mb][b] = zero;
      }
    }
  }
  const int n_tiles = n_tiles_dev[0];
  int ta, tb;
  wave_tile_range(tiles, n_tiles, ta, tb);
  struct Pre {
    TileInfo ti;
    ZaRows<NB> z;        // the stored z_a
    u32x4 rc;
  };
  auto loop = [&](auto&& ld, auto&& bd) {
    if constexpr (STAGE == 2 && CO > 32) run_tiles_single<Pre>(tiles, ta, tb, ld, bd);
    else run_tiles<Pre>(tiles, ta, tb, ld, bd);
  };
  loop([&](const TileInfo& ti, int t) {
    Pre p;
    p.ti = ti;
    const bool ok = j < p.ti.nv;
    const uint32_t view = (uint32_t)(p.ti.v0 + j);
    p.z = load_za<CO>(zst, ti, j, h);
    if (STAGE == 2) p.rc = ld128(RC, ok ? view * 16u : OOB);
    return p;
  }, [&](const Pre& p) {
    const bool ok = j < p.ti.nv;
    const uint32_t keep = ok ? 0xffffffffu : 0u;
    // the handed-over gradient [V][CO] reaches 4 GiB at the headline size (V = 2^25, CO = 64): one descriptor per tile
    const __amdgpu_buffer_rsrc_t DA = make_rsrc(da + (int64_t)p.ti.v0 * CO, (uint64_t)p.ti.nv * CO * 2);
    f32x16 za[NB];
    unpack_za<NB>(p.z, za);
    if constexpr (STAGE == 1) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const uint32_t off = ok ? (uint32_t)j * (uint32_t)(CO * 2) + (uint32_t)(32 * b + 16 * h) * 2u : OOB;
        const u32x4 lo = ld128(DA, off), hi = ld128(DA, ok ? off + 16u : OOB);
        const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        f32x16 dy;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          dy[2 * i] = __uint_as_float(w[i] << 16);
          dy[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
        float dz[16];
        bn_bwd_apply(za[b], dy, s_taba[b], h, dz);
        st128(DA, off, __builtin_bit_cast(u32x4, pack8(&dz[0])));
        st128(DA, ok ? off + 16u : OOB, __builtin_bit_cast(u32x4, pack8(&dz[8])));
      }
    } else {
      bf16_t* tA[NB];
      bf16_t* tB[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        tA[b] = s_ta[wv][b];
        tB[b] = s_tb[wv][b];
      }
      bf16x8 a[NB][2];
      act_a<NB>(za, s_taba, h, keep, a);
#pragma unroll
      for (int b = 0; b < NB; ++b) tileT_put_packed(tB[b], j, h, a[b]);
      f32x16 zb[NB];
      linear_b_std<NB>(s_eops, lane, a, zb);
      // d value[ch] = (gate attention)[g(ch)] grad_out[point][ch] for the lane's channels 32 mb + chan(r, h)
      const uint32_t pid = p.rc.x;
      const float ga4[4] = {__uint_as_float(p.rc.y << 16), __uint_as_float(p.rc.y & 0xffff0000u),
                            __uint_as_float(p.rc.z << 16), __uint_as_float(p.rc.z & 0xffff0000u)};
      bf16x8 dzp[NB][2];
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
        f32x16 dy;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c0 = 32 * mb + 8 * q + 4 * h;            // channels c0 .. c0 + 3 = chan(4 q + i, h) + 32 mb
          const u32x2 gv = ld64(GO, ok ? pid * (uint32_t)(CO * 2) + (uint32_t)c0 * 2u : OOB);
          const float gg = ga4[G == 1 ? 0 : c0 / GS];
          dy[4 * q] = gg * __uint_as_float(gv.x << 16);
          dy[4 * q + 1] = gg * __uint_as_float(gv.x & 0xffff0000u);
          dy[4 * q + 2] = gg * __uint_as_float(gv.y << 16);
          dy[4 * q + 3] = gg * __uint_as_float(gv.y & 0xffff0000u);
        }
        // dy_b = leaky'(y_b) d value, y_b = G_b z_b + B_b;  dz_b = G_b dy_b - K1 - K2 z_b
        float dz[16];
        {
          asm volatile("" ::: "memory");
          float g_[16], b_[16];
          tab16(s_tabb[mb], T_G, h, g_);
          tab16(s_tabb[mb], T_B, h, b_);
#pragma unroll
          for (int r = 0; r < 16; ++r) dy[r] = __builtin_fmaf(zb[mb][r], g_[r], b_[r]) > 0.f ? dy[r] : SLOPE * dy[r];
        }
        bn_bwd_apply(zb[mb], dy, s_tabb[mb], h, dz);
        pack16(dz, keep, dzp[mb]);
        tileT_put_packed(tA[mb], j, h, dzp[mb]);
      }
      // da = W_b^T dz_b, dy_a = leaky'(y_a) da;  S of BatchNorm_a;  hand dy_a over (bf16, position order)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        f32x16 dya = {0};
        asm volatile("" ::: "memory");
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) {
#pragma unroll
          for (int m = 0; m < 2; ++m) dya = CH_MFMA(lds_op(s_eops, op_bwd<NB>(b, mb, m), lane), dzp[mb][m], dya);
        }
        {
          asm volatile("" ::: "memory");
          float g_[16], b_[16];
          tab16(s_taba[b], T_G, h, g_);
          tab16(s_taba[b], T_B, h, b_);
#pragma unroll
          for (int r = 0; r < 16; ++r) dya[r] = __builtin_fmaf(za[b][r], g_[r], b_[r]) > 0.f ? dya[r] : SLOPE * dya[r];
        }
        bn_bwd_stats(za[b], dya, st[b]);
        float t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = dya[r];
        const uint32_t off = ok ? (uint32_t)j * (uint32_t)(CO * 2) + (uint32_t)(32 * b + 16 * h) * 2u : OOB;
        st128(DA, off, __builtin_bit_cast(u32x4, pack8(&t[0])));
        st128(DA, ok ? off + 16u : OOB, __builtin_bit_cast(u32x4, pack8(&t[8])));
      }
      wave_sync();
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
#pragma unroll
        for (int b = 0; b < NB; ++b) accW[mb][b] = wgrad(tA[mb], tB[b], j, h, accW[mb][b]);   // dW_b[out][in]
      }
      wave_sync();
    }
  });
  if constexpr (STAGE == 2) {
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) {
#pragma unroll
      for (int b = 0; b < NB; ++b) flush_matrix(accW[mb][b], dWb + (32 * mb) * CO + 32 * b, CO, D, false, s_red);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) flush_stats_c<2>(st[b], stats_a, CO, 32 * b, s_red);
  }
}

// ------------------------------------------------------------------------------------------------
// E_mod backward for wide rows (C_o >= 128): stage 2 of emod_bwd_kernel split into the two things its registers cannot
// hold together at this width (16 blocks of dW_b = 256 accumulator registers):
//   MODE 1 (8 wavefronts per block, two per SIMD): z_b block by block -> dy_b -> dz_b (packed: the B operands of the
//          second product) -> dy_a = leaky'(y_a) W_b^T dz_b block by block, written as bf16 [V][CO] (position order), S of
//          BatchNorm_a = column sums of the stored rows through a natural LDS tile (col_sums2: no per-lane accumulators);
//   MODE 2 (one wavefront per SIMD, all of dW_b in registers): the same dz_b again -> natural LDS tiles of dz_b and y_a
//          -> dW_b += dz_b^T y_a through the transpose read (wgradN).
// One more evaluation of Linear_b (2 V CO^2 flop) against 8 CO more bytes per view for handing dz_b over.
// ------------------------------------------------------------------------------------------------
//   MODE 3 / MODE 4 (C_o = 256, one wavefront per SIMD): MODE 1 cut in two, because the operands of W_b (128 KB per
//          orientation) do not fit LDS together: MODE 3 = z_b -> dz_b, written as bf16 [V][CO] into the buffer that will hold
//          dy_a; MODE 4 = the stored dz_b -> dy_a IN PLACE (a lane reads and writes the same 32-byte pieces of its view) + S of
//          BatchNorm_a.  Between the two, emodw_wgrad_coop_kernel takes dW_b from the stored dz_b.
template <int CO, int G, int MODE>
__global__ __launch_bounds__(MODE == 2 ? 256 : 512, 1) void emodw_bwd_kernel(
    const int2* __restrict__ tiles, const int32_t* __restrict__ n_tiles_dev, const uint4* __restrict__ eops,
    const float* __restrict__ bna, const float* __restrict__ bnb, const float* __restrict__ smb,
    const uint32_t* __restrict__ rec, const bf16_t* __restrict__ gout, bf16_t* __restrict__ da, float* __restrict__ dWb,
    double* __restrict__ stats_a, const bf16_t* __restrict__ zst, int64_t V, int64_t N) {
  constexpr int NB = CO / 32, GS = CO / G, NW = MODE == 2 ? 4 : 8;
  constexpr bool FWD = MODE != 4;                    // evaluates z_b -> dz_b
  constexpr bool DYA = MODE == 1 || MODE == 4;       // evaluates dy_a
  constexpr int NT = MODE == 2 ? NB : 1;
  constexpr int N_EOPS = (MODE == 1 ? 2 : 1) * NB * NB * 2 * 64;
  constexpr int BWD_BASE = MODE == 4 ? NB * NB * 2 : 0;      // MODE 4 holds the second half of the table only
  __shared__ __attribute__((aligned(16))) float s_taba[NB][TAB_FLOATS];
  __shared__ __attribute__((aligned(16))) float s_tabb[FWD ? NB : 1][FWD ? TAB_FLOATS : 8];
  __shared__ __attribute__((aligned(16))) uint4 s_eops[N_EOPS];
  // MODE 1 / 4: [0] = the dy_a block, [1] = the z_a block;  MODE 2: NB tiles of dz_b, NB tiles of y_a;  MODE 3: unused
  // (MODE 4: ONE tile per wavefront, used twice per block -- dy_a, then the products dy_a z_a: with W_b^T filling 128 KB there
  //  is room for 8 x 1 tiles, not 8 x 2)
  __shared__ __attribute__((aligned(16))) bf16_t s_ta[MODE == 3 ? 1 : NW][NT][MODE == 3 ? 8 : 32 * TSB],
      s_tb[(MODE == 3 || MODE == 4) ? 1 : NW][NT][(MODE == 3 || MODE == 4) ? 8 : 32 * TSB];
  float* s_red = reinterpret_cast<float*>(&s_ta[0][0][0]);     // epilogue: D x D floats | NW x 64 floats
  static_assert(MODE == 3 || sizeof(bf16_t) * NW * NT * 32 * TSB >= sizeof(float) * (MODE == 2 ? D * D : NW * 64),
                "epilogue buffer");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < N_EOPS; i += blockDim.x) s_eops[i] = eops[BWD_BASE * 64 + i];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (FWD) stage_tab_c(s_tabb[b], bnb, CO, 32 * b, smb);
    stage_tab_c(s_taba[b], bna, CO, 32 * b, nullptr);
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t RC = make_rsrc(rec, (uint64_t)V * 16), GO = make_rsrc(gout, (uint64_t)N * CO * 2);
  float sa1[DYA ? NB : 1], sa2[DYA ? NB : 1];
  f32x16 accW[MODE == 2 ? NB : 1][MODE == 2 ? NB : 1];
  if (DYA) {
#pragma unroll
    for (int b = 0; b < NB; ++b) sa1[b] = sa2[b] = 0.f;
  }
  if (MODE == 2) {
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const f32x16 zero = {0};
        accW[mb][b] = zero;
      }
    }
  }
  const int n_tiles = n_tiles_dev[0];
  int ta, tb;
  wave_tile_range(tiles, n_tiles, ta, tb);
  struct Pre {
    TileInfo ti;
    ZaRows<NB> z;        // the stored z_a
    ZaRows<MODE == 4 ? NB : 1> dz;      // MODE 4: the stored dz_b
    u32x4 rc;
  };
  auto loop = [&](auto&& ld, auto&& bd) {
    // (MODE 2 with the prefetch register set: 772 bytes of scratch next to the 256 accumulator registers -- dropped)
    run_tiles_single<Pre>(tiles, ta, tb, ld, bd);
  };
  loop([&](const TileInfo& ti, int t) {
    Pre p;
    p.ti = ti;
    const bool ok = j < p.ti.nv;
    p.z = load_za<CO>(zst, ti, j, h);
    if constexpr (MODE == 4) p.dz = load_za<CO>(da, ti, j, h);
    else p.rc = ld128(RC, ok ? (uint32_t)(p.ti.v0 + j) * 16u : OOB);
    return p;
  }, [&](const Pre& p) {
    const bool ok = j < p.ti.nv;
    const uint32_t keep = ok ? 0xffffffffu : 0u;
    // (the handed-over gradient [V][CO] exceeds 4 GiB at the headline size: one descriptor per tile)
    const __amdgpu_buffer_rsrc_t DA = make_rsrc(da + (int64_t)p.ti.v0 * CO, (uint64_t)p.ti.nv * CO * 2);
    bf16x8 dzp[(MODE == 1 || MODE == 4) ? NB : 1][2];
    if constexpr (FWD) {
      bf16x8 a[NB][2];
      act_a_rows<NB>(p.z, s_taba, h, keep, a);
      if (MODE == 2) {
#pragma unroll
        for (int b = 0; b < NB; ++b) tileN_put_packed(s_tb[wv][b], j, h, a[b]);
      }
      // d value[ch] = (gate attention)[g(ch)] grad_out[point][ch] for the lane's channels 32 mb + chan(r, h)
      const uint32_t pid = p.rc.x;
      const float ga4[4] = {__uint_as_float(p.rc.y << 16), __uint_as_float(p.rc.y & 0xffff0000u),
                            __uint_as_float(p.rc.z << 16), __uint_as_float(p.rc.z & 0xffff0000u)};
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
        const f32x16 zb = linear_b_std_blk<NB>(s_eops, lane, a, mb);
        f32x16 dy;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c0 = 32 * mb + 8 * q + 4 * h;            // channels c0 .. c0 + 3 = chan(4 q + i, h) + 32 mb
          const u32x2 gv = ld64(GO, ok ? pid * (uint32_t)(CO * 2) + (uint32_t)c0 * 2u : OOB);
          const float gg = ga4[G == 1 ? 0 : c0 / GS];
          dy[4 * q] = gg * __uint_as_float(gv.x << 16);
          dy[4 * q + 1] = gg * __uint_as_float(gv.x & 0xffff0000u);
          dy[4 * q + 2] = gg * __uint_as_float(gv.y << 16);
          dy[4 * q + 3] = gg * __uint_as_float(gv.y & 0xffff0000u);
        }
        // dy_b = leaky'(y_b) d value, y_b = G_b z_b + B_b;  dz_b = G_b dy_b - K1 - K2 z_b
        float dz[16];
        {
          asm volatile("" ::: "memory");
          float g_[16], b_[16];
          tab16(s_tabb[mb], T_G, h, g_);
          tab16(s_tabb[mb], T_B, h, b_);
#pragma unroll
          for (int r = 0; r < 16; ++r) dy[r] = __builtin_fmaf(zb[r], g_[r], b_[r]) > 0.f ? dy[r] : SLOPE * dy[r];
        }
        bn_bwd_apply(zb, dy, s_tabb[mb], h, dz);
        if (MODE == 1) {
          pack16(dz, keep, dzp[mb]);
        } else {
          bf16x8 t2[2];
          pack16(dz, keep, t2);
          if (MODE == 2) {
            tileN_put_packed(s_ta[wv][mb], j, h, t2);
          } else {      // MODE 3: hand dz_b over (position order, the lane's 32 bytes of block mb)
            const uint32_t off = ok ? (uint32_t)j * (uint32_t)(CO * 2) + (uint32_t)(32 * mb + 16 * h) * 2u : OOB;
            st128(DA, off, __builtin_bit_cast(u32x4, t2[0]));
            st128(DA, ok ? off + 16u : OOB, __builtin_bit_cast(u32x4, t2[1]));
          }
        }
      }
    } else {
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
        dzp[mb][0] = __builtin_bit_cast(bf16x8, p.dz.q[mb][0]);
        dzp[mb][1] = __builtin_bit_cast(bf16x8, p.dz.q[mb][1]);
      }
    }
    if constexpr (DYA) {
      // da = W_b^T dz_b, dy_a = leaky'(y_a) da, handed over as bf16 (position order);  S of BatchNorm_a from the stored rows
      bf16_t* tdy = s_ta[wv][0];
      bf16_t* tz = s_tb[MODE == 4 ? 0 : wv][0];      // (MODE 4 has no second tile)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        f32x16 dya = {0};
        asm volatile("" ::: "memory");
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) {
#pragma unroll
          for (int m = 0; m < 2; ++m)
            dya = CH_MFMA(lds_op(s_eops, op_bwd<NB>(b, mb, m) - BWD_BASE, lane), dzp[mb][m], dya);
        }
        f32x16 za;
        unpack_za_blk<NB>(p.z, b, za);
        float t[16];
        {
          asm volatile("" ::: "memory");
          float g_[16], b_[16];
          tab16(s_taba[b], T_G, h, g_);
          tab16(s_taba[b], T_B, h, b_);
#pragma unroll
          for (int r = 0; r < 16; ++r) t[r] = __builtin_fmaf(za[r], g_[r], b_[r]) > 0.f ? dya[r] : SLOPE * dya[r];
        }
        bf16x8 pk[2] = {pack8(&t[0]), pack8(&t[8])};     // lanes without a view: dz_b = 0 -> dy_a = 0
        const uint32_t off = ok ? (uint32_t)j * (uint32_t)(CO * 2) + (uint32_t)(32 * b + 16 * h) * 2u : OOB;
        st128(DA, off, __builtin_bit_cast(u32x4, pk[0]));
        st128(DA, ok ? off + 16u : OOB, __builtin_bit_cast(u32x4, pk[1]));
        if constexpr (MODE == 4) {
          tileN_put_packed(tdy, j, h, pk);
          wave_sync();
          col_sum1(tdy, lane, sa1[b]);
          wave_sync();
          float dyr[16], pr[16];      // the stored (rounded) dy_a times the stored z_a, rounded once more for the tile
          unpack8(pk[0], reinterpret_cast<float(&)[8]>(dyr[0]));
          unpack8(pk[1], reinterpret_cast<float(&)[8]>(dyr[8]));
#pragma unroll
          for (int r = 0; r < 16; ++r) pr[r] = dyr[r] * za[r];
          bf16x8 pp[2] = {pack8(&pr[0]), pack8(&pr[8])};
          tileN_put_packed(tdy, j, h, pp);
          wave_sync();
          col_sum1(tdy, lane, sa2[b]);
          wave_sync();
        } else {
          bf16x8 zk[2] = {__builtin_bit_cast(bf16x8, p.z.q[b][0]), __builtin_bit_cast(bf16x8, p.z.q[b][1])};
          tileN_put_packed(tdy, j, h, pk);
          tileN_put_packed(tz, j, h, zk);
          wave_sync();
          col_sums2(tdy, tz, lane, sa1[b], sa2[b]);
          wave_sync();
        }
      }
    } else if constexpr (MODE == 2) {
      wave_sync();
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
#pragma unroll
        for (int b = 0; b < NB; ++b) accW[mb][b] = wgradN(s_ta[wv][mb], s_tb[wv][b], lane, accW[mb][b]);   // dW_b[out][in]
      }
      wave_sync();
    }
  });
  if constexpr (DYA) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float a0 = sa1[b] + other_half(sa1[b]), a1 = sa2[b] + other_half(sa2[b]);
      flush_lane_stats(a0, a1, stats_a, stats_a + CO, 32 * b, s_red, true);
    }
  } else if constexpr (MODE == 2) {
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) {
#pragma unroll
      for (int b = 0; b < NB; ++b)
        flush_matrix_nat(accW[mb][b], dWb + (32 * mb) * CO + 32 * b, CO, D, false, s_red, true);
    }
  }
}

// dW_b += dz_b^T y_a from the STORED dz_b (emodw_bwd MODE 3) and z_a, for widths whose 16 NB^2 accumulator registers no
// wavefront can hold (C_o = 256: 1024): the NB wavefronts of a block work on the SAME 32-view tile -- wavefront w loads
// block w of z_a and of dz_b, applies BatchNorm_a + LeakyReLU, writes both as natural LDS tiles, and owns the rows of
// dW_b of output block w (NB accumulator blocks = 128 registers at C_o = 256): after the block barrier it reads its own
// dz_b tile and the y_a tiles of all input blocks through the transpose read.  No weight operands in LDS at all.
// DYA (round 6, C_o = 256): the same launch also turns the stored dz_b into dy_a IN PLACE -- what emodw_bwd MODE 4 did in a
// pass of its own (z_a + dz_b read once more, 1 KB per view).  Wavefront w keeps its 32 x CO slice of W_b^T in registers
// (weight-stationary: 64 VGPRs at C_o = 256; the 128 KB of the whole orientation do not fit LDS next to the tiles), the
// dz_b tiles of ALL blocks are in LDS anyway (natural tiles: what a lane stored is the packed B operand of its view), so
// da[block w] = sum_mb W_b^T[w][mb] dz_b[mb] is 2 NB more matrix instructions per wavefront and tile, dy_a = leaky'(y_a) da
// goes back over the wavefront's own 32 bytes of the view's dz_b row (every reader of the row takes it from LDS), and S of
// BatchNorm_a are column sums through the wavefront's slot of the idle tile buffer, exactly as MODE 4 took them
// (statistics of the stored, rounded values).
template <int CO, bool DYA>
__global__ __launch_bounds__(CO * 2, 1) void emodw_wgrad_coop_kernel(
    const int2* __restrict__ tiles, const int32_t* __restrict__ n_tiles_dev, const float* __restrict__ bna,
    bf16_t* __restrict__ dzst, const bf16_t* __restrict__ zst, float* __restrict__ dWb, const uint4* __restrict__ eops,
    double* __restrict__ stats_a) {
  constexpr int NB = CO / 32;
  __shared__ __attribute__((aligned(16))) float s_taba[NB][TAB_FLOATS];
  __shared__ __attribute__((aligned(16))) bf16_t s_ta[2][NB][32 * TSB], s_tb[2][NB][32 * TSB];     // double-buffered tiles
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int b = 0; b < NB; ++b) stage_tab_c(s_taba[b], bna, CO, 32 * b, nullptr);
  __syncthreads();
  f32x16 accW[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const f32x16 zero = {0};
    accW[b] = zero;
  }
  // the wavefront's slice of W_b^T: the first k-half of every block in registers (32 VGPRs), the second in LDS (64 KB for
  // the eight wavefronts: all 64 VGPRs next to the 128 accumulators of dW_b spilled)
  __shared__ __attribute__((aligned(16))) uint4 s_wt[DYA ? NB : 1][DYA ? NB : 1][DYA ? 64 : 1];
  bf16x8 wT[DYA ? NB : 1];
  if constexpr (DYA) {
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) {
      wT[mb] = load_op(eops, op_bwd<NB>(w, mb, 0), lane);
      s_wt[w][mb][lane] = eops[op_bwd<NB>(w, mb, 1) * 64 + lane];
    }
  }
  float sa1 = 0.f, sa2 = 0.f;
  const int n_tiles = n_tiles_dev[0];
  const int t0 = (int)((int64_t)n_tiles * blockIdx.x / gridDim.x), t1 = (int)((int64_t)n_tiles * (blockIdx.x + 1) / gridDim.x);
  auto fetch = [&](int t, u32x4 (&z)[2], u32x4 (&d)[2], bool& ok, TileInfo& ti) {
    ti = get_tile(tiles, t);
    ok = j < ti.nv;
    const __amdgpu_buffer_rsrc_t Z = make_rsrc(zst + (int64_t)ti.v0 * CO, (uint64_t)ti.nv * CO * 2),
                                 DZ = make_rsrc(dzst + (int64_t)ti.v0 * CO, (uint64_t)ti.nv * CO * 2);
    const uint32_t off = ok ? (uint32_t)j * (uint32_t)(CO * 2) + (uint32_t)(32 * w + 16 * h) * 2u : OOB;
    z[0] = ld128(Z, off);
    z[1] = ld128(Z, ok ? off + 16u : OOB);
    d[0] = ld128(DZ, off);
    d[1] = ld128(DZ, ok ? off + 16u : OOB);
  };
  u32x4 zq[2], dq[2];
  bool ok = false;
  TileInfo ti_next;
  ti_next.v0 = ti_next.nv = ti_next.frag = 0;
  if (t0 < t1) fetch(t0, zq, dq, ok, ti_next);
  for (int t = t0; t < t1; ++t) {
    const int buf = (t - t0) & 1;
    const bool ok_cur = ok;
    const TileInfo ti = ti_next;
    const u32x4 zk0 = zq[0], zk1 = zq[1];      // the z_a block of this tile (dy_a below needs the sign of y_a)
    {
      // this wavefront's block of the tile: y_a = leaky(BatchNorm_a(z_a)) (0 for lanes without a view), dz_b as stored
      f32x16 za;
      const uint32_t v[8] = {zq[0].x, zq[0].y, zq[0].z, zq[0].w, zq[1].x, zq[1].y, zq[1].z, zq[1].w};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        za[2 * i] = __uint_as_float(v[i] << 16);
        za[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
      }
      bf16x8 a[2];
      act_pack(za, s_taba[w], h, ok ? 0xffffffffu : 0u, a);
      const bf16x8 dzk[2] = {__builtin_bit_cast(bf16x8, dq[0]), __builtin_bit_cast(bf16x8, dq[1])};
      tileN_put_packed(s_tb[buf][w], j, h, a);
      tileN_put_packed(s_ta[buf][w], j, h, dzk);
    }
    if (t + 1 < t1) fetch(t + 1, zq, dq, ok, ti_next);       // the next tile's loads fly during the products
    __syncthreads();       // tiles of all blocks written (the other buffer is free: its readers passed this barrier once more)
#pragma unroll
    for (int b = 0; b < NB; ++b) accW[b] = wgradN(s_ta[buf][w], s_tb[buf][b], lane, accW[b]);      // dW_b[32 w ..][32 b ..]
    if constexpr (DYA) {
      f32x16 dya = {0};
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
        const bf16_t* row = s_ta[buf][mb] + j * TSB + 16 * h;        // the packed dz_b operand of view j, block mb
        dya = CH_MFMA(wT[mb], *reinterpret_cast<const bf16x8*>(row), dya);
        dya = CH_MFMA(__builtin_bit_cast(bf16x8, s_wt[w][mb][lane]), *reinterpret_cast<const bf16x8*>(row + 8), dya);
      }
      const uint32_t zv[8] = {zk0.x, zk0.y, zk0.z, zk0.w, zk1.x, zk1.y, zk1.z, zk1.w};
      float tt[16], zaf[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {        // four channels at a time: two float4 of constants live
        const float4 g4 = *reinterpret_cast<const float4*>(s_taba[w] + T_G * D + 16 * h + 4 * q);
        const float4 b4 = *reinterpret_cast<const float4*>(s_taba[w] + T_B * D + 16 * h + 4 * q);
        const float g_[4] = {g4.x, g4.y, g4.z, g4.w}, b_[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * q + e;
          zaf[r] = (r & 1) ? __uint_as_float(zv[r >> 1] & 0xffff0000u) : __uint_as_float(zv[r >> 1] << 16);
          tt[r] = __builtin_fmaf(zaf[r], g_[e], b_[e]) > 0.f ? dya[r] : SLOPE * dya[r];
        }
      }
      bf16x8 pk[2] = {pack8(&tt[0]), pack8(&tt[8])};     // lanes without a view: dz_b = 0 -> dy_a = 0
      {
        const __amdgpu_buffer_rsrc_t DA = make_rsrc(dzst + (int64_t)ti.v0 * CO, (uint64_t)ti.nv * CO * 2);
        const uint32_t off = ok_cur ? (uint32_t)j * (uint32_t)(CO * 2) + (uint32_t)(32 * w + 16 * h) * 2u : OOB;
        st128(DA, off, __builtin_bit_cast(u32x4, pk[0]));
        st128(DA, ok_cur ? off + 16u : OOB, __builtin_bit_cast(u32x4, pk[1]));
      }
      // S of BatchNorm_a: column sums of the stored dy_a and of dy_a z_a (rounded once more for the tile, as MODE 4 did)
      // through this wavefront's slot of the OTHER buffer: nobody reads it before this wavefront refills it
      bf16_t* tdy = s_ta[buf ^ 1][w];
      tileN_put_packed(tdy, j, h, pk);
      wave_sync();
      col_sum1(tdy, lane, sa1);
      wave_sync();
      float dyr[16], pr[16];
      unpack8(pk[0], reinterpret_cast<float(&)[8]>(dyr[0]));
      unpack8(pk[1], reinterpret_cast<float(&)[8]>(dyr[8]));
#pragma unroll
      for (int r = 0; r < 16; ++r) pr[r] = dyr[r] * zaf[r];
      bf16x8 pp[2] = {pack8(&pr[0]), pack8(&pr[8])};
      tileN_put_packed(tdy, j, h, pp);
      wave_sync();
      col_sum1(tdy, lane, sa2);
      wave_sync();
    }
  }
  // rows of the accumulator = image columns of the dz_b tile, columns = image columns of the y_a tile (cperm)
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      atomicAdd(&dWb[(size_t)(32 * w + cperm(chan(r, h))) * CO + 32 * b + cperm(j)], accW[b][r]);
  }
  if constexpr (DYA) {
    const float a0 = sa1 + other_half(sa1), a1 = sa2 + other_half(sa2);
    if (h == 0) {          // lane n of the first half-wave: image column n of block w = channel 32 w + cperm(n)
      atomicAdd(&stats_a[32 * w + cperm(j)], (double)a0);
      atomicAdd(&stats_a[CO + 32 * w + cperm(j)], (double)a1);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// C_o = 256 backward WITHOUT the dz_b hand-off (round 6): two cooperative kernels that each evaluate dz_b of a 32-view
// tile themselves -- the eight wavefronts of a block own one output block each and share the tile through LDS --
//   WGRAD = false:  z_a -> y_a tiles | barrier | z_b[w] = W_b[w][:] y_a, dz_b[w] (BatchNorm_b backward of the attention's
//                   value gradient) -> dz_b tiles | barrier | dy_a[w] = leaky'(y_a) W_b^T[w][:] dz_b -> bf16 [V][CO] (written
//                   once), S of BatchNorm_a (column sums);
//   WGRAD = true:   the same up to the dz_b tiles | barrier | dW_b[32 w ..][:] += dz_b[w]^T y_a (128 accumulator registers).
// Both keep their W_b slice weight-stationary (WGRAD = false: W_b and W_b^T slices in 128 VGPRs; WGRAD = true: the first
// k-halves of W_b in 32 VGPRs, the second in 64 KB of LDS next to the accumulators).  Against MODE 3 + the merged
// cooperative kernel: 1 KB per view less (dz_b is never written or read: 512 + 512 B), 128 matrix instructions per tile
// more (dz_b evaluated twice).  MEASURED (round 6): 25.9 against 23.3 ms for the pair it replaces -- 51 GB in 25.9 ms is
// 2 TB/s and the matrix pipe is ~30 % busy: the eight wavefronts run in lockstep through two block barriers per tile and
// one block per CU (233 / 246 VGPRs) leaves nothing to overlap them with.  Opt-in (DVA_EMOD_COOP2=1), kept for the A/B.  The bf16 rounding of dz_b is the stored one's (pack16), so the results are those of
// the three-kernel form up to the order of the fp32 atomics.  One tile buffer per kind suffices: two barriers per tile.
template <int CO, int G, bool WGRAD>
__global__ __launch_bounds__(CO * 2, 1) void emodw_coop2_kernel(
    const int2* __restrict__ tiles, const int32_t* __restrict__ n_tiles_dev, const uint4* __restrict__ eops,
    const float* __restrict__ bna, const float* __restrict__ bnb, const float* __restrict__ smb,
    const uint32_t* __restrict__ rec, const bf16_t* __restrict__ gout, bf16_t* __restrict__ da, float* __restrict__ dWb,
    double* __restrict__ stats_a, const bf16_t* __restrict__ zst, int64_t V, int64_t N) {
  constexpr int NB = CO / 32, GS = CO / G;
  __shared__ __attribute__((aligned(16))) float s_taba[NB][TAB_FLOATS], s_tabb[NB][TAB_FLOATS];
  __shared__ __attribute__((aligned(16))) bf16_t s_ta[NB][32 * TSB], s_tb[NB][32 * TSB];     // dz_b tiles | y_a tiles
  __shared__ __attribute__((aligned(16))) uint4 s_wb[WGRAD ? NB : 1][WGRAD ? NB : 1][WGRAD ? 64 : 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    stage_tab_c(s_taba[b], bna, CO, 32 * b, nullptr);
    stage_tab_c(s_tabb[b], bnb, CO, 32 * b, smb);
  }
  // weight-stationary slices of this wavefront's output block
  bf16x8 wB0[NB], wB1[WGRAD ? 1 : NB], wT[WGRAD ? 1 : NB][2];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    wB0[b] = load_op(eops, op_fwd<NB>(w, b, 0), lane);
    if constexpr (WGRAD) {
      s_wb[w][b][lane] = eops[op_fwd<NB>(w, b, 1) * 64 + lane];
    } else {
      wB1[b] = load_op(eops, op_fwd<NB>(w, b, 1), lane);
      wT[b][0] = load_op(eops, op_bwd<NB>(w, b, 0), lane);
      wT[b][1] = load_op(eops, op_bwd<NB>(w, b, 1), lane);
    }
  }
  __syncthreads();
  f32x16 accW[WGRAD ? NB : 1];
#pragma unroll
  for (int b = 0; b < (WGRAD ? NB : 1); ++b) {
    const f32x16 zero = {0};
    accW[b] = zero;
  }
  float sa1 = 0.f, sa2 = 0.f;
  const __amdgpu_buffer_rsrc_t RC = make_rsrc(rec, (uint64_t)V * 16), GO = make_rsrc(gout, (uint64_t)N * CO * 2);
  const int n_tiles = n_tiles_dev[0];
  const int t0 = (int)((int64_t)n_tiles * blockIdx.x / gridDim.x), t1 = (int)((int64_t)n_tiles * (blockIdx.x + 1) / gridDim.x);
  auto fetch = [&](int t, u32x4 (&z)[2], u32x4& rc, bool& ok, TileInfo& ti) {
    ti = get_tile(tiles, t);
    ok = j < ti.nv;
    const __amdgpu_buffer_rsrc_t Z = make_rsrc(zst + (int64_t)ti.v0 * CO, (uint64_t)ti.nv * CO * 2);
    const uint32_t off = ok ? (uint32_t)j * (uint32_t)(CO * 2) + (uint32_t)(32 * w + 16 * h) * 2u : OOB;
    z[0] = ld128(Z, off);
    z[1] = ld128(Z, ok ? off + 16u : OOB);
    rc = ld128(RC, ok ? (uint32_t)(ti.v0 + j) * 16u : OOB);
  };
  u32x4 zq[2], rq;
  bool ok = false;
  TileInfo ti_next;
  ti_next.v0 = ti_next.nv = ti_next.frag = 0;
  if (t0 < t1) fetch(t0, zq, rq, ok, ti_next);
  for (int t = t0; t < t1; ++t) {
    const bool ok_cur = ok;
    const uint32_t keep = ok_cur ? 0xffffffffu : 0u;
    const TileInfo ti = ti_next;
    const u32x4 zk0 = zq[0], zk1 = zq[1], rc = rq;
    {
      // this wavefront's block of y_a = leaky(BatchNorm_a(z_a)) (0 for lanes without a view) -> its tile
      f32x16 za;
      const uint32_t v[8] = {zk0.x, zk0.y, zk0.z, zk0.w, zk1.x, zk1.y, zk1.z, zk1.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        za[2 * i] = __uint_as_float(v[i] << 16);
        za[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
      }
      bf16x8 a[2];
      act_pack(za, s_taba[w], h, keep, a);
      tileN_put_packed(s_tb[w], j, h, a);
    }
    // the value gradient of this block, requested before the first barrier: (gate attention)[g] grad_out[point][ch]
    u32x2 gv[4];
    {
      const uint32_t pid = rc.x;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        gv[q] = ld64(GO, ok_cur ? pid * (uint32_t)(CO * 2) + (uint32_t)(32 * w + 8 * q + 4 * h) * 2u : OOB);
    }
    if (t + 1 < t1) fetch(t + 1, zq, rq, ok, ti_next);       // the next tile's loads fly during the products
    __syncthreads();       // y_a tiles of all blocks written (and: the dz_b tiles of the previous tile are consumed)
    // ---- z_b[w] = sum_b W_b[w][b] y_a[b]
    f32x16 zb = {0};
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const bf16_t* row = s_tb[b] + j * TSB + 16 * h;          // the packed y_a operand of view j, block b
      zb = CH_MFMA(wB0[b], *reinterpret_cast<const bf16x8*>(row), zb);
      if constexpr (WGRAD) zb = CH_MFMA(__builtin_bit_cast(bf16x8, s_wb[w][b][lane]), *reinterpret_cast<const bf16x8*>(row + 8), zb);
      else zb = CH_MFMA(wB1[b], *reinterpret_cast<const bf16x8*>(row + 8), zb);
    }
    // ---- dz_b[w]: dy_b = leaky'(y_b) d value, dz_b = G_b dy_b - K1 - K2 z_b (as emodw_bwd MODE 3)
    {
      const float ga4[4] = {__uint_as_float(rc.y << 16), __uint_as_float(rc.y & 0xffff0000u),
                            __uint_as_float(rc.z << 16), __uint_as_float(rc.z & 0xffff0000u)};
      const float gg = ga4[G == 1 ? 0 : (32 * w) / GS];      // a 32-channel block lies inside one group (GS >= 32)
      static_assert(GS % 32 == 0, "a block inside one channel group");
      float dz[16];
      asm volatile("" ::: "memory");
#pragma unroll
      for (int q = 0; q < 4; ++q) {        // four channels at a time: sixteen constants live instead of 64 (register budget)
        const float dy[4] = {gg * __uint_as_float(gv[q].x << 16), gg * __uint_as_float(gv[q].x & 0xffff0000u),
                             gg * __uint_as_float(gv[q].y << 16), gg * __uint_as_float(gv[q].y & 0xffff0000u)};
        const int o = 16 * h + 4 * q;
        const float4 g4 = *reinterpret_cast<const float4*>(s_tabb[w] + T_G * D + o);
        const float4 b4 = *reinterpret_cast<const float4*>(s_tabb[w] + T_B * D + o);
        const float4 a4 = *reinterpret_cast<const float4*>(s_tabb[w] + T_K1 * D + o);
        const float4 c4 = *reinterpret_cast<const float4*>(s_tabb[w] + T_K2 * D + o);
        const float g_[4] = {g4.x, g4.y, g4.z, g4.w}, b_[4] = {b4.x, b4.y, b4.z, b4.w};
        const float k1[4] = {a4.x, a4.y, a4.z, a4.w}, k2[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * q + e;
          const float dyb = __builtin_fmaf(zb[r], g_[e], b_[e]) > 0.f ? dy[e] : SLOPE * dy[e];
          dz[r] = __builtin_fmaf(-k2[e], zb[r], __builtin_fmaf(g_[e], dyb, -k1[e]));       // = bn_bwd_apply
        }
      }
      bf16x8 t2[2];
      pack16(dz, keep, t2);
      tileN_put_packed(s_ta[w], j, h, t2);
    }
    __syncthreads();       // dz_b tiles of all blocks written (and: the y_a tiles are consumed by the products above)
    if constexpr (WGRAD) {
#pragma unroll
      for (int b = 0; b < NB; ++b) accW[b] = wgradN(s_ta[w], s_tb[b], lane, accW[b]);      // dW_b[32 w ..][32 b ..]
    } else {
      f32x16 dya = {0};
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
        const bf16_t* row = s_ta[mb] + j * TSB + 16 * h;        // the packed dz_b operand of view j, block mb
        dya = CH_MFMA(wT[mb][0], *reinterpret_cast<const bf16x8*>(row), dya);
        dya = CH_MFMA(wT[mb][1], *reinterpret_cast<const bf16x8*>(row + 8), dya);
      }
      const uint32_t zv[8] = {zk0.x, zk0.y, zk0.z, zk0.w, zk1.x, zk1.y, zk1.z, zk1.w};
      float tt[16], zaf[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 g4 = *reinterpret_cast<const float4*>(s_taba[w] + T_G * D + 16 * h + 4 * q);
        const float4 b4 = *reinterpret_cast<const float4*>(s_taba[w] + T_B * D + 16 * h + 4 * q);
        const float g_[4] = {g4.x, g4.y, g4.z, g4.w}, b_[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * q + e;
          zaf[r] = (r & 1) ? __uint_as_float(zv[r >> 1] & 0xffff0000u) : __uint_as_float(zv[r >> 1] << 16);
          tt[r] = __builtin_fmaf(zaf[r], g_[e], b_[e]) > 0.f ? dya[r] : SLOPE * dya[r];
        }
      }
      bf16x8 pk[2] = {pack8(&tt[0]), pack8(&tt[8])};     // lanes without a view: dz_b = 0 -> dy_a = 0
      {
        const __amdgpu_buffer_rsrc_t DA = make_rsrc(da + (int64_t)ti.v0 * CO, (uint64_t)ti.nv * CO * 2);
        const uint32_t off = ok_cur ? (uint32_t)j * (uint32_t)(CO * 2) + (uint32_t)(32 * w + 16 * h) * 2u : OOB;
        st128(DA, off, __builtin_bit_cast(u32x4, pk[0]));
        st128(DA, ok_cur ? off + 16u : OOB, __builtin_bit_cast(u32x4, pk[1]));
      }
      // S of BatchNorm_a through this wavefront's own y_a tile: every reader passed the second barrier, and this
      // wavefront refills the tile before the next first barrier
      bf16_t* tdy = s_tb[w];
      tileN_put_packed(tdy, j, h, pk);
      wave_sync();
      col_sum1(tdy, lane, sa1);
      wave_sync();
      float dyr[16], pr[16];
      unpack8(pk[0], reinterpret_cast<float(&)[8]>(dyr[0]));
      unpack8(pk[1], reinterpret_cast<float(&)[8]>(dyr[8]));
#pragma unroll
      for (int r = 0; r < 16; ++r) pr[r] = dyr[r] * zaf[r];
      bf16x8 pp[2] = {pack8(&pr[0]), pack8(&pr[8])};
      tileN_put_packed(tdy, j, h, pp);
      wave_sync();
      col_sum1(tdy, lane, sa2);
      wave_sync();
    }
  }
  if constexpr (WGRAD) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        atomicAdd(&dWb[(size_t)(32 * w + cperm(chan(r, h))) * CO + 32 * b + cperm(j)], accW[b][r]);
    }
  } else {
    const float a0 = sa1 + other_half(sa1), a1 = sa2 + other_half(sa2);
    if (h == 0) {
      atomicAdd(&stats_a[32 * w + cperm(j)], (double)a0);
      atomicAdd(&stats_a[CO + 32 * w + cperm(j)], (double)a1);
    }
  }
}

}  // namespace emod
}  // namespace dva

using namespace dva;
using namespace dva::chain;
using namespace dva::emod;

extern "C" {

int dva_emod_prep(const float* Wb, int32_t C_out, void* ops, void* stream) {
  if (!Wb || !ops || (C_out != 32 && C_out != 64 && C_out != 128 && C_out != 256)) return DVA_ERR_INVALID;
  const int NB = C_out / 32;
  hipLaunchKernelGGL(emod_prep_kernel, dim3(2 * NB * NB * 2), dim3(64), 0, (hipStream_t)stream, Wb, (int)C_out,
                     (uint4*)ops);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

#define DVA_EMOD_CHECK_SIZES()                                                                            \
  if (n_rows * (int64_t)C_out * 2 > 0xfffffff0ll || n_views * 32 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED

int dva_emod_stats(int32_t layer, const void* Y, const int32_t* tap_rows, const float* tap_weights, const void* tiles,
                   const int32_t* n_tiles, const void* eops, const float* bn_a, double* stats, void* z_a,
                   int64_t n_views, int64_t n_rows, int32_t C_out, void* stream) {
  if (n_views < 0 || (layer != 1 && layer != 2) || (C_out != 32 && C_out != 64 && C_out != 128 && C_out != 256))
    return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!tiles || !n_tiles || !stats) return DVA_ERR_INVALID;
  if (layer == 1 && (!Y || !tap_rows || !tap_weights)) return DVA_ERR_INVALID;
  if (layer == 2 && (!eops || !bn_a || !z_a)) return DVA_ERR_INVALID;
  DVA_EMOD_CHECK_SIZES();
  // (layer 1 at 3 blocks per CU: no gain, the pass is bandwidth-bound; C_out = 256: W_b alone is 128 KB of LDS)
  const dim3 grid(chain_grid(C_out >= 256 && layer == 2 ? 1 : 2)), block(C_out >= 256 && layer == 2 ? 512 : 256);
  hipStream_t s = (hipStream_t)stream;
#define DVA_EMOD_STATS(CO_, L_)                                                                              \
  hipLaunchKernelGGL((emod_stats_kernel<CO_, L_>), grid, block, 0, s, (const bf16_t*)Y, (const int4*)tap_rows, \
                     (const float4*)tap_weights, (const int2*)tiles, n_tiles, (const uint4*)eops, bn_a, stats,  \
                     (bf16_t*)z_a, n_views, n_rows)
  if (C_out == 32 && layer == 1) DVA_EMOD_STATS(32, 1);
  else if (C_out == 32) DVA_EMOD_STATS(32, 2);
  else if (C_out == 64 && layer == 1) DVA_EMOD_STATS(64, 1);
  else if (C_out == 64) DVA_EMOD_STATS(64, 2);
  else if (C_out == 128 && layer == 1) DVA_EMOD_STATS(128, 1);
  else if (C_out == 128) DVA_EMOD_STATS(128, 2);
  else if (layer == 1) DVA_EMOD_STATS(256, 1);
  else DVA_EMOD_STATS(256, 2);
#undef DVA_EMOD_STATS
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_emod_stats1_plan(const void* Y, const int32_t* tap_rows, const float* tap_weights, const int32_t* perm,
                         double* stats, void* z_a, int64_t n_views, int64_t n_rows, int32_t C_out, void* stream) {
  if (n_views < 0 || (C_out != 32 && C_out != 64 && C_out != 128 && C_out != 256)) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!Y || !tap_rows || !tap_weights || !perm || !stats) return DVA_ERR_INVALID;
  if (n_rows * (int64_t)C_out * 2 > 0xfffffff0ll || n_views * 16 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  const dim3 grid(chain_grid(C_out >= 128 ? 3 : 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define DVA_EMOD_S1P(CO_)                                                                                          \
  hipLaunchKernelGGL((emod_stats1_plan_kernel<CO_>), grid, block, 0, s, (const bf16_t*)Y, (const int4*)tap_rows,    \
                     (const float4*)tap_weights, perm, stats, (bf16_t*)z_a, n_views, n_rows)
  if (C_out == 32) DVA_EMOD_S1P(32);
  else if (C_out == 64) DVA_EMOD_S1P(64);
  else if (C_out == 128) DVA_EMOD_S1P(128);
  else DVA_EMOD_S1P(256);
#undef DVA_EMOD_S1P
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_emod_attn_fwd(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                      const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2, const float* bn5,
                      const float* bn6, const float* score_bias, const void* Y, const int32_t* tap_rows,
                      const float* tap_weights, const void* eops, const float* bn_a, const float* bn_b,
                      const int64_t* ptr, const float* gate_w, const float* gate_b, void* out, float* scores_out,
                      const void* z_a, int64_t n_points, int64_t n_views, int64_t n_rows, int32_t C_out, int32_t G,
                      int32_t scaling, float eps, void* stream) {
  if (n_views < 0 || n_points < 0) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!x_map || !view_point || !u || !tiles || !n_tiles || !ops || !bn1 || !bn2 || !bn5 || !bn6 || !score_bias ||
      !eops || !bn_a || !bn_b || !ptr || !out || ((gate_w == nullptr) != (gate_b == nullptr)))
    return DVA_ERR_INVALID;
  if (!z_a && (!Y || !tap_rows || !tap_weights)) return DVA_ERR_INVALID;
  DVA_EMOD_CHECK_SIZES();
  if (n_points * 128 > 0xfffffff0ll || n_points * (int64_t)C_out * 2 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  const dim3 grid(chain_grid(C_out > 64 ? (C_out == 128 ? 2 : 1) : (C_out == 32 && z_a ? (G == 1 ? 3 : 4) : 2))),
      block(C_out >= 256 && z_a ? 512 : 256);
  hipStream_t s = (hipStream_t)stream;
#define DVA_EMOD_FWD_Z(CO_, G_, ZM_)                                                                                  \
  hipLaunchKernelGGL((emod_attn_fwd_kernel<CO_, G_, ZM_>), grid, block, 0, s, x_map, view_point, u,                    \
                     (const int2*)tiles, n_tiles, (const uint4*)ops, bn1, bn2, bn5, bn6, score_bias, (const bf16_t*)Y, \
                     (const int4*)tap_rows, (const float4*)tap_weights, (const uint4*)eops, bn_a, bn_b, ptr,           \
                     gate_w, gate_b, (bf16_t*)out, scores_out, (const bf16_t*)z_a, scaling, eps, n_views, n_points,    \
                     n_rows)
#define DVA_EMOD_FWD(CO_, G_)          \
  do {                                 \
    if (z_a) DVA_EMOD_FWD_Z(CO_, G_, 1); \
    else DVA_EMOD_FWD_Z(CO_, G_, 0);     \
  } while (0)
  switch (C_out * 8 + G) {
    case 32 * 8 + 1: DVA_EMOD_FWD(32, 1); break;
    case 32 * 8 + 2: DVA_EMOD_FWD(32, 2); break;
    case 32 * 8 + 4: DVA_EMOD_FWD(32, 4); break;
    case 64 * 8 + 1: DVA_EMOD_FWD(64, 1); break;
    case 64 * 8 + 2: DVA_EMOD_FWD(64, 2); break;
    case 64 * 8 + 4: DVA_EMOD_FWD(64, 4); break;
    case 128 * 8 + 1: DVA_EMOD_FWD(128, 1); break;
    case 128 * 8 + 2: DVA_EMOD_FWD(128, 2); break;
    case 128 * 8 + 4: DVA_EMOD_FWD(128, 4); break;
    case 256 * 8 + 4: DVA_EMOD_FWD(256, 4); break;
    default: return DVA_ERR_UNSUPPORTED;
  }
#undef DVA_EMOD_FWD
#undef DVA_EMOD_FWD_Z
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_emod_attn_bwd(const float* scores, const int32_t* view_point, const void* tiles, const int32_t* n_tiles,
                      const void* Y, const int32_t* tap_rows, const float* tap_weights, const void* eops,
                      const float* bn_a, const float* bn_b, const int64_t* ptr, const float* gate_w,
                      const float* gate_b, const void* grad_out, const void* out, float* grad_scores, void* view_rec,
                      float* grad_gate_wb, double* stats_b, const void* z_a, int64_t n_points, int64_t n_views,
                      int64_t n_rows, int32_t C_out, int32_t G, int32_t scaling, float eps, void* stream) {
  if (n_views < 0 || n_points < 0) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!scores || !view_point || !tiles || !n_tiles || !z_a || !eops || !bn_a || !bn_b ||
      !ptr || !grad_out || !out || !grad_scores || !view_rec || !stats_b ||
      ((gate_w == nullptr) != (gate_b == nullptr)) || (gate_w && !grad_gate_wb))
    return DVA_ERR_INVALID;
  DVA_EMOD_CHECK_SIZES();
  if (n_points * (int64_t)C_out * 2 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  static const int bpc32 = tune_int("DVA_EMOD_ABWD_BPC", 4);      // read once (getenv), like the other switches
  static const int occ128 = tune_int("DVA_EMOD_ABWD128_OCC", 2);  // C_out = 128: two wavefronts per SIMD (206 - 215 VGPRs)
  const dim3 grid(chain_grid(C_out == 32 ? bpc32 : (C_out == 128 ? occ128 : (C_out == 256 ? 1 : 2)))), block(C_out == 256 ? 512 : 256);
  hipStream_t s = (hipStream_t)stream;
#define DVA_EMOD_BWD(CO_, G_)                                                                                      \
  hipLaunchKernelGGL((emod_attn_bwd_kernel<CO_, G_>), grid, block, 0, s, scores, view_point, (const int2*)tiles,     \
                     n_tiles, (const bf16_t*)Y, (const int4*)tap_rows, (const float4*)tap_weights,                   \
                     (const uint4*)eops, bn_a, bn_b, ptr, gate_w, gate_b, (const bf16_t*)grad_out,                   \
                     (const bf16_t*)out, grad_scores, (uint32_t*)view_rec, grad_gate_wb, stats_b,                     \
                     (const bf16_t*)z_a, scaling, eps, n_views, n_points, n_rows)
  switch (C_out * 8 + G) {
    case 32 * 8 + 1: DVA_EMOD_BWD(32, 1); break;
    case 32 * 8 + 2: DVA_EMOD_BWD(32, 2); break;
    case 32 * 8 + 4: DVA_EMOD_BWD(32, 4); break;
    case 64 * 8 + 1: DVA_EMOD_BWD(64, 1); break;
    case 64 * 8 + 2: DVA_EMOD_BWD(64, 2); break;
    case 64 * 8 + 4: DVA_EMOD_BWD(64, 4); break;
#define DVA_EMOD_BWD_O(CO_, G_, O_)                                                                                \
  hipLaunchKernelGGL((emod_attn_bwd_kernel<CO_, G_, O_>), grid, block, 0, s, scores, view_point, (const int2*)tiles, \
                     n_tiles, (const bf16_t*)Y, (const int4*)tap_rows, (const float4*)tap_weights,                   \
                     (const uint4*)eops, bn_a, bn_b, ptr, gate_w, gate_b, (const bf16_t*)grad_out,                   \
                     (const bf16_t*)out, grad_scores, (uint32_t*)view_rec, grad_gate_wb, stats_b,                     \
                     (const bf16_t*)z_a, scaling, eps, n_views, n_points, n_rows)
#define DVA_EMOD_BWD128(G_)                    \
  do {                                         \
    if (occ128 == 2) DVA_EMOD_BWD_O(128, G_, 2); \
    else DVA_EMOD_BWD_O(128, G_, 1);             \
  } while (0)
    case 128 * 8 + 1: DVA_EMOD_BWD128(1); break;
    case 128 * 8 + 2: DVA_EMOD_BWD128(2); break;
    case 128 * 8 + 4: DVA_EMOD_BWD128(4); break;
    case 256 * 8 + 4: DVA_EMOD_BWD_O(256, 4, 1); break;
#undef DVA_EMOD_BWD128
#undef DVA_EMOD_BWD_O
    default: return DVA_ERR_UNSUPPORTED;
  }
#undef DVA_EMOD_BWD
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_emod_bwd(int32_t stage, const void* Y, const int32_t* tap_rows, const float* tap_weights, const void* tiles,
                 const int32_t* n_tiles, const void* eops, const float* bn_a, const float* bn_b, const float* sm_a,
                 const float* sm_b, const void* view_rec, const void* grad_out, void* da, float* dWb, double* stats_a,
                 const void* z_a, int64_t n_points, int64_t n_views, int64_t n_rows, int32_t C_out, int32_t G,
                 void* stream) {
  if (n_views < 0 || stage < 1 || stage > 4 || (C_out != 32 && C_out != 64 && C_out != 128 && C_out != 256) ||
      (G != 1 && G != 2 && G != 4))
    return DVA_ERR_INVALID;
  if (C_out == 256 && (stage != 2 || G != 4)) return DVA_ERR_UNSUPPORTED;    // (dz_b lives in `da` between its three kernels)
  if (stage == 1 && C_out > 64) return DVA_ERR_UNSUPPORTED;     // (the in-place form; the anchor scatter applies it)
  if (stage > 2 && C_out < 128) return DVA_ERR_UNSUPPORTED;     // the halves of stage 2 exist for wide rows only
  const bool do_dya = stage != 4, do_wgrad = stage != 3;
  if (stage >= 2) {
    if (!eops || !bn_b || !sm_b || !view_rec || !grad_out) return DVA_ERR_INVALID;
    if (do_dya && !stats_a) return DVA_ERR_INVALID;
    if (do_wgrad && !dWb) return DVA_ERR_INVALID;
  }
  if (n_views == 0) return DVA_OK;
  if (!z_a || !tiles || !n_tiles || !bn_a || !da) return DVA_ERR_INVALID;
  if (stage == 1 && !sm_a) return DVA_ERR_INVALID;
  DVA_EMOD_CHECK_SIZES();
  if (n_points * (int64_t)C_out * 2 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  const dim3 block(256);
  hipStream_t s = (hipStream_t)stream;
#define DVA_EMOD_L(CO_, G_, ST_)                                                                                     \
  hipLaunchKernelGGL((emod_bwd_kernel<CO_, G_, ST_>), dim3(chain_grid(2)), block, 0, s, \
                     (const bf16_t*)Y, (const int4*)tap_rows, (const float4*)tap_weights, (const int2*)tiles,         \
                     n_tiles, (const uint4*)eops, bn_a, bn_b, sm_a, sm_b, (const uint32_t*)view_rec,                   \
                     (const bf16_t*)grad_out, (bf16_t*)da, dWb, stats_a, (const bf16_t*)z_a, n_views, n_points,       \
                     n_rows)
  if (stage == 1) {
    if (C_out == 32) DVA_EMOD_L(32, 1, 1);
    else DVA_EMOD_L(64, 1, 1);
  } else if (C_out < 128 && stage != 2) {
    return DVA_ERR_UNSUPPORTED;
  } else {
    switch (C_out * 8 + G) {
      case 32 * 8 + 1: DVA_EMOD_L(32, 1, 2); break;
      case 32 * 8 + 2: DVA_EMOD_L(32, 2, 2); break;
      case 32 * 8 + 4: DVA_EMOD_L(32, 4, 2); break;
      case 64 * 8 + 1: DVA_EMOD_L(64, 1, 2); break;
      case 64 * 8 + 2: DVA_EMOD_L(64, 2, 2); break;
      case 64 * 8 + 4: DVA_EMOD_L(64, 4, 2); break;
      // wide rows: dy_a + S of BatchNorm_a (8 wavefronts per block), then dW_b (one wavefront per SIMD)
#define DVA_EMODW(G_)                                                                                                 \
  do {                                                                                                                \
    if (do_dya)                                                                                                       \
      hipLaunchKernelGGL((emodw_bwd_kernel<128, G_, 1>), dim3(chain_grid(1)), dim3(512), 0, s, (const int2*)tiles,      \
                         n_tiles, (const uint4*)eops, bn_a, bn_b, sm_b, (const uint32_t*)view_rec,                      \
                         (const bf16_t*)grad_out, (bf16_t*)da, dWb, stats_a, (const bf16_t*)z_a, n_views, n_points);    \
    if (do_wgrad)                                                                                                     \
      hipLaunchKernelGGL((emodw_bwd_kernel<128, G_, 2>), dim3(chain_grid(1)), dim3(256), 0, s, (const int2*)tiles,      \
                         n_tiles, (const uint4*)eops, bn_a, bn_b, sm_b, (const uint32_t*)view_rec,                      \
                         (const bf16_t*)grad_out, (bf16_t*)da, dWb, stats_a, (const bf16_t*)z_a, n_views, n_points);    \
  } while (0)
      case 128 * 8 + 1: DVA_EMODW(1); break;
      case 128 * 8 + 2: DVA_EMODW(2); break;
      case 128 * 8 + 4: DVA_EMODW(4); break;
#undef DVA_EMODW
      case 256 * 8 + 4:
        {
          // round 6 (second form, A/B): no dz_b hand-off at all -- two cooperative kernels that each evaluate dz_b themselves
          // (emodw_coop2_kernel).  Parity-green and SLOWER (25.9 against 23.3 ms, profiles/r06_emod_coop2_ab.json): with one
          // 512-thread block per CU nothing overlaps its two barriers per tile.  DVA_EMOD_COOP2=1 selects it; default:
          // MODE 3 + the merged cooperative kernel below
          static const int coop2 = tune_int("DVA_EMOD_COOP2", 0);
          if (coop2) {
            hipLaunchKernelGGL((emodw_coop2_kernel<256, 4, false>), dim3(chain_grid(1)), dim3(512), 0, s, (const int2*)tiles,
                               n_tiles, (const uint4*)eops, bn_a, bn_b, sm_b, (const uint32_t*)view_rec,
                               (const bf16_t*)grad_out, (bf16_t*)da, dWb, stats_a, (const bf16_t*)z_a, n_views, n_points);
            hipLaunchKernelGGL((emodw_coop2_kernel<256, 4, true>), dim3(chain_grid(1)), dim3(512), 0, s, (const int2*)tiles,
                               n_tiles, (const uint4*)eops, bn_a, bn_b, sm_b, (const uint32_t*)view_rec,
                               (const bf16_t*)grad_out, (bf16_t*)da, dWb, stats_a, (const bf16_t*)z_a, n_views, n_points);
            break;
          }
        }
        // W_b (128 KB per orientation) does not fit LDS twice: dz_b -> `da`, dW_b from the stored dz_b, dy_a in place
        hipLaunchKernelGGL((emodw_bwd_kernel<256, 4, 3>), dim3(chain_grid(1)), dim3(512), 0, s, (const int2*)tiles, n_tiles,
                           (const uint4*)eops, bn_a, bn_b, sm_b, (const uint32_t*)view_rec, (const bf16_t*)grad_out,
                           (bf16_t*)da, dWb, stats_a, (const bf16_t*)z_a, n_views, n_points);
        {
          // round 6: dW_b AND dy_a (in place) + S of BatchNorm_a from the stored dz_b in ONE launch (weight-stationary W_b^T
          // slices); DVA_EMOD_COOP_DYA=0: the round-4 pair (emodw_wgrad_coop + emodw_bwd MODE 4), the A/B
          static const int coop_dya = tune_int("DVA_EMOD_COOP_DYA", 1);
          if (coop_dya) {
            hipLaunchKernelGGL((emodw_wgrad_coop_kernel<256, true>), dim3(chain_grid(1)), dim3(512), 0, s,
                               (const int2*)tiles, n_tiles, bn_a, (bf16_t*)da, (const bf16_t*)z_a, dWb, (const uint4*)eops,
                               stats_a);
            break;
          }
        }
        hipLaunchKernelGGL((emodw_wgrad_coop_kernel<256, false>), dim3(chain_grid(1)), dim3(512), 0, s, (const int2*)tiles,
                           n_tiles, bn_a, (bf16_t*)da, (const bf16_t*)z_a, dWb, (const uint4*)eops, stats_a);
        hipLaunchKernelGGL((emodw_bwd_kernel<256, 4, 4>), dim3(chain_grid(1)), dim3(512), 0, s, (const int2*)tiles, n_tiles,
                           (const uint4*)eops, bn_a, bn_b, sm_b, (const uint32_t*)view_rec, (const bf16_t*)grad_out,
                           (bf16_t*)da, dWb, stats_a, (const bf16_t*)z_a, n_views, n_points);
        break;
      default: return DVA_ERR_UNSUPPORTED;
    }
  }
#undef DVA_EMOD_L
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // extern "C"
