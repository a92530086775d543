// The DeepSetFeat chain in fp32: scores and their backward for fp32 features outside torch.autocast -- the reference's
// default arithmetic (models/base_model.py:244 `enabled=is_mixed_precision()`).
//
// Same tile geometry, statistics / hand-over rules and per-point set branch as the bf16 chain of chain_fwd.hip /
// chain_bwd.hip; what differs is the arithmetic: every product runs on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: an exact fp32 fma chain, 64 cycles per instruction and SIMD = the fp32 vector rate),
// BatchNorm + LeakyReLU in fp32 on the accumulators (no BatchNorm folding; leaky' follows the sign of the plain
// pre-activation), and the gradient rows handed between the backward passes are fp32 [V, 32] (128 bytes per view).
//
// Balance.  A 32 x 32 x 32 product costs 16 instructions x 64 cycles per tile, and this instruction does not overlap
// with vector instructions of the SIMD (SQ_VALU_MFMA_COEXEC_CYCLES = 0: a pass costs matrix time + vector time): a pass
// that re-evaluates the chain from x_map ran at a third of the HBM bandwidth (measured: 21.2 ms for the eight passes).
// So two raw layer outputs stay in HBM -- z2 (written by the layer-2 statistics pass) and z5 (written by the layer-5
// statistics pass), fp32 [V, 32] each -- and every later pass starts from one of them: one 128-byte row per view read
// back replaces the 20 .. 52 instructions that would recompute it (15.8 ms, 15.0 with tile-native rows).  The stored-activation kernels of deepset_mfma.hip keep thirteen such
// tensors and are HBM-bound at 24.5 ms.  A fp32-equivalent six-term bf16 product (operands split into three bf16 parts)
// was measured at the same step time and dropped.
//
// Layout.  One wavefront owns a 32-view tile; lane (j, h) = view j, half h.  D[i][j] += sum_{kk<2} A[i][kk] B[kk][j] with
// lane l supplying A[i = l & 31][kk = l >> 5] and B[kk = l >> 5][j = l & 31]; register r of lane (j, h) holds
// D[chan(r, h)][j].  A layer D = W a takes i = output channel, j = view and pairs, in k-step s, the input channels
// (chan(s, 0), chan(s, 1)) -- which is register s of the two half-waves of the previous layer's accumulators (and of
// a stored row loaded in the same order): the B operand of step s IS entry s.  The A operands (weights, one float per
// lane and step) come from an LDS table: per matrix 4 blocks of 64 float4 (steps 4q .. 4q + 3 of lane l at block q).
// Weight gradients dW[n][k] = sum_v dz[v][n] a[v][k] pair the views (s, 16 + s) in step s; both operands come from
// [channel][view] fp32 tiles in LDS (row stride 36 floats: conflict-free scalar writes, ds_read_b128 of four steps).
//   dva_chain3_prep         operand table (26 KiB)
//   dva_chain3_stats2       x_map -> z2 (stored), statistics of layer 2 + per-point extremum (set pooling)
//   dva_chain3_stats        layer 5: z2 -> z5 (stored) + statistics; layer 6: z5 -> statistics of z6
//   dva_chain3_scores       z5 -> scores fp32 [V, 4]  (score layer on the vector units: 4 rows of 32)
//   dva_chain3_score_stats  z5, score gradients -> dWs, dbs, S of layer 6
//   dva_chain3_bwd_layer    stages 6 (from z5), 5 (from z2), 2 (from x_map + z2); hand-offs fp32
//   dva_chain3_set_*        the per-point set branch (chain_set.hip) on the fp32 matrix cores
#include "chain_split.h"

namespace dva {
namespace chain3 {
using namespace dva::chain;

#define F32_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// operand table, in blocks of 64 float4 (1 KiB)
enum {
  Q_W1 = 0,     // 1 block: steps 0..3 = W1[i][4h + s]   (x_map lane (j, h) holds features 4h .. 4h + 3)
  Q_W2 = 1,     // forward, 4 blocks each: step s = W[i][chan(s, h)]
  Q_W5 = 5,
  Q_W6 = 9,
  Q_W6T = 13,   // transposed: step s = W[chan(s, h)][i]
  Q_W5T = 17,
  Q_W2T = 21,
  Q_WSV = 25,   // score layer (forward and backward) for the vector units: entry 2r + h = Ws[0..3][chan(r, h)]
  N_Q = 26
};

__global__ __launch_bounds__(64) void prep3_kernel(const float* __restrict__ W1, const float* __restrict__ W2,
                                                   const float* __restrict__ W5, int ld5,
                                                   const float* __restrict__ W6, const float* __restrict__ Ws, int G,
                                                   float4* __restrict__ ops) {
  const int q = blockIdx.x, lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  float w[4] = {0.f, 0.f, 0.f, 0.f};
  if (q == Q_W1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = W1[i * 8 + 4 * h + e];
  } else if (q == Q_WSV) {
    if (lane < 32) {
      const int r = lane >> 1, hh = lane & 1;
#pragma unroll
      for (int g = 0; g < 4; ++g) w[g] = g < G ? Ws[g * D + chan(r, hh)] : 0.f;
    }
  } else {
    const int mat = (q - Q_W2) / 4, qq = (q - Q_W2) % 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = chan(4 * qq + e, h);
      switch (mat) {
        case 0: w[e] = W2[i * D + c]; break;
        case 1: w[e] = W5[i * ld5 + c]; break;
        case 2: w[e] = W6[i * D + c]; break;
        case 3: w[e] = W6[c * D + i]; break;
        case 4: w[e] = W5[c * ld5 + i]; break;
        default: w[e] = W2[c * D + i]; break;
      }
    }
  }
  ops[q * 64 + lane] = make_float4(w[0], w[1], w[2], w[3]);
}

// copy n blocks of the table to LDS (whole block; __syncthreads() afterwards)
__device__ __forceinline__ void stage_q(float4* s_w, int dst, const float4* __restrict__ ops, int src, int n) {
  for (int i = threadIdx.x; i < n * 64; i += blockDim.x) s_w[dst * 64 + i] = ops[src * 64 + i];
}
// D += W a: 16 steps, the B operand of step s = entry s of `a` (the previous layer's accumulator layout)
template <typename A16>
__device__ __forceinline__ f32x16 mmf(const float4* s_w, int q0, int lane, const A16& a, f32x16 c) {
  asm volatile("" ::: "memory");      // keep the operand reads inside the tile loop
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 w = s_w[(q0 + q) * 64 + lane];
    c = F32_MFMA(w.x, a[4 * q], c);
    c = F32_MFMA(w.y, a[4 * q + 1], c);
    c = F32_MFMA(w.z, a[4 * q + 2], c);
    c = F32_MFMA(w.w, a[4 * q + 3], c);
  }
  return c;
}
// first layer: 8 input features = 4 steps
__device__ __forceinline__ f32x16 mm_x(const float4* s_w, int q0, int lane, const float4& x) {
  asm volatile("" ::: "memory");
  const float4 w = s_w[q0 * 64 + lane];
  f32x16 c = {0};
  c = F32_MFMA(w.x, x.x, c);
  c = F32_MFMA(w.y, x.y, c);
  c = F32_MFMA(w.z, x.z, c);
  c = F32_MFMA(w.w, x.w, c);
  return c;
}
// Score layer + BatchNorm-6 backwards on the vector units, four channels at a time (chain_common.h layer_bwd with
// da6 = Ws^T dc folded in: 4 fma per value from the Q_WSV block, entry 2r + h = Ws[0..3][chan(r, h)]):
//   dy = leaky'(y6) da6;  STATS: st[0] += dy, st[1] += dy z6;  APPLY: dz = G dy - K1 - K2 z6
// keep = ~0 / 0: zeroes dz in the lanes without a view (an AND: the select form of the mask made hipcc spill)
template <bool STATS, bool APPLY>
__device__ __forceinline__ void score_layer_bwd(const f32x16& z, const float4& dc, const float4* wsv, const float* tab,
                                                int h, uint32_t keep, float (&st)[2][16], float (&dz)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    asm volatile("" ::: "memory");
    const int o = 16 * h + 4 * q;
    const float4 g4 = *reinterpret_cast<const float4*>(tab + T_G * D + o);
    const float4 b4 = *reinterpret_cast<const float4*>(tab + T_B * D + o);
    const float g[4] = {g4.x, g4.y, g4.z, g4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
    float k1[4] = {0.f, 0.f, 0.f, 0.f}, k2[4] = {0.f, 0.f, 0.f, 0.f};
    if (APPLY) {
      const float4 a4 = *reinterpret_cast<const float4*>(tab + T_K1 * D + o);
      const float4 c4 = *reinterpret_cast<const float4*>(tab + T_K2 * D + o);
      k1[0] = a4.x; k1[1] = a4.y; k1[2] = a4.z; k1[3] = a4.w;
      k2[0] = c4.x; k2[1] = c4.y; k2[2] = c4.z; k2[3] = c4.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * q + e;
      const float4 w = wsv[2 * r + h];
      const float da = __builtin_fmaf(w.w, dc.w, __builtin_fmaf(w.z, dc.z, __builtin_fmaf(w.y, dc.y, w.x * dc.x)));
      const float y = __builtin_fmaf(z[r], g[e], b[e]);
      const float dy = y > 0.f ? da : SLOPE * da;
      if (STATS) {
        st[0][r] += dy;
        st[1][r] = __builtin_fmaf(dy, z[r], st[1][r]);
      }
      if (APPLY)
        dz[r] = __uint_as_float(__float_as_uint(__builtin_fmaf(-k2[e], z[r], __builtin_fmaf(g[e], dy, -k1[e]))) & keep);
    }
  }
}
// BatchNorm + LeakyReLU in fp32: a = leaky(z G + B)
template <typename Z16>
__device__ __forceinline__ void act(const Z16& z, const float* tab, int h, float (&a)[16]) {
  asm volatile("" ::: "memory");
  float g[16], b[16];
  tab16(tab, T_G, h, g);
  tab16(tab, T_B, h, b);
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = leaky(__builtin_fmaf(z[r], g[r], b[r]));
}
// [channel][view] fp32 tile, row stride TS: lane (j, h) = view j writes its 16 channels as 16 ds_write_b32 (bank =
// 4 chan + j: no conflicts), lane (n, h) reads views 16h .. 16h + 15 of channel n as 4 ds_read_b128
constexpr int TS = 36;
template <typename A16>
__device__ __forceinline__ void tile_put(float* tile, int j, int h, const A16& x, bool ok) {
#pragma unroll
  for (int r = 0; r < 16; ++r) tile[chan(r, h) * TS + j] = ok ? x[r] : 0.f;
}
// acc[r] += sum_v A[chan(r, h)][v] B[j][v]: k-step s pairs the views (s, 16 + s); four steps per pair of LDS reads
__device__ __forceinline__ f32x16 wgradf(const float* ta, const float* tb, int j, int h, f32x16 acc) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    asm volatile("" ::: "memory");      // one chunk of operands in flight: 8 registers, not 32
    const float4 a = *reinterpret_cast<const float4*>(ta + j * TS + 16 * h + 4 * c);
    const float4 b = *reinterpret_cast<const float4*>(tb + j * TS + 16 * h + 4 * c);
    acc = F32_MFMA(a.x, b.x, acc);
    acc = F32_MFMA(a.y, b.y, acc);
    acc = F32_MFMA(a.z, b.z, acc);
    acc = F32_MFMA(a.w, b.w, acc);
  }
  return acc;
}
// the same with a short second tile: its rows >= jb_max are one shared zero row (row jb_max)
__device__ __forceinline__ f32x16 wgradf_short(const float* ta, const float* tb, int j, int jb_max, int h,
                                                f32x16 acc) {
  const int jb = j < jb_max ? j : jb_max;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    asm volatile("" ::: "memory");
    const float4 a = *reinterpret_cast<const float4*>(ta + j * TS + 16 * h + 4 * c);
    const float4 b = *reinterpret_cast<const float4*>(tb + jb * TS + 16 * h + 4 * c);
    acc = F32_MFMA(a.x, b.x, acc);
    acc = F32_MFMA(a.y, b.y, acc);
    acc = F32_MFMA(a.z, b.z, acc);
    acc = F32_MFMA(a.w, b.w, acc);
  }
  return acc;
}
__device__ __forceinline__ f32x16 load_u(__amdgpu_buffer_rsrc_t U, bool ok, int vpj, int h) {
  f32x16 u;
  load_rows16(U, ok, (uint32_t)vpj, h, u);
  return u;
}

template <typename Z16>
__device__ __forceinline__ void add_stats(const Z16& z, bool ok, float (&st)[2][16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float zz = ok ? z[r] : 0.f;
    st[0][r] += zz;
    st[1][r] = __builtin_fmaf(zz, zz, st[1][r]);
  }
}

// a [V][32] fp32 row tensor reaches 4 GiB at V = 2^25: one buffer descriptor per tile (rows of the tile only).
// The rows of a tile are stored in the order the lanes hold them ("tile-native": element (view j, half h, quad q) at
// byte q (32 nv) + 32 j + 16 h of the tile's nv x 128 bytes), so that every load / store instruction of a wavefront
// covers one contiguous run of 32 nv bytes instead of 32 separate 32-byte pieces.  Producers and consumers of these
// tensors (z2, z5, dy5, dy2) are the kernels of this file, all walking the same tile table.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const float* base, const TileInfo& ti) {
  return make_rsrc(base ? base + (int64_t)ti.v0 * D : nullptr, base ? (uint64_t)ti.nv * 128 : 0);
}
__device__ __forceinline__ f32x16 load_tile_rows(const float* base, const TileInfo& ti, int j, int h) {
  const __amdgpu_buffer_rsrc_t R = tile_rsrc(base, ti);
  const bool ok = j < ti.nv;
  const uint32_t o = (uint32_t)j * 32u + 16u * h, qs = (uint32_t)ti.nv * 32u;
  f32x16 r;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = as_f4(ld128(R, ok ? o + q * qs : OOB));
    r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
  }
  return r;
}
template <typename A16>
__device__ __forceinline__ void store_tile_rows(float* base, const TileInfo& ti, int j, int h, const A16& x) {
  const __amdgpu_buffer_rsrc_t R = tile_rsrc(base, ti);
  const bool ok = j < ti.nv;
  const uint32_t o = (uint32_t)j * 32u + 16u * h, qs = (uint32_t)ti.nv * 32u;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    st128(R, ok ? o + q * qs : OOB, as_u4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]));
}

// ------------------------------------------------------------------------------------------------
// layer 2: statistics + per-point extremum of sign(gamma2) z2 (chain_fwd.hip stats2_kernel in fp32); z2 stays in HBM
// ------------------------------------------------------------------------------------------------
struct PreX {
  TileInfo ti;
  float4 x;
  int vpj;
};
__global__ __launch_bounds__(256, 4) void stats2_kernel(
    const float* __restrict__ x_map, const int32_t* __restrict__ vp, const int2* __restrict__ tiles,
    const int32_t* __restrict__ n_tiles_dev, const float4* __restrict__ ops, const float* __restrict__ bn1,
    const float* __restrict__ gamma2, double* __restrict__ stats, float* __restrict__ zstar,
    int32_t* __restrict__ arg, float* __restrict__ z2_out, int64_t V) {
  __shared__ __attribute__((aligned(16))) float s_tab[1][TAB_FLOATS];
  __shared__ __attribute__((aligned(16))) float4 s_w[Q_W5 * 64];
  __shared__ __attribute__((aligned(16))) float s_tile[4][32 * TS];
  __shared__ float s_red[STATS_RED_FLOATS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  stage_q(s_w, 0, ops, 0, Q_W5);
  stage_tab(s_tab[0], bn1, nullptr, false);
  __syncthreads();
  const uint32_t flip = gamma2[j] < 0.f ? 0x80000000u : 0u;
  const __amdgpu_buffer_rsrc_t X = make_rsrc(x_map, (uint64_t)V * 32), P = make_rsrc(vp, (uint64_t)V * 4);
  float st[2][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) st[0][r] = st[1][r] = 0.f;
  float run_m = -INFINITY;
  int run_a = -1;
  float* tz = s_tile[wv];
  const int n_tiles = n_tiles_dev[0];
  int ta, tb;
  wave_tile_range(tiles, n_tiles, ta, tb);
  run_tiles<PreX>(tiles, ta, tb, [&](const TileInfo& ti, int t) {
    PreX p;
    p.ti = ti;
    const bool ok = j < p.ti.nv;
    p.x = as_f4(ld128(X, ok ? (uint32_t)(p.ti.v0 + j) * 32u + 16u * h : OOB));
    p.vpj = (int)ld32(P, ok ? (uint32_t)(p.ti.v0 + j) * 4u : OOB);
    return p;
  }, [&](const PreX& p) {
    const int nv = p.ti.nv;
    const bool ok = j < nv;
    const f32x16 zero = {0};
    float a1[16];
    act(mm_x(s_w, Q_W1, lane, p.x), s_tab[0], h, a1);
    const f32x16 z2 = mmf(s_w, Q_W2, lane, a1, zero);
    add_stats(z2, ok, st);
    store_tile_rows(z2_out, p.ti, j, h, z2);
    tile_put(tz, j, h, z2, true);
    const int nxt = shfl(p.vpj, lane + 1);
    const bool is_end = ok && (j == nv - 1 || nxt != p.vpj);
    uint32_t endmask = (uint32_t)__ballot(is_end);
    if (p.ti.frag == 1 || p.ti.frag == 2) endmask = 0;
    wave_sync();
    float xv[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 t4 = *reinterpret_cast<const float4*>(tz + j * TS + 4 * q);      // channel j, views 4q .. 4q + 3
      xv[4 * q] = __uint_as_float(__float_as_uint(t4.x) ^ flip);
      xv[4 * q + 1] = __uint_as_float(__float_as_uint(t4.y) ^ flip);
      xv[4 * q + 2] = __uint_as_float(__float_as_uint(t4.z) ^ flip);
      xv[4 * q + 3] = __uint_as_float(__float_as_uint(t4.w) ^ flip);
    }
#pragma unroll
    for (int v = 0; v < 32; ++v) {
      if (v < nv) {                      // uniform
        const bool gt = xv[v] > run_m;   // strict: the first extremal view wins (torch_scatter arg semantics)
        run_m = gt ? xv[v] : run_m;
        run_a = gt ? p.ti.v0 + v : run_a;
        if ((endmask >> v) & 1u) {       // uniform
          const int pt = __builtin_amdgcn_readlane(p.vpj, v);
          if (h == 0) {
            zstar[(int64_t)pt * D + j] = __uint_as_float(__float_as_uint(run_m) ^ flip);
            arg[(int64_t)pt * D + j] = run_a;
          }
          run_m = -INFINITY;
          run_a = -1;
        }
      }
    }
    wave_sync();
  });
  flush_stats<2>(st, stats, s_red);
}

// rows of one stored layer output (z2 or z5) per view, prefetched one tile ahead
struct PreR {
  TileInfo ti;
  f32x16 row;
  float4 dc;
  int vpj;
};

// L = 5: z2 -> a2 -> z5 = W5a a2 + u[point]: statistics of layer 5, z5 stays in HBM
// L = 6: z5 -> a5 -> z6 = W6 a5: statistics of layer 6
template <int L>
__global__ __launch_bounds__(256, L == 5 ? 3 : 4) void stats_mid_kernel(
    const float* __restrict__ rows_in, const int32_t* __restrict__ vp, const float* __restrict__ u,
    const int2* __restrict__ tiles, const int32_t* __restrict__ n_tiles_dev, const float4* __restrict__ ops,
    const float* __restrict__ bn, float* __restrict__ rows_out, double* __restrict__ stats, int64_t V, int64_t N) {
  __shared__ __attribute__((aligned(16))) float s_tab[1][TAB_FLOATS];
  __shared__ __attribute__((aligned(16))) float4 s_w[4 * 64];
  __shared__ float s_red[STATS_RED_FLOATS];
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  stage_q(s_w, 0, ops, L == 5 ? Q_W5 : Q_W6, 4);
  stage_tab(s_tab[0], bn, nullptr, false);
  __syncthreads();
  const __amdgpu_buffer_rsrc_t P = make_rsrc(vp, (uint64_t)V * 4), U = make_rsrc(u, (uint64_t)N * 128);
  float st[2][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) st[0][r] = st[1][r] = 0.f;
  const int n_tiles = n_tiles_dev[0];
  int ta, tb;
  wave_tile_range(tiles, n_tiles, ta, tb);
  run_tiles<PreR>(tiles, ta, tb, [&](const TileInfo& ti, int t) {
    PreR p;
    p.ti = ti;
    p.row = load_tile_rows(rows_in, ti, j, h);
    if (L == 5) p.vpj = (int)ld32(P, j < ti.nv ? (uint32_t)(ti.v0 + j) * 4u : OOB);
    return p;
  }, [&](const PreR& p) {
    const bool ok = j < p.ti.nv;
    // (the per-point row is a dependent load: it is added after the product, which therefore does not wait for it;
    //  stage 5 of the backward evaluates z5 in the same order)
    f32x16 uacc = {0};
    if (L == 5) uacc = load_u(U, ok, p.vpj, h);
    float a[16];
    act(p.row, s_tab[0], h, a);
    const f32x16 zero = {0};
    f32x16 z = mmf(s_w, 0, lane, a, zero);
    if (L == 5) z += uacc;
    add_stats(z, ok, st);
    if (L == 5) store_tile_rows(rows_out, p.ti, j, h, z);
  });
  flush_stats<2>(st, stats, s_red);
}

// z5 -> scores [V][4] (columns >= G zero): layers 5 (activation), 6, then the score layer on the vector units
__global__ __launch_bounds__(256, 4) void scores_kernel(
    const float* __restrict__ z5, const int2* __restrict__ tiles, const int32_t* __restrict__ n_tiles_dev,
    const float4* __restrict__ ops, const float* __restrict__ bn5, const float* __restrict__ bn6,
    const float* __restrict__ bs, int G, float* __restrict__ scores, int64_t V) {
  __shared__ __attribute__((aligned(16))) float s_tab[2][TAB_FLOATS];
  __shared__ __attribute__((aligned(16))) float4 s_w[5 * 64];
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  stage_q(s_w, 0, ops, Q_W6, 4);
  stage_q(s_w, 4, ops, Q_WSV, 1);
  stage_tab(s_tab[0], bn5, nullptr, false);
  stage_tab(s_tab[1], bn6, nullptr, false);
  __syncthreads();
  const __amdgpu_buffer_rsrc_t SC = make_rsrc(scores, (uint64_t)V * 16);
  float bias[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bias[g] = g < G ? bs[g] : 0.f;
  const float4* wsv = s_w + 4 * 64;
  const int n_tiles = n_tiles_dev[0];
  int ta, tb;
  wave_tile_range(tiles, n_tiles, ta, tb);
  run_tiles<PreR>(tiles, ta, tb, [&](const TileInfo& ti, int t) {
    PreR p;
    p.ti = ti;
    p.row = load_tile_rows(z5, ti, j, h);
    return p;
  }, [&](const PreR& p) {
    const bool ok = j < p.ti.nv;
    const f32x16 zero = {0};
    float a5[16], a6[16];
    act(p.row, s_tab[0], h, a5);
    act(mmf(s_w, 0, lane, a5, zero), s_tab[1], h, a6);
    asm volatile("" ::: "memory");
    float sc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 w = wsv[2 * r + h];
      sc[0] = __builtin_fmaf(a6[r], w.x, sc[0]);
      sc[1] = __builtin_fmaf(a6[r], w.y, sc[1]);
      sc[2] = __builtin_fmaf(a6[r], w.z, sc[2]);
      sc[3] = __builtin_fmaf(a6[r], w.w, sc[3]);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) sc[g] += __shfl_xor(sc[g], 32);
    st128(SC, ok && h == 0 ? (uint32_t)(p.ti.v0 + j) * 16u : OOB,
          as_u4(sc[0] + bias[0], sc[1] + bias[1], sc[2] + bias[2], sc[3] + bias[3]));
  });
}

// score layer backward + statistics of the BatchNorm-6 backward (from z5)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void score_stats_kernel(
    const float* __restrict__ z5, const int2* __restrict__ tiles, const int32_t* __restrict__ n_tiles_dev,
    const float4* __restrict__ ops, const float* __restrict__ bn5, const float* __restrict__ bn6,
    const float* __restrict__ dc, double* __restrict__ stats6, float* __restrict__ dWs, float* __restrict__ dbs,
    int G, int64_t V) {
  constexpr int L_WSV = 4;
  __shared__ __attribute__((aligned(16))) float s_tab[2][TAB_FLOATS];
  __shared__ __attribute__((aligned(16))) float4 s_w[5 * 64];
  __shared__ __attribute__((aligned(16))) float s_ta[4][32 * TS], s_td[4][5 * TS];      // a6 | dc (4 rows + a zero row)
  float* s_red = &s_ta[0][0];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < 4 * 5 * TS; i += blockDim.x) (&s_td[0][0])[i] = 0.f;
  stage_q(s_w, 0, ops, Q_W6, 4);
  stage_q(s_w, L_WSV, ops, Q_WSV, 1);
  stage_tab(s_tab[0], bn5, nullptr, false);
  stage_tab(s_tab[1], bn6, nullptr, false);
  __syncthreads();
  const __amdgpu_buffer_rsrc_t DC = make_rsrc(dc, (uint64_t)V * 16);
  float* ta_ = s_ta[wv];
  float* td = s_td[wv];
  f32x16 accS = {0};
  float dbsum[4] = {0.f, 0.f, 0.f, 0.f};
  float st[2][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) st[0][r] = st[1][r] = 0.f;
  const int n_tiles = n_tiles_dev[0];
  int ta, tb;
  wave_tile_range(tiles, n_tiles, ta, tb);
  run_tiles<PreR>(tiles, ta, tb, [&](const TileInfo& ti, int t) {
    PreR p;
    p.ti = ti;
    p.row = load_tile_rows(z5, ti, j, h);
    p.dc = as_f4(ld128(DC, j < ti.nv ? (uint32_t)(ti.v0 + j) * 16u : OOB));      // zeros in the lanes without a view
    return p;
  }, [&](const PreR& p) {
    const bool ok = j < p.ti.nv;
    const f32x16 zero = {0};
    float a5[16], a6[16];
    act(p.row, s_tab[0], h, a5);
    const f32x16 z6 = mmf(s_w, 0, lane, a5, zero);
    act(z6, s_tab[1], h, a6);
    tile_put(ta_, j, h, a6, ok);
    if (h == 0) {
      td[0 * TS + j] = p.dc.x; td[1 * TS + j] = p.dc.y; td[2 * TS + j] = p.dc.z; td[3 * TS + j] = p.dc.w;
      dbsum[0] += p.dc.x; dbsum[1] += p.dc.y; dbsum[2] += p.dc.z; dbsum[3] += p.dc.w;
    }
    float unused[16];
    score_layer_bwd<true, false>(z6, p.dc, s_w + L_WSV * 64, s_tab[1], h, 0xffffffffu, st, unused);   // dc = 0 in the lanes without a view
    wave_sync();
    accS = wgradf_short(ta_, td, j, 4, h, accS);      // dWs^T[c][g] = sum_v a6[v][c] dc[v][g]
    wave_sync();
  });
  flush_matrix(accS, dWs, D, G, true, s_red);
  flush_stats<2>(st, stats6, s_red);
  __syncthreads();       // dbs: one atomic per block and group (chain_common.h flush_red)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float v = dbsum[g];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off);
    if (lane == 0) s_red[wv * 4 + g] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < G) atomicAdd(&dbs[threadIdx.x], (s_red[threadIdx.x] + s_red[4 + threadIdx.x]) +
                                                           (s_red[8 + threadIdx.x] + s_red[12 + threadIdx.x]));
}

// ------------------------------------------------------------------------------------------------
// layer passes (chain_bwd.hip layer_bwd_kernel in fp32; hand-offs fp32 [V][32]).  Every pass starts from the stored
// output of its own layer's input side (stage 6: z5, stage 5: z2, stage 2: x_map + z2) instead of re-evaluating
// the chain from x_map: these kernels are bound by the matrix pipe while HBM idles, so one 128-byte row per view read
// back is cheaper than the 16 .. 36 instructions that recompute it.
//   stage 6: z5, dc -> dz6 -> dW6, S5; hands dy5.   stage 5: z2, dy5 -> dz5 -> dW5, du, S2 (view part); hands dy2.
//   stage 2: x_map, z2, dy2 + the routed set-pooling gradient -> dz2 -> dW2, P.
// ------------------------------------------------------------------------------------------------
struct PreB {
  TileInfo ti;
  f32x16 row, din;
  float4 x;       // stage 2: x_map; stage 6: dc
  int vpj;
};
// Two wavefronts per SIMD with the row tensors of the next tile prefetched (two register sets).  Measured alternative:
// rows loaded at the start of the tile's own body, 168 registers, three wavefronts per SIMD -- the same time (stage 6:
// 2.72 vs 2.69 ms), the passes are bound by the dependent VALU / LDS latency between the products, not by occupancy.
template <int STAGE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void layer_bwd_kernel(
    const float* __restrict__ x_map, const float* __restrict__ zrows, const int32_t* __restrict__ vp,
    const float* __restrict__ u, const int2* __restrict__ tiles, const int32_t* __restrict__ n_tiles_dev,
    const float4* __restrict__ ops, const float* __restrict__ bn_lo, const float* __restrict__ bn_hi,
    const float* __restrict__ sm, const float* __restrict__ dc, const int32_t* __restrict__ arg,
    const float* __restrict__ dpooled, const float* __restrict__ da_in, float* __restrict__ da_out,
    float* __restrict__ dW, float* __restrict__ du, float* __restrict__ Pm, double* __restrict__ stats, int64_t V,
    int64_t N) {
  // bn_lo / bn_hi: the BatchNorm tables of the lower / upper layer of the pass (6: bn5, bn6; 5: bn2, bn5; 2: bn1, bn2);
  // sm = S / M of the upper layer.  local operand table:
  //   stage 6: W6 at 0, W6T at 4, WSV at 8;   stage 5: W5 at 0, W5T at 4;   stage 2: W1 at 0, W2T at 1
  constexpr int NQ = STAGE == 6 ? 9 : (STAGE == 5 ? 8 : 5);
  constexpr int TXR = 18;       // stage 2: rows of the [x (8) | 0 (8) | 1 | 0][view] tile (the P layout of dva_chain_dw1)
  __shared__ __attribute__((aligned(16))) float s_tab[2][TAB_FLOATS];
  __shared__ __attribute__((aligned(16))) float4 s_w[NQ * 64];
  __shared__ __attribute__((aligned(16))) float s_ta[4][32 * TS], s_tb[4][32 * TS];
  __shared__ __attribute__((aligned(16))) float s_tc[STAGE == 2 ? 4 : 1][STAGE == 2 ? TXR * TS : 4];
  float* s_red = &s_ta[0][0];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  if (STAGE == 6) {
    stage_q(s_w, 0, ops, Q_W6, 8);            // W6, W6T are contiguous in the global table
    stage_q(s_w, 8, ops, Q_WSV, 1);
  } else if (STAGE == 5) {
    stage_q(s_w, 0, ops, Q_W5, 4);
    stage_q(s_w, 4, ops, Q_W5T, 4);
  } else {
    stage_q(s_w, 0, ops, Q_W1, 1);
    stage_q(s_w, 1, ops, Q_W2T, 4);
  }
  stage_tab(s_tab[0], bn_lo, nullptr, false);
  stage_tab(s_tab[1], bn_hi, sm, false);
  if (STAGE == 2) {
    for (int i = threadIdx.x; i < 4 * TXR * TS; i += blockDim.x) (&s_tc[0][0])[i] = 0.f;
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t X = make_rsrc(x_map, (uint64_t)V * 32), P = make_rsrc(vp, (uint64_t)V * 4),
                               U = make_rsrc(u, (uint64_t)N * 128), DC = make_rsrc(dc, (uint64_t)V * 16),
                               AR = make_rsrc(arg, (uint64_t)N * 128), DP = make_rsrc(dpooled, (uint64_t)N * 128);
  float st[2][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) st[0][r] = st[1][r] = 0.f;
  f32x16 accW = {0}, accS = {0};
  float* ta_ = s_ta[wv];
  float* tb_ = s_tb[wv];
  float* tc = s_tc[STAGE == 2 ? wv : 0];
  const int n_tiles = n_tiles_dev[0];
  int t0, t1;
  wave_tile_range(tiles, n_tiles, t0, t1);
  run_tiles<PreB>(tiles, t0, t1, [&](const TileInfo& ti, int t) {
    PreB p;
    p.ti = ti;
    const bool ok = j < ti.nv;
    const uint32_t view = (uint32_t)(ti.v0 + j);
    p.row = load_tile_rows(zrows, ti, j, h);
    if (STAGE != 6) p.din = load_tile_rows(da_in, ti, j, h);
    if (STAGE == 6) p.x = as_f4(ld128(DC, ok ? view * 16u : OOB));
    if (STAGE == 2) p.x = as_f4(ld128(X, ok ? view * 32u + 16u * h : OOB));
    if (STAGE != 6) p.vpj = (int)ld32(P, ok ? view * 4u : OOB);
    return p;
  }, [&](const PreB& p) {
    const int nv = p.ti.nv;
    const bool ok = j < nv;
    const f32x16 zero = {0};
    float dz[16];
    if constexpr (STAGE == 6) {
      // p.row = z5
      float a5[16];
      act(p.row, s_tab[0], h, a5);
      tile_put(tb_, j, h, a5, ok);
      {
        const f32x16 z6 = mmf(s_w, 0, lane, a5, zero);
        float unused[2][16];
        score_layer_bwd<false, true>(z6, p.x, s_w + 8 * 64, s_tab[1], h, ok ? 0xffffffffu : 0u, unused, dz);
      }
      tile_put(ta_, j, h, dz, true);
      const f32x16 da5 = mmf(s_w, 4, lane, dz, zero);
      float dy5[16];
      layer_bwd<true, false>(p.row, da5, s_tab[0], h, ok, st, dy5);
      store_tile_rows(da_out, p.ti, j, h, dy5);
      wave_sync();
      accW = wgradf(ta_, tb_, j, h, accW);      // dW6[n][k] = sum_v dz6[v][n] a5[v][k]
      wave_sync();
    } else if constexpr (STAGE == 5) {
      // p.row = z2, p.din = dy5
      const f32x16 uacc = load_u(U, ok, p.vpj, h);
      float a2[16];
      act(p.row, s_tab[0], h, a2);
      tile_put(tb_, j, h, a2, ok);
      {
        f32x16 z5 = mmf(s_w, 0, lane, a2, zero);
        z5 += uacc;
        bn_bwd_apply(z5, p.din, s_tab[1], h, dz);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) dz[r] = ok ? dz[r] : 0.f;
      tile_put(ta_, j, h, dz, true);
      const f32x16 da2 = mmf(s_w, 4, lane, dz, zero);
      float dy2[16];
      layer_bwd<true, false>(p.row, da2, s_tab[0], h, ok, st, dy2);
      store_tile_rows(da_out, p.ti, j, h, dy2);
      // du[p][c] = sum of dz5 over the views of point p: segmented scan over the lanes of each half-wave, the last view
      // of a point stores its 16 channels (a point in several tiles: its fragments add up in the caller-zeroed row)
      {
        const SegInfo sg = seg_setup(p.vpj, j, lane, nv);
        const SegMaskF sm_ = seg_mask_f(sg);
        float tot[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[r] = seg_scan_sum_f(dz[r], sm_);
        const bool wr = ok && ((sg.emask >> j) & 1u);
        if (p.ti.frag == 0) {
          const __amdgpu_buffer_rsrc_t DU = make_rsrc(du, (uint64_t)N * 128);
          store_rows16(DU, wr, (uint32_t)p.vpj, h, tot);
        } else if (wr) {
#pragma unroll
          for (int r = 0; r < 16; ++r) atomicAdd(&du[(size_t)p.vpj * D + chan(r, h)], tot[r]);
        }
      }
      wave_sync();
      accW = wgradf(ta_, tb_, j, h, accW);      // dW5a[n][k] = sum_v dz5[v][n] a2[v][k]
      wave_sync();
    } else {
      // p.row = z2, p.din = dy2 (view path), p.x = x_map
      u32x4 arq[4], dpq[4];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const uint32_t off = ok ? (uint32_t)p.vpj * 128u + (8u * qq + 4u * h) * 4u : OOB;
        arq[qq] = ld128(AR, off);
        dpq[qq] = ld128(DP, off);
      }
      const f32x16 z1 = mm_x(s_w, 0, lane, p.x);
      float a1[16];
      act(z1, s_tab[0], h, a1);
      tile_put(tb_, j, h, a1, ok);
      f32x16 dy = p.din;
      {
        const int vg = p.ti.v0 + j;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const uint32_t ai[4] = {arq[qq].x, arq[qq].y, arq[qq].z, arq[qq].w};
          const uint32_t di[4] = {dpq[qq].x, dpq[qq].y, dpq[qq].z, dpq[qq].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) dy[4 * qq + e] += (int)ai[e] == vg ? __uint_as_float(di[e]) : 0.f;
        }
      }
      bn_bwd_apply(p.row, dy, s_tab[1], h, dz);
#pragma unroll
      for (int r = 0; r < 16; ++r) dz[r] = ok ? dz[r] : 0.f;
      tile_put(ta_, j, h, dz, true);
      const f32x16 da1 = mmf(s_w, 1, lane, dz, zero);
      float dy1[16];
      {
        asm volatile("" ::: "memory");
        float g_[16], b_[16];
        tab16(s_tab[0], T_G, h, g_);
        tab16(s_tab[0], T_B, h, b_);
#pragma unroll
        for (int r = 0; r < 16; ++r) dy1[r] = da1[r] * dleaky(__builtin_fmaf(z1[r], g_[r], b_[r]));
      }
      wave_sync();
      accW = wgradf(ta_, tb_, j, h, accW);       // dW2[n][k] = sum_v dz2[v][n] a1[v][k]
      wave_sync();
      // P[n][f] = sum_v dy1[v][n] [x | 0 | 1][v][f]: the first-layer weight gradient and the statistics of layer 1
      tile_put(ta_, j, h, dy1, true);            // (lanes without a view: da1 = 0)
      tc[(4 * h + 0) * TS + j] = p.x.x;      // lanes without a view: zeros
      tc[(4 * h + 1) * TS + j] = p.x.y;
      tc[(4 * h + 2) * TS + j] = p.x.z;
      tc[(4 * h + 3) * TS + j] = p.x.w;
      if (h == 0) tc[16 * TS + j] = ok ? 1.f : 0.f;
      wave_sync();
      accS = wgradf_short(ta_, tc, j, 17, h, accS);
      wave_sync();
    }
  });
  flush_matrix(accW, dW, STAGE == 5 ? 2 * D : D, D, false, s_red);
  if (STAGE == 2) flush_matrix(accS, Pm, 20, 17, false, s_red);
  else flush_stats<2>(st, stats, s_red);
}

// ------------------------------------------------------------------------------------------------
// per-point set branch in fp32 (chain_set.hip set_kernel on the fp32 matrix cores): pooled [N, 32] (+ the set-size
// feature) -> mlp_set -> u = Wc[:, 32:] . s.  lane (j, h) = point j of a 32-point tile.
// ------------------------------------------------------------------------------------------------
enum { SQ_WSA = 0, SQ_WSB = 4, SQ_WCB = 8, SQ_WCBT = 12, SQ_WSBT = 16, SQ_WSAT = 20, N_SQ = 24 };

__global__ __launch_bounds__(64) void set3_prep_kernel(const float* __restrict__ Wsa, int ldsa,
                                                       const float* __restrict__ Wsb, const float* __restrict__ Wc,
                                                       int ldc, float4* __restrict__ ops) {
  const int q = blockIdx.x, lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  const int mat = q / 4, qq = q % 4;
  float w[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = chan(4 * qq + e, h);
    switch (mat) {
      case 0: w[e] = Wsa[i * ldsa + c]; break;             // forward: W[out = i][in = c]
      case 1: w[e] = Wsb[i * D + c]; break;
      case 2: w[e] = Wc[i * ldc + D + c]; break;
      case 3: w[e] = Wc[c * ldc + D + i]; break;           // transposed: W[out = c][in = i]
      case 4: w[e] = Wsb[c * D + i]; break;
      default: w[e] = Wsa[c * ldsa + i]; break;
    }
  }
  ops[q * 64 + lane] = make_float4(w[0], w[1], w[2], w[3]);
}

// w33 [32]: the set-size column of Wsa (use_num), nullptr otherwise.  DIR = 0 forward, 1 backward.
template <int DIR, int STAGE>
__global__ __launch_bounds__(256, 2) void set3_kernel(
    const float* __restrict__ pooled, const int64_t* __restrict__ ptr, const float* __restrict__ w33,
    const float4* __restrict__ ops, const float* __restrict__ bn_s1, const float* __restrict__ bn_s2,
    const float* __restrict__ sm_s1, const float* __restrict__ sm_s2, const float* __restrict__ du,
    float* __restrict__ u_out, float* __restrict__ dpooled, float* __restrict__ dW, int ld_dw,
    float* __restrict__ dw33, double* __restrict__ stats, int64_t N) {
  __shared__ __attribute__((aligned(16))) float s_tab[2][TAB_FLOATS];
  __shared__ __attribute__((aligned(16))) float s_w33[D];
  __shared__ __attribute__((aligned(16))) float4 s_w[N_SQ * 64];
  __shared__ __attribute__((aligned(16))) float s_ta[DIR == 1 ? 4 : 1][DIR == 1 ? 32 * TS : 4],
      s_tb[DIR == 1 ? 4 : 1][DIR == 1 ? 32 * TS : 4];
  __shared__ float s_red[D * D];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  stage_q(s_w, 0, ops, 0, N_SQ);
  stage_tab(s_tab[0], bn_s1, (DIR == 1 && STAGE == 3) ? sm_s1 : nullptr, false);
  stage_tab(s_tab[1], bn_s2, (DIR == 1 && STAGE >= 2) ? sm_s2 : nullptr, false);
  for (int i = threadIdx.x; i < D; i += blockDim.x) s_w33[i] = w33 ? w33[chan(i & 15, i >> 4)] : 0.f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t PL = make_rsrc(pooled, (uint64_t)N * 128), DU = make_rsrc(du, (uint64_t)N * 128),
                               UO = make_rsrc(u_out, (uint64_t)N * 128), DPO = make_rsrc(dpooled, (uint64_t)N * 128);
  float st[2][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) st[0][r] = st[1][r] = 0.f;
  f32x16 accW = {0};
  float acc33[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc33[r] = 0.f;
  float* ta_ = s_ta[DIR == 1 ? wv : 0];
  float* tb_ = s_tb[DIR == 1 ? wv : 0];
  const f32x16 zero = {0};
  const int64_t tiles = (N + 31) / 32;
  const int64_t wave = rfl((int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave; t < tiles; t += n_waves) {
    const int64_t p = t * 32 + j;
    const bool ok = p < N;
    const uint32_t row = (uint32_t)p;
    float x[16], w3[16];
    load_rows16(PL, ok, row, h, x);
    float num = 0.f;
    if (w33 && ok) num = sqrtf(1.f / ((float)(ptr[p + 1] - ptr[p]) + 1e-3f));
    tab16(s_w33, 0, h, w3);
    // ---- forward: s1 = Wsa [pooled | num], a1 = act(BN(s1)), s2 = Wsb a1, a2 = act(BN(s2)), u = WcB a2
    f32x16 z1 = mmf(s_w, SQ_WSA, lane, x, zero);
#pragma unroll
    for (int r = 0; r < 16; ++r) z1[r] = __builtin_fmaf(num, w3[r], z1[r]);
    if (DIR == 0 && STAGE == 1) { add_stats(z1, ok, st); continue; }
    float a1[16], a2[16];
    act(z1, s_tab[0], h, a1);
    const f32x16 z2 = mmf(s_w, SQ_WSB, lane, a1, zero);
    if (DIR == 0 && STAGE == 2) { add_stats(z2, ok, st); continue; }
    act(z2, s_tab[1], h, a2);
    if (DIR == 0) {
      store_rows16(UO, ok, row, h, mmf(s_w, SQ_WCB, lane, a2, zero));
      continue;
    }
    // ---- backward
    float d[16], dz[16], unused_st[2][16];
    load_rows16(DU, ok, row, h, d);                    // zeros in the lanes without a point
    const f32x16 da2 = mmf(s_w, SQ_WCBT, lane, d, zero);
    if (STAGE == 1) {
      layer_bwd<true, false>(z2, da2, s_tab[1], h, ok, st, dz);
      tile_put(ta_, j, h, d, true);
      tile_put(tb_, j, h, a2, ok);
      wave_sync();
      accW = wgradf(ta_, tb_, j, h, accW);          // dWcB[n][k] = sum_p du[p][n] a2[p][k]
      wave_sync();
      continue;
    }
    layer_bwd<false, true>(z2, da2, s_tab[1], h, ok, unused_st, dz);
#pragma unroll
    for (int r = 0; r < 16; ++r) dz[r] = ok ? dz[r] : 0.f;
    const f32x16 da1 = mmf(s_w, SQ_WSBT, lane, dz, zero);
    if (STAGE == 2) {
      float tmp[16];
      layer_bwd<true, false>(z1, da1, s_tab[0], h, ok, st, tmp);
      tile_put(ta_, j, h, dz, true);
      tile_put(tb_, j, h, a1, ok);
      wave_sync();
      accW = wgradf(ta_, tb_, j, h, accW);          // dWsb[n][k] = sum_p dz2[p][n] a1[p][k]
      wave_sync();
      continue;
    }
    float dz1[16];
    layer_bwd<false, true>(z1, da1, s_tab[0], h, ok, unused_st, dz1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dz1[r] = ok ? dz1[r] : 0.f;
      acc33[r] = __builtin_fmaf(dz1[r], num, acc33[r]);
    }
    store_rows16(DPO, ok, row, h, mmf(s_w, SQ_WSAT, lane, dz1, zero));
    tile_put(ta_, j, h, dz1, true);
    tile_put(tb_, j, h, x, ok);
    wave_sync();
    accW = wgradf(ta_, tb_, j, h, accW);            // dWsa[n][k] = sum_p dz1[p][n] pooled[p][k]
    wave_sync();
  }
  if (DIR == 1) flush_matrix(accW, dW, ld_dw, D, false, s_red);
  if (DIR == 1 && STAGE == 3 && dw33) {
    // d Wsa[:, 32] = sum_p dz1[p] num_p
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += blockDim.x) s_red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc33[r];
#pragma unroll
      for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off);
      if (j == 0) atomicAdd(&s_red[chan(r, h)], v);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += blockDim.x) atomicAdd(&dw33[i * ld_dw], s_red[i]);
  }
  if (!(DIR == 0 && STAGE == 3) && !(DIR == 1 && STAGE == 3)) flush_stats<2>(st, stats, s_red);
}

}  // namespace chain3
}  // namespace dva

using namespace dva;
using namespace dva::chain;

extern "C" {

int dva_chain3_prep(const float* W1, const float* W2, const float* W5, int32_t ld5, const float* W6, const float* Ws,
                    int32_t G, void* ops, void* stream) {
  if (!W1 || !W2 || !W5 || !W6 || !Ws || !ops || G < 1 || G > 4 || ld5 < D) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(chain3::prep3_kernel, dim3(chain3::N_Q), dim3(64), 0, (hipStream_t)stream, W1, W2, W5, ld5, W6,
                     Ws, G, (float4*)ops);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain3_stats2(const float* x_map, const int32_t* view_point, const void* tiles, const int32_t* n_tiles,
                      const void* ops, const float* bn1, const float* gamma2, double* stats, float* zstar,
                      int32_t* arg, float* z2, int64_t n_views, void* stream) {
  if (n_views < 0) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!x_map || !view_point || !tiles || !n_tiles || !ops || !bn1 || !gamma2 || !stats || !zstar || !arg || !z2)
    return DVA_ERR_INVALID;
  if (n_views * 32 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(chain3::stats2_kernel, dim3(chain_grid(4)), dim3(256), 0, (hipStream_t)stream, x_map, view_point,
                     (const int2*)tiles, n_tiles, (const float4*)ops, bn1, gamma2, stats, zstar, arg, z2, n_views);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain3_stats(int32_t layer, const float* rows_in, const int32_t* view_point, const float* u,
                     const void* tiles, const int32_t* n_tiles, const void* ops, const float* bn, float* z5,
                     double* stats, int64_t n_views, int64_t n_points, void* stream) {
  if (n_views < 0 || (layer != 5 && layer != 6)) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!rows_in || !tiles || !n_tiles || !ops || !bn || !stats || (layer == 5 && (!view_point || !u || !z5)))
    return DVA_ERR_INVALID;
  if (n_views * 16 > 0xfffffff0ll || n_points * 128 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  const dim3 grid(chain_grid(layer == 5 ? 3 : 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (layer == 5)
    hipLaunchKernelGGL((chain3::stats_mid_kernel<5>), grid, block, 0, s, rows_in, view_point, u, (const int2*)tiles,
                       n_tiles, (const float4*)ops, bn, z5, stats, n_views, n_points);
  else
    hipLaunchKernelGGL((chain3::stats_mid_kernel<6>), grid, block, 0, s, rows_in, view_point, u, (const int2*)tiles,
                       n_tiles, (const float4*)ops, bn, (float*)nullptr, stats, n_views, n_points);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain3_scores(const float* z5, const void* tiles, const int32_t* n_tiles, const void* ops, const float* bn5,
                      const float* bn6, const float* score_bias, int32_t G, float* scores, int64_t n_views,
                      void* stream) {
  if (n_views < 0 || G < 1 || G > 4) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!z5 || !tiles || !n_tiles || !ops || !bn5 || !bn6 || !score_bias || !scores) return DVA_ERR_INVALID;
  if (n_views * 16 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(chain3::scores_kernel, dim3(chain_grid(4)), dim3(256), 0, (hipStream_t)stream, z5,
                     (const int2*)tiles, n_tiles, (const float4*)ops, bn5, bn6, score_bias, (int)G, scores, n_views);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain3_score_stats(const float* z5, const void* tiles, const int32_t* n_tiles, const void* ops,
                           const float* bn5, const float* bn6, const float* grad_scores, double* stats6, float* dWs,
                           float* dbs, int32_t G, int64_t n_views, void* stream) {
  if (n_views < 0 || G < 1 || G > 4) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!z5 || !tiles || !n_tiles || !ops || !bn5 || !bn6 || !grad_scores || !stats6 || !dWs || !dbs)
    return DVA_ERR_INVALID;
  if (n_views * 16 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(chain3::score_stats_kernel, dim3(chain_grid(2)), dim3(256), 0, (hipStream_t)stream, z5,
                     (const int2*)tiles, n_tiles, (const float4*)ops, bn5, bn6, grad_scores, stats6, dWs, dbs, (int)G,
                     n_views);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain3_bwd_layer(int32_t stage, const float* x_map, const float* z_rows, const int32_t* view_point,
                         const float* u, const void* tiles, const int32_t* n_tiles, const void* ops,
                         const float* bn_lo, const float* bn_hi, const float* sm, const float* grad_scores,
                         const int32_t* arg, const float* dpooled, const float* da_in, float* da_out, float* dW,
                         float* du, float* P, double* stats, int64_t n_views, int64_t n_points, void* stream) {
  if (n_views < 0 || (stage != 6 && stage != 5 && stage != 2)) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!z_rows || !tiles || !n_tiles || !ops || !bn_lo || !bn_hi || !sm || !dW || (stage != 2 && !stats))
    return DVA_ERR_INVALID;
  if (stage == 6 && (!grad_scores || !da_out)) return DVA_ERR_INVALID;
  if (stage == 5 && (!view_point || !u || !du || !da_in || !da_out)) return DVA_ERR_INVALID;
  if (stage == 2 && (!x_map || !view_point || !arg || !dpooled || !P || !da_in)) return DVA_ERR_INVALID;
  if (n_views * 32 > 0xfffffff0ll || n_points * 128 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  const dim3 block(256);
  hipStream_t s = (hipStream_t)stream;
#define DVA_L3(ST_)                                                                                               \
  hipLaunchKernelGGL((chain3::layer_bwd_kernel<ST_>), dim3(chain_grid(2)), block, 0, s, x_map, z_rows, view_point, u, \
                     (const int2*)tiles, n_tiles, (const float4*)ops, bn_lo, bn_hi, sm, grad_scores, arg, dpooled,    \
                     da_in, da_out, dW, du, P, stats, n_views, n_points)
  if (stage == 6) DVA_L3(6);
  else if (stage == 5) DVA_L3(5);
  else DVA_L3(2);
#undef DVA_L3
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain3_set_prep(const float* Wsa, int32_t ld_sa, const float* Wsb, const float* Wc, int32_t ld_c, void* ops,
                        void* stream) {
  if (!Wsa || !Wsb || !Wc || !ops || ld_sa < D || ld_c < 2 * D) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(chain3::set3_prep_kernel, dim3(chain3::N_SQ), dim3(64), 0, (hipStream_t)stream, Wsa, ld_sa, Wsb,
                     Wc, ld_c, (float4*)ops);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain3_set_fwd(int32_t stage, const float* pooled, const int64_t* ptr, const float* w33, const void* ops,
                       const float* bn_s1, const float* bn_s2, float* u, double* stats, int64_t n_points,
                       void* stream) {
  if (n_points < 0 || stage < 1 || stage > 3) return DVA_ERR_INVALID;
  if (n_points == 0) return DVA_OK;
  if (!pooled || !ptr || !ops || (stage >= 2 && !bn_s1) || (stage == 3 && (!bn_s2 || !u)) || (stage < 3 && !stats))
    return DVA_ERR_INVALID;
  if (n_points * 128 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  const int64_t tiles = (n_points + 31) / 32;
  const int cap = chain_grid(2);
  const dim3 grid((int)((tiles + 3) / 4 < cap ? (tiles + 3) / 4 : cap)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define DVA_SET3_FWD(ST_)                                                                                          \
  hipLaunchKernelGGL((chain3::set3_kernel<0, ST_>), grid, block, 0, s, pooled, ptr, w33, (const float4*)ops, bn_s1, \
                     bn_s2, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, u,                  \
                     (float*)nullptr, (float*)nullptr, 0, (float*)nullptr, stats, n_points)
  if (stage == 1) DVA_SET3_FWD(1);
  else if (stage == 2) DVA_SET3_FWD(2);
  else DVA_SET3_FWD(3);
#undef DVA_SET3_FWD
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain3_set_bwd(int32_t stage, const float* pooled, const int64_t* ptr, const float* w33, const void* ops,
                       const float* bn_s1, const float* bn_s2, const float* sm_s1, const float* sm_s2,
                       const float* du, float* dpooled, float* dW, int32_t ld_dw, float* dw33, double* stats,
                       int64_t n_points, void* stream) {
  if (n_points < 0 || stage < 1 || stage > 3) return DVA_ERR_INVALID;
  if (n_points == 0) return DVA_OK;
  if (!pooled || !ptr || !ops || !bn_s1 || !bn_s2 || !du || !dW || ld_dw < D || (stage < 3 && !stats) ||
      (stage >= 2 && !sm_s2) || (stage == 3 && (!sm_s1 || !dpooled)))
    return DVA_ERR_INVALID;
  if (n_points * 128 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  const int64_t tiles = (n_points + 31) / 32;
  const int cap = chain_grid(2);
  const dim3 grid((int)((tiles + 3) / 4 < cap ? (tiles + 3) / 4 : cap)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define DVA_SET3_BWD(ST_)                                                                                          \
  hipLaunchKernelGGL((chain3::set3_kernel<1, ST_>), grid, block, 0, s, pooled, ptr, w33, (const float4*)ops, bn_s1, \
                     bn_s2, sm_s1, sm_s2, du, (float*)nullptr, dpooled, dW, ld_dw, dw33, stats, n_points)
  if (stage == 1) DVA_SET3_BWD(1);
  else if (stage == 2) DVA_SET3_BWD(2);
  else DVA_SET3_BWD(3);
#undef DVA_SET3_BWD
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // extern "C"
