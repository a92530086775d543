// Shared device helpers of the recompute chain (chain_fwd.hip / chain_bwd.hip): DeepSetFeat re-evaluated from the
// raw mapping features inside every pass instead of being streamed through HBM between passes.
//
// Geometry.  One wavefront owns one TILE = up to 32 consecutive views made of whole points (tile table built by
// dva_chain_tile_build; points with more than 32 views are cut into consecutive fragment tiles that one
// wavefront processes in order, carrying its per-point state in registers).  Lane l = (j, h) = (l & 31, l >> 5):
// j = view of the tile, h = half.  A 32x32 layer is two v_mfma_f32_32x32x16_bf16 whose B operand is the packed
// activation of the view (lane (j, h) supplies 8 input channels per k-block) and whose result lands as
//   z[r] = Z[channel chan(r, h)][view j],  chan(r, h) = (r & 3) + 8 (r >> 2) + 4 h,   r < 16,
// which is exactly the k-slot order the weight operands of the NEXT layer are prepared in (dva_chain_prep), so
// BatchNorm + LeakyReLU + bf16 packing are register-to-register and layers chain without any data movement.
// bf16 operands (activations and weights rounded to bf16 like the reference under torch.autocast(bfloat16); layers whose
// raw output a pass does not need take the BatchNorm scale inside the rounded operand: "BatchNorm folded" below), fp32
// accumulation, BatchNorm / softmax / statistics in fp32.
#pragma once
#include <cstdlib>
#include "dva_common.h"

namespace dva {
namespace chain {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define CH_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int D = 32;
constexpr float SLOPE = 0.2f;
constexpr uint32_t OOB = 0x7ffffff0u;  // byte offset beyond every buffer: loads return 0, stores are dropped

__host__ __device__ __forceinline__ int chan(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// weight operand table (dva_chain_prep): N_OPS x 64 lanes x 16 bytes
enum {
  OP_W1 = 0,    // x_map (8) -> 32: slots 0..3 = W1[c][4h + s], 4..7 = the same (x enters as hi | lo)
  OP_W2 = 1,    // +m: W2[c][chan(8m + s, h)]
  OP_W5 = 3,    // concatenation layer, h1 half: Wc[c][chan(8m + s, h)]
  OP_W6 = 5,
  OP_WS = 7,    // score rows (rows >= G are zero)
  OP_W6T = 9,   // transposed operands of the input-gradient products: W[chan(8m + s, h)][c]
  OP_W5T = 11,
  OP_W2T = 13,
  OP_WST = 15,  // Ws^T for da6 = Ws^T dc: h = 0: slots 0..3 = Ws[s][c], 4..7 = the same (dc enters as hi | lo)
  OP_WKT = 16,  // +m: the FULL transposed last layer Ws[chan(8m + s, h)][c] (G = 32: the key layer of QKVBimodalCSRPool,
                //     whose gradient arrives as a 32-wide row instead of 4 scores; zero otherwise)
  N_OPS = 18
};

// per-layer constant table in LDS, accumulator-permuted (index 16 h + r <-> channel chan(r, h)):
//   G = gamma * invstd | B = beta - mean * G | I = invstd | M = -mean * invstd |
//   K1 = G (S1/M + M S2/M) | K2 = G I S2/M   (BatchNorm backward dz = G dy - K1 - K2 z) |
//   G6 = 0.6 G | B6 = 0.6 B  (activation rows: leaky(y) = 0.6 y + 0.4 |y| = t + 2/3 |t| with t = 0.6 y)
enum { T_G = 0, T_B = 1, T_I = 2, T_M = 3, T_K1 = 4, T_K2 = 5, T_G6 = 6, T_B6 = 7, T_ROWS = 8 };
constexpr int TAB_FLOATS = T_ROWS * D;

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint64_t bytes) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = rfl((uint32_t)a), hi = rfl((uint32_t)(a >> 32));
  const uint32_t n = rfl((uint32_t)(bytes > 0xfffffff0ull ? 0xfffffff0ull : bytes));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, (int)n, 0x00020000);
}
__device__ __forceinline__ u32x4 ld128(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
}
__device__ __forceinline__ u32x2 ld64(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
}
__device__ __forceinline__ uint32_t ld32(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0);
}
__device__ __forceinline__ void st128(__amdgpu_buffer_rsrc_t r, uint32_t off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, 0);
}
__device__ __forceinline__ void st32(__amdgpu_buffer_rsrc_t r, uint32_t off, uint32_t v) {
  __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)off, 0, 0);
}
__device__ __forceinline__ float4 as_f4(u32x4 v) { return __builtin_bit_cast(float4, v); }
__device__ __forceinline__ u32x4 as_u4(float a, float b, float c, float d) {
  const float4 f = make_float4(a, b, c, d);
  return __builtin_bit_cast(u32x4, f);
}

// DS operations of one wavefront execute in order: a wave-level hand-off through LDS needs the counter wait
// and a scheduling barrier, no s_barrier.
__device__ __forceinline__ void wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// plain v_max_f32: fmaxf() makes hipcc quiet both operands first (v_max x, x) -- three instructions for one
__device__ __forceinline__ float vmaxf(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float leaky(float y) { return fmaxf(y, SLOPE * y); }
__device__ __forceinline__ float dleaky(float y) { return y > 0.f ? 1.f : SLOPE; }

__device__ __forceinline__ void swap_halves(uint32_t& a, uint32_t& b) {
  // lanes [32, 64) of a <-> lanes [0, 32) of b
  const u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r.x;
  b = r.y;
}
__device__ __forceinline__ float shfl(float v, int src_lane) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)__float_as_uint(v)));
}
__device__ __forceinline__ int shfl(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }

// ---- tiles -------------------------------------------------------------------------------------------------
// int2 {v0, nv | frag << 8}: frag 0 = whole points, 1 / 2 / 3 = first / middle / last fragment of a long point
struct TileInfo {
  int v0, nv, frag;
};
__device__ __forceinline__ TileInfo get_tile(const int2* __restrict__ tiles, int t) {
  const int2 d = tiles[t];
  TileInfo ti;
  ti.v0 = rfl(d.x);
  const int m = rfl(d.y);
  ti.nv = m & 0xff;
  ti.frag = m >> 8;
  return ti;
}
// Flat tile range of this wavefront; continuation fragments stay with the wavefront that owns the first one.
__device__ __forceinline__ void wave_tile_range(const int2* __restrict__ tiles, int n_tiles, int& ta, int& tb) {
  const int wave = rfl((int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const int n_waves = (int)(((int64_t)gridDim.x * blockDim.x) >> 6);
  ta = (int)((int64_t)n_tiles * wave / n_waves);
  tb = (int)((int64_t)n_tiles * (wave + 1) / n_waves);
  while (ta < n_tiles && (rfl(tiles[ta].y) >> 8) >= 2) ++ta;
  while (tb < n_tiles && (rfl(tiles[tb].y) >> 8) >= 2) ++tb;
  if (ta > tb) ta = tb;
}

// Tile descriptors of 64 consecutive tiles in one register pair (lane i <-> tile base + i): the per-tile
// descriptor is two v_readlane instead of a dependent global load in front of every prefetch (measured: that load
// was a full memory latency per tile on the critical path).
struct TileWindow {
  int base;
  int2 w;
};
__device__ __forceinline__ TileInfo window_tile(TileWindow& tw, const int2* __restrict__ tiles, int t, int last,
                                                int lane) {
  if (t - tw.base >= 64 || t < tw.base) {      // uniform
    tw.base = t;
    const int i = t + lane;
    tw.w = tiles[i < last ? i : last];
  }
  const int idx = t - tw.base;
  TileInfo ti;
  ti.v0 = __builtin_amdgcn_readlane(tw.w.x, idx);
  const int m = __builtin_amdgcn_readlane(tw.w.y, idx);
  ti.nv = m & 0xff;
  ti.frag = m >> 8;
  return ti;
}

// Software-pipelined loop over the tiles [ta, tb): the loads of tile t + 1 are in flight while tile t is
// computed; two register sets in ping-pong (a rotating copy would wait for the prefetch it has just issued).
// load(TileInfo, t) issues the loads of a tile, body(pre) computes it.
template <typename Pre, typename LoadF, typename BodyF>
__device__ __forceinline__ void run_tiles(const int2* __restrict__ tiles, int ta, int tb, LoadF&& load,
                                          BodyF&& body) {
  if (ta >= tb) return;
  const int lane = threadIdx.x & 63;
  const int last = tb - 1;
  TileWindow tw;
  tw.base = ta - 64;
  auto ld = [&](int t) {
    const int tt = t < tb ? t : last;
    return load(window_tile(tw, tiles, tt, last, lane), tt);
  };
  Pre a = ld(ta);
  int t = ta + 1;
  Pre b = ld(t);
  body(a);
  while (t < tb) {
    a = ld(t + 1);
    body(b);
    ++t;
    if (t >= tb) break;
    b = ld(t + 1);
    body(a);
    ++t;
  }
}

// the same loop without the prefetch (one register set): for kernels whose register budget decides the occupancy
template <typename Pre, typename LoadF, typename BodyF>
__device__ __forceinline__ void run_tiles_single(const int2* __restrict__ tiles, int ta, int tb, LoadF&& load,
                                                 BodyF&& body) {
  if (ta >= tb) return;
  const int lane = threadIdx.x & 63;
  const int last = tb - 1;
  TileWindow tw;
  tw.base = ta - 64;
  for (int t = ta; t < tb; ++t) {
    const Pre a = load(window_tile(tw, tiles, t, last, lane), t);
    body(a);
  }
}

// ---- layer pieces --------------------------------------------------------------------------------------------
struct WOp {
  bf16x8 m[2];
};
__device__ __forceinline__ bf16x8 load_op(const uint4* __restrict__ ops, int op, int lane) {
  return __builtin_bit_cast(bf16x8, ops[op * 64 + lane]);
}
__device__ __forceinline__ WOp load_wop(const uint4* __restrict__ ops, int op, int lane) {
  WOp w;
  w.m[0] = load_op(ops, op, lane);
  w.m[1] = load_op(ops, op + 1, lane);
  return w;
}
// weight operands staged in LDS (backward kernels: 64 registers per lane less than holding them)
__device__ __forceinline__ void stage_ops(uint4* s_ops, const uint4* __restrict__ ops) {
  for (int i = threadIdx.x; i < N_OPS * 64; i += blockDim.x) s_ops[i] = ops[i];
}
__device__ __forceinline__ bf16x8 lds_op(const uint4* s_ops, int op, int lane) {
  return __builtin_bit_cast(bf16x8, s_ops[op * 64 + lane]);
}
__device__ __forceinline__ f32x16 mm32_lds(const uint4* s_ops, int op, int lane, const bf16x8 (&a)[2], f32x16 c) {
  asm volatile("" ::: "memory");   // keep the two ds_read_b128 inside the tile loop
  const bf16x8 w0 = lds_op(s_ops, op, lane), w1 = lds_op(s_ops, op + 1, lane);
  c = CH_MFMA(w0, a[0], c);
  c = CH_MFMA(w1, a[1], c);
  return c;
}
__device__ __forceinline__ f32x16 mm32(const WOp& w, const bf16x8 (&a)[2], f32x16 c) {
  c = CH_MFMA(w.m[0], a[0], c);
  c = CH_MFMA(w.m[1], a[1], c);
  return c;
}
__device__ __forceinline__ bf16x8 pack8(const float* x) {
  const u32x4 v = {pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]), pack_bf16x2(x[4], x[5]),
                   pack_bf16x2(x[6], x[7])};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 mask8(bf16x8 a, uint32_t keep) {  // keep = 0 or ~0
  u32x4 v = __builtin_bit_cast(u32x4, a);
  v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;      // (compiled to v_cndmask; a v_and_b32 forced through inline asm
                                                          //  broke tests/test_gpu_fullsize.py::test_fused_bilinear_full_size_slice)
  return __builtin_bit_cast(bf16x8, v);
}
// 4 features per lane as the 8 k-slots of its half: hi(x) | lo(x) = x - hi(x): the first layer sees x to 16 bits
__device__ __forceinline__ bf16x8 pack_x(const float4& x) {
  const uint32_t h0 = pack_bf16x2(x.x, x.y), h1 = pack_bf16x2(x.z, x.w);
  const float r0 = x.x - __uint_as_float(h0 << 16), r1 = x.y - __uint_as_float(h0 & 0xffff0000u);
  const float r2 = x.z - __uint_as_float(h1 << 16), r3 = x.w - __uint_as_float(h1 & 0xffff0000u);
  const u32x4 v = {h0, h1, pack_bf16x2(r0, r1), pack_bf16x2(r2, r3)};
  return __builtin_bit_cast(bf16x8, v);
}

// the constants of this lane's 16 accumulator channels from an LDS table row
__device__ __forceinline__ void tab16(const float* tab, int row, int h, float (&c)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = *reinterpret_cast<const float4*>(tab + row * D + 16 * h + 4 * q);
    c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w;
  }
}
// Fill the table of one layer from bn = [4][32] (mean | invstd | gamma | beta) and sm = [2][32] (S1/M | S2/M,
// nullable).  Call from the whole block, then __syncthreads().
// ext: bn has a fifth row = the 0.6-scaled shift of the layer's (possibly folded) product (dva_chain_bn_consts).
__device__ __forceinline__ void stage_tab(float* tab, const float* __restrict__ bn, const float* __restrict__ sm,
                                          bool ext = true) {
  if (!bn) return;      // a layer the pass does not reach
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    const int c = chan(i & 15, i >> 4);
    const float mean = bn[c], inv = bn[D + c], gam = bn[2 * D + c], bet = bn[3 * D + c];
    const float g = gam * inv;
    tab[T_G * D + i] = g;
    tab[T_B * D + i] = bet - mean * g;
    tab[T_I * D + i] = inv;
    tab[T_M * D + i] = -mean * inv;
    const float s1 = sm ? sm[c] : 0.f, s2 = sm ? sm[D + c] : 0.f;
    tab[T_K1 * D + i] = g * (s1 - mean * inv * s2);
    tab[T_K2 * D + i] = g * inv * s2;
    tab[T_G6 * D + i] = 0.6f * g;
    tab[T_B6 * D + i] = ext ? bn[4 * D + c] : 0.6f * (bet - mean * g);
  }
}

// forward-only variant: two rows, 0.6 G | 0.6 B (act_pack with rg = 0, rb = 1)
__device__ __forceinline__ void stage_tab_fwd(float* tab, const float* __restrict__ bn) {
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    const int c = chan(i & 15, i >> 4);
    const float g = bn[2 * D + c] * bn[D + c];
    tab[0 * D + i] = 0.6f * g;
    tab[1 * D + i] = bn[4 * D + c];
  }
}

// BatchNorm + LeakyReLU(0.2) + bf16 packing as the B operand of the next layer, two VALU operations per value:
//   t = z * (0.6 G) + 0.6 B = 0.6 y;   leaky(y) = 0.6 y + 0.4 |y| = t + (2/3) |t|   (one fma with an |.| modifier)
// keep = 0 zeroes the operand of a lane without a view.  The LDS reads stay inside the tile loop (asm barrier):
// hoisting 32 constants per layer into registers costs an occupancy step.
// MASK = false (round 6): the operand of a lane without a view is NOT zeroed.  A column of the B operand only reaches the
// same column of the product, so the garbage (finite: such a lane loaded zeros) stays in lanes that own no view; what has
// to be exact is every REDUCTION OVER VIEWS, and each of those is guarded where it happens (statistics under `if (ok)`,
// gradient rows masked by pack16 before the weight-gradient products).  v_cndmask costs two issue slots (tools/ubench):
// the eight per layer and tile were a quarter of the vector time of a statistics pass.
#ifndef DVA_MASK_ACT
#define DVA_MASK_ACT 0      // 1: the A/B build (csrc/Makefile EXTRA=-DDVA_MASK_ACT=1) keeps every activation mask
#endif
template <bool MASK_ = true>
__device__ __forceinline__ void act_pack(const f32x16& z, const float* tab, int h, uint32_t keep, bf16x8 (&a)[2],
                                         int rg = T_G6, int rb = T_B6, float* sum = nullptr) {
  asm volatile("" ::: "memory");
  float g[16], b[16], av[16];
  tab16(tab, rg, h, g);
  tab16(tab, rb, h, b);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float t = __builtin_fmaf(z[r], g[r], b[r]);
    av[r] = __builtin_fmaf(__builtin_fabsf(t), 0.6666667f, t);
  }
  if (sum && keep) {      // per-channel sums of the activation (the mean the next layer's folded shift needs)
#pragma unroll
    for (int r = 0; r < 16; ++r) sum[r] += av[r];
  }
  constexpr bool MASK = MASK_ || DVA_MASK_ACT;
  a[0] = MASK ? mask8(pack8(&av[0]), keep) : pack8(&av[0]);
  a[1] = MASK ? mask8(pack8(&av[8]), keep) : pack8(&av[8]);
}

// ---- BatchNorm folded into the weight operand ---------------------------------------------------------------------
// A layer whose raw output a pass does not need (only its activation) takes the BatchNorm scale inside the product:
// W' = bf16(0.6 G W) (rows of the A operand: every 16-byte entry of lane l belongs to output channel l & 31), the
// shift 0.6 B as the accumulator the MFMA starts from, so the product IS t = 0.6 y and the activation is the single
// operation leaky(y) = t + (2/3) |t|.  fold_ops builds the folded operand `op` (n_blocks blocks) at block `dst` of the
// LDS table from the fp32 entries dva_chain_prep appends to the table (one rounding, like the plain operand).
constexpr int N_OPS32 = 7;                       // W1 | W2 (2) | W5 (2) | W6 (2): the forward operands that can fold
constexpr int OPS_UINT4 = N_OPS * 64 + N_OPS32 * 64 * 2;
__device__ __forceinline__ void fold_ops(uint4* s_ops, int dst, const uint4* __restrict__ ops, int op, int n_blocks,
                                         const float* __restrict__ bn) {
  const float4* w32 = reinterpret_cast<const float4*>(ops + N_OPS * 64);
  for (int i = threadIdx.x; i < n_blocks * 64; i += blockDim.x) {
    const int n = i & 31;
    const float s = 0.6f * bn[2 * D + n] * bn[D + n];
    const float4 a = w32[(op * 64 + i) * 2], b = w32[(op * 64 + i) * 2 + 1];
    s_ops[dst * 64 + i] = make_uint4(pack_bf16x2(a.x * s, a.y * s), pack_bf16x2(a.z * s, a.w * s),
                                     pack_bf16x2(b.x * s, b.y * s), pack_bf16x2(b.z * s, b.w * s));
  }
}
// the same for a kernel that keeps its operands in registers: this lane's entry of block `op`, folded
__device__ __forceinline__ bf16x8 load_op_fold(const uint4* __restrict__ ops, int op, int lane,
                                               const float* __restrict__ bn) {
  const float4* w32 = reinterpret_cast<const float4*>(ops + N_OPS * 64) + (op * 64 + lane) * 2;
  const int n = lane & 31;
  const float s = 0.6f * bn[2 * D + n] * bn[D + n];
  const float4 a = w32[0], b = w32[1];
  const u32x4 v = {pack_bf16x2(a.x * s, a.y * s), pack_bf16x2(a.z * s, a.w * s), pack_bf16x2(b.x * s, b.y * s),
                   pack_bf16x2(b.z * s, b.w * s)};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ WOp load_wop_fold(const uint4* __restrict__ ops, int op, int lane,
                                             const float* __restrict__ bn) {
  WOp w;
  w.m[0] = load_op_fold(ops, op, lane, bn);
  w.m[1] = load_op_fold(ops, op + 1, lane, bn);
  return w;
}
// the accumulator a folded layer starts from: row `row` of the table (0.6 B) for this lane's 16 channels
__device__ __forceinline__ f32x16 bias_acc(const float* tab, int row, int h) {
  asm volatile("" ::: "memory");
  float b[16];
  tab16(tab, row, h, b);
  f32x16 c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = b[r];
  return c;
}
// activation of a folded layer (t = 0.6 y from the product) + bf16 packing as the next B operand
template <bool MASK_ = true>
__device__ __forceinline__ void act_fold(const f32x16& t, uint32_t keep, bf16x8 (&a)[2], float* sum = nullptr) {
  float av[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) av[r] = __builtin_fmaf(__builtin_fabsf(t[r]), 0.6666667f, t[r]);
  if (sum && keep) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sum[r] += av[r];
  }
  constexpr bool MASK = MASK_ || DVA_MASK_ACT;
  a[0] = MASK ? mask8(pack8(&av[0]), keep) : pack8(&av[0]);
  a[1] = MASK ? mask8(pack8(&av[8]), keep) : pack8(&av[8]);
}

// ---- statistics ------------------------------------------------------------------------------------------------
// per-lane fp32 partial sums of the 16 accumulator channels -> per wavefront (shuffles) -> per block in a FIXED order
// (one LDS slot per wavefront, summed in fp64: no LDS float atomics, whose order would vary from run to run and
// move the fp32 BatchNorm constants by an ulp -- enough to change bf16 roundings downstream) -> fp64 atomics (their
// order varies, at 1e-16: the statistics are reproducible after rounding to fp32, and with them the whole forward).
// s_red: 4 * NV * D floats (blocks of 4 wavefronts).
constexpr int STATS_RED_FLOATS = 4 * 3 * D;
template <int NV>
__device__ __forceinline__ void flush_stats(float (&st)[NV][16], double* __restrict__ out, float* s_red) {
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
    atomicAdd(&out[i], acc);
  }
}

// transposed bf16 tile [channel][view], row stride 40 (80 bytes): the 8 consecutive views a lane feeds to the
// weight-gradient MFMA are one ds_read_b128
constexpr int TSB = 40;
__device__ __forceinline__ void tileT_put(bf16_t* tile, int c0, int c1, int v, float x0, float x1) {
  const uint32_t d = pack_bf16x2(x0, x1);
  tile[c0 * TSB + v] = (bf16_t)(d & 0xffffu);
  tile[c1 * TSB + v] = (bf16_t)(d >> 16);
}
__device__ __forceinline__ void tileT_put_acc(bf16_t* tile, int v, int h, const float* x) {
#pragma unroll
  for (int r = 0; r < 16; r += 2) tileT_put(tile, chan(r, h), chan(r + 1, h), v, x[r], x[r + 1]);
}
__device__ __forceinline__ bf16x8 tileT_get(const bf16_t* tile, int c, int h, int m) {
  return *reinterpret_cast<const bf16x8*>(tile + c * TSB + 16 * m + 8 * h);
}

// One BatchNorm + LeakyReLU layer backwards: dy = leaky'(y) da, then
//   STATS: st[0] += dy, st[1] += dy * z   (the caller turns sum dy z into S2 = sum dy z_hat = I (sum dy z - mean S1))
//   APPLY: dz = G (dy - S1/M - z_hat S2/M) = G dy - K1 - K2 z   (lanes without a view: masked when packed)
// six VALU operations per value each.
template <bool STATS, bool APPLY>
__device__ __forceinline__ void layer_bwd(const f32x16& z, const f32x16& da, const float* tab, int h, bool ok,
                                          float (&st)[2][16], float (&dz)[16]) {
  asm volatile("" ::: "memory");
  // four channels at a time: a few float4 of constants live instead of 64+ registers
#pragma unroll
  for (int q = 0; q < 4; ++q) {
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
      const float y = __builtin_fmaf(z[r], g[e], b[e]);
      const float dy = y > 0.f ? da[r] : SLOPE * da[r];
      if (STATS) {
        st[0][r] += dy;
        st[1][r] = __builtin_fmaf(dy, z[r], st[1][r]);
      }
      if (APPLY) dz[r] = __builtin_fmaf(-k2[e], z[r], __builtin_fmaf(g[e], dy, -k1[e]));
      else dz[r] = dy;        // statistics-only call: hands dy back (dead code where the caller ignores it)
    }
  }
}
// ---- backward of a layer whose activation the forward took from ANOTHER product than the raw z (BatchNorm folded
// into the operand): the derivative of leaky follows the sign of what the forward's activation saw (t = 0.6 y of the
// folded product), the BatchNorm-backward terms follow the raw z.  Taking the sign from G z + B instead flips
// leaky' on the ~0.3 % of values where the two pre-activations straddle zero: measured 3-10 % on the parameter
// gradients (tests/test_gpu_chain.py::test_chain_matches_bf16_emulation).
__device__ __forceinline__ void dleaky_mul(const f32x16& t, f32x16& da) {
#pragma unroll
  for (int r = 0; r < 16; ++r) da[r] = t[r] > 0.f ? da[r] : SLOPE * da[r];
}
// statistics of a BatchNorm backward from dy = leaky' da: st[0] += dy, st[1] += dy * z (raw layer output)
__device__ __forceinline__ void bn_bwd_stats(const f32x16& z, const f32x16& dy, float (&st)[2][16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    st[0][r] += dy[r];
    st[1][r] = __builtin_fmaf(dy[r], z[r], st[1][r]);
  }
}
// dz = G dy - K1 - K2 z (four channels at a time: a few float4 of constants live)
__device__ __forceinline__ void bn_bwd_apply(const f32x16& z, const f32x16& dy, const float* tab, int h,
                                             float (&dz)[16]) {
  asm volatile("" ::: "memory");
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int o = 16 * h + 4 * q;
    const float4 g4 = *reinterpret_cast<const float4*>(tab + T_G * D + o);
    const float4 a4 = *reinterpret_cast<const float4*>(tab + T_K1 * D + o);
    const float4 c4 = *reinterpret_cast<const float4*>(tab + T_K2 * D + o);
    const float g[4] = {g4.x, g4.y, g4.z, g4.w}, k1[4] = {a4.x, a4.y, a4.z, a4.w}, k2[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * q + e;
      dz[r] = __builtin_fmaf(-k2[e], z[r], __builtin_fmaf(g[e], dy[r], -k1[e]));
    }
  }
}
__device__ __forceinline__ void pack16(const float (&x)[16], uint32_t keep, bf16x8 (&a)[2]) {
  a[0] = mask8(pack8(&x[0]), keep);
  a[1] = mask8(pack8(&x[8]), keep);
}
// packed activation (k-slot s of block m = accumulator register 8m + s) -> transposed tile
__device__ __forceinline__ void tileT_put_packed(bf16_t* tile, int v, int h, const bf16x8 (&a)[2]) {
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const u32x4 w = __builtin_bit_cast(u32x4, a[m]);
    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      tile[chan(8 * m + 2 * i, h) * TSB + v] = (bf16_t)(ww[i] & 0xffffu);
      tile[chan(8 * m + 2 * i + 1, h) * TSB + v] = (bf16_t)(ww[i] >> 16);
    }
  }
}
// weight gradient of one layer: acc[r] += sum_v A[chan(r, h)][v] B[j][v] from two transposed tiles
__device__ __forceinline__ f32x16 wgrad(const bf16_t* ta, const bf16_t* tb, int j, int h, f32x16 acc) {
  acc = CH_MFMA(tileT_get(ta, j, h, 0), tileT_get(tb, j, h, 0), acc);
  acc = CH_MFMA(tileT_get(ta, j, h, 1), tileT_get(tb, j, h, 1), acc);
  return acc;
}
// the same with a short second tile: its rows >= jb_max are one shared zero row (row jb_max)
__device__ __forceinline__ f32x16 wgrad_short(const bf16_t* ta, const bf16_t* tb, int j, int jb_max, int h,
                                              f32x16 acc) {
  const int jb = j < jb_max ? j : jb_max;
  acc = CH_MFMA(tileT_get(ta, j, h, 0), tileT_get(tb, jb, h, 0), acc);
  acc = CH_MFMA(tileT_get(ta, j, h, 1), tileT_get(tb, jb, h, 1), acc);
  return acc;
}
// ---- natural tiles + LDS transpose read (gfx950 ds_read_b64_tr_b16) ---------------------------------------------------
// [view][column] bf16 tile, row stride TSB: a lane writes its packed operand (columns 16 h .. 16 h + 15 = its
// accumulator registers in order) as two 16-byte stores instead of sixteen 2-byte ones; the MFMA operand of a k-block
// (lane l: row / column n = 16 (g & 1) + (l & 15), views 16 m + 8 (g >> 1) .. + 7, g = l >> 4) comes back through the
// transpose read: within a 16-lane group, lane (a, b) = 4 a + b receives element b of the 8-byte chunks addressed by the
// lanes (0, a) .. (3, a), so the lane 4 i + a points at the four columns 4 a .. 4 a + 3 of view i of the group's four.
// Column n of the image holds channel chan(n & 15, n >> 4) = cperm(n): outputs indexed by image columns go through it.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int cperm(int n) { return chan(n & 15, n >> 4); }
__device__ __forceinline__ void tileN_put_packed(bf16_t* tile, int v, int h, const bf16x8 (&a)[2]) {
  *reinterpret_cast<bf16x8*>(tile + v * TSB + 16 * h) = a[0];
  *reinterpret_cast<bf16x8*>(tile + v * TSB + 16 * h + 8) = a[1];
}
__device__ __forceinline__ bf16x8 tileN_get(const bf16_t* tile, int lane, int m) {
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const int ip = lane & 15, g = lane >> 4;
  const bf16_t* base = tile + (16 * m + 8 * (g >> 1) + (ip >> 2)) * TSB + 16 * (g & 1) + 4 * (ip & 3);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)base);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + 4 * TSB));
  const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
  const u32x4 v = {l2.x, l2.y, h2.x, h2.y};
  return __builtin_bit_cast(bf16x8, v);
}
// acc[r] += sum_v A[v][chan(r, h)] B[v][j] over image columns, both tiles natural
__device__ __forceinline__ f32x16 wgradN(const bf16_t* ta, const bf16_t* tb, int lane, f32x16 acc) {
  acc = CH_MFMA(tileN_get(ta, lane, 0), tileN_get(tb, lane, 0), acc);
  acc = CH_MFMA(tileN_get(ta, lane, 1), tileN_get(tb, lane, 1), acc);
  return acc;
}
// Sums over the 32 views of a natural tile this wavefront has just written (tileN_put_packed): lane (n, hh) receives the
// views 8 hh .. 8 hh + 7 and 16 + 8 hh .. of column n through the transpose read -- 16 in-lane additions; the two
// half-waves hold the two halves of the column sum (added at the flush).  Values as stored (bf16).
__device__ __forceinline__ void unpack8(const bf16x8& v, float (&f)[8]) {
  const u32x4 u = __builtin_bit_cast(u32x4, v);
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(w[i] << 16);
    f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void col_sum1(const bf16_t* tx, int lane, float& s) {        // sum x
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    float x[8];
    unpack8(tileN_get(tx, lane, m), x);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
  }
}
// first tile natural, second a transposed tile ([row][view]: indicator / short tiles; rows >= jb_max share row jb_max)
__device__ __forceinline__ f32x16 wgradN_T(const bf16_t* ta, const bf16_t* tb, int lane, int j, int jb_max, int h,
                                           f32x16 acc) {
  const int jb = j < jb_max ? j : jb_max;
  acc = CH_MFMA(tileN_get(ta, lane, 0), tileT_get(tb, jb, h, 0), acc);
  acc = CH_MFMA(tileN_get(ta, lane, 1), tileT_get(tb, jb, h, 1), acc);
  return acc;
}
// The block's D x D sums in s_red ([row][col], cols < ncol) -> global fp32 atomics.  Atomic REQUESTS (one per wavefront
// instruction and cache line touched) are served one after the other per line, ~4-9 ns each, by every block of the
// grid at the end of the kernel: the lanes of an instruction therefore cover consecutive addresses (measured on the
// [D][G] score-weight gradient: 8 scattered lanes per instruction = 64 requests per block = 113 us of a 254 us kernel;
// per-wavefront scalar atomics of the attention backward: 180 us of 340).
__device__ __forceinline__ void flush_red(float* __restrict__ out, int ld, int ncol, bool transpose,
                                          const float* s_red) {
  for (int i = threadIdx.x; i < ncol * D; i += blockDim.x) {
    const int row = transpose ? i % D : i / ncol, col = transpose ? i / D : i % ncol;
    atomicAdd(&out[transpose ? col * ld + row : row * ld + col], s_red[row * D + col]);
  }
}

// flush_matrix for accumulators whose rows (and, cols_nat, columns) are image columns of natural tiles
__device__ __forceinline__ void flush_matrix_nat(const f32x16& acc, float* __restrict__ out, int ld, int ncol,
                                                 bool transpose, float* s_red, bool cols_nat) {
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  __syncthreads();
  for (int i = threadIdx.x; i < D * D; i += blockDim.x) s_red[i] = 0.f;
  __syncthreads();
  if (j < ncol) {
    const int col = cols_nat ? cperm(j) : j;
#pragma unroll
    for (int r = 0; r < 16; ++r) atomicAdd(&s_red[cperm(chan(r, h)) * D + col], acc[r]);
  }
  __syncthreads();
  flush_red(out, ld, ncol, transpose, s_red);
}

// acc[r] = M[chan(r, h)][j] of every wavefront -> out[row * ld + col] (fp32 atomics), cols < ncol only
__device__ __forceinline__ void flush_matrix(const f32x16& acc, float* __restrict__ out, int ld, int ncol,
                                             bool transpose, float* s_red) {
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  __syncthreads();
  for (int i = threadIdx.x; i < D * D; i += blockDim.x) s_red[i] = 0.f;
  __syncthreads();
  if (j < ncol) {
#pragma unroll
    for (int r = 0; r < 16; ++r) atomicAdd(&s_red[chan(r, h) * D + j], acc[r]);
  }
  __syncthreads();
  flush_red(out, ld, ncol, transpose, s_red);
}


static inline int chain_grid(int blocks_per_cu) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
  }
  return cus * blocks_per_cu;
}

}  // namespace chain
}  // namespace dva

// ---- segments of a tile (lane j = view j; both half-waves hold the same bookkeeping) -------------------------
namespace dva {
namespace chain {

struct SegInfo {
  bool valid;        // lane has a view
  bool pd[4];        // scan step k: lane j - (1 << k) lies in the same 16-lane row and in the same point
  bool pb;           // row 1 lane whose point started in row 0 (takes the row-0 tail in the cross-row step)
  int ss, se;        // first / last lane of this lane's point inside the tile
  uint32_t smask;    // bit j: view j is the first view (in the tile) of its point
  uint32_t emask;    // bit j: view j is the last view (in the tile) of its point
  int nseg;          // points in the tile (uniform)
};

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_keep(int old, int x) {     // lanes without a source lane keep `old`
  return __builtin_amdgcn_update_dpp(old, x, CTRL, ROW_MASK, 0xf, false);
}

// Point structure of a tile from the view -> point index: neighbours through DPP wave shifts (both half-waves hold
// the same 32 views), first / last views through two ballots; no LDS crossbar operation.
__device__ __forceinline__ SegInfo seg_setup(int vpj, int j, int lane, int nv) {
  SegInfo s;
  s.valid = j < nv;
  const int lp = s.valid ? vpj : -1 - j;
  const int prv = dpp_keep<0x138>(lp, lp);      // wave_shr:1
  const int nxt = dpp_keep<0x130>(lp, lp);      // wave_shl:1
  const bool is_start = (j == 0) || (prv != lp);
  const bool is_end = (j == 31) || (nxt != lp);
  s.smask = (uint32_t)__ballot(is_start && s.valid);
  s.emask = (uint32_t)__ballot(is_end && s.valid);
  // invalid lanes: their own one-lane segments
  const uint32_t below = s.smask & (0xffffffffu >> (31 - j));
  s.ss = s.valid ? 31 - __clz((int)below) : j;
  s.se = s.valid ? j + (__ffs((int)(s.emask >> j)) - 1) : j;
  s.nseg = __popc(s.smask);
  const int dist = j - s.ss, jr = j & 15;
#pragma unroll
  for (int k = 0; k < 4; ++k) s.pd[k] = dist >= (1 << k) && jr >= (1 << k);
  s.pb = j >= 16 && dist > jr;
  return s;
}
// inclusive segmented scans over the 32 lanes of a half-wave (the last lane of a point ends up with the reduction over
// the point): four DPP row shifts inside the 16-lane rows, then lane 15 of row 0 / 2 broadcast into row 1 / 3
template <typename Op>
__device__ __forceinline__ float seg_scan(float v, const SegInfo& s, Op op) {
  float t;
  t = __int_as_float(dpp_keep<0x111>(__float_as_int(v), __float_as_int(v)));  v = s.pd[0] ? op(v, t) : v;
  t = __int_as_float(dpp_keep<0x112>(__float_as_int(v), __float_as_int(v)));  v = s.pd[1] ? op(v, t) : v;
  t = __int_as_float(dpp_keep<0x114>(__float_as_int(v), __float_as_int(v)));  v = s.pd[2] ? op(v, t) : v;
  t = __int_as_float(dpp_keep<0x118>(__float_as_int(v), __float_as_int(v)));  v = s.pd[3] ? op(v, t) : v;
  t = __int_as_float(dpp_keep<0x142, 0xa>(__float_as_int(v), __float_as_int(v)));   // row_bcast:15 into rows 1, 3
  return s.pb ? op(v, t) : v;
}
__device__ __forceinline__ float seg_scan_max(float v, const SegInfo& s, int lane) {
  return seg_scan(v, s, [](float a, float b) { return vmaxf(a, b); });
}
__device__ __forceinline__ float seg_scan_sum(float v, const SegInfo& s, int lane) {
  return seg_scan(v, s, [](float a, float b) { return a + b; });
}
// the sum scan with float masks (1 / 0 per step) instead of selects: v += shifted(v) * mask -- a DPP move (lanes without
// a source read 0) and one fma per step, which hipcc may fold into a single v_fmac_f32_dpp
struct SegMaskF {
  float pd[4], pb;
};
__device__ __forceinline__ SegMaskF seg_mask_f(const SegInfo& s) {
  SegMaskF m;
#pragma unroll
  for (int k = 0; k < 4; ++k) m.pd[k] = s.pd[k] ? 1.f : 0.f;
  m.pb = s.pb ? 1.f : 0.f;
  return m;
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_zero(float x) {       // lanes without a source lane (or outside ROW_MASK) read 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float seg_scan_sum_f(float v, const SegMaskF& m) {
  v = __builtin_fmaf(dpp_zero<0x111>(v), m.pd[0], v);
  v = __builtin_fmaf(dpp_zero<0x112>(v), m.pd[1], v);
  v = __builtin_fmaf(dpp_zero<0x114>(v), m.pd[2], v);
  v = __builtin_fmaf(dpp_zero<0x118>(v), m.pd[3], v);
  return __builtin_fmaf(dpp_zero<0x142, 0xa>(v), m.pb, v);     // row_bcast:15 into rows 1, 3
}
// value of the segment's last lane, in every lane of the segment
__device__ __forceinline__ float seg_total(float scanned, const SegInfo& s, int h) {
  return shfl(scanned, 32 * h + s.se);
}

// all-reduce over the 32 lanes of a half-wave without the LDS crossbar: quad_perm xor 1 / xor 2, row_half_mirror,
// row_mirror (DPP, folded into the VALU operation), then v_permlane16_swap for the two rows of the half
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), CTRL, 0xf, 0xf, false));
}
template <typename Op>
__device__ __forceinline__ float half_allreduce(float v, Op op) {
  v = op(v, dpp_mov<0xB1>(v));    // quad_perm [1, 0, 3, 2]
  v = op(v, dpp_mov<0x4E>(v));    // quad_perm [2, 3, 0, 1]
  v = op(v, dpp_mov<0x141>(v));   // row_half_mirror
  v = op(v, dpp_mov<0x140>(v));   // row_mirror
  const uint32_t x = __float_as_uint(v);
  const u32x2 r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  return op(__uint_as_float(r.x), __uint_as_float(r.y));
}
// max: fmaxf() makes hipcc quiet both operands first (v_max x, x) and keeps the DPP move separate -- five
// instructions per step; the DPP form of v_max_f32 is written out (hipcc pads nothing inside asm: the two wait states
// a DPP read needs after a VALU write of its source are the leading s_nop)
#define DVA_MAX_DPP(ctrl)                                                                          \
  asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v)); \
  v = r;
__device__ __forceinline__ float half_max(float v) {
  float r;
  DVA_MAX_DPP("quad_perm:[1,0,3,2]")
  DVA_MAX_DPP("quad_perm:[2,3,0,1]")
  DVA_MAX_DPP("row_half_mirror")
  DVA_MAX_DPP("row_mirror")
  asm volatile("s_nop 1" ::: );
  const uint32_t x = __float_as_uint(v);
  const u32x2 sw = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  return vmaxf(__uint_as_float(sw.x), __uint_as_float(sw.y));
}
#undef DVA_MAX_DPP
__device__ __forceinline__ float half_sum(float v) {
  return half_allreduce(v, [](float a, float b) { return a + b; });
}

// tanh for x >= 0 (the gate is tanh(relu(.))): 1 - 2 / (e^2x + 1), absolute error ~1e-7 (the result multiplies
// features that are stored in bf16); e^2x = inf gives exactly 1
__device__ __forceinline__ float tanh_pos(float x) {
  const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return __builtin_fmaf(-2.f, __builtin_amdgcn_rcpf(t + 1.f), 1.f);
}

}  // namespace chain
}  // namespace dva
