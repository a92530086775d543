// DeepSetFeat layer kernels on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 fma chain).
//
// Second generation of the row-streaming kernels of deepset.hip.  The first generation broadcast the
// inputs of a row to 32 lanes through LDS and spent 32 v_fmac + 8 ds_read_b128 per row pair: rocprofv3
// showed ~1000-1600 VALU and ~150-350 LDS instructions per 32-row tile, i.e. LDS/VALU-bound at ~2 TB/s
// of HBM traffic.  Here one wavefront owns a 32-view tile and every 32x32 product is 16 MFMA
// instructions whose operands are the registers the global loads landed in:
//
//   D[i][j] += sum_{kk<2} A[i][kk] * B[kk][j];  lane l supplies A[i=l&31][kk=l>>5], B[kk=l>>5][j=l&31];
//   lane l holds D[i=(r&3)+8*(r>>2)+4*(l>>5)][j=l&31] in register r (r < 16).
//
// view-major products (forward, dx):  j = view, i = output channel, k-step s pairs input channels
//   (s, s+16): lane (v, h) loads ITS half row X[v][16h .. 16h+16) as 4 float4 (BN + LeakyReLU applied in
//   registers) and lane (n, h) holds W[n][16h + s].  The result lands as 4 groups of 4 consecutive
//   channels per lane -> float4 stores.  Chaining a second layer needs no data movement: k-step r of
//   the next product pairs channels (n(r,0), n(r,1)), which is exactly register r of the two half-waves.
// channel-major product (weight gradient): i = n, j = k, k-step s pairs views (2s, 2s+1): plain
//   coalesced row loads, the 16 accumulators live across all tiles of the wavefront.
// No LDS on the data path; the matrix pipe needs 1024 cycles per product and tile, well below the HBM
// time of a pass, so these kernels are HBM-bound.
#include "dva_common.h"

namespace dva {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int DM = 32;
constexpr float SLOPE_M = 0.2f;

// slope < 1: leaky(z) = max(z, slope*z) -- one v_mul + one v_max, no compare / select
__device__ __forceinline__ float leaky_m(float z) { return fmaxf(z, SLOPE_M * z); }
__device__ __forceinline__ float dleaky_m(float z) { return z > 0.f ? 1.f : SLOPE_M; }
__device__ __forceinline__ int acc_chan(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// bn arrays are [4][32] = mean | invstd | gamma | beta.  Constants of the 16 channels of a half row
// (c = 16h + s) or of the accumulator layout (c = acc_chan(r, h)).
struct BN16 {
  float mean[16], invstd[16], gamma[16], beta[16];
};
__device__ __forceinline__ void load_bn_half(const float* __restrict__ bn, int h, BN16& b) {
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int c = 16 * h + s;
    b.mean[s] = bn[c]; b.invstd[s] = bn[DM + c]; b.gamma[s] = bn[2 * DM + c]; b.beta[s] = bn[3 * DM + c];
  }
}
__device__ __forceinline__ void load_bn_acc(const float* __restrict__ bn, int h, BN16& b) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = acc_chan(r, h);
    b.mean[r] = bn[c]; b.invstd[r] = bn[DM + c]; b.gamma[r] = bn[2 * DM + c]; b.beta[r] = bn[3 * DM + c];
  }
}

__device__ __forceinline__ void load16(const float* __restrict__ p, float (&x)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
    x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
  }
}
// accumulator layout <-> memory: 4 float4 at channels 8q + 4h
__device__ __forceinline__ void load_acc_layout(const float* __restrict__ row, int h, float (&x)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = *reinterpret_cast<const float4*>(row + 8 * q + 4 * h);
    x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
  }
}
__device__ __forceinline__ void store_acc_layout(float* __restrict__ row, int h, const f32x16& a) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<float4*>(row + 8 * q + 4 * h) =
        make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
}

__device__ __forceinline__ void store16(float* __restrict__ p, const float (&x)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<float4*>(p + 4 * q) = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
}

// bf16 storage of the [V, 32] activation / gradient tensors (arithmetic stays fp32): a half row is
// 32 bytes (2 x 16-byte accesses), a group of 4 accumulator channels 8 bytes.
__device__ __forceinline__ void unpack_bf16x2(uint32_t u, float& lo, float& hi) {
  lo = __uint_as_float(u << 16);
  hi = __uint_as_float(u & 0xffff0000u);
}
__device__ __forceinline__ void load16(const bf16_t* __restrict__ p, float (&x)[16]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const uint4 v = *reinterpret_cast<const uint4*>(p + 8 * q);
    unpack_bf16x2(v.x, x[8 * q], x[8 * q + 1]);
    unpack_bf16x2(v.y, x[8 * q + 2], x[8 * q + 3]);
    unpack_bf16x2(v.z, x[8 * q + 4], x[8 * q + 5]);
    unpack_bf16x2(v.w, x[8 * q + 6], x[8 * q + 7]);
  }
}
__device__ __forceinline__ void store16(bf16_t* __restrict__ p, const float (&x)[16]) {
#pragma unroll
  for (int q = 0; q < 2; ++q)
    *reinterpret_cast<uint4*>(p + 8 * q) =
        make_uint4(pack_bf16x2(x[8 * q], x[8 * q + 1]), pack_bf16x2(x[8 * q + 2], x[8 * q + 3]),
                   pack_bf16x2(x[8 * q + 4], x[8 * q + 5]), pack_bf16x2(x[8 * q + 6], x[8 * q + 7]));
}
__device__ __forceinline__ void load_acc_layout(const bf16_t* __restrict__ row, int h, float (&x)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint2 v = *reinterpret_cast<const uint2*>(row + 8 * q + 4 * h);
    unpack_bf16x2(v.x, x[4 * q], x[4 * q + 1]);
    unpack_bf16x2(v.y, x[4 * q + 2], x[4 * q + 3]);
  }
}
__device__ __forceinline__ void store_acc_layout(bf16_t* __restrict__ row, int h, const f32x16& a) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<uint2*>(row + 8 * q + 4 * h) =
        make_uint2(pack_bf16x2(a[4 * q], a[4 * q + 1]), pack_bf16x2(a[4 * q + 2], a[4 * q + 3]));
}

// Sum vals[r] over the 32 lanes of a half-wave; lanes 0 / 32 then add channel acc_chan(r,h) to s_red.
__device__ __forceinline__ void reduce_acc_channels(float (&vals)[16], float* s_red, int lane) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = vals[r];
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off);
    if ((lane & 31) == 0) atomicAdd(&s_red[acc_chan(r, lane >> 5)], v);
  }
}

template <int NV>
__device__ __forceinline__ void flush_stats(float (&st)[NV][16], double* __restrict__ out, float* s_red,
                                            int lane) {
  for (int i = threadIdx.x; i < NV * DM; i += blockDim.x) s_red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NV; ++j) reduce_acc_channels(st[j], s_red + j * DM, lane);
  __syncthreads();
  for (int i = threadIdx.x; i < NV * DM; i += blockDim.x) atomicAdd(&out[i], (double)s_red[i]);
}

#define DVA_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// bf16 matrix cores for the bf16-storage instantiations.  v_mfma_f32_32x32x16_bf16 does 8x the work of the
// fp32 instruction in half the cycles; rocprofv3 (profiles/r01c_sq_counters_fp32_mfma.txt) showed the fp32 chain as
// the limiter once the traffic was halved: 55-68 % of the wave cycles were MFMA issue stalls.  fp32
// accuracy is kept with the split  x = hi + lo  (hi = bf16(x), lo = bf16(x - hi)) and three products
// hi*hi + lo*hi + hi*lo (the dropped lo*lo term is < 2^-16 relative).  Lane l supplies the 8 k-slots
// 8*(l>>5) .. +7 of row / column l & 31; the accumulator layout is that of the fp32 instruction.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define DVA_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void split8(const float* x, bf16x8& hi, bf16x8& lo) {
  uint32_t H[4], L[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    H[p] = pack_bf16x2(x[2 * p], x[2 * p + 1]);
    const float h0 = __uint_as_float(H[p] << 16), h1 = __uint_as_float(H[p] & 0xffff0000u);
    L[p] = pack_bf16x2(x[2 * p] - h0, x[2 * p + 1] - h1);
  }
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  const u4 hv = {H[0], H[1], H[2], H[3]}, lv = {L[0], L[1], L[2], L[3]};
  hi = __builtin_bit_cast(bf16x8, hv);
  lo = __builtin_bit_cast(bf16x8, lv);
}
__device__ __forceinline__ bf16x8 round8(const float* x) {
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  const u4 v = {pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]), pack_bf16x2(x[4], x[5]),
                pack_bf16x2(x[6], x[7])};
  return __builtin_bit_cast(bf16x8, v);
}

// View-major 32x32 product  D[n][view] += sum_c W[n][c] X[c][view]:  lane (j, h) holds w[s] = W[n = j][c(s,h)]
// and x[s] = X[c(s,h)][view = j] for the same 16 channels c(s, h) (half-row or accumulator layout).
template <typename AT> struct ViewProd;
template <> struct ViewProd<float> {
  float w[16];
  __device__ __forceinline__ void prep(const float (&wf)[16]) {
#pragma unroll
    for (int s = 0; s < 16; ++s) w[s] = wf[s];
  }
  __device__ __forceinline__ f32x16 mul(const float (&x)[16], f32x16 acc) const {
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = DVA_MFMA(w[s], x[s], acc);
    return acc;
  }
};
template <> struct ViewProd<bf16_t> {
  bf16x8 hi[2], lo[2];
  __device__ __forceinline__ void prep(const float (&wf)[16]) {
    split8(&wf[0], hi[0], lo[0]);
    split8(&wf[8], hi[1], lo[1]);
  }
  __device__ __forceinline__ f32x16 mul(const float (&x)[16], f32x16 acc) const {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      bf16x8 xh, xl;
      split8(&x[8 * m], xh, xl);
      acc = DVA_MFMA_BF16(hi[m], xh, acc);
      acc = DVA_MFMA_BF16(lo[m], xh, acc);
      acc = DVA_MFMA_BF16(hi[m], xl, acc);
    }
    return acc;
  }
};
// The same with 4 channels per lane (x_map rows: 8 features).
template <typename AT> struct ViewProd4;
template <> struct ViewProd4<float> {
  float w[4];
  __device__ __forceinline__ void prep(const float (&wf)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) w[s] = wf[s];
  }
  __device__ __forceinline__ f32x16 mul(const float4& x, f32x16 acc) const {
    acc = DVA_MFMA(w[0], x.x, acc);
    acc = DVA_MFMA(w[1], x.y, acc);
    acc = DVA_MFMA(w[2], x.z, acc);
    return DVA_MFMA(w[3], x.w, acc);
  }
};
template <> struct ViewProd4<bf16_t> {
  bf16x8 hi, lo;
  __device__ __forceinline__ void prep(const float (&wf)[4]) {
    const float w8[8] = {wf[0], wf[1], wf[2], wf[3], 0.f, 0.f, 0.f, 0.f};
    split8(w8, hi, lo);
  }
  __device__ __forceinline__ f32x16 mul(const float4& x, f32x16 acc) const {
    const float x8[8] = {x.x, x.y, x.z, x.w, 0.f, 0.f, 0.f, 0.f};
    bf16x8 xh, xl;
    split8(x8, xh, xl);
    acc = DVA_MFMA_BF16(hi, xh, acc);
    acc = DVA_MFMA_BF16(lo, xh, acc);
    return DVA_MFMA_BF16(hi, xl, acc);
  }
};

// Intra-wavefront LDS hand-off (DS ops of one wavefront execute in order; see deepset.hip).
__device__ __forceinline__ void wave_sync_m() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
// 32x32 fp32 tile staged per wavefront for the view-major -> channel-major transposition.
// Row stride 36 floats: 16-byte aligned rows, float4 writes of 8 consecutive lanes hit 32 distinct
// banks, and the column reads (lane = channel) are conflict free.
constexpr int TS = 36;
__device__ __forceinline__ void tile_put_half(float* tile, int v, int h, const float (&x)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<float4*>(tile + v * TS + 16 * h + 4 * q) =
        make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
}
__device__ __forceinline__ void tile_put_acc(float* tile, int v, int h, const float (&x)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<float4*>(tile + v * TS + 8 * q + 4 * h) =
        make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
}

// bf16 instantiations: the tile is staged TRANSPOSED ([channel][view], bf16, row stride 40 = 80 bytes) so
// that the 8 consecutive views a lane feeds to v_mfma_f32_32x32x16_bf16 are one ds_read_b128.  The weight
// gradient takes the operands rounded to bf16 (no split): it is a sum over all V views, the rounding
// errors are unbiased and average out (the reference under autocast rounds the same operands).
constexpr int TSB = 40;
__device__ __forceinline__ void tileT_put(bf16_t* tile, int c0, int c1, int v, float x0, float x1) {
  const uint32_t d = pack_bf16x2(x0, x1);
  tile[c0 * TSB + v] = (bf16_t)(d & 0xffffu);
  tile[c1 * TSB + v] = (bf16_t)(d >> 16);
}
__device__ __forceinline__ void tileT_put_half(bf16_t* tile, int v, int h, const float (&x)[16]) {
#pragma unroll
  for (int s = 0; s < 16; s += 2) tileT_put(tile, 16 * h + s, 16 * h + s + 1, v, x[s], x[s + 1]);
}
__device__ __forceinline__ void tileT_put_acc(bf16_t* tile, int v, int h, const float (&x)[16]) {
#pragma unroll
  for (int r = 0; r < 16; r += 2) tileT_put(tile, acc_chan(r, h), acc_chan(r + 1, h), v, x[r], x[r + 1]);
}
// views 16m + 8h .. +7 of channel c: the 8 k-slots of lane (c, h) in MFMA m
__device__ __forceinline__ bf16x8 tileT_get(const bf16_t* tile, int c, int h, int m) {
  return *reinterpret_cast<const bf16x8*>(tile + c * TSB + 16 * m + 8 * h);
}

// Tile loop: every full 32-row tile runs a body instantiated with TAIL = false (no predication at all:
// the first versions predicated each load / store, which made hipcc (a) branch around every access and
// (b) lose count of the outstanding accesses and fall back to s_waitcnt vmcnt(0) -- i.e. wait for the
// tile's STORES before touching the prefetched rows).  The one partial tile runs the TAIL = true body.
struct TailNo { static constexpr bool value = false; };
struct TailYes { static constexpr bool value = true; };
template <typename Body>
__device__ __forceinline__ void for_each_tile(int64_t V, Body&& body) {
  // (a two-body variant -- unpredicated full tiles + a predicated tail body -- doubled the register
  // footprint; one body with clamped row indices costs a v_min per address and keeps loads branch-free)
  const int64_t tiles = (V + 31) / 32;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave; t < tiles; t += n_waves) body(t, TailYes());
}

// Raw (not yet converted) registers of half a row / of a row in the accumulator layout, so that the loads
// of the NEXT tile can be in flight while the current one is computed (for_each_tile_pf).
template <typename AT> struct HalfRow;
template <> struct HalfRow<float> { float4 q[4]; };
template <> struct HalfRow<bf16_t> { uint4 q[2]; };
template <typename AT> struct AccRow;
template <> struct AccRow<float> { float4 q[4]; };
template <> struct AccRow<bf16_t> { uint4 q[2]; };  // raw half row; half-waves exchange at unpack

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// The <= 32 rows [row0, row0 + 32) of a row-major array with `row_bytes` per row, as a raw buffer
// resource (row0 is wave-uniform: the descriptor lives in SGPRs).  The hardware bounds check replaces
// every predicate of the partial last tile: loads past the last row return 0, stores are dropped.  That
// matters beyond the saved compares -- a store inside an `if (row < V)` branch makes hipcc lose count of
// the outstanding memory operations and emit s_waitcnt vmcnt(0), which also waits for the prefetch of
// the next tile.
struct RowTile {
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ RowTile(const void* base, int64_t row0, int64_t V, int row_bytes) {
    int64_t rows = V - row0;
    rows = rows > 32 ? 32 : (rows < 0 ? 0 : rows);
    // readfirstlane: tell the compiler the descriptor is wave-uniform (else every access is wrapped in
    // a waterfall loop)
    const uint64_t addr = (uint64_t)base + (uint64_t)(row0 * row_bytes);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)addr);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
    const int n = __builtin_amdgcn_readfirstlane((int)rows * row_bytes);
    r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
  }
  __device__ __forceinline__ uint32_t b32(int off) const { return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0); }
  __device__ __forceinline__ u32x2 b64(int off) const { return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0); }
  __device__ __forceinline__ u32x4 b128(int off) const { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); }
  __device__ __forceinline__ void st64(int off, u32x2 v) const { __builtin_amdgcn_raw_buffer_store_b64(v, r, off, 0, 0); }
  __device__ __forceinline__ void st128(int off, u32x4 v) const { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0); }
};
__device__ __forceinline__ float4 as_f4(u32x4 v) { return __builtin_bit_cast(float4, v); }
__device__ __forceinline__ u32x4 as_u4(float a, float b, float c, float d) {
  const float4 f = make_float4(a, b, c, d);
  return __builtin_bit_cast(u32x4, f);
}
constexpr int OOB = 0x7ffffff0;  // byte offset no tile reaches: a store there is dropped (branch-free masking)

template <typename AT> __device__ __forceinline__ HalfRow<AT> tile_load_half(const RowTile& T, int j, int h);
template <> __device__ __forceinline__ HalfRow<float> tile_load_half<float>(const RowTile& T, int j, int h) {
  HalfRow<float> r;
#pragma unroll
  for (int q = 0; q < 4; ++q) r.q[q] = as_f4(T.b128(j * 128 + h * 64 + 16 * q));
  return r;
}
template <> __device__ __forceinline__ HalfRow<bf16_t> tile_load_half<bf16_t>(const RowTile& T, int j, int h) {
  HalfRow<bf16_t> r;
#pragma unroll
  for (int q = 0; q < 2; ++q) r.q[q] = __builtin_bit_cast(uint4, T.b128(j * 64 + h * 32 + 16 * q));
  return r;
}
template <typename AT> __device__ __forceinline__ AccRow<AT> tile_load_acc(const RowTile& T, int j, int h);
template <> __device__ __forceinline__ AccRow<float> tile_load_acc<float>(const RowTile& T, int j, int h) {
  AccRow<float> r;
#pragma unroll
  for (int q = 0; q < 4; ++q) r.q[q] = as_f4(T.b128(j * 128 + (8 * q + 4 * h) * 4));
  return r;
}
// bf16 rows are 64 bytes: in the accumulator layout a lane owns 4 groups of 4 channels = 4 x 8 bytes with
// a 16-byte stride, and 8-byte accesses cost twice the issue slots of 16-byte ones.  Instead every lane
// moves its contiguous HALF row (2 x 16 bytes) and the two half-waves trade two groups with
// v_permlane32_swap (lanes [32,64) of the first operand <-> lanes [0,32) of the second):
//   half row of (j, h=0) = channels 0..15 = own q0 | partner q0 | own q1 | partner q1
//   half row of (j, h=1) = channels 16..31 = partner q2 | own q2 | partner q3 | own q3
__device__ __forceinline__ void swap_halves(uint32_t& a, uint32_t& b) {
  const u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r.x;
  b = r.y;
}
template <> __device__ __forceinline__ AccRow<bf16_t> tile_load_acc<bf16_t>(const RowTile& T, int j, int h) {
  AccRow<bf16_t> r;
#pragma unroll
  for (int q = 0; q < 2; ++q) r.q[q] = __builtin_bit_cast(uint4, T.b128(j * 64 + h * 32 + 16 * q));
  return r;
}
template <typename AT> __device__ __forceinline__ void tile_store_acc(const RowTile& T, int j, int h, const f32x16& a);
template <> __device__ __forceinline__ void tile_store_acc<float>(const RowTile& T, int j, int h, const f32x16& a) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    T.st128(j * 128 + (8 * q + 4 * h) * 4, as_u4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]));
}
template <> __device__ __forceinline__ void tile_store_acc<bf16_t>(const RowTile& T, int j, int h, const f32x16& a) {
  uint32_t p[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    p[q][0] = pack_bf16x2(a[4 * q], a[4 * q + 1]);
    p[q][1] = pack_bf16x2(a[4 * q + 2], a[4 * q + 3]);
  }
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    swap_halves(p[0][d], p[2][d]);
    swap_halves(p[1][d], p[3][d]);
  }
  const u32x4 v0 = {p[0][0], p[0][1], p[2][0], p[2][1]};
  const u32x4 v1 = {p[1][0], p[1][1], p[3][0], p[3][1]};
  T.st128(j * 64 + h * 32, v0);
  T.st128(j * 64 + h * 32 + 16, v1);
}
template <typename AT> __device__ __forceinline__ void tile_store_half(const RowTile& T, int j, int h, const float (&x)[16]);
template <> __device__ __forceinline__ void tile_store_half<float>(const RowTile& T, int j, int h, const float (&x)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    T.st128(j * 128 + h * 64 + 16 * q, as_u4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]));
}
template <> __device__ __forceinline__ void tile_store_half<bf16_t>(const RowTile& T, int j, int h, const float (&x)[16]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const u32x4 v = {pack_bf16x2(x[8 * q], x[8 * q + 1]), pack_bf16x2(x[8 * q + 2], x[8 * q + 3]),
                     pack_bf16x2(x[8 * q + 4], x[8 * q + 5]), pack_bf16x2(x[8 * q + 6], x[8 * q + 7])};
    T.st128(j * 64 + h * 32 + 16 * q, v);
  }
}
__device__ __forceinline__ void unpack(const HalfRow<float>& r, float (&x)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    x[4 * q] = r.q[q].x; x[4 * q + 1] = r.q[q].y; x[4 * q + 2] = r.q[q].z; x[4 * q + 3] = r.q[q].w;
  }
}
__device__ __forceinline__ void unpack(const HalfRow<bf16_t>& r, float (&x)[16]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    unpack_bf16x2(r.q[q].x, x[8 * q], x[8 * q + 1]);
    unpack_bf16x2(r.q[q].y, x[8 * q + 2], x[8 * q + 3]);
    unpack_bf16x2(r.q[q].z, x[8 * q + 4], x[8 * q + 5]);
    unpack_bf16x2(r.q[q].w, x[8 * q + 6], x[8 * q + 7]);
  }
}
__device__ __forceinline__ void unpack(const AccRow<float>& r, float (&x)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    x[4 * q] = r.q[q].x; x[4 * q + 1] = r.q[q].y; x[4 * q + 2] = r.q[q].z; x[4 * q + 3] = r.q[q].w;
  }
}
__device__ __forceinline__ void unpack(const AccRow<bf16_t>& r, float (&x)[16]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    // (A, B) = the two 8-byte groups of 16 bytes: after the swap A = group q = i, B = group q = i + 2
    uint32_t a0 = r.q[i].x, a1 = r.q[i].y, b0 = r.q[i].z, b1 = r.q[i].w;
    swap_halves(a0, b0);
    swap_halves(a1, b1);
    unpack_bf16x2(a0, x[4 * i], x[4 * i + 1]);
    unpack_bf16x2(a1, x[4 * i + 2], x[4 * i + 3]);
    unpack_bf16x2(b0, x[4 * (i + 2)], x[4 * (i + 2) + 1]);
    unpack_bf16x2(b1, x[4 * (i + 2) + 2], x[4 * (i + 2) + 3]);
  }
}

// Software-pipelined tile loop: load(t) issues the global loads of tile t and returns the raw registers,
// compute(t, raw) consumes them.  The loads of the wavefront's next tile are issued before the current
// tile is computed, so one memory latency per tile is hidden behind the MFMA / VALU work of the previous
// one (3 wavefronts per SIMD do not hide it on their own).  Past the last tile the prefetch re-reads the
// last tile (valid memory, result unused).
template <bool PIPE = true, typename Load, typename Compute>
__device__ __forceinline__ void for_each_tile_pf(int64_t V, Load&& load, Compute&& compute) {
  const int64_t tiles = (V + 31) / 32;
  // readfirstlane: the tile index must be provably wave-uniform (buffer descriptors live in SGPRs)
  const int64_t wave = __builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t last = tiles - 1;
  if (!PIPE) {
    // plain loop for bodies with memory operations under divergent control flow (run-length atomics):
    // hipcc waits with vmcnt(0) there anyway, a second register set would only cost occupancy
    for (int64_t t = wave; t < tiles; t += n_waves) compute(t, load(t));
    return;
  }
  if (wave < tiles) {
    // Two register sets in ping-pong (a rotating `cur = nxt` copy would make the loop wait for the
    // prefetch it has just issued: the copy reads the registers the loads are landing in).  The first
    // tile is peeled so that the loop is ENTERED in the same memory state as it is RE-ENTERED (one
    // prefetch in flight, then the stores of a tile): hipcc's s_waitcnt insertion merges the two paths
    // conservatively, and with a plain prologue it waited for the previous tile's stores at the loop top.
    auto a = load(wave);
    int64_t t = wave + n_waves;
    auto b = load(t < tiles ? t : last);
    compute(wave, a);
    while (t < tiles) {
      a = load(t + n_waves < tiles ? t + n_waves : last);
      compute(t, b);
      t += n_waves;
      if (t >= tiles) break;
      b = load(t + n_waves < tiles ? t + n_waves : last);
      compute(t, a);
      t += n_waves;
    }
  }
}

// BatchNorm constants as an LDS table [4][32] (mean | invstd | gamma | beta): 16 ds_read_b128 per tile
// instead of 64 live registers per lane (occupancy).  `base` = first channel of each group of 4:
// half-row layout 16h + 4q, accumulator layout 8q + 4h.
__device__ __forceinline__ void stage_bn(float (*t)[DM], const float* __restrict__ bn) {
  for (int i = threadIdx.x; i < 4 * DM; i += blockDim.x) t[i / DM][i % DM] = bn ? bn[i] : ((i / DM) == 1 || (i / DM) == 2 ? 1.f : 0.f);
}
template <bool ACC_LAYOUT>
__device__ __forceinline__ void bn_norm16(const float (*t)[DM], int h, const float (&x)[16], float (&ah)[16],
                                          float (&z)[16]) {
  // keep the table in LDS: without the barrier hipcc hoists the 64 constants of a lane out of the tile
  // loop into registers, which costs a whole occupancy step (16 ds_read_b128 per tile are nothing)
  asm volatile("" ::: "memory");
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int base = ACC_LAYOUT ? 8 * q + 4 * h : 16 * h + 4 * q;
    const float4 m = *reinterpret_cast<const float4*>(&t[0][base]);
    const float4 iv = *reinterpret_cast<const float4*>(&t[1][base]);
    const float4 g = *reinterpret_cast<const float4*>(&t[2][base]);
    const float4 b = *reinterpret_cast<const float4*>(&t[3][base]);
    ah[4 * q] = (x[4 * q] - m.x) * iv.x;         z[4 * q] = ah[4 * q] * g.x + b.x;
    ah[4 * q + 1] = (x[4 * q + 1] - m.y) * iv.y; z[4 * q + 1] = ah[4 * q + 1] * g.y + b.y;
    ah[4 * q + 2] = (x[4 * q + 2] - m.z) * iv.z; z[4 * q + 2] = ah[4 * q + 2] * g.z + b.z;
    ah[4 * q + 3] = (x[4 * q + 3] - m.w) * iv.w; z[4 * q + 3] = ah[4 * q + 3] * g.w + b.w;
  }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------

// x_map [V,8] -> a1 (stats only) or -> a1 -> BN1 -> leaky -> a2 (written, with stats)
template <typename AT, bool STATS_ONLY>
__global__ __launch_bounds__(256) void dsm_fwd_first_kernel(const float* __restrict__ x_map,
                                                             const float* __restrict__ Wa,
                                                             const float* __restrict__ bn1,
                                                             const float* __restrict__ Wb,
                                                             AT* __restrict__ a2,
                                                             double* __restrict__ stats, int64_t V) {
  __shared__ float s_red[2 * DM];
  __shared__ __attribute__((aligned(16))) float s_bn[4][DM];
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  ViewProd4<AT> pa;  // Wa[n=j][4h + s]
  {
    float wa[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) wa[s] = Wa[j * 8 + 4 * h + s];
    pa.prep(wa);
  }
  ViewProd<AT> pb;  // Wb[n2=j][acc_chan(r,h)]
  if (!STATS_ONLY) {
    float wb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) wb[r] = Wb[j * DM + acc_chan(r, h)];
    pb.prep(wb);
    stage_bn(s_bn, bn1);
    __syncthreads();
  }
  float st[2][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) st[0][r] = st[1][r] = 0.f;
  for_each_tile_pf<sizeof(AT) == 2>(V, [&](int64_t t) {
    return as_f4(RowTile(x_map, t * 32, V, 32).b128(j * 32 + h * 16));   // rows beyond V read as zeros
  }, [&](int64_t t, float4 x) {
    const bool ok = t * 32 + j < V;
    f32x16 acc = {0};
    acc = pa.mul(x, acc);
    if (STATS_ONLY) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[0][r] += acc[r];            // rows beyond V are exact zeros
        st[1][r] = fmaf(acc[r], acc[r], st[1][r]);
      }
    } else {
      f32x16 acc2 = {0};
      float a1v[16], ah1[16], z1[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) a1v[r] = acc[r];
      bn_norm16<true>(s_bn, h, a1v, ah1, z1);
      float xin[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) xin[r] = ok ? leaky_m(z1[r]) : 0.f;
      acc2 = pb.mul(xin, acc2);
      tile_store_acc<AT>(RowTile(a2, t * 32, V, DM * (int)sizeof(AT)), j, h, acc2);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[0][r] += acc2[r];
        st[1][r] = fmaf(acc2[r], acc2[r], st[1][r]);
      }
    }
  });
  flush_stats<2>(st, stats, s_red, lane);
}

// a_out[v] = leaky(BN_in(a_in[v])) . W^T (+ addend[vp[v]]) ; statistics of a_out.
// SCORE: W has G <= 32 valid rows (others zero), bias added, only columns < G stored, no statistics.
// STORE = false: statistics of a_out only (the activation is recomputed by its consumers instead of being
//   stored: one [V, 32] write now and one read per consumer saved).
// PRE (with SCORE): the scores are those of the layer AFTER this one: a_mid = leaky(BN_in(a_in)).W^T is
//   recomputed in registers (it was never stored) and out = leaky(BN2(a_mid)).W2^T + bias.
template <typename AT, bool HAS_ADD, bool SCORE, bool STORE = true, bool PRE = false>
__global__ __launch_bounds__(256, (HAS_ADD ? 2 : 3)) void dsm_fwd_layer_kernel(
    const AT* __restrict__ a_in, const float* __restrict__ bn_in, const float* __restrict__ W,
    const float* __restrict__ addend, const int32_t* __restrict__ vp, const float* __restrict__ bias,
    void* __restrict__ a_out_, double* __restrict__ stats, int64_t V, int G,
    const float* __restrict__ bn2, const float* __restrict__ W2) {
  static_assert(!PRE || SCORE, "PRE is a score variant");
  __shared__ __attribute__((aligned(16))) float s_bn2[PRE ? 4 : 1][DM];
  // SCORE writes the fp32 compatibilities [V, G]; layers write activations in the storage type
  float* __restrict__ c_out = reinterpret_cast<float*>(a_out_);
  AT* __restrict__ a_out = reinterpret_cast<AT*>(a_out_);
  __shared__ float s_red[2 * DM];
  __shared__ __attribute__((aligned(16))) float s_bn[4][DM];
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const bool ident = bn_in == nullptr;  // raw input (no BatchNorm / activation), e.g. pooled set features
  stage_bn(s_bn, bn_in);
  if (PRE) stage_bn(s_bn2, bn2);
  __syncthreads();
  ViewProd<AT> pw;  // W[n=j][16h + s]
  {
    float w[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) w[s] = (!SCORE || PRE || j < G) ? W[j * DM + 16 * h + s] : 0.f;
    pw.prep(w);
  }
  ViewProd<AT> pw2;  // PRE: W2[g=j][acc_chan(r, h)] (the recomputed layer comes out in the accumulator layout)
  if (PRE) {
    float w[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) w[r] = j < G ? W2[j * DM + acc_chan(r, h)] : 0.f;
    pw2.prep(w);
  }
  float bia[16];
  if (SCORE) {
#pragma unroll
    for (int r = 0; r < 16; ++r) bia[r] = acc_chan(r, h) < G ? bias[acc_chan(r, h)] : 0.f;
  }
  float st[2][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) st[0][r] = st[1][r] = 0.f;
  struct Raw {
    HalfRow<AT> x;
    int32_t p;
  };
  constexpr int RB = DM * (int)sizeof(AT);
  for_each_tile_pf<sizeof(AT) == 2>(V, [&](int64_t t) {
    Raw r;
    r.x = tile_load_half<AT>(RowTile(a_in, t * 32, V, RB), j, h);
    r.p = HAS_ADD ? (int32_t)RowTile(vp, t * 32, V, 4).b32(j * 4) : 0;
    return r;
  }, [&](int64_t t, const Raw& raw) {
    const bool ok = t * 32 + j < V;
    float x[16];
    unpack(raw.x, x);
    // the addend row is shared by the views of a point (mostly one point per tile): L2 hit
    float ad[16];
    if (HAS_ADD) load_acc_layout(addend + (int64_t)raw.p * DM, h, ad);
    f32x16 acc = {0};
    float ahx[16], zx[16];
    bn_norm16<false>(s_bn, h, x, ahx, zx);
    float xin[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) xin[s] = ok ? (ident ? x[s] : leaky_m(zx[s])) : 0.f;
    acc = pw.mul(xin, acc);
    if (PRE) {
      float am[16], ah2[16], z2[16], x2[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) am[r] = acc[r];
      bn_norm16<true>(s_bn2, h, am, ah2, z2);
#pragma unroll
      for (int r = 0; r < 16; ++r) x2[r] = leaky_m(z2[r]);
      f32x16 acc2 = {0};
      acc = pw2.mul(x2, acc2);
    }
    if (SCORE) {
      // scores [V, G] fp32; lanes / registers without a score column store out of bounds (dropped)
      const RowTile C(c_out, t * 32, V, G * 4);
      if (G == 4) {
        C.st128(h == 0 ? j * 16 : OOB, as_u4(acc[0] + bia[0], acc[1] + bia[1], acc[2] + bia[2], acc[3] + bia[3]));
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = acc_chan(r, h);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[r] + bia[r]), C.r,
                                                c < G ? (j * G + c) * 4 : OOB, 0, 0);
        }
      }
    } else {
      if (HAS_ADD) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += ad[r];
      }
      if (STORE) tile_store_acc<AT>(RowTile(a_out, t * 32, V, RB), j, h, acc);
      if (ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          st[0][r] += acc[r];
          st[1][r] = fmaf(acc[r], acc[r], st[1][r]);
        }
      }
    }
  });
  if (!SCORE) flush_stats<2>(st, stats, s_red, lane);
}

// ------------------------------------------------------------------------------------------------
// backward (same contract as dsf_bwd_layer_kernel in deepset.hip)
// ------------------------------------------------------------------------------------------------
// FUSE1 (bf16 storage, PREV_XMAP): the first layer's weight gradient is folded in.  BN1-backward needs the
// statistics this very pass produces, but it is linear in them:
//   dWa[n][f] = sum_v da1[v][n] x[v][f],  da1 = gsc (dz1 - S1/M - a1_hat S2/M)
//             = gsc[n] (P[n][f] - (S1/M)[n] SX[f] - (S2/M)[n] Q[n][f])
//   P = sum_v dz1 x^T,  Q = sum_v a1_hat x^T,  SX = sum_v x     (first = P[32][8] | Q[32][8] | SX[8])
// so P, Q, SX are accumulated here (two more channel-major bf16 products per tile) and dz1 is never
// written: saves the dva_deepset_bwd_first pass and 2 x V x 64 bytes of traffic.
// RC (bf16 storage): a_L is not read but RECOMPUTED from the layer input, a_L = leaky(BN_prev(a_prev)).W_L^T
//   (+ addend[point] for the concatenation layer): 3 tensors per view instead of 4 cross the memory system
//   for one more product on the (idle) matrix cores.  Everything then lives in the accumulator layout:
//   dz_L is loaded in it, da and the dx product use W_L rows indexed by acc_chan.
template <typename AT, bool PREV_XMAP, bool RAW_OUT, bool HAS_DT, bool FUSE1 = false, bool RC = false>
__global__ __launch_bounds__(256, 2) void dsm_bwd_layer_kernel(
    const AT* __restrict__ dz_L, const AT* __restrict__ a_L, const float* __restrict__ bn_L,
    const float* __restrict__ sm_L, const float* __restrict__ W_L, const void* __restrict__ a_prev_,
    const float* __restrict__ Wa, const float* __restrict__ bn_prev, AT* __restrict__ out,
    float* __restrict__ dW, double* __restrict__ st_prev, float* __restrict__ dt,
    const int32_t* __restrict__ vp, float* __restrict__ first, const float* __restrict__ addend, int64_t V) {
  static_assert(!FUSE1 || (PREV_XMAP && !RAW_OUT && !HAS_DT && sizeof(AT) == 2), "FUSE1 variant");
  static_assert(!RC || sizeof(AT) == 2, "RC variant: bf16 storage");
  // per-wavefront transposed bf16 tiles of the fused first-layer products: dz1 | a1_hat | x (32 rows each)
  __shared__ __attribute__((aligned(16))) bf16_t s_f1[FUSE1 ? 4 : 1][FUSE1 ? 96 * TSB : 8];
  // the layer input: raw x_map rows (fp32 [V, 8]) or activations in the storage type
  const float* __restrict__ x_prev = reinterpret_cast<const float*>(a_prev_);
  const AT* __restrict__ a_prev = reinterpret_cast<const AT*>(a_prev_);
  __shared__ float s_red[DM * DM];
  __shared__ __attribute__((aligned(16))) float s_c[5][DM];  // BN_L: gsc | mean | invstd | S1/M | S2/M
  __shared__ __attribute__((aligned(16))) float s_p[4][DM];  // BN_prev table
  __shared__ __attribute__((aligned(16))) float s_da[4][32 * TS];  // per-wavefront da tile
  __shared__ __attribute__((aligned(16))) float s_x[4][32 * TS];   // per-wavefront x_L tile
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  if (threadIdx.x < DM) {
    const int c = threadIdx.x;
    s_c[0][c] = bn_L[2 * DM + c] * bn_L[DM + c];
    s_c[1][c] = bn_L[c];
    s_c[2][c] = bn_L[DM + c];
    s_c[3][c] = sm_L[c];
    s_c[4][c] = sm_L[DM + c];
  }
  stage_bn(s_p, bn_prev);
  __syncthreads();
  const bool pident = bn_prev == nullptr;  // the layer input is raw (only valid with RAW_OUT)
  constexpr bool BF = sizeof(AT) == 2;
  ViewProd<AT> pt;  // W_L[n = 16h + s][k = j]  (RC: n = acc_chan(s, h))
  {
    float wt[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) wt[s] = W_L[(RC ? acc_chan(s, h) : 16 * h + s) * DM + j];
    pt.prep(wt);
  }
  ViewProd<AT> pf;  // RC: forward weights W_L[n = j][c = acc_chan(r, h)]
  if (RC) {
    float wf[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) wf[r] = W_L[j * DM + acc_chan(r, h)];
    pf.prep(wf);
  }
  ViewProd4<AT> p4;
  if (PREV_XMAP) {
    float wa4[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) wa4[s] = Wa[j * 8 + 4 * h + s];  // view-major recompute of a1
    p4.prep(wa4);
  }
  f32x16 accW = {0};
  f32x16 accP = {0}, accQ = {0};
  float4 sx4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float st[2][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) st[0][r] = st[1][r] = 0.f;
  float* tda = s_da[wv];
  float* tx = s_x[wv];
  bf16_t* f1 = s_f1[FUSE1 ? wv : 0];
  if (FUSE1) {
    // x^T tile: only the 8 feature rows are rewritten per tile, rows 8..31 stay zero
    uint32_t* z = reinterpret_cast<uint32_t*>(f1 + 64 * TSB);
    for (int i = lane; i < 32 * TSB / 2; i += 64) z[i] = 0u;
    wave_sync_m();
  }

  struct Raw {
    HalfRow<AT> dz, al;      // (RC: dz holds the row in the accumulator layout, al is unused)
    AccRow<AT> ap;
    float4 xm;
    int32_t pnt;
  };
  constexpr int RB = DM * (int)sizeof(AT);
  constexpr bool NEED_P = HAS_DT;
  for_each_tile_pf<(sizeof(AT) == 2) && !HAS_DT && !FUSE1>(V, [&](int64_t t) {
    // ---- the tile is read ONCE, view-major (each lane: half a row of dz_L, a_L; a_prev in the
    //      accumulator layout); the channel-major operands of the weight gradient come from LDS
    Raw r;
    r.dz = tile_load_half<AT>(RowTile(dz_L, t * 32, V, RB), j, h);   // RC: swapped into acc layout at unpack
    if (!RC) r.al = tile_load_half<AT>(RowTile(a_L, t * 32, V, RB), j, h);
    if (PREV_XMAP) r.xm = as_f4(RowTile(x_prev, t * 32, V, 32).b128(j * 32 + h * 16));
    else r.ap = tile_load_acc<AT>(RowTile(a_prev, t * 32, V, RB), j, h);
    r.pnt = NEED_P ? (int32_t)RowTile(vp, t * 32, V, 4).b32(j * 4) : 0;
    return r;
  }, [&](int64_t t, const Raw& raw) {
    const int64_t row0 = t * 32;
    const int64_t v = row0 + j;
    const bool ok = v < V;
    float dzv[16], alv[16], ap[16];
    if (RC) {
      AccRow<AT> dq;                   // same bytes, accumulator-layout view (bf16: half row + permlane swap)
      dq.q[0] = raw.dz.q[0];
      dq.q[1] = raw.dz.q[1];
      unpack(dq, dzv);
    } else {
      unpack(raw.dz, dzv);
      unpack(raw.al, alv);
    }
    if (!PREV_XMAP) unpack(raw.ap, ap);
    const float4 xm = raw.xm;
    const int32_t pnt = raw.pnt;
    // ---------------- (PREV_XMAP) a1 = x_map . Wa^T;  x_L = leaky(BN_prev(a_prev)), accumulator layout.
    // RC needs x_L first (a_L is recomputed from it); the other variants compute it after the dx product,
    // which keeps 48 registers out of the da / dx phase.
    float ahp[16], zp[16], xl[16];
    auto input_side = [&]() {
      if (PREV_XMAP) {
        f32x16 a1 = {0};
        a1 = p4.mul(xm, a1);
#pragma unroll
        for (int r = 0; r < 16; ++r) ap[r] = a1[r];
      }
      bn_norm16<true>(s_p, h, ap, ahp, zp);
#pragma unroll
      for (int r = 0; r < 16; ++r) xl[r] = ok ? (pident ? ap[r] : leaky_m(zp[r])) : 0.f;
    };
    if (RC) input_side();
    if (RC) {
      // ---------------- a_L recomputed: x_L . W_L^T (+ the per-point addend of the concatenation layer)
      f32x16 aL = {0};
      aL = pf.mul(xl, aL);
      if (HAS_DT && addend) {
        float ad[16];
        load_acc_layout(addend + (int64_t)pnt * DM, h, ad);
#pragma unroll
        for (int r = 0; r < 16; ++r) aL[r] += ad[r];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) alv[r] = aL[r];
    }
    // ---------------- da = BN_L-backward(dz_L); view-major product dx = da . W_L
    float da[16];
    f32x16 accx = {0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int base = RC ? 8 * q + 4 * h : 16 * h + 4 * q;
      const float4 g4 = *reinterpret_cast<const float4*>(&s_c[0][base]);
      const float4 m4 = *reinterpret_cast<const float4*>(&s_c[1][base]);
      const float4 i4 = *reinterpret_cast<const float4*>(&s_c[2][base]);
      const float4 a4 = *reinterpret_cast<const float4*>(&s_c[3][base]);
      const float4 b4 = *reinterpret_cast<const float4*>(&s_c[4][base]);
      const float gs[4] = {g4.x, g4.y, g4.z, g4.w}, ms[4] = {m4.x, m4.y, m4.z, m4.w};
      const float is[4] = {i4.x, i4.y, i4.z, i4.w}, s1[4] = {a4.x, a4.y, a4.z, a4.w};
      const float s2[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int s = 4 * q + e;
        const float ah = (alv[s] - ms[e]) * is[e];
        da[s] = ok ? gs[e] * (dzv[s] - s1[e] - ah * s2[e]) : 0.f;
      }
    }
    accx = pt.mul(da, accx);
    if (RC) tileT_put_acc(reinterpret_cast<bf16_t*>(tda), j, h, da);
    else if (BF) tileT_put_half(reinterpret_cast<bf16_t*>(tda), j, h, da);
    else tile_put_half(tda, j, h, da);
    // ---------------- x_L -> LDS; dz_prev
    if (!RC) input_side();
    if (BF) tileT_put_acc(reinterpret_cast<bf16_t*>(tx), j, h, xl);
    else tile_put_acc(tx, j, h, xl);
    if (!RAW_OUT) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = ok ? accx[r] * dleaky_m(zp[r]) : 0.f;
        accx[r] = d;
        st[0][r] += d;
        st[1][r] = fmaf(d, ahp[r], st[1][r]);
      }
    }
    if (FUSE1) {
      float d1[16], ah1[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        d1[r] = accx[r];                 // dz1, already zero for rows beyond V
        ah1[r] = ok ? ahp[r] : 0.f;
      }
      tileT_put_acc(f1, j, h, d1);
      tileT_put_acc(f1 + 32 * TSB, j, h, ah1);
      tileT_put(f1 + 64 * TSB, 4 * h, 4 * h + 1, j, xm.x, xm.y);     // rows beyond V read as zeros
      tileT_put(f1 + 64 * TSB, 4 * h + 2, 4 * h + 3, j, xm.z, xm.w);
      sx4.x += xm.x; sx4.y += xm.y; sx4.z += xm.z; sx4.w += xm.w;
    }
    wave_sync_m();
    // ---------------- channel-major: dW_L[n][k] += sum_v da[v][n] * x_L[v][k]  (operands from LDS)
    if (BF) {
      const bf16_t* ta = reinterpret_cast<const bf16_t*>(tda);
      const bf16_t* tb = reinterpret_cast<const bf16_t*>(tx);
#pragma unroll
      for (int m = 0; m < 2; ++m) accW = DVA_MFMA_BF16(tileT_get(ta, j, h, m), tileT_get(tb, j, h, m), accW);
      if (FUSE1) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const bf16x8 bx = tileT_get(f1 + 64 * TSB, j, h, m);      // x[view][f = j] (zero for j >= 8)
          accP = DVA_MFMA_BF16(tileT_get(f1, j, h, m), bx, accP);
          accQ = DVA_MFMA_BF16(tileT_get(f1 + 32 * TSB, j, h, m), bx, accQ);
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int row = 2 * s + h;
        accW = DVA_MFMA(tda[row * TS + j], tx[row * TS + j], accW);
      }
    }
    // ---------------- dt[p][n] += da[v][n]: run-length sum over the tile's rows, lane = channel
    if (HAS_DT) {
      const int64_t left = V - row0;
      const int nvalid = left < 32 ? (int)left : 32;
      const int32_t p_first = __shfl(pnt, 0), p_last = __shfl(pnt, nvalid - 1);
      if (BF && p_first == p_last) {
        // all the tile's views belong to one point (rows are sorted by point; the usual case with tens
        // of views per point): column sums of the transposed tile.  Lane (c = j, h) owns the 16 views
        // 8h..8h+7, 16+8h..16+8h+7 (rows beyond V hold zeros).
        const bf16_t* ta = reinterpret_cast<const bf16_t*>(tda);
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const uint4 q = *reinterpret_cast<const uint4*>(ta + j * TSB + 16 * m + 8 * h);
          const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) sum += __uint_as_float(w[e] << 16) + __uint_as_float(w[e] & 0xffff0000u);
        }
        sum += __shfl_xor(sum, 32);
        if (h == 0) atomicAdd(&dt[(int64_t)p_first * DM + j], sum);
      } else {
        // point id of row (2s + h) lives in lane (2s + h) of either half: fetch with a shuffle
        int32_t cur_p = -1;
        float cur_s = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const int row = 2 * s + h;
          const int32_t p = __shfl(pnt, row);
          if (row0 + row < V) {
            if (p != cur_p) {
              if (cur_p >= 0) atomicAdd(&dt[(int64_t)cur_p * DM + j], cur_s);
              cur_p = p;
              cur_s = 0.f;
            }
            cur_s += BF ? bf2f(reinterpret_cast<const bf16_t*>(tda)[j * TSB + row]) : tda[row * TS + j];
          }
        }
        if (cur_p >= 0) atomicAdd(&dt[(int64_t)cur_p * DM + j], cur_s);
      }
    }
    // ---------------- the tile's only stores, last: nothing in this iteration waits for them
    if (!FUSE1) tile_store_acc<AT>(RowTile(out, row0, V, RB), j, h, accx);
    wave_sync_m();
  });
  // dW: accW[r] = dW[n = acc_chan(r,h)][k = j]; block reduction in LDS, one atomic per element per block
  for (int i = threadIdx.x; i < DM * DM; i += blockDim.x) s_red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) atomicAdd(&s_red[acc_chan(r, h) * DM + j], accW[r]);
  __syncthreads();
  for (int i = threadIdx.x; i < DM * DM; i += blockDim.x) atomicAdd(&dW[i], s_red[i]);
  if (!RAW_OUT) {
    __syncthreads();
    flush_stats<2>(st, st_prev, s_red, lane);
  }
  if (FUSE1) {
    __syncthreads();
    for (int i = threadIdx.x; i < 520; i += blockDim.x) s_red[i] = 0.f;
    __syncthreads();
    if (j < 8) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        atomicAdd(&s_red[acc_chan(r, h) * 8 + j], accP[r]);
        atomicAdd(&s_red[256 + acc_chan(r, h) * 8 + j], accQ[r]);
      }
    }
    float sx[4] = {sx4.x, sx4.y, sx4.z, sx4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = sx[e];
#pragma unroll
      for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off);
      if (j == 0) atomicAdd(&s_red[512 + 4 * h + e], v);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 520; i += blockDim.x) atomicAdd(&first[i], s_red[i]);
  }
}

// Sum vals[s] over the 32 lanes of a half-wave; channel of register s is 16h + s (half-row layout).
__device__ __forceinline__ void reduce_half_channels(float (&vals)[16], float* s_red, int lane) {
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    float v = vals[s];
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off);
    if ((lane & 31) == 0) atomicAdd(&s_red[16 * (lane >> 5) + s], v);
  }
}

// dz2[v,c] = (dcat[v,c] + [arg[p,c]==v] dpooled[p,c]) * leaky'(BN2(a2[v,c])), p = vp[v]; S1, S2 of BN2.
// View-major, 16 channels per lane, 16-byte accesses only.
template <typename AT>
__global__ __launch_bounds__(256) void dsm_bwd_max_kernel(
    const AT* __restrict__ dcat, const AT* __restrict__ a2, const float* __restrict__ bn2,
    const int32_t* __restrict__ arg, const float* __restrict__ dpooled, const int32_t* __restrict__ vp,
    AT* __restrict__ dz2, double* __restrict__ st, int64_t V) {
  __shared__ float s_red[2 * DM];
  __shared__ __attribute__((aligned(16))) float s_p[4][DM];
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  stage_bn(s_p, bn2);
  __syncthreads();
  float acc[2][16];
#pragma unroll
  for (int s = 0; s < 16; ++s) acc[0][s] = acc[1][s] = 0.f;
  struct Raw {
    HalfRow<AT> g, a;
    int32_t p;
  };
  constexpr int RB = DM * (int)sizeof(AT);
  for_each_tile_pf<sizeof(AT) == 2>(V, [&](int64_t t) {
    Raw r;
    r.g = tile_load_half<AT>(RowTile(dcat, t * 32, V, RB), j, h);
    r.a = tile_load_half<AT>(RowTile(a2, t * 32, V, RB), j, h);
    r.p = (int32_t)RowTile(vp, t * 32, V, 4).b32(j * 4);
    return r;
  }, [&](int64_t t, const Raw& raw) {
    const int64_t v = t * 32 + j;
    const bool ok = v < V;
    const int64_t p = raw.p;
    float g[16], a[16], dp[16];
    load16(dpooled + p * DM + 16 * h, dp);   // per-point rows: shared by the views of a point (L2)
    unpack(raw.g, g);
    unpack(raw.a, a);
    int32_t ag[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int4 w = *reinterpret_cast<const int4*>(arg + p * DM + 16 * h + 4 * q);
      ag[4 * q] = w.x; ag[4 * q + 1] = w.y; ag[4 * q + 2] = w.z; ag[4 * q + 3] = w.w;
    }
    float d[16], ahv[16], zv[16];
    bn_norm16<false>(s_p, h, a, ahv, zv);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float gg = g[s] + ((int64_t)ag[s] == v ? dp[s] : 0.f);
      d[s] = ok ? gg * dleaky_m(zv[s]) : 0.f;
      acc[0][s] += d[s];
      acc[1][s] = fmaf(d[s], ahv[s], acc[1][s]);
    }
    tile_store_half<AT>(RowTile(dz2, t * 32, V, RB), j, h, d);
  });
  for (int i = threadIdx.x; i < 2 * DM; i += blockDim.x) s_red[i] = 0.f;
  __syncthreads();
  reduce_half_channels(acc[0], s_red, lane);
  reduce_half_channels(acc[1], s_red + DM, lane);
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * DM; i += blockDim.x) atomicAdd(&st[i], (double)s_red[i]);
}

// Score layer backward on the matrix cores: out = f(a4).Ws^T + bs with f = leaky(BN4(.)).
//   dz4[v,k] = (sum_g dc[v,g] Ws[g,k]) * leaky'(z4[v,k])   (view-major, ceil(G/2) k-steps)
//   dWs[g,k] = sum_v dc[v,g] f(a4)[v,k],  dbs[g] = sum_v dc[v,g]   (channel-major, 16 k-steps)
// PRE: `a` is the input of the layer before the scores (never-stored activation recomputed here as
//   leaky(BN_pre(a)).W_pre^T, see dsm_fwd_layer_kernel).
template <typename AT, bool PRE = false>
__global__ __launch_bounds__(256) void dsm_bwd_score_kernel(
    const float* __restrict__ dcompat, const AT* __restrict__ a, const float* __restrict__ bn,
    const float* __restrict__ Ws, AT* __restrict__ dz, float* __restrict__ dWs,
    float* __restrict__ dbs, double* __restrict__ st, int64_t V, int G, const float* __restrict__ bn_pre,
    const float* __restrict__ W_pre) {
  __shared__ float s_red[DM * DM];
  __shared__ __attribute__((aligned(16))) float s_p[4][DM];
  __shared__ __attribute__((aligned(16))) float s_pre[PRE ? 4 : 1][DM];
  __shared__ __attribute__((aligned(16))) float s_x[4][32 * TS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  const int GH = (G + 1) / 2;  // k-step s pairs score columns (s, s + GH)
  stage_bn(s_p, bn);
  if (PRE) stage_bn(s_pre, bn_pre);
  __syncthreads();
  ViewProd<AT> ppre;  // W_pre[n = j][acc_chan(r, h)]
  if (PRE) {
    float w[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) w[r] = W_pre[j * DM + acc_chan(r, h)];
    ppre.prep(w);
  }
  float* tx = s_x[wv];
  f32x16 accW = {0};
  float db = 0.f;
  float stv[2][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) stv[0][r] = stv[1][r] = 0.f;
  float wsv[16];  // Ws[g = s + GH*h][k = j] for the (at most 16) view-major k-steps
#pragma unroll
  for (int s = 0; s < 16; ++s) wsv[s] = (s < GH && s + GH * h < G) ? Ws[(s + GH * h) * DM + j] : 0.f;
  struct Raw {
    AccRow<AT> ap;
    float dcr[16];
    float2 dc2;
  };
  const int jc = j < G ? j : 0;
  constexpr int RB = DM * (int)sizeof(AT);
  constexpr bool BF = sizeof(AT) == 2;
  for_each_tile_pf<sizeof(AT) == 2>(V, [&](int64_t t) {
    Raw r;
    r.ap = tile_load_acc<AT>(RowTile(a, t * 32, V, RB), j, h);
    const RowTile D(dcompat, t * 32, V, G * 4);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      // score gradient of column g = j: fp32 k-step s takes rows (2s, 2s+1); bf16 MFMA m = s / 8 takes
      // rows 16m + 8h + (s & 7).  Rows beyond V read as zeros.
      const int row = BF ? 16 * (s >> 3) + 8 * h + (s & 7) : 2 * s + h;
      r.dcr[s] = __uint_as_float(D.b32((row * G + jc) * 4));
    }
    if (G == 4) r.dc2 = __builtin_bit_cast(float2, D.b64(j * 16 + h * 8));
    return r;
  }, [&](int64_t t, const Raw& raw) {
    const int64_t row0 = t * 32;
    const int64_t v = row0 + j;
    const bool ok = v < V;
    const int64_t vc = ok ? v : V - 1;
    float ap[16];
    unpack(raw.ap, ap);
    if (PRE) {
      float ah0[16], z0[16], x0[16];
      bn_norm16<true>(s_pre, h, ap, ah0, z0);
#pragma unroll
      for (int r = 0; r < 16; ++r) x0[r] = leaky_m(z0[r]);
      f32x16 am = {0};
      am = ppre.mul(x0, am);
#pragma unroll
      for (int r = 0; r < 16; ++r) ap[r] = am[r];
    }
    const float (&dcr)[16] = raw.dcr;
    f32x16 accx = {0};
    if (G == 4) {
      const float2 dc2 = raw.dc2;
      accx = DVA_MFMA(wsv[0], ok ? dc2.x : 0.f, accx);
      accx = DVA_MFMA(wsv[1], ok ? dc2.y : 0.f, accx);
    } else {
      for (int s = 0; s < GH; ++s) {
        const int g = s + GH * h;
        const float wv = g < G ? Ws[g * DM + j] : 0.f;                   // A[i = k = j][kk = h]
        const float dc = (ok && g < G) ? dcompat[vc * G + g] : 0.f;      // B[kk = h][j = v]
        accx = DVA_MFMA(wv, dc, accx);
      }
    }
    float ahp[16], zp[16], xl[16];
    bn_norm16<true>(s_p, h, ap, ahp, zp);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      xl[r] = ok ? leaky_m(zp[r]) : 0.f;
      const float d = ok ? accx[r] * dleaky_m(zp[r]) : 0.f;
      accx[r] = d;
      stv[0][r] += d;
      stv[1][r] = fmaf(d, ahp[r], stv[1][r]);
    }
    if (BF) tileT_put_acc(reinterpret_cast<bf16_t*>(tx), j, h, xl);
    else tile_put_acc(tx, j, h, xl);
    wave_sync_m();
    float dcm[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      dcm[s] = j < G ? dcr[s] : 0.f;      // rows beyond V were read as zeros
      db += dcm[s];
    }
    if (BF) {
      const bf16_t* tb = reinterpret_cast<const bf16_t*>(tx);
#pragma unroll
      for (int m = 0; m < 2; ++m) accW = DVA_MFMA_BF16(round8(&dcm[8 * m]), tileT_get(tb, j, h, m), accW);
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) accW = DVA_MFMA(dcm[s], tx[(2 * s + h) * TS + j], accW);  // B[kk = h][j = k]
    }
    tile_store_acc<AT>(RowTile(dz, row0, V, RB), j, h, accx);
    wave_sync_m();
  });
  for (int i = threadIdx.x; i < DM * DM; i += blockDim.x) s_red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if (acc_chan(r, h) < G) atomicAdd(&s_red[acc_chan(r, h) * DM + j], accW[r]);
  __syncthreads();
  for (int i = threadIdx.x; i < G * DM; i += blockDim.x) atomicAdd(&dWs[i], s_red[i]);
  __syncthreads();       // dbs: the wavefronts' sums meet in LDS, one global atomic per block and group
  if (threadIdx.x < DM) s_red[threadIdx.x] = 0.f;      // G <= DM
  __syncthreads();
  if (j < G && db != 0.f) atomicAdd(&s_red[j], db);
  __syncthreads();
  if ((int)threadIdx.x < G && s_red[threadIdx.x] != 0.f) atomicAdd(&dbs[threadIdx.x], s_red[threadIdx.x]);
  flush_stats<2>(stv, st, s_red, lane);
}

static inline int grid_tiles(int64_t V) {
  int64_t b = ((V + 31) / 32 + 3) / 4;
  if (b > 256 * 8) b = 256 * 8;
  if (b < 1) b = 1;
  return (int)b;
}

// launchers used by the C entry points in deepset.hip; `bf` = the [V, 32] activation / gradient tensors
// are stored as bf16 (else fp32)
#define DVA_ACT(bf, CALL_F, CALL_B) \
  do {                              \
    if (bf) { CALL_B; } else { CALL_F; } \
  } while (0)

int dsm_launch_fwd_first(const float* x_map, const float* Wa, const float* bn1, const float* Wb, void* a2,
                         double* stats, int64_t V, int stats_only, int bf, hipStream_t s) {
  const dim3 grid(grid_tiles(V)), block(256);
  if (stats_only)
    hipLaunchKernelGGL((dsm_fwd_first_kernel<float, true>), grid, block, 0, s, x_map, Wa,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr, stats, V);
  else
    DVA_ACT(bf,
            hipLaunchKernelGGL((dsm_fwd_first_kernel<float, false>), grid, block, 0, s, x_map, Wa, bn1, Wb,
                               (float*)a2, stats, V),
            hipLaunchKernelGGL((dsm_fwd_first_kernel<bf16_t, false>), grid, block, 0, s, x_map, Wa, bn1, Wb,
                               (bf16_t*)a2, stats, V));
  return 0;
}

template <typename AT>
static void launch_fwd_layer(const void* a_in, const float* bn_in, const float* W, const float* addend,
                             const int32_t* vp, void* a_out, double* stats, int64_t V, hipStream_t s) {
  const dim3 grid(grid_tiles(V)), block(256);
  const float* nf = nullptr;
#define DVA_F(ADD, ST)                                                                                \
  hipLaunchKernelGGL((dsm_fwd_layer_kernel<AT, ADD, false, ST, false>), grid, block, 0, s, (const AT*)a_in, \
                     bn_in, W, addend, vp, nf, a_out, stats, V, 32, nf, nf)
  if (addend && a_out) DVA_F(true, true);
  else if (addend) DVA_F(true, false);
  else if (a_out) DVA_F(false, true);
  else DVA_F(false, false);
#undef DVA_F
}
int dsm_launch_fwd_layer(const void* a_in, const float* bn_in, const float* W, const float* addend,
                         const int32_t* vp, void* a_out, double* stats, int64_t V, int bf, hipStream_t s) {
  DVA_ACT(bf, launch_fwd_layer<float>(a_in, bn_in, W, addend, vp, a_out, stats, V, s),
          launch_fwd_layer<bf16_t>(a_in, bn_in, W, addend, vp, a_out, stats, V, s));
  return 0;
}

template <typename AT>
static void launch_fwd_score(const void* a, const float* bn, const float* Ws, const float* bs, float* compat,
                             int64_t V, int G, const float* bn_pre, const float* W_pre, hipStream_t s) {
  const dim3 grid(grid_tiles(V)), block(256);
  const float* nf = nullptr;
  const int32_t* ni = nullptr;
  if (W_pre)   // a = input of the layer BEFORE the scores: (bn_pre, W_pre) first, then (bn, Ws)
    hipLaunchKernelGGL((dsm_fwd_layer_kernel<AT, false, true, true, true>), grid, block, 0, s, (const AT*)a,
                       bn_pre, W_pre, nf, ni, bs, (void*)compat, (double*)nullptr, V, G, bn, Ws);
  else
    hipLaunchKernelGGL((dsm_fwd_layer_kernel<AT, false, true, true, false>), grid, block, 0, s, (const AT*)a, bn,
                       Ws, nf, ni, bs, (void*)compat, (double*)nullptr, V, G, nf, nf);
}
int dsm_launch_fwd_score(const void* a, const float* bn, const float* Ws, const float* bs, float* compat,
                         int64_t V, int G, const float* bn_pre, const float* W_pre, int bf, hipStream_t s) {
  DVA_ACT(bf, launch_fwd_score<float>(a, bn, Ws, bs, compat, V, G, bn_pre, W_pre, s),
          launch_fwd_score<bf16_t>(a, bn, Ws, bs, compat, V, G, bn_pre, W_pre, s));
  return 0;
}

template <typename AT>
static int launch_bwd_layer(const void* dz_L, const void* a_L, const float* bn_L, const float* sm_L,
                            const float* W_L, const void* a_prev, const float* Wa, const float* bn_prev,
                            void* out, float* dW, double* st_prev, float* dt, const int32_t* vp,
                            float* first, const float* addend, int64_t V, int prev_is_xmap, int raw_out,
                            hipStream_t s) {
  const dim3 grid(grid_tiles(V)), block(256);
#define DVA_L(P, R, T, F, C)                                                                            \
  hipLaunchKernelGGL((dsm_bwd_layer_kernel<AT, P, R, T, F, C>), grid, block, 0, s, (const AT*)dz_L,     \
                     (const AT*)a_L, bn_L, sm_L, W_L, a_prev, Wa, bn_prev, (AT*)out, dW, st_prev, dt, vp, \
                     first, addend, V)
  if constexpr (sizeof(AT) == 2) {
    if (!a_L) {   // recompute variants (bf16 storage): the three layers of the element-wise MLPs
      if (first && prev_is_xmap && !raw_out && !dt) DVA_L(true, false, false, true, true);
      else if (!prev_is_xmap && raw_out && dt) DVA_L(false, true, true, false, true);
      else if (!prev_is_xmap && !raw_out && !dt && !first) DVA_L(false, false, false, false, true);
      else return -2;
      return 0;
    }
    if (first) {
      DVA_L(true, false, false, true, false);
      return 0;
    }
  }
  if (!a_L) return -2;
  if (prev_is_xmap && raw_out) DVA_L(true, true, false, false, false);
  else if (prev_is_xmap) DVA_L(true, false, false, false, false);
  else if (raw_out && dt) DVA_L(false, true, true, false, false);
  else if (raw_out) DVA_L(false, true, false, false, false);
  else if (dt) DVA_L(false, false, true, false, false);
  else DVA_L(false, false, false, false, false);
#undef DVA_L
  return 0;
}
int dsm_launch_bwd_layer(const void* dz_L, const void* a_L, const float* bn_L, const float* sm_L,
                         const float* W_L, const void* a_prev, const float* Wa, const float* bn_prev,
                         void* out, float* dW, double* st_prev, float* dt, const int32_t* vp, float* first,
                         const float* addend, int64_t V, int prev_is_xmap, int raw_out, int bf, hipStream_t s) {
  if (bf)
    return launch_bwd_layer<bf16_t>(dz_L, a_L, bn_L, sm_L, W_L, a_prev, Wa, bn_prev, out, dW, st_prev, dt, vp,
                                    first, addend, V, prev_is_xmap, raw_out, s);
  return launch_bwd_layer<float>(dz_L, a_L, bn_L, sm_L, W_L, a_prev, Wa, bn_prev, out, dW, st_prev, dt, vp,
                                 first, addend, V, prev_is_xmap, raw_out, s);
}

int dsm_launch_bwd_max(const void* dcat, const void* a2, const float* bn2, const int32_t* arg,
                       const float* dpooled, const int32_t* vp, void* dz2, double* st, int64_t V, int bf,
                       hipStream_t s) {
  const dim3 grid(grid_tiles(V)), block(256);
  DVA_ACT(bf,
          hipLaunchKernelGGL((dsm_bwd_max_kernel<float>), grid, block, 0, s, (const float*)dcat,
                             (const float*)a2, bn2, arg, dpooled, vp, (float*)dz2, st, V),
          hipLaunchKernelGGL((dsm_bwd_max_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)dcat,
                             (const bf16_t*)a2, bn2, arg, dpooled, vp, (bf16_t*)dz2, st, V));
  return 0;
}

int dsm_launch_bwd_score(const float* dcompat, const void* a, const float* bn, const float* Ws, void* dz,
                         float* dWs, float* dbs, double* st, int64_t V, int G, const float* bn_pre,
                         const float* W_pre, int bf, hipStream_t s) {
  const dim3 grid(grid_tiles(V)), block(256);
  if (W_pre) {   // bf16 storage only (checked by the caller)
    hipLaunchKernelGGL((dsm_bwd_score_kernel<bf16_t, true>), grid, block, 0, s, dcompat, (const bf16_t*)a, bn, Ws,
                       (bf16_t*)dz, dWs, dbs, st, V, G, bn_pre, W_pre);
    return 0;
  }
  const float* nf = nullptr;
  DVA_ACT(bf,
          hipLaunchKernelGGL((dsm_bwd_score_kernel<float, false>), grid, block, 0, s, dcompat, (const float*)a,
                             bn, Ws, (float*)dz, dWs, dbs, st, V, G, nf, nf),
          hipLaunchKernelGGL((dsm_bwd_score_kernel<bf16_t, false>), grid, block, 0, s, dcompat,
                             (const bf16_t*)a, bn, Ws, (bf16_t*)dz, dWs, dbs, st, V, G, nf, nf));
  return 0;
}

}  // namespace dva
