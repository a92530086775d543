// DeepViewAgg view attention for gfx950: segment softmax + attention-weighted segment sum +
// rectified-tanh gating, forward and backward (reference: modules/multimodal/pooling.py:284-300,
// :514-530, :690-715, :737-810).
//
// HBM-bound.  Algorithmic bytes per launch (s = bytes/element of val):
//   fwd: V*(C*s + 2*G*4) + N*(C*s + 8 + 2*G*4)     (val, compat in, att out | out, ptr, gate+amax)
//   bwd: V*(2*C*s + 4*G*4) + N*(C*s + 8 + 2*G*4)   (val in, grad_val out, compat/att in,
//                                                   grad_compat written twice (scratch + final))
//
// Fused fast path ("team" kernels): a team of TS = LPR*R lanes owns one point; LPR lanes cover one
// view row with 16-byte loads (VEC elements each), R rows are in flight per step, so a wavefront
// issues 1 KiB of contiguous val per load instruction when R*LPR = 64.  Reductions over views are
// per-lane fp32 accumulators + xor-shuffles across the R row slots; reductions over the channels of
// a group are xor-shuffles inside a row.  No LDS, no atomics on the data path (gating-parameter
// gradients: LDS partials per block, then one atomic per block).
// The generic path (one thread per (point, channel)) covers every other shape.
#include "dva_common.h"

namespace dva {

// ------------------------------------------------------------------------------------------------
// generic path
// ------------------------------------------------------------------------------------------------

// one thread per (point, group): softmax over the point's views, gate, argmax
__global__ __launch_bounds__(256) void att_scores_kernel(const float* __restrict__ compat,
                                                          const int64_t* __restrict__ ptr,
                                                          const float* __restrict__ gw,
                                                          const float* __restrict__ gb,
                                                          float* __restrict__ att,
                                                          float* __restrict__ gate,
                                                          int32_t* __restrict__ amax, int64_t N,
                                                          int G, int scaling, float eps) {
  const int64_t total = N * (int64_t)G;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = t / G;
    const int g = (int)(t - p * G);
    const int64_t beg = ptr[p], end = ptr[p + 1];
    float m = 0.f;
    int64_t am = -1;
    for (int64_t r = beg; r < end; ++r) {
      const float v = compat[r * G + g];
      if (r == beg || v > m) {
        m = v;
        am = r;
      }
    }
    if (end > beg) {
      const float d = scaling ? sqrtf((float)(end - beg)) : 1.f;
      float s = 0.f;
      for (int64_t r = beg; r < end; ++r) s += expf((compat[r * G + g] - m) / d);
      s += eps;
      for (int64_t r = beg; r < end; ++r) att[r * G + g] = expf((compat[r * G + g] - m) / d) / s;
    }
    if (amax) amax[t] = (int32_t)am;
    if (gate) {
      float gt = 1.f;
      if (gw) {
        const float pre = gw[g] * m + gb[g];
        gt = tanhf(fmaxf(pre, 0.f));
      }
      gate[t] = gt;
    }
  }
}

// one thread per (point, channel)
template <typename T>
__global__ __launch_bounds__(256) void att_wsum_kernel(const T* __restrict__ val,
                                                        const int32_t* __restrict__ row_idx,
                                                        const float* __restrict__ att,
                                                        const float* __restrict__ gate,
                                                        const int64_t* __restrict__ ptr,
                                                        T* __restrict__ out, int64_t N, int C,
                                                        int G) {
  const int64_t total = N * (int64_t)C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = t / C;
    const int c = (int)(t - p * C);
    const int g = group_of_channel(c, C, G);
    const int64_t beg = ptr[p], end = ptr[p + 1];
    float acc = 0.f;
    for (int64_t r = beg; r < end; ++r) {
      const int64_t vr = row_idx ? (int64_t)row_idx[r] : r;
      acc = fmaf(att[r * G + g], Elt<T>::ld(val, vr * C + c), acc);
    }
    if (gate) acc *= gate[p * G + g];
    Elt<T>::st(out, t, acc);
  }
}

// backward, one thread per (point, group): d[v,g] = sum_{c in g} go[p,c]*val[v,c] (scratch in
// gcompat), softmax backward, gating backward
template <typename T>
__global__ __launch_bounds__(256) void att_bwd_scores_kernel(
    const T* __restrict__ gout, const T* __restrict__ val, const int32_t* __restrict__ row_idx,
    const float* __restrict__ compat,
    const float* __restrict__ att, const float* __restrict__ gate, const int32_t* __restrict__ amax,
    const int64_t* __restrict__ ptr, const float* __restrict__ gw, float* __restrict__ gcompat,
    float* __restrict__ gwb, int64_t N, int C, int G, int scaling) {
  extern __shared__ float s_wb[];  // [2*G] block partials of d/dw, d/db
  for (int i = threadIdx.x; i < 2 * G; i += blockDim.x) s_wb[i] = 0.f;
  __syncthreads();
  const int64_t total = N * (int64_t)G;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = t / G;
    const int g = (int)(t - p * G);
    const int64_t beg = ptr[p], end = ptr[p + 1];
    if (end <= beg) continue;
    const int c0 = group_begin(g, C, G), c1 = group_begin(g + 1, C, G);
    float sum_ad = 0.f;
    for (int64_t r = beg; r < end; ++r) {
      float d = 0.f;
      const int64_t vr = row_idx ? (int64_t)row_idx[r] : r;
      for (int c = c0; c < c1; ++c) d += Elt<T>::ld(gout, p * C + c) * Elt<T>::ld(val, vr * C + c);
      gcompat[r * G + g] = d;
      sum_ad += att[r * G + g] * d;
    }
    const float gt = gate ? gate[t] : 1.f;
    const float dn = scaling ? sqrtf((float)(end - beg)) : 1.f;
    float g_mx = 0.f;
    const int64_t am = amax[t];
    if (gw) {
      const float g_pre = (gt > 0.f) ? sum_ad * (1.f - gt * gt) : 0.f;
      const float mx = compat[am * G + g];
      atomicAdd(&s_wb[g], g_pre * mx);
      atomicAdd(&s_wb[G + g], g_pre);
      g_mx = g_pre * gw[g];
    }
    const float tt = gt * sum_ad;
    for (int64_t r = beg; r < end; ++r) {
      const float d = gcompat[r * G + g];
      float gc = att[r * G + g] * (gt * d - tt) / dn;
      if (r == am) gc += g_mx;
      gcompat[r * G + g] = gc;
    }
  }
  __syncthreads();
  if (gwb)
    for (int i = threadIdx.x; i < 2 * G; i += blockDim.x)
      if (s_wb[i] != 0.f) atomicAdd(&gwb[i], s_wb[i]);
}

// backward, one thread per (point, channel): grad_val[v,c] = go[p,c]*gate[p,g]*att[v,g]
template <typename T>
__global__ __launch_bounds__(256) void att_bwd_val_kernel(const T* __restrict__ gout,
                                                           const int32_t* __restrict__ row_idx,
                                                           float* __restrict__ grows,
                                                           const float* __restrict__ att,
                                                           const float* __restrict__ gate,
                                                           const int64_t* __restrict__ ptr,
                                                           T* __restrict__ gval, int64_t N, int C,
                                                           int G) {
  const int64_t total = N * (int64_t)C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = t / C;
    const int c = (int)(t - p * C);
    const int g = group_of_channel(c, C, G);
    const int64_t beg = ptr[p], end = ptr[p + 1];
    float go = Elt<T>::ld(gout, t);
    if (gate) go *= gate[p * G + g];
    for (int64_t r = beg; r < end; ++r) {
      if (row_idx)
        atomicAdd(&grows[(int64_t)row_idx[r] * C + c], go * att[r * G + g]);
      else
        Elt<T>::st(gval, r * C + c, go * att[r * G + g]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fused team path
// ------------------------------------------------------------------------------------------------

template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
  static constexpr int N = 4;
  typedef float4 raw;
  static __device__ __forceinline__ void unpack(const raw& r, float* f) {
    f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w;
  }
  static __device__ __forceinline__ raw pack(const float* f) { return make_float4(f[0], f[1], f[2], f[3]); }
};
template <>
struct Vec16<bf16_t> {
  static constexpr int N = 8;
  typedef uint4 raw;
  static __device__ __forceinline__ void unpack(const raw& r, float* f) {
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
    f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
    f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
  }
  static __device__ __forceinline__ raw pack(const float* f) {
    uint4 r;
    r.x = pack_bf16x2(f[0], f[1]);
    r.y = pack_bf16x2(f[2], f[3]);
    r.z = pack_bf16x2(f[4], f[5]);
    r.w = pack_bf16x2(f[6], f[7]);
    return r;
  }
};

struct TeamGeom {
  int lpr;      // lanes per view row (power of two, <= 64)
  int rows;     // view rows in flight per team (power of two)
  int ts;       // team size = lpr * rows (<= 64)
  int lpg;      // lanes per group inside a row = lpr / G
};

// ---- reductions across the row slots of a team without the LDS crossbar (compile-time geometry only):
// lanes l and l ^ OFF exchange through DPP (OFF = 8: row_ror:8 inside a 16-lane row) or through the gfx950 row
// swaps v_permlane16_swap / v_permlane32_swap (both operands = the value: afterwards one register holds the value
// of the lower row of each pair in every lane, the other the upper one, so op(a, b) is the pair's reduction).
template <int OFF>
__device__ __forceinline__ void xor_pair(float v, float& a, float& b) {
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  const uint32_t x = __float_as_uint(v);
  if constexpr (OFF == 32) {
    const u2 r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    a = __uint_as_float(r.x);
    b = __uint_as_float(r.y);
  } else if constexpr (OFF == 16) {
    const u2 r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    a = __uint_as_float(r.x);
    b = __uint_as_float(r.y);
  } else if constexpr (OFF == 8) {
    a = v;
    b = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, false));  // row_ror:8
  } else {
    a = v;
    b = __shfl_xor(v, OFF);
  }
}
template <int OFF>
__device__ __forceinline__ void xor_pair(int v, int& a, int& b) {
  float fa, fb;
  xor_pair<OFF>(__int_as_float(v), fa, fb);
  a = __float_as_int(fa);
  b = __float_as_int(fb);
}
// sum over the ROWS row slots (lane stride LPR) of a team
template <int LPR, int ROWS>
__device__ __forceinline__ float team_sum(float v) {
  float a, b;
  if constexpr (ROWS >= 2) { xor_pair<LPR>(v, a, b); v = a + b; }
  if constexpr (ROWS >= 4) { xor_pair<2 * LPR>(v, a, b); v = a + b; }
  if constexpr (ROWS >= 8) { xor_pair<4 * LPR>(v, a, b); v = a + b; }
  if constexpr (ROWS >= 16) { xor_pair<8 * LPR>(v, a, b); v = a + b; }
  return v;
}
// lexicographic (max value, then smallest index) over the row slots
template <int OFF>
__device__ __forceinline__ void argmax_step(float& m, int& am) {
  float ma, mb;
  int ia, ib;
  xor_pair<OFF>(m, ma, mb);
  xor_pair<OFF>(am, ia, ib);
  const bool take_b = mb > ma || (mb == ma && ib < ia);
  m = take_b ? mb : ma;
  am = take_b ? ib : ia;
}
template <int LPR, int ROWS>
__device__ __forceinline__ void team_argmax(float& m, int& am) {
  if constexpr (ROWS >= 2) argmax_step<LPR>(m, am);
  if constexpr (ROWS >= 4) argmax_step<2 * LPR>(m, am);
  if constexpr (ROWS >= 8) argmax_step<4 * LPR>(m, am);
  if constexpr (ROWS >= 16) argmax_step<8 * LPR>(m, am);
}

// Forward.  Requirements (checked by the host): C % VEC == 0, C/VEC == lpr exactly (power of two),
// C % G == 0, (C/G) % VEC == 0, G <= ts, G power of two.
template <typename T, int LPR, int ROWS>   // LPR > 0: compile-time team geometry (LPR lanes per row x ROWS rows)
__global__ __launch_bounds__(256) void att_fwd_team_kernel(
    const T* __restrict__ val, const int32_t* __restrict__ row_idx, const float* __restrict__ compat,
    const int64_t* __restrict__ ptr,
    const float* __restrict__ gw, const float* __restrict__ gb, T* __restrict__ out,
    float* __restrict__ att, float* __restrict__ gate, int32_t* __restrict__ amax, int64_t N, int C,
    int G, int scaling, float eps, TeamGeom tg) {
  constexpr int VEC = Vec16<T>::N;
  // team geometry: constants in the specialised instances (divisions become shifts, the reduction loops unroll,
  // the shuffle offsets are immediates), taken from `tg` in the generic one
  const int tg_lpr = LPR > 0 ? LPR : tg.lpr, tg_rows = LPR > 0 ? ROWS : tg.rows;
  const int tg_ts = tg_lpr * tg_rows, tg_lpg = tg_lpr / G;
  typedef typename Vec16<T>::raw raw_t;
  const int lane = threadIdx.x & 63;
  const int li = lane & (tg_ts - 1);        // lane in team
  const int team_base = lane - li;          // first lane of the team inside the wave
  const int lane_r = li & (tg_lpr - 1);     // position inside the row
  const int row_slot = li / tg_lpr;         // which of the R rows in flight
  const int g_lane = lane_r / tg_lpg;       // group of this lane's channels
  const bool g_first = (lane_r % tg_lpg) == 0;
  // softmax-statistics mapping: lane li <-> (view slot, group)
  const int sg = li & (G - 1);
  const int vslot = li / G;
  const int VS = tg_ts / G;

  const int teams_per_block = blockDim.x / tg_ts;
  // WAVE: a team is a whole wavefront, so the point, its CSR range and every base address derived from them
  // are wave-uniform: they are kept in SGPRs (readfirstlane) and the per-lane part of an address is a 32-bit
  // offset (global_load ... v_off, s[base]) instead of 64-bit vector arithmetic per access.
  constexpr bool WAVE = LPR > 0 && LPR * ROWS == 64;
  const int64_t team0 = WAVE ? (int64_t)blockIdx.x * teams_per_block +
                                   __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))
                             : (int64_t)blockIdx.x * teams_per_block + threadIdx.x / tg_ts;
  const int64_t team_stride = (int64_t)gridDim.x * teams_per_block;
  auto uniform64 = [](int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
  };

  // Software pipeline over the points of a team, three stages deep, so that the three dependent loads of
  // a point (CSR pointers -> row indices + scores -> value rows) of three consecutive points are in flight
  // together: one memory latency per point instead of three.
  // U views per row slot on the short-segment path; 8 where a row takes 16 lanes (fp32, C = 64: 4 row slots), so that the
  // 32-view points of the headline shape stay on it (with U = 4 they took the chunked long-segment path: 2.24 ms)
  constexpr int U = (LPR == 16 && ROWS == 4 && sizeof(T) == 4) ? 8 : 4;
  struct StageB {          // row indices and scores of a short segment (n <= U * rows): 8 registers
    int32_t ri[U];
    float cg[U];
  };
  auto is_small = [&](int n) { return n > 0 && n <= U * tg_rows; };
  auto load_a = [&](int64_t p, int64_t& beg, int& n) {
    beg = 0;
    n = 0;
    if (p < N) {
      beg = ptr[p];
      n = (int)(ptr[p + 1] - beg);
      if (WAVE) {
        beg = uniform64(beg);
        n = __builtin_amdgcn_readfirstlane(n);
      }
    }
  };
  auto load_b = [&](int64_t beg, int n, StageB& b) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      b.ri[u] = 0;
      b.cg[u] = 0.f;
    }
    if (is_small(n)) {
      if (WAVE) {
        const int32_t* ri_base = row_idx ? row_idx + beg : nullptr;   // uniform bases, 32-bit lane offsets
        const float* c_base = compat + beg * G;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t v = (uint32_t)(row_slot + u * tg_rows);
          const uint32_t vv = v < (uint32_t)n ? v : 0u;
          if (row_idx) b.ri[u] = ri_base[vv];
          b.cg[u] = c_base[vv * (uint32_t)G + (uint32_t)g_lane];
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int v = row_slot + u * tg_rows;
          const int64_t r = beg + (v < n ? v : 0);
          if (row_idx) b.ri[u] = row_idx[r];
          b.cg[u] = compat[r * G + g_lane];
        }
      }
    }
  };
  int64_t p = team0, p1 = team0 + team_stride, p2 = team0 + 2 * team_stride;
  int64_t beg, beg1, beg2;
  int n, n1, n2;
  StageB sb, sb1;
  load_a(p, beg, n);
  load_a(p1, beg1, n1);
  load_b(beg, n, sb);
  for (; p < N; p = p1, beg = beg1, n = n1, sb = sb1, p1 = p2, beg1 = beg2, n1 = n2, p2 += team_stride) {
    load_b(beg1, n1, sb1);      // next point: row indices + scores (its pointers came one iteration ago)
    load_a(p2, beg2, n2);       // the point after: CSR pointers
    if (is_small(n)) {
      // ---- short segments (the common case): every lane derives the softmax statistics of ITS channel
      //      group from the scores it needs anyway; the value rows are the only loads left to issue
      const int64_t col = (int64_t)lane_r * VEC;
      bool ok[U];
      int64_t rr[U];
      const float (&cg)[U] = sb.cg;
      raw_t x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int v = row_slot + u * tg_rows;
        ok[u] = v < n;
        rr[u] = beg + (ok[u] ? v : 0);
        const int64_t ri = row_idx ? (int64_t)sb.ri[u] : rr[u];
        x[u] = *reinterpret_cast<const raw_t*>(val + ri * C + col);
      }
      float m = -INFINITY;
      int am = 0x7fffffff;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (ok[u] && cg[u] > m) {     // views ascend with u inside a lane: strict > keeps the first
          m = cg[u];
          am = row_slot + u * tg_rows;
        }
      }
      if constexpr (LPR > 0) {
        team_argmax<LPR, ROWS>(m, am);
      } else {
        for (int off = tg_lpr; off < tg_ts; off <<= 1) {
          const float m2 = __shfl_xor(m, off);
          const int a2 = __shfl_xor(am, off);
          if (m2 > m || (m2 == m && a2 < am)) {
            m = m2;
            am = a2;
          }
        }
      }
      // one reciprocal per point instead of a division per view, hardware exp2: the kernel is VALU-bound (its
      // row gathers hit the cache hierarchy), and the IEEE divisions + expf were ~40 % of its instructions
      const float inv_dn = scaling ? 1.f / sqrtf((float)n) : 1.f;
      float e[U], s = 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        e[u] = ok[u] ? __expf((cg[u] - m) * inv_dn) : 0.f;
        s += e[u];
      }
      if constexpr (LPR > 0) s = team_sum<LPR, ROWS>(s);
      else
        for (int off = tg_lpr; off < tg_ts; off <<= 1) s += __shfl_xor(s, off);
      s += eps;
      float gt = 1.f;
      if (gw) gt = tanhf(fmaxf(gw[g_lane] * m + gb[g_lane], 0.f));
      if (row_slot == 0 && g_first) {
        if (gate) gate[p * G + g_lane] = gt;
        if (amax) amax[p * G + g_lane] = (int32_t)(beg + am);
      }
      float acc[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
      const float inv_s = 1.f / s;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float a = e[u] * inv_s;
        if (ok[u] && g_first) {
          if (WAVE)
            (att + beg * G)[(uint32_t)(row_slot + u * tg_rows) * (uint32_t)G + (uint32_t)g_lane] = a;
          else
            att[rr[u] * G + g_lane] = a;
        }
        float f[VEC];
        Vec16<T>::unpack(x[u], f);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = fmaf(a, f[k], acc[k]);
      }
      if constexpr (LPR > 0) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = team_sum<LPR, ROWS>(acc[k]);
      } else {
        for (int off = tg_lpr; off < tg_ts; off <<= 1) {
#pragma unroll
          for (int k = 0; k < VEC; ++k) acc[k] += __shfl_xor(acc[k], off);
        }
      }
      if (row_slot == 0) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] *= gt;
        *reinterpret_cast<raw_t*>(out + p * C + col) = Vec16<T>::pack(acc);
      }
      continue;
    }
    // ---- per-group max (+ first arg) over the point's views
    float m = -INFINITY;
    int am = 0x7fffffff;
    for (int v = vslot; v < n; v += VS) {
      const float c = compat[(beg + v) * G + sg];
      if (c > m) {
        m = c;
        am = v;
      }
    }
    for (int off = G; off < tg_ts; off <<= 1) {
      const float m2 = __shfl_xor(m, off);
      const int a2 = __shfl_xor(am, off);
      if (m2 > m || (m2 == m && a2 < am)) {
        m = m2;
        am = a2;
      }
    }
    if (n == 0) m = 0.f;
    const float inv_dn = (scaling && n > 0) ? 1.f / sqrtf((float)n) : 1.f;
    float s = 0.f;
    for (int v = vslot; v < n; v += VS) s += __expf((compat[(beg + v) * G + sg] - m) * inv_dn);
    for (int off = G; off < tg_ts; off <<= 1) s += __shfl_xor(s, off);
    s += eps;
    float gt = 1.f;
    if (gw) gt = tanhf(fmaxf(gw[sg] * m + gb[sg], 0.f));
    if (vslot == 0) {
      if (gate) gate[p * G + sg] = gt;
      if (amax) amax[p * G + sg] = n > 0 ? (int32_t)(beg + am) : -1;
    }
    // broadcast the statistics of this lane's own channel group
    const float m_l = __shfl(m, team_base + g_lane);
    const float inv_s_l = 1.f / __shfl(s, team_base + g_lane);
    const float gt_l = __shfl(gt, team_base + g_lane);

    // ---- attention-weighted sum over views
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    const int64_t col = (int64_t)lane_r * VEC;
    // chunks of 4 rows per lane: row indices + scores, then the value rows, are issued before the first use
    for (int v0 = 0; v0 < n; v0 += 4 * tg_rows) {
      constexpr int U = 4;
      bool ok[U];
      int64_t rr[U], ri[U];
      float cg[U];
      raw_t x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int v = v0 + row_slot + u * tg_rows;
        ok[u] = v < n;
        rr[u] = beg + (ok[u] ? v : 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ri[u] = row_idx ? (int64_t)row_idx[rr[u]] : rr[u];  // fused view gather: row of the value map
        cg[u] = compat[rr[u] * G + g_lane];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = *reinterpret_cast<const raw_t*>(val + ri[u] * C + col);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float a = ok[u] ? __expf((cg[u] - m_l) * inv_dn) * inv_s_l : 0.f;
        if (ok[u] && g_first) att[rr[u] * G + g_lane] = a;
        float f[VEC];
        Vec16<T>::unpack(x[u], f);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = fmaf(a, f[k], acc[k]);
      }
    }
    for (int off = tg_lpr; off < tg_ts; off <<= 1) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += __shfl_xor(acc[k], off);
    }
    if (row_slot == 0) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] *= gt_l;
      *reinterpret_cast<raw_t*>(out + p * C + col) = Vec16<T>::pack(acc);
    }
  }
}

// Backward, same geometry.  gcompat doubles as scratch for d[v,g] between the two passes (each
// (v,g) slot is written and re-read by the same lane).  With row_idx (fused view gather) the value
// gradient is scatter-added into the fp32 value map: that phase switches to a channel-per-lane
// layout so that one atomic instruction covers whole contiguous rows (2 cache lines per 64 atomics
// instead of 16 with the 16-byte-per-lane layout: measured 8.7x on the first version).
template <typename T, int LPR, int ROWS>   // LPR > 0: compile-time team geometry (LPR lanes per row x ROWS rows)
__global__ __launch_bounds__(256, (LPR == 16 && ROWS == 4 && sizeof(T) == 4) ? 2 : 3) void att_bwd_team_kernel(
    const T* __restrict__ gout, const T* __restrict__ val, const int32_t* __restrict__ row_idx,
    float* __restrict__ grows, const float* __restrict__ compat,
    const float* __restrict__ att, const float* __restrict__ gate, const int32_t* __restrict__ amax,
    const int64_t* __restrict__ ptr, const float* __restrict__ gw, T* __restrict__ gval,
    float* __restrict__ gcompat, float* __restrict__ gwb, float* __restrict__ rec, int rs, int64_t N,
    int C, int G, int scaling, TeamGeom tg) {
  constexpr int VEC = Vec16<T>::N;
  // team geometry: constants in the specialised instances (divisions become shifts, the reduction loops unroll,
  // the shuffle offsets are immediates), taken from `tg` in the generic one
  const int tg_lpr = LPR > 0 ? LPR : tg.lpr, tg_rows = LPR > 0 ? ROWS : tg.rows;
  const int tg_ts = tg_lpr * tg_rows, tg_lpg = tg_lpr / G;
  typedef typename Vec16<T>::raw raw_t;
  __shared__ float s_wb[64];  // [2*G], G <= 32
  // long segments of the fused-gather form: d[v, g] of a point is parked in LDS between the two passes
  // (DBUF floats per wavefront, shared by its teams) instead of making a round trip through grad_compat
  constexpr int DBUF = 1024;
  __shared__ float s_dbuf[4][DBUF];
  if (threadIdx.x < 64) s_wb[threadIdx.x] = 0.f;
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int li = lane & (tg_ts - 1);
  const int lane_r = li & (tg_lpr - 1);
  const int row_slot = li / tg_lpr;
  const int g_lane = lane_r / tg_lpg;
  const bool g_first = (lane_r % tg_lpg) == 0;
  const int teams_per_wave = 64 / tg_ts;
  const int team_in_wave = lane / tg_ts;

  // WAVE: a team is a whole wavefront -- point, CSR range and the base addresses derived from them live in
  // SGPRs, the per-lane part of an address is a 32-bit offset (see the forward kernel)
  constexpr bool WAVE = LPR > 0 && LPR * ROWS == 64;
  auto uniform64 = [](int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
  };
  // wave-uniform outer loop: the teams of one wavefront always iterate together
  const int64_t wave = WAVE ? (int64_t)blockIdx.x * (blockDim.x >> 6) +
                                  __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))
                            : ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  // Three-stage software pipeline over the points of a team (same scheme as the forward kernel): CSR
  // pointers of point i+2, row indices / attentions / grad_out row / gate of point i+1 and the value rows of
  // point i are in flight together.
  constexpr int U = (LPR == 16 && ROWS == 4 && sizeof(T) == 4) ? 8 : 4;      // see the forward kernel
  struct StageB {
    int32_t ri[U];
    float av[U];
    typename Vec16<T>::raw go;
    float gt;
  };
  auto load_a = [&](int64_t p, int64_t& beg, int& n) {
    beg = 0;
    n = 0;
    if (p < N) {
      beg = ptr[p];
      n = (int)(ptr[p + 1] - beg);
      if (WAVE) {
        beg = uniform64(beg);
        n = __builtin_amdgcn_readfirstlane(n);
      }
    }
  };
  auto load_b = [&](int64_t p, int64_t beg, int n, StageB& b) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      b.ri[u] = 0;
      b.av[u] = 0.f;
    }
    b.go = raw_t();
    b.gt = 1.f;
    if (n > 0) {
      b.go = *reinterpret_cast<const raw_t*>(gout + p * C + (int64_t)lane_r * VEC);
      if (gate) b.gt = gate[p * G + g_lane];
      if (n <= U * tg_rows) {
        if (WAVE) {
          const int32_t* ri_base = row_idx ? row_idx + beg : nullptr;
          const float* a_base = att + beg * G;
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const uint32_t v = (uint32_t)(row_slot + u * tg_rows);
            const uint32_t vv = v < (uint32_t)n ? v : 0u;
            if (row_idx) b.ri[u] = ri_base[vv];
            b.av[u] = a_base[vv * (uint32_t)G + (uint32_t)g_lane];
          }
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int v = row_slot + u * tg_rows;
            const int64_t r = beg + (v < n ? v : 0);
            if (row_idx) b.ri[u] = row_idx[r];
            b.av[u] = att[r * G + g_lane];
          }
        }
      }
    }
  };
  const int64_t pstep = n_waves * teams_per_wave;
  int64_t pw = wave * teams_per_wave;
  int64_t beg, beg1, beg2;
  int n, n1, n2;
  StageB sb, sb1;
  load_a(pw + team_in_wave, beg, n);
  load_a(pw + pstep + team_in_wave, beg1, n1);
  load_b(pw + team_in_wave, beg, n, sb);
  for (; pw < N; pw += pstep, beg = beg1, n = n1, sb = sb1, beg1 = beg2, n1 = n2) {
    const int64_t p = pw + team_in_wave;
    load_b(p + pstep, beg1, n1, sb1);
    load_a(p + 2 * pstep, beg2, n2);
    const int64_t col = (int64_t)lane_r * VEC;
    float go[VEC];
    Vec16<T>::unpack(sb.go, go);
    const float gt = sb.gt;

    // ---- pass 1: d[v,g] = sum_{c in g} go[c]*val[v,c];  sum_ad[g] = sum_v att[v,g]*d[v,g]
    float sum_ad = 0.f;
    // short segments (the common case): row indices, value rows and attentions of the whole point are
    // loaded before the first use and d stays in registers (no round trip through grad_compat)
    const bool small = n <= U * tg_rows;
    // LDS form of the long path: fused gather only (no per-row grad_val writes), segment fits the team's share
    const int team_cap = DBUF / (64 / tg_ts);
    const bool lds_d = !small && row_idx && n * G <= team_cap;
    float* sd = s_dbuf[threadIdx.x >> 6] + (lane / tg_ts) * team_cap;
    float dreg[U], areg[U];
    bool okr[U];
    if (small) {
      raw_t x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int v = row_slot + u * tg_rows;
        okr[u] = v < n;
        areg[u] = sb.av[u];
        const int64_t rr = beg + (okr[u] ? v : 0);
        const int64_t ri = row_idx ? (int64_t)sb.ri[u] : rr;
        if (n > 0) x[u] = *reinterpret_cast<const raw_t*>(val + ri * C + col);
        else x[u] = raw_t();
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float f[VEC];
        Vec16<T>::unpack(x[u], f);
        float d = 0.f;
#pragma unroll
        for (int k = 0; k < VEC; ++k) d = fmaf(go[k], f[k], d);
        for (int off = 1; off < tg_lpg; off <<= 1) d += __shfl_xor(d, off);
        dreg[u] = d;
        if (g_first && okr[u]) sum_ad += areg[u] * d;
      }
    } else {
      // long segments: same load-first chunks, d goes through grad_compat (re-read by the same lane in 2a)
      for (int v0 = 0; v0 < n; v0 += U * tg_rows) {
        bool ok[U];
        int64_t rr[U], ri[U];
        float av[U];
        raw_t x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int v = v0 + row_slot + u * tg_rows;
          ok[u] = v < n;
          rr[u] = beg + (ok[u] ? v : 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          ri[u] = row_idx ? (int64_t)row_idx[rr[u]] : rr[u];
          av[u] = att[rr[u] * G + g_lane];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = *reinterpret_cast<const raw_t*>(val + ri[u] * C + col);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          float f[VEC];
          Vec16<T>::unpack(x[u], f);
          float d = 0.f;
#pragma unroll
          for (int k = 0; k < VEC; ++k) d = fmaf(go[k], f[k], d);
          for (int off = 1; off < tg_lpg; off <<= 1) d += __shfl_xor(d, off);
          if (g_first && ok[u]) {
            if (lds_d) sd[(rr[u] - beg) * G + g_lane] = d;
            else gcompat[rr[u] * G + g_lane] = d;
            sum_ad += av[u] * d;
          }
        }
      }
    }
    // lanes of one column position in the R row slots hold partials of the same group
    if constexpr (LPR > 0) sum_ad = team_sum<LPR, ROWS>(sum_ad);
    else
      for (int off = tg_lpr; off < tg_ts; off <<= 1) sum_ad += __shfl_xor(sum_ad, off);
    // make the group total visible to every lane of the group (only g_first lanes accumulated)
    sum_ad = __shfl(sum_ad, lane - (lane_r % tg_lpg));

    const float inv_dn = scaling ? 1.f / sqrtf((float)(n > 0 ? n : 1)) : 1.f;
    float g_mx = 0.f;
    const int64_t am = n > 0 ? amax[p * G + g_lane] : -1;
    if (gw && n > 0) {
      const float g_pre = (gt > 0.f) ? sum_ad * (1.f - gt * gt) : 0.f;
      if (g_first && row_slot == 0) {
        const float mx = compat[am * G + g_lane];
        atomicAdd(&s_wb[g_lane], g_pre * mx);
        atomicAdd(&s_wb[G + g_lane], g_pre);
      }
      g_mx = g_pre * gw[g_lane];
    }
    const float tt = gt * sum_ad;

    // ---- pass 2a: grad_compat (and grad_val when it is a dense [V, C] tensor)
#pragma unroll
    for (int k = 0; k < VEC; ++k) go[k] *= gt;
    auto emit = [&](int64_t r, float a, float d) {
      if (g_first) {
        float gc = a * (gt * d - tt) * inv_dn;
        if (r == am) gc += g_mx;
        if (WAVE) {
          const uint32_t vl = (uint32_t)(r - beg);   // view inside the point: 32-bit lane offsets, SGPR bases
          (gcompat + beg * G)[vl * (uint32_t)G + (uint32_t)g_lane] = gc;
          if (rec) {
            float* rb = rec + beg * rs;
            rb[vl * (uint32_t)rs + 1u + (uint32_t)g_lane] = a * gt;
            if (g_lane == 0) rb[vl * (uint32_t)rs] = __int_as_float((int)p);
          }
        } else {
          gcompat[r * G + g_lane] = gc;
          if (rec) {
            // view record for dva_view_gather_rows_grad: point id | gate * attention per group, one
            // 32-byte sector per view instead of three scattered reads (view_point, att, gate)
            rec[r * rs + 1 + g_lane] = a * gt;
            if (g_lane == 0) rec[r * rs] = __int_as_float((int)p);
          }
        }
      }
      if (!row_idx) {
        float f[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) f[k] = go[k] * a;
        *reinterpret_cast<raw_t*>(gval + r * C + col) = Vec16<T>::pack(f);
      }
    };
    if (small) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (okr[u]) emit(beg + row_slot + u * tg_rows, areg[u], dreg[u]);
    } else if (lds_d) {
      // lane i of the team takes (view, group) pair i of each batch of ts / G views: coalesced reads of the
      // attentions, coalesced writes of grad_compat, two iterations for 32 views instead of 32 serial ones.
      // The per-group scalars live in the lanes of that group: fetched from the group's first lane.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      const int g2 = li & (G - 1), vs = li / G, per = tg_ts / G;
      const int src = (lane - li) + g2 * tg_lpg;
      const float gt2 = __shfl(gt, src), tt2 = __shfl(tt, src), gmx2 = __shfl(g_mx, src);
      const int am2 = __shfl((int)(am - beg), src);
      for (int v0 = 0; v0 < n; v0 += per) {
        const int v = v0 + vs;
        if (v < n) {
          const int64_t r = beg + v;
          const float a = att[r * G + g2];
          float gc = a * (gt2 * sd[v * G + g2] - tt2) * inv_dn;
          if (v == am2) gc += gmx2;
          gcompat[r * G + g2] = gc;
          if (rec) {
            rec[r * rs + 1 + g2] = a * gt2;
            if (g2 == 0) rec[r * rs] = __int_as_float((int)p);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the buffer is reused by the team's next point
      __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll 2
      for (int v = row_slot; v < n; v += tg_rows) {
        const int64_t r = beg + v;
        emit(r, att[r * G + g_lane], g_first ? gcompat[r * G + g_lane] : 0.f);
      }
    }

    // ---- pass 2b: fused-gather scatter: grows[row_idx[v], c] += go[p,c]*gate[p,g(c)]*att[v,g(c)]
    if (row_idx && grows) {
      const int cpg = C / G;
      for (int t = 0; t < teams_per_wave; ++t) {
        const int src = t * tg_ts;
        const int nb = __shfl(n, src);
        if (nb == 0) continue;  // wave-uniform
        const int64_t pp = pw + t;
        const int64_t pb = __shfl((long long)beg, src);
        if (C <= 64) {
          const int c = lane & (C - 1), rsub = lane / C, rpi = 64 / C;
          const int gc = c / cpg;
          const float gq = Elt<T>::ld(gout, pp * C + c) * (gate ? gate[pp * G + gc] : 1.f);
          for (int v = rsub; v < nb; v += rpi) {
            const int64_t r = pb + v;
            atomicAdd(&grows[(int64_t)row_idx[r] * C + c], gq * att[r * G + gc]);
          }
        } else {
          for (int c = lane; c < C; c += 64) {
            const int gc = c / cpg;
            const float gq = Elt<T>::ld(gout, pp * C + c) * (gate ? gate[pp * G + gc] : 1.f);
            for (int v = 0; v < nb; ++v) {
              const int64_t r = pb + v;
              atomicAdd(&grows[(int64_t)row_idx[r] * C + c], gq * att[r * G + gc]);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  if (gwb && threadIdx.x < 2 * G && s_wb[threadIdx.x] != 0.f)
    atomicAdd(&gwb[threadIdx.x], s_wb[threadIdx.x]);
}

static inline bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

// Decide whether the fused team path applies and with which geometry.
template <typename T>
static bool team_geometry(int C, int G, int64_t N, int64_t V, TeamGeom* tg) {
  constexpr int VEC = Vec16<T>::N;
  if (C % VEC) return false;
  const int lpr = C / VEC;
  if (!is_pow2(lpr) || lpr > 64) return false;
  if (!is_pow2(G) || G > 32 || C % G) return false;
  if ((C / G) % VEC) return false;
  int rows = 64 / lpr;  // default: a whole wavefront per point
  // shrink the team for short segments so that lanes are not idle: the short-segment path covers 4 * rows
  // views, so rows ~ 0.75 x the average segment keeps ~95 % of geometric-tailed segments on it while
  // several points share a wavefront
  const double avg = N > 0 ? (double)V / (double)N : 0.0;
  while (rows > 1 && (double)(rows / 2) >= 0.75 * avg && lpr * (rows / 2) >= G) rows >>= 1;
  while (lpr * rows < G) rows <<= 1;
  if (lpr * rows > 64) return false;
  tg->lpr = lpr;
  tg->rows = rows;
  tg->ts = lpr * rows;
  tg->lpg = lpr / G;
  return tg->lpg >= 1;
}

static inline int grid_cap(int64_t blocks) {
  const int64_t cap = 256 * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

template <typename T>
static int fwd_impl(const void* val, const int32_t* row_idx, const float* compat, const int64_t* ptr,
                    const float* gw,
                    const float* gb, void* out, float* att, float* gate, int32_t* amax, int64_t N,
                    int64_t V_hint, int C, int G, int scaling, float eps, int algo, hipStream_t s) {
  TeamGeom tg;
  const bool team_ok = team_geometry<T>(C, G, N, V_hint, &tg);
  if (algo == 2 && !team_ok) return DVA_ERR_UNSUPPORTED;
  if (algo != 1 && team_ok) {
    const int tpb = 256 / tg.ts;
    const int grid = grid_cap((N + tpb - 1) / tpb);
#define DVA_FWD_TEAM(L, R)                                                                          \
  hipLaunchKernelGGL((att_fwd_team_kernel<T, L, R>), dim3(grid), dim3(256), 0, s, (const T*)val,    \
                     row_idx, compat, ptr, gw, gb, (T*)out, att, gate, amax, N, C, G, scaling, eps, tg)
    if (tg.lpr == 8 && tg.rows == 8) DVA_FWD_TEAM(8, 8);          // C = 64 bf16, ~32 views per point
    else if (tg.lpr == 16 && tg.rows == 4) DVA_FWD_TEAM(16, 4);   // C = 64 fp32 / C = 128 bf16
    else if (tg.lpr == 8 && tg.rows == 4) DVA_FWD_TEAM(8, 4);     // the same with ~4-8 views per point (ragged)
    else if (tg.lpr == 16 && tg.rows == 2) DVA_FWD_TEAM(16, 2);
    else DVA_FWD_TEAM(0, 0);
#undef DVA_FWD_TEAM
    return DVA_OK;
  }
  hipLaunchKernelGGL(att_scores_kernel, dim3(grid_cap((N * G + 255) / 256)), dim3(256), 0, s, compat,
                     ptr, gw, gb, att, gate, amax, N, G, scaling, eps);
  hipLaunchKernelGGL((att_wsum_kernel<T>), dim3(grid_cap((N * C + 255) / 256)), dim3(256), 0, s,
                     (const T*)val, row_idx, att, gw ? gate : nullptr, ptr, (T*)out, N, C, G);
  return DVA_OK;
}

template <typename T>
static int bwd_impl(const void* gout, const void* val, const int32_t* row_idx, float* grows,
                    const float* compat, const float* att,
                    const float* gate, const int32_t* amax, const int64_t* ptr, const float* gw,
                    void* gval, float* gcompat, float* gwb, float* rec, int rs, int64_t N, int64_t V_hint,
                    int C, int G, int scaling, int algo, hipStream_t s) {
  TeamGeom tg;
  const bool team_ok = team_geometry<T>(C, G, N, V_hint, &tg);
  if (algo == 2 && !team_ok) return DVA_ERR_UNSUPPORTED;
  if (rec && !(algo != 1 && team_ok)) return DVA_ERR_UNSUPPORTED;  // records come from the team kernel
  if (algo != 1 && team_ok) {
    const int tpb = 256 / tg.ts;
    const int grid = grid_cap((N + tpb - 1) / tpb);
#define DVA_BWD_TEAM(L, R)                                                                          \
  hipLaunchKernelGGL((att_bwd_team_kernel<T, L, R>), dim3(grid), dim3(256), 0, s, (const T*)gout,   \
                     (const T*)val, row_idx, grows, compat, att, gw ? gate : nullptr, amax, ptr, gw, \
                     (T*)gval, gcompat, gwb, rec, rs, N, C, G, scaling, tg)
    if (tg.lpr == 8 && tg.rows == 8) DVA_BWD_TEAM(8, 8);
    else if (tg.lpr == 16 && tg.rows == 4) DVA_BWD_TEAM(16, 4);
    else if (tg.lpr == 8 && tg.rows == 4) DVA_BWD_TEAM(8, 4);
    else if (tg.lpr == 16 && tg.rows == 2) DVA_BWD_TEAM(16, 2);
    else DVA_BWD_TEAM(0, 0);
#undef DVA_BWD_TEAM
    return DVA_OK;
  }
  hipLaunchKernelGGL((att_bwd_scores_kernel<T>), dim3(grid_cap((N * G + 255) / 256)), dim3(256),
                     2 * G * sizeof(float), s, (const T*)gout, (const T*)val, row_idx, compat, att,
                     gw ? gate : nullptr, amax, ptr, gw, gcompat, gwb, N, C, G, scaling);
  if (!row_idx || grows)
    hipLaunchKernelGGL((att_bwd_val_kernel<T>), dim3(grid_cap((N * C + 255) / 256)), dim3(256), 0, s,
                       (const T*)gout, row_idx, grows, att, gw ? gate : nullptr, ptr, (T*)gval, N, C, G);
  return DVA_OK;
}

// ------------------------------------------------------------------------------------------------
// rows gradient of the fused gather as a segmented reduction (no atomics, deterministic):
//   grows[r, c] = sum_{i in [row_ptr[r], row_ptr[r+1])} gout[p, c] * gate[p, g(c)] * att[v, g(c)],
//   v = perm[i], p = view_point[v].  (perm, row_ptr) = dva_row_plan of the row index.
// One wavefront per feature-map row, 64/lpr views in flight, 4 loads deep: the dependent chain
// perm -> view_point -> gout row is issued for 4 views before the first use.
// ------------------------------------------------------------------------------------------------
// REC16: rec holds packed 16-byte records {int32 point, 4 x bf16 gate * attention, 4 bytes unused} (G <= 4) instead of
// fp32 records of stride rs.
// TO = float: fp32 rows out; TO = T (bf16 grad_out, REC16 callers whose map is bf16): the row is rounded where it is
// summed -- no fp32 [R, C] tensor and no conversion pass behind the kernel (round 5: 64 us of a 3.3 ms step at C = 512)
template <typename T, bool REC16 = false, typename TO = float>
__global__ __launch_bounds__(256) void rows_grad_team_kernel(
    const T* __restrict__ gout, const float* __restrict__ att, const float* __restrict__ gate,
    const int32_t* __restrict__ vp, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ row_ptr, const float* __restrict__ rec, int rs,
    TO* __restrict__ grows, int64_t R, int C, int G, int lpr, int lpg, int c0 = 0) {
  // C = row stride in elements; a launch covers the channels [c0, c0 + lpr * VEC) of every row (c0 = 0, lpr * VEC = C:
  // whole rows; channel slabs: dva_view_gather_rows_grad_rec16)
  constexpr int VEC = Vec16<T>::N;
  constexpr int U = 4;
  typedef typename Vec16<T>::raw raw_t;
  const int lane = threadIdx.x & 63;
  const int lane_r = lane & (lpr - 1);
  const int slot = lane / lpr;
  const int slots = 64 / lpr;
  const int g_lane = (c0 / VEC + lane_r) / lpg;
  const int64_t col = (int64_t)c0 + (int64_t)lane_r * VEC;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < R; r += n_waves) {
    const int beg = row_ptr[r], end = row_ptr[r + 1];
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    for (int i0 = beg; i0 < end; i0 += slots * U) {
      int v[U], p[U];
      bool ok[U];
      float sc[U];
      raw_t raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * slots + slot;
        ok[u] = i < end;
        v[u] = (REC16 && !perm) ? (ok[u] ? i : beg) : perm[ok[u] ? i : beg];   // perm == NULL: records in plan order
      }
      if (REC16) {
        // one 16-byte record per view: point id | gate * attention per group as bf16
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t* rv = reinterpret_cast<const uint32_t*>(rec) + (int64_t)v[u] * 4;
          p[u] = (int)rv[0];
          const uint32_t w2 = rv[1 + (g_lane >> 1)];
          sc[u] = ok[u] ? __uint_as_float((g_lane & 1) ? (w2 & 0xffff0000u) : (w2 << 16)) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) raw[u] = *reinterpret_cast<const raw_t*>(gout + (int64_t)p[u] * C + col);
      } else if (rec) {
        // one 32-byte record per view: point id | gate * attention per group
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float* rv = rec + (int64_t)v[u] * rs;
          p[u] = __float_as_int(rv[0]);
          sc[u] = ok[u] ? rv[1 + g_lane] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) raw[u] = *reinterpret_cast<const raw_t*>(gout + (int64_t)p[u] * C + col);
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) p[u] = vp[v[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          raw[u] = *reinterpret_cast<const raw_t*>(gout + (int64_t)p[u] * C + col);
          const float a = att[(int64_t)v[u] * G + g_lane];
          const float gt = gate ? gate[(int64_t)p[u] * G + g_lane] : 1.f;
          sc[u] = ok[u] ? a * gt : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float f[VEC];
        Vec16<T>::unpack(raw[u], f);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = fmaf(f[k], sc[u], acc[k]);
      }
    }
    for (int off = lpr; off < 64; off <<= 1) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += __shfl_xor(acc[k], off);
    }
    if (slot == 0) {
      if constexpr (sizeof(TO) == 4) {
        float* dst = reinterpret_cast<float*>(grows) + r * C + col;
#pragma unroll
        for (int k = 0; k < VEC; k += 4)
          *reinterpret_cast<float4*>(dst + k) = make_float4(acc[k], acc[k + 1], acc[k + 2], acc[k + 3]);
      } else {
        *reinterpret_cast<raw_t*>(reinterpret_cast<T*>(grows) + r * C + col) = Vec16<T>::pack(acc);
      }
    }
  }
}

// Sparse row plans (at most ~2 views per row: the identity gather of a view-level tensor, ops.gather_segment_max): one
// lane team per ROW instead of one wavefront per row -- 64 / lpr rows per wavefront, no cross-slot reduction.
template <typename T, typename TO = float>
__global__ __launch_bounds__(256) void rows_grad_short_rec16_kernel(const T* __restrict__ gout,
                                                                     const int32_t* __restrict__ perm,
                                                                     const int32_t* __restrict__ row_ptr,
                                                                     const uint32_t* __restrict__ rec,
                                                                     TO* __restrict__ grows, int64_t R, int C, int lpr,
                                                                     int lpg) {
  constexpr int VEC = Vec16<T>::N;
  typedef typename Vec16<T>::raw raw_t;
  const int lane = threadIdx.x & 63;
  const int lane_r = lane & (lpr - 1), slot = lane / lpr, slots = 64 / lpr;
  const int g_lane = lane_r / lpg;
  const int64_t col = (int64_t)lane_r * VEC;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave * slots + slot; r < R; r += n_waves * slots) {
    const int beg = row_ptr[r], end = row_ptr[r + 1];
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    for (int i = beg; i < end; ++i) {
      const int64_t v = perm ? perm[i] : i;
      const uint32_t* rv = rec + v * 4;
      const int64_t p = (int)rv[0];
      const uint32_t w2 = rv[1 + (g_lane >> 1)];
      const float sc = __uint_as_float((g_lane & 1) ? (w2 & 0xffff0000u) : (w2 << 16));
      float f[VEC];
      Vec16<T>::unpack(*reinterpret_cast<const raw_t*>(gout + p * C + col), f);
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] = fmaf(f[k], sc, acc[k]);
    }
    if constexpr (sizeof(TO) == 4) {
      float* dst = reinterpret_cast<float*>(grows) + r * C + col;
#pragma unroll
      for (int k = 0; k < VEC; k += 4)
        *reinterpret_cast<float4*>(dst + k) = make_float4(acc[k], acc[k + 1], acc[k + 2], acc[k + 3]);
    } else {
      *reinterpret_cast<raw_t*>(reinterpret_cast<T*>(grows) + r * C + col) = Vec16<T>::pack(acc);
    }
  }
}

// any C / G: one thread per (row, channel)
template <typename T>
__global__ __launch_bounds__(256) void rows_grad_generic_kernel(
    const T* __restrict__ gout, const float* __restrict__ att, const float* __restrict__ gate,
    const int32_t* __restrict__ vp, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ row_ptr, const float* __restrict__ rec, int rs,
    float* __restrict__ grows, int64_t R, int C, int G) {
  const int64_t total = R * (int64_t)C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / C;
    const int c = (int)(t - r * C);
    const int g = group_of_channel(c, C, G);
    float acc = 0.f;
    for (int i = row_ptr[r]; i < row_ptr[r + 1]; ++i) {
      const int64_t v = perm[i];
      const int64_t p = rec ? (int64_t)__float_as_int(rec[v * rs]) : (int64_t)vp[v];
      const float s = rec ? rec[v * rs + 1 + g] : att[v * G + g] * (gate ? gate[p * G + g] : 1.f);
      acc = fmaf(Elt<T>::ld(gout, p * C + c), s, acc);
    }
    grows[t] = acc;
  }
}

// Backward of a plain nearest gather (x_mod[p] = rows[row_idx[p]]) over the row plan:
// grows[r, :] = sum over the atoms i of row r of gout[perm[i], :]   (deterministic, no atomics).
// With `weights`: the plan is over ENTRIES (entry e belongs to atom e >> shift and carries weights[e]), e.g.
// the 4 bilinear corner taps of every atom (shift = 2).
template <typename T>
__global__ __launch_bounds__(256) void rows_sum_team_kernel(const T* __restrict__ gout,
                                                             const int32_t* __restrict__ perm,
                                                             const int32_t* __restrict__ row_ptr,
                                                             const float* __restrict__ weights, int shift,
                                                             float* __restrict__ grows, int64_t R, int C,
                                                             int lpr) {
  constexpr int VEC = Vec16<T>::N;
  constexpr int U = 4;
  typedef typename Vec16<T>::raw raw_t;
  const int lane = threadIdx.x & 63;
  const int lane_r = lane & (lpr - 1), slot = lane / lpr, slots = 64 / lpr;
  const int64_t col = (int64_t)lane_r * VEC;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < R; r += n_waves) {
    const int beg = row_ptr[r], end = row_ptr[r + 1];
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    for (int i0 = beg; i0 < end; i0 += slots * U) {
      int v[U];
      bool ok[U];
      raw_t raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * slots + slot;
        ok[u] = i < end;
        v[u] = perm[ok[u] ? i : beg];
      }
      float wu[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        raw[u] = *reinterpret_cast<const raw_t*>(gout + (int64_t)(v[u] >> shift) * C + col);
        wu[u] = ok[u] ? (weights ? weights[v[u]] : 1.f) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float f[VEC];
        Vec16<T>::unpack(raw[u], f);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = fmaf(wu[u], f[k], acc[k]);
      }
    }
    for (int off = lpr; off < 64; off <<= 1) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += __shfl_xor(acc[k], off);
    }
    if (slot == 0) {
      float* dst = grows + r * C + col;
#pragma unroll
      for (int k = 0; k < VEC; k += 4)
        *reinterpret_cast<float4*>(dst + k) = make_float4(acc[k], acc[k + 1], acc[k + 2], acc[k + 3]);
    }
  }
}

// Backward of the bilinear gather over the ANCHOR plan (views grouped by the padded cell of their top-left tap):
// S[a][k][:] = sum over the views v of anchor a of weights[v][k] * gout[v, :]  for the four taps k -- every view row
// is read ONCE (the plan over the 4 V tap entries reads it four times and sorts four times as many keys); the map
// gradient follows from S by dva_anchor_combine.  Deterministic, no atomics.
// AFF (the fused bilinear path, csrc/chain_emod.hip): the rows are dy_a in the chain kernels' position order and the
// BatchNorm_a backward dz_a = G dy_a - K1 - K2 z_a (constants as chain_emod.hip stage_tab_c builds them from bn =
// mean | invstd | gamma | beta and sm = S1 / M | S2_hat / M, natural channel order) is folded into the scatter: the
// separate in-place pass over [V][C] (read 2 rows, write 1) disappears.
//   AFF = 1: applied to every row as it is read, from the stored z_a row of the view (a second random row per view);
//   AFF = 2: at the level of the anchor.  All views of an anchor interpolate the SAME four rows of Y, z_a[v] =
//            sum_k' w_k'(v) Y[r_k'], so  sum_v w_k(v) z_a[v] = sum_k' M[k][k'] Y[r_k']  with the 4 x 4 Gram matrix
//            M = sum_v w(v) w(v)^T of the anchor's tap weights -- accumulated here from the weights the kernel reads
//            anyway -- and  S[a][k] = G sum_v w_k dy_a[v] - K1 sum_v w_k - K2 sum_k' M[k][k'] Y[r_k']:
//            one random row per view again, plus four rows of Y per anchor.
__device__ __forceinline__ int position_channel(int p) {      // position 32 b + 16 h + r holds channel 32 b + chan(r, h)
  const int r = p & 15, h = (p >> 4) & 1;
  return (p & ~31) + (r & 3) + 8 * (r >> 2) + 4 * h;
}
template <typename T, int AFF>
__global__ __launch_bounds__(256) void anchor_rows_sum_kernel(const T* __restrict__ gout,
                                                               const int32_t* __restrict__ perm,
                                                               const int32_t* __restrict__ row_ptr,
                                                               const float4* __restrict__ weights,
                                                               float* __restrict__ S, int64_t R, int C, int lpr,
                                                               const T* __restrict__ zrows, const float* __restrict__ bn,
                                                               const float* __restrict__ sm,
                                                               const int4* __restrict__ tap_rows,
                                                               const T* __restrict__ Y) {
  constexpr int VEC = Vec16<T>::N;
  constexpr int U = 2;
  typedef typename Vec16<T>::raw raw_t;
  const int lane = threadIdx.x & 63;
  const int lane_r = lane & (lpr - 1), slot = lane / lpr, slots = 64 / lpr;
  const int64_t col = (int64_t)lane_r * VEC;
  // AFF = 1: the constants of the lane's VEC positions live in registers (used per row); AFF = 2: they are needed
  // once per anchor, by the slot-0 lanes: an LDS table [3][C] instead of 24 registers (118 -> 94 VGPRs)
  float cg[AFF == 1 ? VEC : 1], ck1[AFF == 1 ? VEC : 1], ck2[AFF == 1 ? VEC : 1];
  __shared__ float s_aff[AFF == 2 ? 3 * 512 : 1];
  if (AFF == 1) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int c = position_channel((int)col + e);
      const float mean = bn[c], inv = bn[C + c], g = bn[2 * C + c] * inv, s1 = sm[c], s2 = sm[C + c];
      cg[e] = g;
      ck1[e] = g * (s1 - mean * inv * s2);
      ck2[e] = g * inv * s2;
    }
  }
  if (AFF == 2) {
    for (int p = threadIdx.x; p < C; p += blockDim.x) {
      const int c = position_channel(p);
      const float mean = bn[c], inv = bn[C + c], g = bn[2 * C + c] * inv, s1 = sm[c], s2 = sm[C + c];
      s_aff[p] = g;
      s_aff[512 + p] = g * (s1 - mean * inv * s2);
      s_aff[1024 + p] = g * inv * s2;
    }
    __syncthreads();
  }
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < R; r += n_waves) {
    const int beg = row_ptr[r], end = row_ptr[r + 1];
    float acc[4][VEC];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[k][e] = 0.f;
    }
    float gm[AFF == 2 ? 10 : 1], w1[AFF == 2 ? 4 : 1];      // upper triangle of M, sum of the weights
    if (AFF == 2) {
#pragma unroll
      for (int i = 0; i < 10; ++i) gm[i] = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) w1[k] = 0.f;
    }
    for (int i0 = beg; i0 < end; i0 += slots * U) {
      int v[U];
      bool ok[U];
      raw_t raw[U], zraw[AFF == 1 ? U : 1];
      float4 wu[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * slots + slot;
        ok[u] = i < end;
        v[u] = perm[ok[u] ? i : beg];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        raw[u] = *reinterpret_cast<const raw_t*>(gout + (int64_t)v[u] * C + col);
        if (AFF == 1) zraw[u] = *reinterpret_cast<const raw_t*>(zrows + (int64_t)v[u] * C + col);
        wu[u] = ok[u] ? weights[v[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float f[VEC];
        Vec16<T>::unpack(raw[u], f);
        if (AFF == 1) {
          float zf[VEC];
          Vec16<T>::unpack(zraw[u], zf);
#pragma unroll
          for (int e = 0; e < VEC; ++e) f[e] = fmaf(-ck2[e], zf[e], fmaf(cg[e], f[e], -ck1[e]));
        }
        const float ww[4] = {wu[u].x, wu[u].y, wu[u].z, wu[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[k][e] = fmaf(ww[k], f[e], acc[k][e]);
        }
        if (AFF == 2) {
          int t = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            w1[k] += ww[k];
#pragma unroll
            for (int k2 = k; k2 < 4; ++k2, ++t) gm[t] = fmaf(ww[k], ww[k2], gm[t]);
          }
        }
      }
    }
    for (int off = lpr; off < 64; off <<= 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[k][e] += __shfl_xor(acc[k][e], off);
      }
      if (AFF == 2) {
#pragma unroll
        for (int i = 0; i < 10; ++i) gm[i] += __shfl_xor(gm[i], off);
#pragma unroll
        for (int k = 0; k < 4; ++k) w1[k] += __shfl_xor(w1[k], off);
      }
    }
    if (slot == 0) {
      if (AFF == 2 && end > beg) {
        // the four rows of Y every view of this anchor interpolates (border-replicated rows may coincide)
        const int4 tr = tap_rows[perm[beg]];
        const int rr[4] = {tr.x, tr.y, tr.z, tr.w};
        const float m4[4][4] = {{gm[0], gm[1], gm[2], gm[3]}, {gm[1], gm[4], gm[5], gm[6]},
                                {gm[2], gm[5], gm[7], gm[8]}, {gm[3], gm[6], gm[8], gm[9]}};
        float k2_[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float g_ = s_aff[col + e], k1_ = s_aff[512 + col + e];
          k2_[e] = s_aff[1024 + col + e];
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k][e] = fmaf(g_, acc[k][e], -k1_ * w1[k]);
        }
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {       // one row of Y at a time: 8 registers instead of 32
          float y[VEC];
          Vec16<T>::unpack(*reinterpret_cast<const raw_t*>(Y + (int64_t)rr[k2] * C + col), y);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[k][e] = fmaf(-k2_[e] * m4[k][k2], y[e], acc[k][e]);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float* dst = S + (r * 4 + k) * C + col;
#pragma unroll
        for (int e = 0; e < VEC; e += 4)
          *reinterpret_cast<float4*>(dst + e) = make_float4(acc[k][e], acc[k][e + 1], acc[k][e + 2], acc[k][e + 3]);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void rows_sum_generic_kernel(const T* __restrict__ gout,
                                                                const int32_t* __restrict__ perm,
                                                                const int32_t* __restrict__ row_ptr,
                                                                const float* __restrict__ weights, int shift,
                                                                float* __restrict__ grows, int64_t R, int C) {
  const int64_t total = R * (int64_t)C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / C;
    const int c = (int)(t - r * C);
    float acc = 0.f;
    for (int i = row_ptr[r]; i < row_ptr[r + 1]; ++i) {
      const int e = perm[i];
      acc = fmaf(weights ? weights[e] : 1.f, Elt<T>::ld(gout, (int64_t)(e >> shift) * C + c), acc);
    }
    grows[t] = acc;
  }
}

template <typename T>
static int rows_sum_impl(const void* gout, const int32_t* perm, const int32_t* row_ptr, const float* weights,
                         int shift, float* grows, int64_t R, int C, hipStream_t s) {
  constexpr int VEC = Vec16<T>::N;
  const int lpr = C / VEC;
  const bool team_ok = (C % VEC) == 0 && is_pow2(lpr) && lpr <= 64 && ((uintptr_t)gout % 16 == 0) &&
                       ((uintptr_t)grows % 16 == 0);
  if (team_ok)
    hipLaunchKernelGGL((rows_sum_team_kernel<T>), dim3(grid_cap((R + 3) / 4)), dim3(256), 0, s, (const T*)gout,
                       perm, row_ptr, weights, shift, grows, R, C, lpr);
  else
    hipLaunchKernelGGL((rows_sum_generic_kernel<T>), dim3(grid_cap((R * C + 255) / 256)), dim3(256), 0, s,
                       (const T*)gout, perm, row_ptr, weights, shift, grows, R, C);
  return DVA_OK;
}

template <typename T>
static int rows_grad_impl(const void* gout, const float* att, const float* gate, const int32_t* vp,
                          const int32_t* perm, const int32_t* row_ptr, const float* rec, int rs,
                          float* grows, int64_t R, int C, int G, hipStream_t s) {
  constexpr int VEC = Vec16<T>::N;
  const int lpr = C / VEC;
  const bool team_ok = (C % VEC) == 0 && is_pow2(lpr) && lpr <= 64 && is_pow2(G) && C % G == 0 &&
                       (C / G) % VEC == 0 && ((uintptr_t)gout % 16 == 0) && ((uintptr_t)grows % 16 == 0);
  if (team_ok) {
    hipLaunchKernelGGL((rows_grad_team_kernel<T>), dim3(grid_cap((R + 3) / 4)), dim3(256), 0, s,
                       (const T*)gout, att, gate, vp, perm, row_ptr, rec, rs, grows, R, C, G, lpr, lpr / G);
  } else {
    hipLaunchKernelGGL((rows_grad_generic_kernel<T>), dim3(grid_cap((R * C + 255) / 256)), dim3(256),
                       0, s, (const T*)gout, att, gate, vp, perm, row_ptr, rec, rs, grows, R, C, G);
  }
  return DVA_OK;
}

}  // namespace dva

using namespace dva;

extern "C" {

static int attention_fwd_entry(const void* val, const int32_t* row_idx, const float* compat,
                               const int64_t* ptr, const float* gate_w, const float* gate_b, void* out,
                               float* att, float* gate, int32_t* amax, int64_t n_points,
                               int64_t n_views, int32_t C, int32_t G, int32_t scaling, float eps,
                               int32_t dtype, int32_t algo, void* stream) {
  if (n_points < 0 || n_views < 0 || C <= 0 || G <= 0 || G > C || !ptr) return DVA_ERR_INVALID;
  if ((gate_w == nullptr) != (gate_b == nullptr)) return DVA_ERR_INVALID;
  if (!out || !att || !gate || !amax) return DVA_ERR_INVALID;
  if (algo < 0 || algo > 2) return DVA_ERR_INVALID;
  if (n_points == 0) return DVA_OK;
  int rc;
  if (dtype == DVA_F32)
    rc = fwd_impl<float>(val, row_idx, compat, ptr, gate_w, gate_b, out, att, gate, amax, n_points,
                         n_views, C, G, scaling, eps, algo, (hipStream_t)stream);
  else if (dtype == DVA_BF16)
    rc = fwd_impl<bf16_t>(val, row_idx, compat, ptr, gate_w, gate_b, out, att, gate, amax, n_points,
                          n_views, C, G, scaling, eps, algo, (hipStream_t)stream);
  else
    return DVA_ERR_INVALID;
  if (rc) return rc;
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

static int attention_bwd_entry(const void* grad_out, const void* val, const int32_t* row_idx,
                               float* grad_rows, const float* compat, const float* att,
                               const float* gate, const int32_t* amax, const int64_t* ptr,
                               const float* gate_w, const float* gate_b, void* grad_val,
                               float* grad_compat, float* grad_gate_wb, float* view_rec,
                               int32_t rec_stride, int64_t n_points, int64_t n_views, int32_t C,
                               int32_t G, int32_t scaling, int32_t dtype, int32_t algo, void* stream) {
  (void)gate_b;
  if (view_rec && rec_stride < G + 1) return DVA_ERR_INVALID;
  if (n_points < 0 || n_views < 0 || C <= 0 || G <= 0 || G > C || !ptr) return DVA_ERR_INVALID;
  if (!att || !gate || !amax || !grad_compat) return DVA_ERR_INVALID;
  if (!row_idx && !grad_val) return DVA_ERR_INVALID;
  if (gate_w && !grad_gate_wb) return DVA_ERR_INVALID;
  if (algo < 0 || algo > 2) return DVA_ERR_INVALID;
  if (n_points == 0) return DVA_OK;
  int rc;
  if (dtype == DVA_F32)
    rc = bwd_impl<float>(grad_out, val, row_idx, grad_rows, compat, att, gate, amax, ptr, gate_w,
                         grad_val, grad_compat, grad_gate_wb, view_rec, rec_stride, n_points, n_views, C, G,
                         scaling, algo, (hipStream_t)stream);
  else if (dtype == DVA_BF16)
    rc = bwd_impl<bf16_t>(grad_out, val, row_idx, grad_rows, compat, att, gate, amax, ptr, gate_w,
                          grad_val, grad_compat, grad_gate_wb, view_rec, rec_stride, n_points, n_views, C, G,
                          scaling, algo, (hipStream_t)stream);
  else
    return DVA_ERR_INVALID;
  if (rc) return rc;
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_view_attention_fwd(const void* val, const float* compat, const int64_t* ptr,
                           const float* gate_w, const float* gate_b, void* out, float* att,
                           float* gate, int32_t* amax, int64_t n_points, int64_t n_views, int32_t C,
                           int32_t G, int32_t scaling, float eps, int32_t dtype, int32_t algo,
                           void* stream) {
  return attention_fwd_entry(val, nullptr, compat, ptr, gate_w, gate_b, out, att, gate, amax, n_points,
                             n_views, C, G, scaling, eps, dtype, algo, stream);
}

int dva_view_attention_bwd(const void* grad_out, const void* val, const float* compat,
                           const float* att, const float* gate, const int32_t* amax,
                           const int64_t* ptr, const float* gate_w, const float* gate_b,
                           void* grad_val, float* grad_compat, float* grad_gate_wb,
                           int64_t n_points, int64_t n_views, int32_t C, int32_t G, int32_t scaling,
                           int32_t dtype, int32_t algo, void* stream) {
  return attention_bwd_entry(grad_out, val, nullptr, nullptr, compat, att, gate, amax, ptr, gate_w,
                             gate_b, grad_val, grad_compat, grad_gate_wb, nullptr, 0, n_points, n_views, C,
                             G, scaling, dtype, algo, stream);
}

int dva_view_gather_attention_fwd(const void* rows, const int32_t* row_idx, const float* compat,
                                  const int64_t* ptr, const float* gate_w, const float* gate_b,
                                  void* out, float* att, float* gate, int32_t* amax,
                                  int64_t n_points, int64_t n_views, int32_t C, int32_t G,
                                  int32_t scaling, float eps, int32_t dtype, int32_t algo,
                                  void* stream) {
  if (!row_idx && n_views > 0) return DVA_ERR_INVALID;
  return attention_fwd_entry(rows, row_idx, compat, ptr, gate_w, gate_b, out, att, gate, amax, n_points,
                             n_views, C, G, scaling, eps, dtype, algo, stream);
}

int dva_view_gather_attention_bwd(const void* grad_out, const void* rows, const int32_t* row_idx,
                                  const float* compat, const float* att, const float* gate,
                                  const int32_t* amax, const int64_t* ptr, const float* gate_w,
                                  const float* gate_b, float* grad_rows, float* grad_compat,
                                  float* grad_gate_wb, float* view_rec, int32_t rec_stride,
                                  int64_t n_points, int64_t n_views, int32_t C, int32_t G,
                                  int32_t scaling, int32_t dtype, int32_t algo, void* stream) {
  if (!row_idx && n_views > 0) return DVA_ERR_INVALID;
  return attention_bwd_entry(grad_out, rows, row_idx, grad_rows, compat, att, gate, amax, ptr, gate_w,
                             gate_b, nullptr, grad_compat, grad_gate_wb, view_rec, rec_stride, n_points,
                             n_views, C, G, scaling, dtype, algo, stream);
}

static int rows_grad_rec16_impl(const void* grad_out, const int32_t* perm, const int32_t* row_ptr,
                                const void* view_rec16, void* grad_rows, int32_t out_dtype, int64_t n_rows,
                                int64_t n_views, int32_t C, int32_t G, int32_t dtype, void* stream) {
  if (n_rows < 0 || n_views < 0 || C <= 0 || G <= 0 || G > 4 || (G & (G - 1))) return DVA_ERR_INVALID;
  if (out_dtype != DVA_F32 && out_dtype != DVA_BF16) return DVA_ERR_INVALID;
  if (n_views > 0x7fffffffLL) return DVA_ERR_UNSUPPORTED;
  if (n_rows == 0) return DVA_OK;
  if (!row_ptr || !grad_rows) return DVA_ERR_INVALID;
  if (n_views > 0 && (!grad_out || !view_rec16)) return DVA_ERR_INVALID;     // perm NULL: records in plan order
  if (dtype != DVA_BF16) return DVA_ERR_UNSUPPORTED;
  const int lpr = C / 8;
  if ((C % 8) || !is_pow2(lpr) || lpr > 64 || (C % G) || ((C / G) % 8) || ((uintptr_t)grad_out % 16) ||
      ((uintptr_t)grad_rows % 16))
    return DVA_ERR_UNSUPPORTED;
  // wide rows in channel slabs (one launch per slab: the slab of grad_out, n_points x slab x 2 bytes, is what the
  // 32 re-reads of a point's row hit): DVA_ROWS_GRAD_SLAB = channels per slab, 0 = whole rows (A/B in profiles/r04*)
  static const int slab_env = tune_int("DVA_ROWS_GRAD_SLAB", 0);
  int slab = slab_env;
  if (slab <= 0 || slab >= C || (slab % 8) || !is_pow2(slab / 8) || (C % slab)) slab = C;
  const int lpr_s = slab / 8;
  const bool to_bf16 = out_dtype == DVA_BF16;
  if (n_views <= 2 * n_rows && lpr < 64) {
    // sparse plan (view-level identity gathers): a lane team per row
    const int64_t waves = (n_rows + (64 / lpr) - 1) / (64 / lpr);
    if (to_bf16)
      hipLaunchKernelGGL((rows_grad_short_rec16_kernel<bf16_t, bf16_t>), dim3(grid_cap((waves + 3) / 4)), dim3(256), 0,
                         (hipStream_t)stream, (const bf16_t*)grad_out, perm, row_ptr, (const uint32_t*)view_rec16,
                         (bf16_t*)grad_rows, n_rows, (int)C, lpr, lpr / G);
    else
      hipLaunchKernelGGL((rows_grad_short_rec16_kernel<bf16_t>), dim3(grid_cap((waves + 3) / 4)), dim3(256), 0,
                         (hipStream_t)stream, (const bf16_t*)grad_out, perm, row_ptr, (const uint32_t*)view_rec16,
                         (float*)grad_rows, n_rows, (int)C, lpr, lpr / G);
    DVA_CHECK_LAUNCH();
    return DVA_OK;
  }
  for (int c0 = 0; c0 < C; c0 += slab) {
    if (to_bf16)
      hipLaunchKernelGGL((rows_grad_team_kernel<bf16_t, true, bf16_t>), dim3(grid_cap((n_rows + 3) / 4)),
                         dim3(256), 0, (hipStream_t)stream, (const bf16_t*)grad_out, (const float*)nullptr,
                         (const float*)nullptr, (const int32_t*)nullptr, perm, row_ptr, (const float*)view_rec16, 4,
                         (bf16_t*)grad_rows, n_rows, (int)C, (int)G, lpr_s, lpr / G, c0);
    else
      hipLaunchKernelGGL((rows_grad_team_kernel<bf16_t, true>), dim3(grid_cap((n_rows + 3) / 4)),
                         dim3(256), 0, (hipStream_t)stream, (const bf16_t*)grad_out, (const float*)nullptr,
                         (const float*)nullptr, (const int32_t*)nullptr, perm, row_ptr, (const float*)view_rec16, 4,
                         (float*)grad_rows, n_rows, (int)C, (int)G, lpr_s, lpr / G, c0);
  }
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_view_gather_rows_grad_rec16(const void* grad_out, const int32_t* perm, const int32_t* row_ptr,
                                    const void* view_rec16, float* grad_rows, int64_t n_rows, int64_t n_views,
                                    int32_t C, int32_t G, int32_t dtype, void* stream) {
  return rows_grad_rec16_impl(grad_out, perm, row_ptr, view_rec16, grad_rows, DVA_F32, n_rows, n_views, C, G, dtype,
                              stream);
}

int dva_view_gather_rows_grad_rec16_to(const void* grad_out, const int32_t* perm, const int32_t* row_ptr,
                                       const void* view_rec16, void* grad_rows, int32_t out_dtype, int64_t n_rows,
                                       int64_t n_views, int32_t C, int32_t G, int32_t dtype, void* stream) {
  return rows_grad_rec16_impl(grad_out, perm, row_ptr, view_rec16, grad_rows, out_dtype, n_rows, n_views, C, G, dtype,
                              stream);
}

int dva_view_gather_rows_grad(const void* grad_out, const float* att, const float* gate,
                              const int32_t* view_point, const int32_t* perm, const int32_t* row_ptr,
                              const float* view_rec, int32_t rec_stride, float* grad_rows,
                              int64_t n_rows, int64_t n_views, int32_t C, int32_t G, int32_t dtype,
                              void* stream) {
  if (n_rows < 0 || n_views < 0 || C <= 0 || G <= 0 || G > C) return DVA_ERR_INVALID;
  if (n_views > 0x7fffffffLL) return DVA_ERR_UNSUPPORTED;
  if (n_rows == 0) return DVA_OK;
  if (!row_ptr || !grad_rows) return DVA_ERR_INVALID;
  if (view_rec && rec_stride < G + 1) return DVA_ERR_INVALID;
  if (n_views > 0 && (!grad_out || !perm)) return DVA_ERR_INVALID;
  if (n_views > 0 && !view_rec && (!att || !view_point)) return DVA_ERR_INVALID;
  int rc;
  if (dtype == DVA_F32)
    rc = rows_grad_impl<float>(grad_out, att, gate, view_point, perm, row_ptr, view_rec, rec_stride,
                               grad_rows, n_rows, C, G, (hipStream_t)stream);
  else if (dtype == DVA_BF16)
    rc = rows_grad_impl<bf16_t>(grad_out, att, gate, view_point, perm, row_ptr, view_rec, rec_stride,
                                grad_rows, n_rows, C, G, (hipStream_t)stream);
  else
    return DVA_ERR_INVALID;
  if (rc) return rc;
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_anchor_rows_sum(const void* grad_out, const int32_t* perm, const int32_t* row_ptr, const float* weights,
                        float* S, int64_t n_anchors, int64_t n_views, int32_t C, int32_t dtype, void* stream) {
  if (n_anchors < 0 || n_views < 0 || C <= 0) return DVA_ERR_INVALID;
  if (n_anchors == 0) return DVA_OK;
  if (!row_ptr || !S || (n_views > 0 && (!grad_out || !perm || !weights))) return DVA_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DVA_BF16) {
    const int lpr = C / 8;
    if ((C % 8) || !is_pow2(lpr) || lpr > 64 || ((uintptr_t)grad_out % 16) || ((uintptr_t)S % 16)) return DVA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((anchor_rows_sum_kernel<bf16_t, 0>), dim3(grid_cap((n_anchors + 3) / 4)), dim3(256), 0, s,
                       (const bf16_t*)grad_out, perm, row_ptr, (const float4*)weights, S, n_anchors, (int)C, lpr,
                       (const bf16_t*)nullptr, (const float*)nullptr, (const float*)nullptr, (const int4*)nullptr,
                       (const bf16_t*)nullptr);
  } else if (dtype == DVA_F32) {
    const int lpr = C / 4;
    if ((C % 4) || !is_pow2(lpr) || lpr > 64 || ((uintptr_t)grad_out % 16) || ((uintptr_t)S % 16)) return DVA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((anchor_rows_sum_kernel<float, 0>), dim3(grid_cap((n_anchors + 3) / 4)), dim3(256), 0, s,
                       (const float*)grad_out, perm, row_ptr, (const float4*)weights, S, n_anchors, (int)C, lpr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const int4*)nullptr,
                       (const float*)nullptr);
  } else {
    return DVA_ERR_INVALID;
  }
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_anchor_rows_sum_bn(const void* dy_a, const void* z_a, const float* bn_a, const float* sm_a, const int32_t* perm,
                           const int32_t* row_ptr, const float* weights, const int32_t* tap_rows, const void* Y,
                           float* S, int64_t n_anchors, int64_t n_views, int32_t C, void* stream) {
  if (n_anchors < 0 || n_views < 0 || C <= 0) return DVA_ERR_INVALID;
  if (n_anchors == 0) return DVA_OK;
  if (!row_ptr || !S || !bn_a || !sm_a || (n_views > 0 && (!dy_a || !perm || !weights))) return DVA_ERR_INVALID;
  if ((z_a == nullptr) == (Y == nullptr) || (Y && !tap_rows)) return DVA_ERR_INVALID;      // exactly one of the two forms
  const int lpr = C / 8;
  if ((C % 32) || C > 512 || !is_pow2(lpr) || lpr > 64 || ((uintptr_t)dy_a % 16) || ((uintptr_t)z_a % 16) || ((uintptr_t)Y % 16) ||
      ((uintptr_t)S % 16) || ((uintptr_t)tap_rows % 16))
    return DVA_ERR_UNSUPPORTED;
  const dim3 grid(grid_cap((n_anchors + 3) / 4)), block(256);
  if (Y)
    hipLaunchKernelGGL((anchor_rows_sum_kernel<bf16_t, 2>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)dy_a,
                       perm, row_ptr, (const float4*)weights, S, n_anchors, (int)C, lpr, (const bf16_t*)nullptr, bn_a,
                       sm_a, (const int4*)tap_rows, (const bf16_t*)Y);
  else
    hipLaunchKernelGGL((anchor_rows_sum_kernel<bf16_t, 1>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)dy_a,
                       perm, row_ptr, (const float4*)weights, S, n_anchors, (int)C, lpr, (const bf16_t*)z_a, bn_a,
                       sm_a, (const int4*)nullptr, (const bf16_t*)nullptr);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_gather_rows_sum(const void* grad_out, const int32_t* perm, const int32_t* row_ptr,
                        const float* weights, int32_t atom_shift, float* grad_rows, int64_t n_rows,
                        int64_t n_atoms, int32_t C, int32_t dtype, void* stream) {
  if (n_rows < 0 || n_atoms < 0 || C <= 0 || atom_shift < 0 || atom_shift > 8) return DVA_ERR_INVALID;
  if (n_atoms > 0x7fffffffLL) return DVA_ERR_UNSUPPORTED;
  if (n_rows == 0) return DVA_OK;
  if (!row_ptr || !grad_rows) return DVA_ERR_INVALID;
  if (n_atoms > 0 && (!grad_out || !perm)) return DVA_ERR_INVALID;
  int rc;
  if (dtype == DVA_F32)
    rc = rows_sum_impl<float>(grad_out, perm, row_ptr, weights, atom_shift, grad_rows, n_rows, C, (hipStream_t)stream);
  else if (dtype == DVA_BF16)
    rc = rows_sum_impl<bf16_t>(grad_out, perm, row_ptr, weights, atom_shift, grad_rows, n_rows, C, (hipStream_t)stream);
  else
    return DVA_ERR_INVALID;
  if (rc) return rc;
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // extern "C"
