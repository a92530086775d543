// Recompute chain, backward side (see chain_common.h for the geometry, chain_fwd.hip for the forward).
// Every chain pass re-evaluates the DeepSetFeat chain of a tile from x_map in registers and walks the gradient back as
// far as the BatchNorm-backward statistics allow; a pass ends where the next global sum (S1 = sum dy,
// S2 = sum dy * z_hat of a BatchNorm layer) is needed:
//   dva_chain_attn_bwd     attention + gate backward from the scores the forward left: score gradients dc [V, 4], view
//                          records (no chain evaluation in this kernel)
//   dva_chain_score_stats  score layer: dWs, dbs, S of layer 6
//   dva_chain_bwd_layer    stage 6: dW6, S of layer 5
//                          stage 5: dW5 (per-view half), du [N, 32] (gradient of the per-point half), S of layer 2 (view part)
//                          stage 2: set-pooling gradient routed to the arg views, dW2, P = sum dy1 [x_hi | x_lo | 1]^T
//   dva_chain_stats1       S of layer 1 from P (z1 is linear in x_map);  dva_chain_dw1: dW1 from P and the moments
//   dva_chain_route_stats  S of layer 2, per-point part (the routed set-pooling gradient)
// BatchNorm backward per layer:  dz = G (dy - S1/M - z_hat S2/M),  dy = leaky'(.) da with the sign of the pre-activation
// the forward's activation saw (the folded product t of layers 1, 2, 6; G z + B of layer 5),  G = gamma * invstd.
// Reference maths: autograd of modules/multimodal/pooling.py:263-315, :658-669.
#include "chain_common.h"

namespace dva {
namespace chain {

struct ChainKeep {
  f32x16 z5, z6, t6;
  bf16x8 a2[2], a5[2], a6[2];
};
// The forward chain as the forward passes evaluate it: layers 1 and 2 with BatchNorm folded into the operand (their
// LDS blocks hold the folded operands), layer 5 plain (the per-point row enters before its BatchNorm), layer 6 twice:
// the raw output z6 (BatchNorm backward, statistics) and the folded product t6 = 0.6 y6 (block W6F) whose sign the
// forward's activation saw and whose activation a6 feeds the score layer.  tabs[0..3] = layers 1, 2, 5, 6.
// L6: 0 = z6 only, 1 = + t6, 2 = + t6 and a6.
// A2IN (round 6, the stored-a2 hybrid): k.a2 already holds the layer-2 activation row the forward stored
// (stats_mid_kernel<5, 1>: the very operand computed below): layers 1 and 2 are skipped, x is not read.
// MASK = false: activations of lanes without a view are not zeroed (chain_common.h act_pack) -- for passes whose every sum
// over views is guarded on the gradient side (score gradients of such a lane are zero, gradient rows go through pack16).
template <int W6F, int L6, bool A2IN = false, bool MASK = true>
__device__ __forceinline__ void chain_forward(const uint4* s_ops, int lane, const float (*tabs)[TAB_FLOATS], int h,
                                              uint32_t keep, const float4& x, const f32x16& uacc, ChainKeep& k) {
  const f32x16 zero = {0};
  if constexpr (!A2IN) {
    bf16x8 a1[2];
    asm volatile("" ::: "memory");
    const f32x16 t1 = CH_MFMA(lds_op(s_ops, OP_W1, lane), pack_x(x), bias_acc(tabs[0], T_B6, h));
    act_fold<MASK>(t1, keep, a1);
    const f32x16 t2 = mm32_lds(s_ops, OP_W2, lane, a1, bias_acc(tabs[1], T_B6, h));
    act_fold<MASK>(t2, keep, k.a2);
  }
  k.z5 = mm32_lds(s_ops, OP_W5, lane, k.a2, uacc);
  act_pack<MASK>(k.z5, tabs[2], h, keep, k.a5);
  k.z6 = mm32_lds(s_ops, OP_W6, lane, k.a5, zero);
  if (L6 >= 1) k.t6 = mm32_lds(s_ops, W6F, lane, k.a5, bias_acc(tabs[3], T_B6, h));
  if (L6 >= 2) act_fold<MASK>(k.t6, keep, k.a6);
}
__device__ __forceinline__ f32x16 load_u(__amdgpu_buffer_rsrc_t U, bool ok, int vpj, int h) {
  f32x16 uacc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = as_f4(ld128(U, ok ? (uint32_t)vpj * 128u + (8u * q + 4u * h) * 4u : OOB));
    uacc[4 * q] = v.x; uacc[4 * q + 1] = v.y; uacc[4 * q + 2] = v.z; uacc[4 * q + 3] = v.w;
  }
  return uacc;
}
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// sum of the 8 bf16 products of two 16-byte chunks (v_dot2c_f32_bf16, fp32 accumulation)
__device__ __forceinline__ float dot8(const u32x4& a, const u32x4& b) {
  const uint32_t aa[4] = {a.x, a.y, a.z, a.w}, bb[4] = {b.x, b.y, b.z, b.w};
  float d = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, aa[i]), __builtin_bit_cast(bf16x2_t, bb[i]), d, false);
  return d;
}
// da6 = Ws^T dc: the score gradients of the view enter as hi | lo in the k-slots of the h = 0 lane
template <int WST>
__device__ __forceinline__ f32x16 score_bwd(const uint4* s_ops, int lane, const float (&dc)[4], int h) {
  const uint32_t h0 = pack_bf16x2(dc[0], dc[1]), h1 = pack_bf16x2(dc[2], dc[3]);
  const float r0 = dc[0] - __uint_as_float(h0 << 16), r1 = dc[1] - __uint_as_float(h0 & 0xffff0000u);
  const float r2 = dc[2] - __uint_as_float(h1 << 16), r3 = dc[3] - __uint_as_float(h1 & 0xffff0000u);
  const uint32_t keep = h == 0 ? 0xffffffffu : 0u;
  const u32x4 v = {h0 & keep, h1 & keep, pack_bf16x2(r0, r1) & keep, pack_bf16x2(r2, r3) & keep};
  const f32x16 zero = {0};
  asm volatile("" ::: "memory");
  return CH_MFMA(lds_op(s_ops, WST, lane), __builtin_bit_cast(bf16x8, v), zero);
}
// ------------------------------------------------------------------------------------------------
// attention backward (round 3: no chain in this kernel).  The fused forward kernel leaves the scores c [V, 4] (16 bytes
// per view) in training mode; the softmax / gate backward needs only them, the value rows and the two per-point rows
// grad_out / out: per view it writes the score gradients dc [V, 4] and the 16-byte record of the rows-gradient pass.
// What the old kernel also did -- re-evaluating the DeepSetFeat chain to get the BatchNorm-6 statistics and the
// score-layer weight gradient -- is the separate pass score_stats_kernel below: 253 VGPRs / 2 wavefronts per SIMD became
// two kernels that each fit a register budget with twice the occupancy.
// ------------------------------------------------------------------------------------------------
// T = bf16_t (the chain path: 8 channels per 16-byte lane, 16-byte records with bf16 weights) or float (the fp32 path of
// ops.view_gather_attention: 4 channels per lane, 32-byte records {point | fp32 gate * attention per group | pad} as
// dva_view_gather_rows_grad reads them).
template <typename T>
__device__ __forceinline__ float dotv(const u32x4& a, const u32x4& b);
template <>
__device__ __forceinline__ float dotv<bf16_t>(const u32x4& a, const u32x4& b) { return dot8(a, b); }
template <>
__device__ __forceinline__ float dotv<float>(const u32x4& a, const u32x4& b) {
  return __builtin_fmaf(__uint_as_float(a.w), __uint_as_float(b.w),
                        __builtin_fmaf(__uint_as_float(a.z), __uint_as_float(b.z),
                                       __builtin_fmaf(__uint_as_float(a.y), __uint_as_float(b.y),
                                                      __uint_as_float(a.x) * __uint_as_float(b.x))));
}

template <typename T, int LPR, int G>
__global__ __launch_bounds__(256, (sizeof(T) == 2 && LPR <= 8) ? 4 : 3) void attn_bwd_kernel(
    const float* __restrict__ compat, const int32_t* __restrict__ vp, const int2* __restrict__ tiles,
    const int32_t* __restrict__ n_tiles_dev, const T* __restrict__ rows, const int32_t* __restrict__ row_idx,
    const int64_t* __restrict__ ptr, const float* __restrict__ gw, const float* __restrict__ gb,
    const T* __restrict__ gout, const T* __restrict__ out, float* __restrict__ dc_out,
    uint32_t* __restrict__ rec, float* __restrict__ gwb, int scaling, float eps, int64_t V, int64_t N, int64_t R,
    const int32_t* __restrict__ rec_pos = nullptr) {
  // rec_pos (nullable, 16-byte records only): the record of view v goes to slot rec_pos[v] (= its position in the row
  // plan: dva_plan_inverse) instead of slot v -- the rows-gradient pass then streams the records (A/B of round 4)
  constexpr int VEC = 16 / (int)sizeof(T), C = LPR * VEC, ROWS = 64 / LPR, KV = 32 / ROWS;
  constexpr uint32_t RB = C * sizeof(T);       // bytes of a value / gradient row
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int KB = KV < 4 ? KV : 4, NB = KV / KB;
  constexpr int NE = G == 1 ? 1 : 2;
  constexpr int LPG = LPR / G;                 // lanes per channel group inside a row
  __shared__ __attribute__((aligned(16))) float s_q[4][4 * 32];
  __shared__ __attribute__((aligned(16))) int s_pid[4][32], s_ri[4][32];
  __shared__ __attribute__((aligned(16))) float s_E[4][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  const __amdgpu_buffer_rsrc_t CP = make_rsrc(compat, (uint64_t)V * 16), P = make_rsrc(vp, (uint64_t)V * 4),
                               RI = make_rsrc(row_idx, (uint64_t)V * 4),
                               RW = make_rsrc(rows, (uint64_t)R * RB), GO = make_rsrc(gout, (uint64_t)N * RB),
                               OU = make_rsrc(out, (uint64_t)N * RB), DC = make_rsrc(dc_out, (uint64_t)V * 16),
                               RC = make_rsrc(rec, (uint64_t)V * (F32 ? 32 : 16));
  // softmax side: lane (j, h) owns the groups gl[e] of view j (G = 4: 2 h, 2 h + 1; G <= 2: the h = 0 lanes)
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
  const int slot = lane / LPR, q = lane % LPR, sv0 = slot * KV;
  const int tg = q / LPG;
  float* q_t = s_q[wv];
  int* pid_t = s_pid[wv];
  int* ri_t = s_ri[wv];
  float dwa[NE], dba[NE];
  // state of a long point across its fragments
  float glob_m[NE], glob_s[NE], glob_E[NE];
  bool seen[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) { dwa[e] = dba[e] = 0.f; glob_m[e] = glob_s[e] = glob_E[e] = 0.f; seen[e] = false; }

  const int n_tiles = n_tiles_dev[0];
  int ta, tb;
  wave_tile_range(tiles, n_tiles, ta, tb);
  struct Pre {
    TileInfo ti;
    int t;
    u32x2 cc;
    int vpj, rij;
  };
  run_tiles<Pre>(tiles, ta, tb, [&](const TileInfo& ti, int t) {
    Pre p;
    p.ti = ti;
    p.t = t;
    const bool ok = j < p.ti.nv;
    p.cc = ld64(CP, ok ? (uint32_t)(p.ti.v0 + j) * 16u + coff : OOB);
    p.vpj = (int)ld32(P, ok ? (uint32_t)(p.ti.v0 + j) * 4u : OOB);
    p.rij = (int)ld32(RI, ok ? (uint32_t)(p.ti.v0 + j) * 4u : OOB);
    return p;
  }, [&](const Pre& p) {
    const int nv = p.ti.nv, frag = p.ti.frag;
    const bool ok = j < nv;
    if (h == 0) {
      ri_t[j] = p.rij;
      pid_t[j] = p.vpj;
    }
    wave_sync();
    u32x4 xr[2][KB], go[2][KB];
    auto issue_rows = [&](int b) {
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) {
        const int vt = sv0 + b * KB + kk;
        xr[b & 1][kk] = ld128(RW, (uint32_t)ri_t[vt] * RB + (uint32_t)q * 16u);
        go[b & 1][kk] = ld128(GO, vt < nv ? (uint32_t)pid_t[vt] * RB + (uint32_t)q * 16u : OOB);
      }
    };
    issue_rows(0);
    float c[NE];
    c[0] = __uint_as_float(p.cc.x);
    if (NE > 1) c[NE - 1] = __uint_as_float(p.cc.y);
    const int vp0 = __builtin_amdgcn_readfirstlane(p.vpj);
    const bool single = frag != 0 || __ballot(ok && p.vpj != vp0) == 0;     // one point in the tile
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
    const float isl = isn * 1.44269504f;        // exp(x isn) = exp2(x isl)
    auto red_max = [&](float v) { return single ? half_max(v) : seg_total(seg_scan_max(v, sg, lane), sg, h); };
    auto red_sum = [&](float v) { return single ? half_sum(v) : seg_total(seg_scan_sum(v, sg, lane), sg, h); };
    if (frag == 1) {
      // ---- long point: softmax statistics over ALL its fragments first (their scores are 16 bytes per view), and
      //      E = sum_v a q from the saved forward output: E_g = sum_{ch in g} gout out / gate
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
      const uint32_t pid0 = (uint32_t)__builtin_amdgcn_readfirstlane(p.vpj);
      const u32x4 go0 = ld128(GO, pid0 * RB + (uint32_t)q * 16u);
      const u32x4 ou0 = ld128(OU, pid0 * RB + (uint32_t)q * 16u);
      float d = dotv<T>(go0, ou0);
#pragma unroll
      for (int off = 1; off < LPG; off <<= 1) d += __shfl_xor(d, off);
      if (slot == 0 && (q % LPG) == 0) s_E[wv][tg] = d;
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
      float s;
      if (frag == 0) {
        m[e] = red_max(ok ? c[e] : -INFINITY);
        const float ev = ok ? __builtin_amdgcn_exp2f((c[e] - m[e]) * isl) : 0.f;
        s = red_sum(ev);
        a[e] = ev * __builtin_amdgcn_rcpf(s + eps);
      } else {
        m[e] = glob_m[e];
        a[e] = ok ? __builtin_amdgcn_exp2f((c[e] - m[e]) * isl) * __builtin_amdgcn_rcpf(glob_s[e] + eps) : 0.f;
      }
      pre[e] = __builtin_fmaf(gwl[e], m[e], gbl[e]);
      gt[e] = gw ? tanh_pos(fmaxf(pre[e], 0.f)) : 1.f;
    }
    // ---- team side: q[v][g] = sum_{ch in g} gout[p][ch] rows[v][ch]
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b + 1 < NB) issue_rows(b + 1);
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) {
        const int vt = sv0 + b * KB + kk;
        float d = dotv<T>(go[b & 1][kk], xr[b & 1][kk]);
#pragma unroll
        for (int off = 1; off < LPG; off <<= 1) d += __shfl_xor(d, off);
        if ((q % LPG) == 0) q_t[tg * 32 + vt] = d;
      }
    }
    wave_sync();
    // ---- softmax + gate backward
    float dcv[NE], gav[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const float qv = q_t[gl[e] * 32 + j];
      float E;
      if (frag == 0) E = red_sum(a[e] * qv);
      else E = glob_E[e];
      const float dpre = (gw && pre[e] > 0.f) ? E * (1.f - gt[e] * gt[e]) : 0.f;
      // first view of the point that attains the maximum (ties -> lowest index, like segment_csr 'max')
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
      // d gate_w, d gate_b: once per point, at the last view of the point
      const bool last_view = ok && j == sg.se && (frag == 0 || frag == 3);
      if (last_view) {
        dwa[e] += dpre * m[e];
        dba[e] += dpre;
      }
    }
    // ---- per-view outputs: dc [V, 4] and the 16-byte record {point | gate * attention per group as bf16}
    float dc4[4] = {0.f, 0.f, 0.f, 0.f}, ga4[4] = {0.f, 0.f, 0.f, 0.f};
    if (G == 4) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        uint32_t x0 = __float_as_uint(dcv[e]), x1 = x0;
        swap_halves(x0, x1);         // x1 in the h = 0 lanes = value of the h = 1 lane
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
    if constexpr (F32) {   // 32-byte record: point | gate * attention per group (fp32) | pad
      const u32x4 r0 = {(uint32_t)p.vpj, __float_as_uint(ga4[0]), __float_as_uint(ga4[1]), __float_as_uint(ga4[2])};
      const u32x4 r1 = {__float_as_uint(ga4[3]), 0u, 0u, 0u};
      st128(RC, wr ? vg * 32u : OOB, r0);
      st128(RC, wr ? vg * 32u + 16u : OOB, r1);
    } else {   // 16-byte record: the rows gradient is rounded to bf16 anyway, its weights travel as bf16
      // word 3 = the view's row key: the split plan (plan_split.hip) sorts the records themselves by it
      const u32x4 r = {(uint32_t)p.vpj, pack_bf16x2(ga4[0], ga4[1]), pack_bf16x2(ga4[2], ga4[3]), (uint32_t)p.rij};
      uint32_t slot = vg;
      if (rec_pos) slot = wr ? (uint32_t)rec_pos[vg] : 0u;       // (uniform branch: a kernel argument)
      st128(RC, wr ? slot * 16u : OOB, r);
    }
    wave_sync();
  });
  if (gw) {      // (uniform: a kernel argument)
    // one atomic per block and address: the wavefronts' sums meet in LDS first (every block adds to the same 2 G
    // addresses; per-wavefront atomics queue up 4 x grid deep on each of them at the end of the kernel)
    __syncthreads();
    float* s_wb = &s_q[0][0];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const float dw = half_sum(dwa[e]), db = half_sum(dba[e]);
      if (j == 0 && s_active && ((G == 4 ? 2 * h : 0) + e) < G) {
        s_wb[wv * 8 + gl[e]] = dw;
        s_wb[wv * 8 + 4 + gl[e]] = db;
      }
    }
    __syncthreads();
    if (threadIdx.x < 2 * G) {
      const int which = threadIdx.x / G, g = threadIdx.x % G;
      float v = 0.f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) v += s_wb[w * 8 + 4 * which + g];
      atomicAdd(&gwb[which * G + g], v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// score layer backward + statistics of the BatchNorm-6 backward: one chain evaluation per view (x_map 32 B + the
// view -> point index 4 B + the score gradients 16 B in; nothing view-sized out).
//   da6 = Ws^T dc;  dy6 = leaky'(t6) da6  (t6 = the folded product the forward's activation saw);
//   S6 = sum dy6 | sum dy6 z6;   dWs^T[k][g] = sum_v a6[v][k] dc[v][g];   dbs = sum_v dc
// LDS operand table: chain positions 0..6 (W1', W2', W5, W6), 7..8 = W6' (folded), 9 = Ws^T.
// ------------------------------------------------------------------------------------------------
// KEYS (the key layer of QKVBimodalCSRPool): the gradient of the last layer is a [32]-row dK per view, built in registers
// from the compatibility gradient dc [V][4] and the point's query row (dkeys_operand; a stored bf16 [V][32] row until
// round 4): da6 = W_k^T dK through the full transposed operand OP_WKT, dW_k [32][32] = dK^T a6 from two natural tiles,
// db_k = column sums of the dK tile.  G = the number of query-key groups (1, 2, 4).
// KEYS: dK' of (view, half h) as the packed B operand, from the compatibility gradient dc [V][4] and the point's query row
// Q' fp32 [N][32] (position order): dK'[16 h + r] = (scale dc[g(r)]) Q'[p][16 h + r], g(r) = (r >> 2) G / 4 -- the product
// qkv.hip's dkeys kernel stored as a bf16 [V][32] row until round 4 (same roundings: scale first, one bf16 rounding).
__device__ __forceinline__ void dkeys_operand(const float4& dcv, __amdgpu_buffer_rsrc_t QP, bool ok, int vpj, int h, int G,
                                              float scale, bf16x8 (&dkp)[2]) {
  const float d4[4] = {dcv.x * scale, dcv.y * scale, dcv.z * scale, dcv.w * scale};
  float dq[4];
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) dq[qd] = G == 4 ? d4[qd] : (G == 2 ? d4[qd >> 1] : d4[0]);
  float d[16];
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const float4 q = as_f4(ld128(QP, ok ? (uint32_t)vpj * 128u + 64u * h + 16u * qq : OOB));
    d[4 * qq] = dq[qq] * q.x;
    d[4 * qq + 1] = dq[qq] * q.y;
    d[4 * qq + 2] = dq[qq] * q.z;
    d[4 * qq + 3] = dq[qq] * q.w;
  }
  dkp[0] = pack8(&d[0]);
  dkp[1] = pack8(&d[8]);
}

template <bool KEYS, bool A2IN = false>
__global__ __launch_bounds__(256, 3) void score_stats_kernel(
    const float* __restrict__ x_map, const int32_t* __restrict__ vp, const float* __restrict__ u,
    const int2* __restrict__ tiles, const int32_t* __restrict__ n_tiles_dev, const uint4* __restrict__ ops,
    const float* __restrict__ bn1, const float* __restrict__ bn2, const float* __restrict__ bn5,
    const float* __restrict__ bn6, const float* __restrict__ dc, double* __restrict__ stats6,
    float* __restrict__ dWs, float* __restrict__ dbs, int G, int64_t V, int64_t N, const float* __restrict__ qp,
    float qscale, const bf16_t* __restrict__ a2buf = nullptr) {
  constexpr int L_W6F = 7, L_WST = 9, NOPS = KEYS ? 11 : 10;
  __shared__ __attribute__((aligned(16))) float s_tab[4][TAB_FLOATS];
  __shared__ __attribute__((aligned(16))) uint4 s_ops[NOPS * 64];
  // transposed a6 tile + a 4-row score-gradient tile (+ one shared zero row), as in the layer passes
  __shared__ __attribute__((aligned(16))) bf16_t s_tc[4][32 * TSB], s_td[4][(KEYS ? 32 : 5) * TSB];
  float* s_red = reinterpret_cast<float*>(&s_tc[0][0]);        // epilogue only (D x D floats <= the tile buffers)
  static_assert(sizeof(bf16_t) * 4 * 32 * TSB >= sizeof(float) * D * D, "epilogue buffer");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < 4 * (KEYS ? 32 : 5) * TSB; i += blockDim.x) (&s_td[0][0])[i] = 0;
  for (int i = threadIdx.x; i < 4 * 64; i += blockDim.x) s_ops[OP_W5 * 64 + i] = ops[OP_W5 * 64 + i];   // W5, W6
  for (int i = threadIdx.x; i < (KEYS ? 128 : 64); i += blockDim.x)
    s_ops[L_WST * 64 + i] = ops[(KEYS ? OP_WKT : OP_WST) * 64 + i];
  if (!A2IN) {
    fold_ops(s_ops, OP_W1, ops, OP_W1, 1, bn1);
    fold_ops(s_ops, OP_W2, ops, OP_W2, 2, bn2);
  }
  fold_ops(s_ops, L_W6F, ops, OP_W6, 2, bn6);
  if (!A2IN) {
    stage_tab(s_tab[0], bn1, nullptr);
    stage_tab(s_tab[1], bn2, nullptr);
  }
  stage_tab(s_tab[2], bn5, nullptr);
  stage_tab(s_tab[3], bn6, nullptr);
  __syncthreads();
  const __amdgpu_buffer_rsrc_t X = make_rsrc(x_map, A2IN ? 0 : (uint64_t)V * 32), P = make_rsrc(vp, (uint64_t)V * 4),
                               U = make_rsrc(u, (uint64_t)N * 128), DC = make_rsrc(dc, (uint64_t)V * 16),
                               QP = make_rsrc(qp, KEYS ? (uint64_t)N * 128 : 0),
                               A2 = make_rsrc(a2buf, A2IN ? (uint64_t)V * 64 : 0);
  bf16_t* tc = s_tc[wv];
  bf16_t* td = s_td[wv];
  f32x16 accS = {0};
  float dbsum[4] = {0.f, 0.f, 0.f, 0.f};      // KEYS: [0] = this lane's share of the column sum of the dK tile
  float st[2][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) st[0][r] = st[1][r] = 0.f;
  const int n_tiles = n_tiles_dev[0];
  int ta, tb;
  wave_tile_range(tiles, n_tiles, ta, tb);
  struct Pre {
    TileInfo ti;
    float4 x, dc;
    u32x4 alo, ahi;
    int vpj;
  };
  run_tiles<Pre>(tiles, ta, tb, [&](const TileInfo& ti, int t) {
    Pre p;
    p.ti = ti;
    const bool ok = j < p.ti.nv;
    const uint32_t view = (uint32_t)(p.ti.v0 + j);
    if constexpr (A2IN) {
      p.alo = ld128(A2, ok ? view * 64u + 32u * h : OOB);
      p.ahi = ld128(A2, ok ? view * 64u + 32u * h + 16u : OOB);
    } else {
      p.x = as_f4(ld128(X, ok ? view * 32u + 16u * h : OOB));
    }
    p.vpj = (int)ld32(P, ok ? view * 4u : OOB);
    p.dc = as_f4(ld128(DC, ok && (KEYS || h == 0) ? view * 16u : OOB));      // KEYS: both halves need the four groups
    return p;
  }, [&](const Pre& p) {
    const bool ok = j < p.ti.nv;
    const uint32_t keep = ok ? 0xffffffffu : 0u;
    const f32x16 uacc = load_u(U, ok, p.vpj, h);
    ChainKeep k;
    if constexpr (A2IN) {
      k.a2[0] = __builtin_bit_cast(bf16x8, p.alo);
      k.a2[1] = __builtin_bit_cast(bf16x8, p.ahi);
    }
    // unmasked activations: the score gradients (KEYS: the dK operand) of a lane without a view are zero, so dy6 is, and
    // the a6 / z6 garbage of such a lane meets a zero in every product and sum below
    chain_forward<L_W6F, 2, A2IN, false>(s_ops, lane, s_tab, h, keep, p.x, uacc, k);
    if constexpr (KEYS) {
      // dK of the view as the packed B operand (zeros for lanes without a view)
      bf16x8 dkp[2];
      dkeys_operand(p.dc, QP, ok, p.vpj, h, G, qscale, dkp);
      tileN_put_packed(tc, j, h, k.a6);
      tileN_put_packed(td, j, h, dkp);
      const f32x16 zero = {0};
      f32x16 dy6 = mm32_lds(s_ops, L_WST, lane, dkp, zero);       // da6 = W_k^T dK
      dleaky_mul(k.t6, dy6);
      bn_bwd_stats(k.z6, dy6, st);
      wave_sync();
      accS = wgradN(td, tc, lane, accS);          // dW_k[row][k] = sum_v dK[v][row] a6[v][k]
      col_sum1(td, lane, dbsum[0]);
      wave_sync();
      return;
    }
    const float dc4[4] = {p.dc.x, p.dc.y, p.dc.z, p.dc.w};      // zeros in the lanes without a view and for h = 1
    tileN_put_packed(tc, j, h, k.a6);
    if (h == 0) {
      const uint32_t d01 = pack_bf16x2(dc4[0], dc4[1]), d23 = pack_bf16x2(dc4[2], dc4[3]);
      td[0 * TSB + j] = (bf16_t)(d01 & 0xffffu);
      td[1 * TSB + j] = (bf16_t)(d01 >> 16);
      td[2 * TSB + j] = (bf16_t)(d23 & 0xffffu);
      td[3 * TSB + j] = (bf16_t)(d23 >> 16);
#pragma unroll
      for (int g = 0; g < 4; ++g) dbsum[g] += dc4[g];
    }
    f32x16 dy6 = score_bwd<L_WST>(s_ops, lane, dc4, h);
    dleaky_mul(k.t6, dy6);
    bn_bwd_stats(k.z6, dy6, st);
    wave_sync();
    accS = wgradN_T(tc, td, lane, j, 4, h, accS);
    wave_sync();
  });
  if constexpr (KEYS) {
    flush_matrix_nat(accS, dWs, D, D, false, s_red, true);
    flush_stats<2>(st, stats6, s_red);
    __syncthreads();     // db_k: the two half-waves hold the two halves of a column sum; column n = key channel cperm(n)
    {
      uint32_t a = __float_as_uint(dbsum[0]), b = a;
      swap_halves(a, b);
      const float tot = dbsum[0] + __uint_as_float((threadIdx.x & 32) ? a : b);
      if (h == 0) s_red[wv * 32 + j] = tot;
    }
    __syncthreads();
    if (threadIdx.x < 32)
      atomicAdd(&dbs[cperm(threadIdx.x)], (s_red[threadIdx.x] + s_red[32 + threadIdx.x]) +
                                             (s_red[64 + threadIdx.x] + s_red[96 + threadIdx.x]));
    return;
  }
  flush_matrix_nat(accS, dWs, D, G, true, s_red, false);
  flush_stats<2>(st, stats6, s_red);
  __syncthreads();       // dbs: one atomic per block and group (see the attention backward)
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

// column sums of natural tiles this wavefront has just written: lane (n, hh) receives 16 of the 32 views of image column
// n through the transpose read; two registers per statistic pair instead of 32 per-lane accumulators (merged stage 5)
__device__ __forceinline__ void col_sums_xy(const bf16_t* tx, const bf16_t* ty, int lane, float& sx, float& sxy) {
  // packed bf16 pairs straight into v_dot2c_f32_bf16 (fp32 accumulation): four dot products per eight values for
  // sum x y, four against a pair of ones for sum x -- no unpacking
  const uint32_t ones = 0x3f803f80u;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const u32x4 x = __builtin_bit_cast(u32x4, tileN_get(tx, lane, m)), y = __builtin_bit_cast(u32x4, tileN_get(ty, lane, m));
    const uint32_t xx[4] = {x.x, x.y, x.z, x.w}, yy[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sx = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, xx[i]), __builtin_bit_cast(bf16x2_t, ones), sx, false);
      sxy = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, xx[i]), __builtin_bit_cast(bf16x2_t, yy[i]), sxy, false);
    }
  }
}
// the two halves of the column sums (lanes n and n + 32) -> fp64 atomics in a fixed per-block order; column n of a
// natural tile = channel cperm(n).  s_red: 4 x 2 x 32 floats.
__device__ __forceinline__ void flush_col_stats(float v0, float v1, double* __restrict__ out, float* s_red) {
  const int wv = threadIdx.x >> 6, n = threadIdx.x & 31;
  {
    uint32_t a = __float_as_uint(v0), b = a;
    swap_halves(a, b);
    v0 += __uint_as_float((threadIdx.x & 32) ? a : b);
    uint32_t c = __float_as_uint(v1), d = c;
    swap_halves(c, d);
    v1 += __uint_as_float((threadIdx.x & 32) ? c : d);
  }
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
    atomicAdd(&out[which * D + cperm(n)], acc);
  }
}

// ------------------------------------------------------------------------------------------------
// Merged backward (round 5): the score pass ALSO produces what stage 6 existed for -- the statistics S5 of the
// BatchNorm-5 backward -- so that stage 6 disappears and stage 5 starts from the score gradients.
//   dz6 = G6 dy6 - K1 - K2 z6 is linear in the two constants this very pass is still summing (K1, K2 per channel from
//   S6), so S5 = sum_v dy5 | sum_v dy5 z5 with dy5 = m5 (W6^T dz6), m5 = leaky'(y5) in {1, 0.2}, splits into
//     S5a[c] = e1[c] - n5[c] sum_j W6[j][c] K1[j] - sum_j W6[j][c] K2[j] P2[c][j]
//     S5b[c] = e2[c] - q5[c] sum_j W6[j][c] K1[j] - sum_j W6[j][c] K2[j] Q2[c][j]
//   with the sums this kernel accumulates per view from quantities it has in registers anyway:
//     e = W6^T (G6 dy6)  (one more product),  e1 = sum m5 e,  e2 = sum (m5 z5) e,  n5 = sum m5,  q5 = sum m5 z5  (vectors),
//     P2[c][j] = sum_v m5[c] z6[j],  Q2[c][j] = sum_v (m5 z5)[c] z6[j]   (two 32 x 32 products over natural LDS tiles).
//   l6_consts_kernel finishes S5 once S6 is complete.  Stage 5 (layer_bwd_kernel<5, ., false, true>) then evaluates
//   dy6 -> dz6 -> da5 -> dy5 itself (three more products on idle matrix cores; it reads 16 bytes of score gradients per
//   view instead of the 64-byte dy5 row) and takes dW6 along.  Gone: one chain evaluation, the [V, 32] bf16 dy5 tensor
//   (2.1 GB written + read), one launch.  Roundings: the operands of the new products are bf16 like every other
//   weight-gradient operand (sums over all views: unbiased, they average out); S5 is no longer the sum of the very dy5
//   values stage 5 uses but of the same expression before its bf16 operand rounding -- a difference of 2^-9 / sqrt(V)
//   relative in the subtracted means.
// acc5 fp32 [2][32][32] = P2 | Q2 (channel indices, caller-zeroed), vec5 fp64 [4][32] = e1 | e2 | n5 | q5 (caller-zeroed).
// LDS operand table: chain positions 0..6 (W1', W2', W5, W6), 7..8 = W6' (folded), 9 = Ws^T, 10..11 = W6^T.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void score_l6_kernel(
    const float* __restrict__ x_map, const int32_t* __restrict__ vp, const float* __restrict__ u,
    const int2* __restrict__ tiles, const int32_t* __restrict__ n_tiles_dev, const uint4* __restrict__ ops,
    const float* __restrict__ bn1, const float* __restrict__ bn2, const float* __restrict__ bn5,
    const float* __restrict__ bn6, const float* __restrict__ dc, double* __restrict__ stats6,
    float* __restrict__ dWs, float* __restrict__ dbs, float* __restrict__ acc5, double* __restrict__ vec5, int G,
    int64_t V, int64_t N) {
  constexpr int L_W6F = 7, L_WST = 9, L_W6T = 10, NOPS = 12;
  __shared__ __attribute__((aligned(16))) float s_tab[4][TAB_FLOATS];
  __shared__ __attribute__((aligned(16))) uint4 s_ops[NOPS * 64];
  // THREE natural tile buffers per wavefront, used in turn (the LDS budget of three blocks per CU), and the 4-row
  // transposed score-gradient tile (+ one zero row).  Every vector statistic is a column sum of such tiles (two registers
  // per pair, col_sums_xy) instead of 16 per-lane accumulators: the register budget of three wavefronts per SIMD.
  __shared__ __attribute__((aligned(16))) bf16_t s_t0[4][32 * TSB], s_t1[4][32 * TSB], s_t2[4][32 * TSB], s_td[4][5 * TSB];
  float* s_red = reinterpret_cast<float*>(&s_t0[0][0]);        // epilogue only
  static_assert(sizeof(bf16_t) * 4 * 32 * TSB >= sizeof(float) * D * D, "epilogue buffer");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < 4 * 5 * TSB; i += blockDim.x) (&s_td[0][0])[i] = 0;
  for (int i = threadIdx.x; i < 4 * 64; i += blockDim.x) s_ops[OP_W5 * 64 + i] = ops[OP_W5 * 64 + i];   // W5, W6
  for (int i = threadIdx.x; i < 64; i += blockDim.x) s_ops[L_WST * 64 + i] = ops[OP_WST * 64 + i];
  for (int i = threadIdx.x; i < 2 * 64; i += blockDim.x) s_ops[L_W6T * 64 + i] = ops[OP_W6T * 64 + i];
  fold_ops(s_ops, OP_W1, ops, OP_W1, 1, bn1);
  fold_ops(s_ops, OP_W2, ops, OP_W2, 2, bn2);
  fold_ops(s_ops, L_W6F, ops, OP_W6, 2, bn6);
  stage_tab(s_tab[0], bn1, nullptr);
  stage_tab(s_tab[1], bn2, nullptr);
  stage_tab(s_tab[2], bn5, nullptr);
  stage_tab(s_tab[3], bn6, nullptr);
  __syncthreads();
  const __amdgpu_buffer_rsrc_t X = make_rsrc(x_map, (uint64_t)V * 32), P = make_rsrc(vp, (uint64_t)V * 4),
                               U = make_rsrc(u, (uint64_t)N * 128), DC = make_rsrc(dc, (uint64_t)V * 16);
  bf16_t* b0 = s_t0[wv];
  bf16_t* b1 = s_t1[wv];
  bf16_t* b2 = s_t2[wv];
  bf16_t* td = s_td[wv];
  f32x16 accS = {0}, accP = {0}, accQ = {0};
  float dbsum[4] = {0.f, 0.f, 0.f, 0.f};
  float s6a = 0.f, s6b = 0.f, n5 = 0.f, e1 = 0.f, q5 = 0.f, e2 = 0.f;       // column sums (lane = image column, half)
  const int n_tiles = n_tiles_dev[0];
  int ta, tb;
  wave_tile_range(tiles, n_tiles, ta, tb);
  struct Pre {
    TileInfo ti;
    float4 x, dc;
    int vpj;
  };
  run_tiles_single<Pre>(tiles, ta, tb, [&](const TileInfo& ti, int t) {
    Pre p;
    p.ti = ti;
    const bool ok = j < p.ti.nv;
    const uint32_t view = (uint32_t)(p.ti.v0 + j);
    p.x = as_f4(ld128(X, ok ? view * 32u + 16u * h : OOB));
    p.vpj = (int)ld32(P, ok ? view * 4u : OOB);
    p.dc = as_f4(ld128(DC, ok && h == 0 ? view * 16u : OOB));
    return p;
  }, [&](const Pre& p) {
    const bool ok = j < p.ti.nv;
    const uint32_t keep = ok ? 0xffffffffu : 0u;
    const f32x16 uacc = load_u(U, ok, p.vpj, h);
    ChainKeep k;
    chain_forward<L_W6F, 2>(s_ops, lane, s_tab, h, keep, p.x, uacc, k);
    const float dc4[4] = {p.dc.x, p.dc.y, p.dc.z, p.dc.w};      // zeros in the lanes without a view and for h = 1
    // ---- phase 1: b0 = a6, td = dc (score-weight gradient); b1 = z6, b2 = dy6 (S6 as column sums of the rounded values)
    tileN_put_packed(b0, j, h, k.a6);
    if (h == 0) {
      const uint32_t d01 = pack_bf16x2(dc4[0], dc4[1]), d23 = pack_bf16x2(dc4[2], dc4[3]);
      td[0 * TSB + j] = (bf16_t)(d01 & 0xffffu);
      td[1 * TSB + j] = (bf16_t)(d01 >> 16);
      td[2 * TSB + j] = (bf16_t)(d23 & 0xffffu);
      td[3 * TSB + j] = (bf16_t)(d23 >> 16);
#pragma unroll
      for (int g = 0; g < 4; ++g) dbsum[g] += dc4[g];
    }
    f32x16 dy6 = score_bwd<L_WST>(s_ops, lane, dc4, h);
    dleaky_mul(k.t6, dy6);
    f32x16 e;
    {
      // z6 and dy6 tiles (zeros for lanes without a view: a5 is masked and dc reads 0); G6 dy6 as the operand of
      // e = W6^T (G6 dy6); constants four channels at a time
      float t16[16], gd[16];
      asm volatile("" ::: "memory");
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 g4 = *reinterpret_cast<const float4*>(s_tab[3] + T_G * D + 16 * h + 4 * q);
        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          t16[4 * q + c] = dy6[4 * q + c];
          gd[4 * q + c] = dy6[4 * q + c] * g[c];
        }
      }
      bf16x8 pk[2];
      pack16(t16, 0xffffffffu, pk);
      tileN_put_packed(b2, j, h, pk);
#pragma unroll
      for (int r = 0; r < 16; ++r) t16[r] = k.z6[r];
      pack16(t16, 0xffffffffu, pk);
      tileN_put_packed(b1, j, h, pk);
      pack16(gd, 0xffffffffu, pk);
      const f32x16 zero = {0};
      e = mm32_lds(s_ops, L_W6T, lane, pk, zero);
    }
    wave_sync();
    accS = wgradN_T(b0, td, lane, j, 4, h, accS);
    col_sums_xy(b2, b1, lane, s6a, s6b);                        // sum dy6 | sum dy6 z6
    wave_sync();
    // ---- phase 2: b0 = m5, b2 = m5 z5 (b1 = z6 stays): P2, Q2
    {
      const float one = ok ? 1.f : 0.f, low = ok ? SLOPE : 0.f;
      float m16[16], q16[16];
      asm volatile("" ::: "memory");
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 g4 = *reinterpret_cast<const float4*>(s_tab[2] + T_G * D + 16 * h + 4 * q);
        const float4 b4 = *reinterpret_cast<const float4*>(s_tab[2] + T_B * D + 16 * h + 4 * q);
        const float g[4] = {g4.x, g4.y, g4.z, g4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int r = 4 * q + c;
          const float y5 = __builtin_fmaf(k.z5[r], g[c], b[c]);      // layer 5 is evaluated plain: sign of G5 z5 + B5
          m16[r] = y5 > 0.f ? one : low;
          q16[r] = m16[r] * k.z5[r];
        }
      }
      bf16x8 mp[2], qp[2];
      pack16(m16, 0xffffffffu, mp);
      pack16(q16, 0xffffffffu, qp);
      tileN_put_packed(b0, j, h, mp);
      tileN_put_packed(b2, j, h, qp);
    }
    wave_sync();
    {
      const bf16x8 z0 = tileN_get(b1, lane, 0), z1 = tileN_get(b1, lane, 1);
      accP = CH_MFMA(tileN_get(b0, lane, 0), z0, accP);       // P2[c][j] = sum_v m5[v][c] z6[v][j]
      accP = CH_MFMA(tileN_get(b0, lane, 1), z1, accP);
      accQ = CH_MFMA(tileN_get(b2, lane, 0), z0, accQ);       // Q2[c][j] = sum_v (m5 z5)[v][c] z6[v][j]
      accQ = CH_MFMA(tileN_get(b2, lane, 1), z1, accQ);
    }
    wave_sync();
    // ---- phase 3: b1 = e: n5 | e1 = sum m5 | sum m5 e,  q5 | e2 = sum m5 z5 | sum (m5 z5) e
    {
      float t16[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) t16[r] = e[r];
      bf16x8 ep[2];
      pack16(t16, 0xffffffffu, ep);
      tileN_put_packed(b1, j, h, ep);
    }
    wave_sync();
    col_sums_xy(b0, b1, lane, n5, e1);
    col_sums_xy(b2, b1, lane, q5, e2);
    wave_sync();
  });
  flush_matrix_nat(accS, dWs, D, G, true, s_red, false);
  flush_matrix_nat(accP, acc5, D, D, false, s_red, true);
  flush_matrix_nat(accQ, acc5 + D * D, D, D, false, s_red, true);
  flush_col_stats(s6a, s6b, stats6, s_red);
  flush_col_stats(e1, e2, vec5, s_red);
  flush_col_stats(n5, q5, vec5 + 2 * D, s_red);
  __syncthreads();       // dbs: one atomic per block and group (see the attention backward)
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

// S5 (fp64 [64] = sum dy5 | sum dy5 z5, the form the layer passes accumulate) from the sums of score_l6_kernel once S6 is
// complete: sm6 fp32 [64] = S1/M | S2_hat/M of layer 6 (dva_bn_bwd_consts), bn6 [4+][32], W6 fp32 [32][32] (rounded to
// bf16 here like the operand of the W6^T product), acc5 = P2 | Q2, vec5 = e1 | e2 | n5 | q5.  One block of 32 threads.
__global__ void l6_consts_kernel(const float* __restrict__ sm6, const float* __restrict__ bn6,
                                 const float* __restrict__ W6, const float* __restrict__ acc5,
                                 const double* __restrict__ vec5, double* __restrict__ s5) {
  __shared__ double k1[D], k2[D];
  const int c = threadIdx.x;
  if (c < D) {
    const double mean = bn6[c], inv = bn6[D + c], g = (double)bn6[2 * D + c] * inv;
    const double s1 = sm6[c], s2 = sm6[D + c];
    k1[c] = g * (s1 - mean * inv * s2);          // dz = G dy - K1 - K2 z (stage_tab)
    k2[c] = g * inv * s2;
  }
  __syncthreads();
  if (c >= D) return;
  double t1 = 0.0, tp = 0.0, tq = 0.0;
  for (int jj = 0; jj < D; ++jj) {
    const double w = (double)bf2f(f2bf(W6[jj * D + c]));
    t1 += w * k1[jj];
    tp += w * k2[jj] * (double)acc5[c * D + jj];
    tq += w * k2[jj] * (double)acc5[D * D + c * D + jj];
  }
  s5[c] = vec5[c] - vec5[2 * D + c] * t1 - tp;
  s5[D + c] = vec5[D + c] - vec5[3 * D + c] * t1 - tq;
}

// ------------------------------------------------------------------------------------------------
// layer passes.  STAGE 6: dz6 -> dW6, S5; hands dy5 = leaky'(y5) da5 (bf16 [V, 32]) to the next pass.
// STAGE 5: dy5 -> dz5 -> dW5, du, S2 (view part); hands dy2 = leaky'(t2) da2 to the next pass.
// STAGE 2: dy2 + set-pooling gradient -> dz2 -> dW2, dz1 statistics S1, P = sum dy1 x^T.
// Each pass re-evaluates only the layers it differentiates (x_map -> ... -> its own layer); the gradient with
// respect to the layer's BatchNorm output crosses the BatchNorm barrier as one bf16 row per view (64 bytes), in the
// accumulator order of the lane that wrote it: re-deriving it through the later layers cost 0.8 ms per pass.
// The row is handed over AFTER the derivative of the activation (the pass that produces it has the pre-activation
// whose sign decides it: y5 in stage 6, the folded product t2 in stage 5).
// ------------------------------------------------------------------------------------------------
template <typename A16>
__device__ __forceinline__ void store_da(__amdgpu_buffer_rsrc_t R, bool ok, uint32_t view, int h, const A16& da) {
  float t[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) t[r] = da[r];
  const bf16x8 lo = pack8(&t[0]), hi = pack8(&t[8]);
  const uint32_t off = ok ? view * 64u + 32u * h : OOB;
  st128(R, off, __builtin_bit_cast(u32x4, lo));
  st128(R, ok ? off + 16u : OOB, __builtin_bit_cast(u32x4, hi));
}
__device__ __forceinline__ f32x16 unpack_da(const u32x4& lo, const u32x4& hi) {
  const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  f32x16 d;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    d[2 * i] = __uint_as_float(w[i] << 16);
    d[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
  return d;
}

// MERGED (STAGE 5 only, round 5): no stage 6 ran -- the pass starts from the score gradients dc [V, 4] and the constants
// of the BatchNorm-6 backward (sm6), evaluates dy6 -> dz6 -> da5 -> dy5 itself and takes dW6 along (written to `Pm`);
// S5 came from score_l6_kernel + l6_consts_kernel.
template <int STAGE, int OCC, bool KEYS = false, bool MERGED = false, bool A2IN = false>
__global__ __launch_bounds__(256, OCC) void layer_bwd_kernel(
    const float* __restrict__ x_map, const int32_t* __restrict__ vp, const float* __restrict__ u,
    const int2* __restrict__ tiles, const int32_t* __restrict__ n_tiles_dev, const uint4* __restrict__ ops,
    const float* __restrict__ bn1, const float* __restrict__ bn2, const float* __restrict__ bn5,
    const float* __restrict__ bn6, const float* __restrict__ sm2, const float* __restrict__ sm5,
    const float* __restrict__ sm6, const float* __restrict__ dc, const int32_t* __restrict__ arg,
    const float* __restrict__ dpooled, const bf16_t* __restrict__ da_in, bf16_t* __restrict__ da_out,
    float* __restrict__ dW,
    float* __restrict__ du, float* __restrict__ Pm, double* __restrict__ stats, int G, int64_t V, int64_t N,
    const float* __restrict__ qp, float qscale, const bf16_t* __restrict__ a2buf = nullptr) {
  static_assert(!A2IN || STAGE == 6, "the stored a2 row replaces x_map in stage 6 only (stages 5, 2 need layers 1-2)");
  constexpr bool A2_PREFETCH = false;     // the row is loaded in the body: a second prefetch set of it spills (56 bytes of scratch)
  __shared__ __attribute__((aligned(16))) float s_tab[4][TAB_FLOATS];
  __shared__ __attribute__((aligned(16))) bf16_t s_ta[4][32 * TSB], s_tb[4][32 * TSB];
  // second operand tile of the small products: 4 (score gradients) / 17 (x_map hi | lo | ones) rows + one shared zero row
  // stage 2: x_map as hi | lo (8 + 8 rows) and a row of ones (P then also carries sum dy1: the statistics of layer 1)
  constexpr int TD_ROWS = 17;
  __shared__ __attribute__((aligned(16))) bf16_t s_tc[STAGE == 2 ? 4 : 1][STAGE == 2 ? 32 * TSB : 8], s_td[STAGE == 2 ? 4 : 1][STAGE == 2 ? (TD_ROWS + 1) * TSB : 8];
  // STAGE 5: indicator tile [local point][view] (bf16 1.0 where the view belongs to the point) and the point ids
  __shared__ __attribute__((aligned(16))) bf16_t s_ind[STAGE == 5 ? 4 : 1][STAGE == 5 ? 32 * TSB : 8];
  __shared__ int s_plp[STAGE == 5 ? 4 : 1][32];
  // (the D x D reduction buffer of the epilogue lives in the first tile buffer: the tiles are dead by then)
  static_assert(sizeof(bf16_t) * 4 * 32 * TSB >= sizeof(float) * D * D, "epilogue buffer");
  float* s_red = reinterpret_cast<float*>(&s_ta[0][0]);
  // only the operands of the pass (LDS budget: three blocks per CU for stages 5 and 2):
  // stage 5: W1 W2 W5 | W5T -> local 5, 6;  stage 2: W1 W2 | W2T -> local 3, 4
  // plus the BatchNorm-folded operands (the forward passes' activations): stage 6 as chain_forward; stage 5: W1 in
  // place, W2 as a second operand (the raw z2 feeds the statistics) -> local 7, 8; stage 2: W1 as a second operand
  // -> local 5
  // stage 6: W1' W2' W5 W6 at their table positions 0..6, then W6T -> 7, 8; WST -> 9; W6' (folded) -> 10, 11
  static_assert(!MERGED || STAGE == 5, "the merged form is a stage-5 instance");
  constexpr int NOPS = STAGE == 6 ? (KEYS ? 13 : 12) : (STAGE == 5 ? (MERGED ? 16 : 9) : 6);      // (KEYS: W_k^T takes two blocks at L6_WST, the folded W6 moves up one)
  constexpr int L_W5T = 5, L_W2T = 3, L_W2F = 7, L_W1F = 5, L6_W6T = 7, L6_WST = 9, L6_W6F = KEYS ? 11 : 10;
  // merged stage 5: W6 -> 9, 10; W6' (folded) -> 11, 12; W6^T -> 13, 14; Ws^T -> 15
  constexpr int L5_W6 = 9, L5_W6F = 11, L5_W6T = 13, L5_WST = 15;
  __shared__ __attribute__((aligned(16))) uint4 s_ops[NOPS * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  if (STAGE == 6) {
    for (int i = threadIdx.x; i < 5 * 64; i += blockDim.x) {
      const int blk = i >> 6, l = i & 63;
      if (blk < 2) s_ops[(OP_W5 + blk) * 64 + l] = ops[(OP_W5 + blk) * 64 + l];
      else if (blk < 4) s_ops[(OP_W6 + blk - 2) * 64 + l] = ops[(OP_W6 + blk - 2) * 64 + l];
      else if (!KEYS) s_ops[L6_WST * 64 + l] = ops[OP_WST * 64 + l];
    }
    if (KEYS) {
      for (int i = threadIdx.x; i < 2 * 64; i += blockDim.x) s_ops[L6_WST * 64 + i] = ops[OP_WKT * 64 + i];
    }
    for (int i = threadIdx.x; i < 2 * 64; i += blockDim.x) s_ops[L6_W6T * 64 + i] = ops[OP_W6T * 64 + i];
    if (!A2IN) {
      fold_ops(s_ops, OP_W1, ops, OP_W1, 1, bn1);
      fold_ops(s_ops, OP_W2, ops, OP_W2, 2, bn2);
    }
    fold_ops(s_ops, L6_W6F, ops, OP_W6, 2, bn6);
  } else {
    for (int i = threadIdx.x; i < (STAGE == 5 ? 7 : 5) * 64; i += blockDim.x) {
      int op = i >> 6;
      if (STAGE == 5 && op >= 5) op = OP_W5T + (op - 5);
      if (STAGE == 2 && op >= 3) op = OP_W2T + (op - 3);
      s_ops[i] = ops[op * 64 + (i & 63)];
    }
    if (MERGED) {
      for (int i = threadIdx.x; i < 5 * 64; i += blockDim.x) {
        const int blk = i >> 6, l = i & 63;
        if (blk < 2) s_ops[(L5_W6 + blk) * 64 + l] = ops[(OP_W6 + blk) * 64 + l];
        else if (blk < 4) s_ops[(L5_W6T + blk - 2) * 64 + l] = ops[(OP_W6T + blk - 2) * 64 + l];
        else s_ops[L5_WST * 64 + l] = ops[OP_WST * 64 + l];
      }
    }
    __syncthreads();
    if (STAGE == 5) {
      fold_ops(s_ops, OP_W1, ops, OP_W1, 1, bn1);
      fold_ops(s_ops, L_W2F, ops, OP_W2, 2, bn2);
      if (MERGED) fold_ops(s_ops, L5_W6F, ops, OP_W6, 2, bn6);
    } else {
      fold_ops(s_ops, L_W1F, ops, OP_W1, 1, bn1);
    }
  }
  if (!A2IN) {
    stage_tab(s_tab[0], bn1, nullptr);
    stage_tab(s_tab[1], bn2, STAGE == 2 ? sm2 : nullptr);
  }
  stage_tab(s_tab[2], bn5, STAGE == 5 ? sm5 : nullptr);
  stage_tab(s_tab[3], bn6, (STAGE == 6 || MERGED) ? sm6 : nullptr);
  // second operand tiles hold rows that are never rewritten (score gradients: rows >= 4, x_map: rows >= 8)
  for (int i = threadIdx.x; i < 4 * 32 * TSB; i += blockDim.x) {
    if (STAGE == 2) {
      if (i < 4 * (TD_ROWS + 1) * TSB) (&s_td[0][0])[i] = 0;
    } else if (STAGE == 5) {
      (&s_ind[0][0])[i] = 0;
    }
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t X = make_rsrc(x_map, A2IN ? 0 : (uint64_t)V * 32), P = make_rsrc(vp, (uint64_t)V * 4),
                               U = make_rsrc(u, (uint64_t)N * 128), DC = make_rsrc(dc, (uint64_t)V * 16), QP = make_rsrc(qp, KEYS ? (uint64_t)N * 128 : 0),
                               AR = make_rsrc(arg, (uint64_t)N * 128), DP = make_rsrc(dpooled, (uint64_t)N * 128),
                               DI = make_rsrc(da_in, (uint64_t)V * 64), DO = make_rsrc(da_out, (uint64_t)V * 64),
                               A2 = make_rsrc(a2buf, A2IN ? (uint64_t)V * 64 : 0);
  float st[2][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) st[0][r] = st[1][r] = 0.f;
  float cs1 = 0.f, cs2 = 0.f;         // merged stage 5: S2 (view part) as column sums of natural tiles
  f32x16 accW = {0}, accS = {0};      // layer weight gradient; P (STAGE 2) / dW6 (merged stage 5)
  bf16_t* ta = s_ta[wv];
  bf16_t* tb_ = s_tb[wv];
  bf16_t* tc = s_tc[STAGE == 2 ? wv : 0];
  bf16_t* td = s_td[STAGE == 2 ? wv : 0];
  bf16_t* ind = s_ind[STAGE == 5 ? wv : 0];
  int* plp = s_plp[STAGE == 5 ? wv : 0];

  const int n_tiles = n_tiles_dev[0];
  int t0, t1;
  wave_tile_range(tiles, n_tiles, t0, t1);
  struct Pre {
    TileInfo ti;
    float4 x, dc;
    u32x4 dlo, dhi;
    int vpj;
  };
  auto loop = [&](auto&& load, auto&& body) {
    if constexpr (MERGED) run_tiles_single<Pre>(tiles, t0, t1, load, body);      // one register set: the occupancy step
    else run_tiles<Pre>(tiles, t0, t1, load, body);
  };
  loop([&](const TileInfo& ti, int t) {
    Pre p;
    p.ti = ti;
    const bool ok = j < p.ti.nv;
    const uint32_t view = (uint32_t)(p.ti.v0 + j);
    if constexpr (A2IN) {      // the stored layer-2 activation row instead of x_map
      if constexpr (A2_PREFETCH) {      // (dlo | dhi are free in stage 6)
        p.dlo = ld128(A2, ok ? view * 64u + 32u * h : OOB);
        p.dhi = ld128(A2, ok ? view * 64u + 32u * h + 16u : OOB);
      }
    } else {
      p.x = as_f4(ld128(X, ok ? view * 32u + 16u * h : OOB));
    }
    p.vpj = (int)ld32(P, ok ? view * 4u : OOB);
    if (STAGE == 2) {
      p.dlo = ld128(DI, ok ? view * 64u + 32u * h : OOB);
      p.dhi = ld128(DI, ok ? view * 64u + 32u * h + 16u : OOB);
    }
    // (stage 6: the score gradients, stage 5: the gradient row are loaded in the body: they are consumed late, and
    //  keeping two prefetch sets of them costs the occupancy step)
    return p;
  }, [&](const Pre& p) {
    const int nv = p.ti.nv;
    const bool ok = j < nv;
    const uint32_t keep = ok ? 0xffffffffu : 0u;
    const uint32_t view = (uint32_t)(p.ti.v0 + j);
    const f32x16 zero = {0};
    float dz[16];
    bf16x8 dzp[2];
    if constexpr (STAGE == 6) {
      const f32x16 uacc = load_u(U, ok, p.vpj, h);
      const float4 dcv = as_f4(ld128(DC, ok && (KEYS || h == 0) ? view * 16u : OOB));
      // forward up to a5 (layers 1, 2 folded, layer 5 plain); z5 stays for the statistics of layer 5
      bf16x8 a5[2];
      f32x16 z5;
      {
        bf16x8 a1[2], a2[2];
        asm volatile("" ::: "memory");
        if constexpr (A2IN && A2_PREFETCH) {
          a2[0] = __builtin_bit_cast(bf16x8, p.dlo);
          a2[1] = __builtin_bit_cast(bf16x8, p.dhi);
        } else if constexpr (A2IN) {
          a2[0] = __builtin_bit_cast(bf16x8, ld128(A2, ok ? view * 64u + 32u * h : OOB));
          a2[1] = __builtin_bit_cast(bf16x8, ld128(A2, ok ? view * 64u + 32u * h + 16u : OOB));
        } else {
          const f32x16 t1 = CH_MFMA(lds_op(s_ops, OP_W1, lane), pack_x(p.x), bias_acc(s_tab[0], T_B6, h));
          act_fold<false>(t1, keep, a1);
          const f32x16 t2 = mm32_lds(s_ops, OP_W2, lane, a1, bias_acc(s_tab[1], T_B6, h));
          act_fold<false>(t2, keep, a2);
        }
        z5 = mm32_lds(s_ops, OP_W5, lane, a2, uacc);
        act_pack<false>(z5, s_tab[2], h, keep, a5);      // unmasked: dz6 below is (pack16), so da5 = dy5 = 0 without a view
      }
      tileN_put_packed(tb_, j, h, a5);
      const float dc4[4] = {dcv.x, dcv.y, dcv.z, dcv.w};
      {
        // dy6 = leaky'(t6) Ws^T dc with the sign the forward's activation saw (the folded product), one 16-register
        // block at a time: t6, then the raw z6 for the BatchNorm backward
        f32x16 dy6;
        if constexpr (KEYS) {
          bf16x8 dkp[2];
          dkeys_operand(dcv, QP, ok, p.vpj, h, G, qscale, dkp);
          dy6 = mm32_lds(s_ops, L6_WST, lane, dkp, zero);       // da6 = W_k^T dK
        } else {
          dy6 = score_bwd<L6_WST>(s_ops, lane, dc4, h);
        }
        {
          const f32x16 t6 = mm32_lds(s_ops, L6_W6F, lane, a5, bias_acc(s_tab[3], T_B6, h));
          dleaky_mul(t6, dy6);
        }
        const f32x16 z6 = mm32_lds(s_ops, OP_W6, lane, a5, zero);
        bn_bwd_apply(z6, dy6, s_tab[3], h, dz);
      }
      pack16(dz, keep, dzp);
      tileN_put_packed(ta, j, h, dzp);
      const f32x16 da5 = mm32_lds(s_ops, L6_W6T, lane, dzp, zero);
      layer_bwd<true, false>(z5, da5, s_tab[2], h, ok, st, dz);        // layer 5 is evaluated plain: sign of G5 z5 + B5
      store_da(DO, ok, view, h, dz);                                    // dy5
      wave_sync();
      accW = wgradN(ta, tb_, lane, accW);      // dW6[n][k] = sum_v dz6[v][n] a5[v][k]
      wave_sync();
    } else if constexpr (STAGE == 5) {
      f32x16 uacc = load_u(U, ok, p.vpj, h);
      u32x4 dlo = {0, 0, 0, 0}, dhi = {0, 0, 0, 0};
      float4 dcv = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (MERGED) {
        dcv = as_f4(ld128(DC, ok && h == 0 ? view * 16u : OOB));
      } else {
        dlo = ld128(DI, ok ? view * 64u + 32u * h : OOB);
        dhi = ld128(DI, ok ? view * 64u + 32u * h + 16u : OOB);
      }
      // forward up to layer 5
      bf16x8 a1[2], a2[2];
      asm volatile("" ::: "memory");
      const f32x16 t1 = CH_MFMA(lds_op(s_ops, OP_W1, lane), pack_x(p.x), bias_acc(s_tab[0], T_B6, h));
      act_fold(t1, 0xffffffffu, a1);
      {
        const f32x16 t2 = mm32_lds(s_ops, L_W2F, lane, a1, bias_acc(s_tab[1], T_B6, h));
        act_fold<MERGED>(t2, keep, a2);      // unmasked: dz5 is masked (pack16) before it meets a2 in the dW5 product
      }
      {
        const f32x16 z5 = mm32_lds(s_ops, OP_W5, lane, a2, uacc);
        f32x16 dy5;
        if constexpr (MERGED) {
          // what stage 6 did for this view: dy6 = leaky'(t6) Ws^T dc, dz6 = BatchNorm-6 backward, da5 = W6^T dz6,
          // dy5 = leaky'(y5) da5 (fp32: no bf16 row in between); dW6 from the two natural tiles
          bf16x8 a5[2];
          act_pack(z5, s_tab[2], h, keep, a5);
          const float dc4[4] = {dcv.x, dcv.y, dcv.z, dcv.w};
          f32x16 dy6 = score_bwd<L5_WST>(s_ops, lane, dc4, h);
          {
            const f32x16 t6 = mm32_lds(s_ops, L5_W6F, lane, a5, bias_acc(s_tab[3], T_B6, h));
            dleaky_mul(t6, dy6);
          }
          float dz6[16];
          {
            const f32x16 z6 = mm32_lds(s_ops, L5_W6, lane, a5, zero);
            bn_bwd_apply(z6, dy6, s_tab[3], h, dz6);
          }
          bf16x8 dzp6[2];
          pack16(dz6, keep, dzp6);
          tileN_put_packed(ta, j, h, dzp6);
          tileN_put_packed(tb_, j, h, a5);
          const f32x16 da5 = mm32_lds(s_ops, L5_W6T, lane, dzp6, zero);
          asm volatile("" ::: "memory");
#pragma unroll
          for (int q = 0; q < 4; ++q) {        // four channels at a time: two float4 of constants live
            const float4 g4 = *reinterpret_cast<const float4*>(s_tab[2] + T_G * D + 16 * h + 4 * q);
            const float4 b4 = *reinterpret_cast<const float4*>(s_tab[2] + T_B * D + 16 * h + 4 * q);
            const float g[4] = {g4.x, g4.y, g4.z, g4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * q + e;
              dy5[r] = __builtin_fmaf(z5[r], g[e], b[e]) > 0.f ? da5[r] : SLOPE * da5[r];
            }
          }
          wave_sync();
          accS = wgradN(ta, tb_, lane, accS);      // dW6[n][k] = sum_v dz6[v][n] a5[v][k]
          wave_sync();
        } else {
          dy5 = unpack_da(dlo, dhi);           // stage 6 hands leaky'(y5) da5
        }
        bn_bwd_apply(z5, dy5, s_tab[2], h, dz);
      }
      pack16(dz, keep, dzp);
      tileN_put_packed(ta, j, h, dzp);
      if constexpr (!MERGED) tileN_put_packed(tb_, j, h, a2);      // (merged: a2 is packed again below, see there)
      // du[p][c] = sum of dz5 over the views of point p = dz5^T . indicator: one more product on the matrix cores
      // (operands: the transposed dz5 tile and a [local point][view] indicator tile of 1.0 / 0)
      const int prv = shfl(p.vpj, lane - 1);
      const bool is_start = ok && (j == 0 || prv != p.vpj);
      const uint32_t smask = (uint32_t)__ballot(is_start);
      const int lpj = __popc(smask & (0xffffffffu >> (31 - j))) - 1;
      const int nseg = __popc(smask);
      if (h == 0 && ok) {
        ind[lpj * TSB + j] = (bf16_t)0x3f80;
        if (is_start) plp[lpj] = p.vpj;
      }
      f32x16 dy2 = mm32_lds(s_ops, L_W5T, lane, dzp, zero);      // da2
      {
        // layer 2 was evaluated folded: leaky' follows the sign of t2 (evaluated again here: a1 is 8 registers, t2 16)
        const f32x16 t2 = mm32_lds(s_ops, L_W2F, lane, a1, bias_acc(s_tab[1], T_B6, h));
        dleaky_mul(t2, dy2);
        if constexpr (MERGED) {
          // the a2 operand of the dW5 product from the same t2: 8 registers that do not live across the layer-6 block
          bf16x8 a2b[2];
          act_fold(t2, keep, a2b);
          tileN_put_packed(tb_, j, h, a2b);
        }
      }
      if constexpr (!MERGED) {
        store_da(DO, ok, view, h, dy2);
        const f32x16 z2 = mm32_lds(s_ops, OP_W2, lane, a1, zero);     // the raw output: sum dy2 z2
        bn_bwd_stats(z2, dy2, st);
      }
      wave_sync();
      accW = wgradN(ta, tb_, lane, accW);      // dW5a[n][k] = sum_v dz5[v][n] a2[v][k]
      const int frag = p.ti.frag;
      const f32x16 accU = wgradN_T(ta, ind, lane, j, 32, h, zero);      // du[image column][local point j] of this tile
      if (h == 0 && ok) ind[lpj * TSB + j] = 0;           // leave the indicator tile clean for the next tile
      if constexpr (MERGED) {
        // the row that is handed over (bf16) and the raw z2 as natural tiles in the buffers the products have just read:
        // S2 = sum dy2 | sum dy2 z2 through column sums -- two registers instead of the 32 per-lane accumulators
        // (the third wavefront per SIMD of this instance); statistics of the rounded values
        wave_sync();
        float t16[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) t16[r] = dy2[r];
        bf16x8 dyp[2];
        pack16(t16, keep, dyp);
        const uint32_t off = ok ? view * 64u + 32u * h : OOB;
        st128(DO, off, __builtin_bit_cast(u32x4, dyp[0]));
        st128(DO, ok ? off + 16u : OOB, __builtin_bit_cast(u32x4, dyp[1]));
        tileN_put_packed(ta, j, h, dyp);
        {
          const f32x16 z2 = mm32_lds(s_ops, OP_W2, lane, a1, zero);
#pragma unroll
          for (int r = 0; r < 16; ++r) t16[r] = z2[r];
          bf16x8 zp[2];
          pack16(t16, keep, zp);
          tileN_put_packed(tb_, j, h, zp);
        }
        wave_sync();
        col_sums_xy(ta, tb_, lane, cs1, cs2);
      }
      {
        const bool wr = j < nseg;
        const uint32_t pt = wr ? (uint32_t)plp[j] : 0u;
        if (frag == 0) {
          const __amdgpu_buffer_rsrc_t DU = make_rsrc(du, (uint64_t)N * 128);
#pragma unroll
          for (int qq = 0; qq < 4; ++qq)      // registers 4 qq .. 4 qq + 3 = image columns -> channels cperm(chan(4 qq, h)) ..
            st128(DU, wr ? pt * 128u + (uint32_t)(4 * (qq >> 1) + 8 * h + 16 * (qq & 1)) * 4u : OOB,
                  as_u4(accU[4 * qq], accU[4 * qq + 1], accU[4 * qq + 2], accU[4 * qq + 3]));
        } else if (wr) {
          // a point with more than 32 views: its fragments add up in the (caller-zeroed) row -- no accumulator carried
          // across the tiles (16 registers for the sake of the rare long point cost the third wavefront per SIMD)
#pragma unroll
          for (int r = 0; r < 16; ++r) atomicAdd(&du[(size_t)pt * D + cperm(chan(r, h))], accU[r]);
        }
      }
      wave_sync();
    } else {
      // Register diet of this stage (three wavefronts per SIMD need <= 168 VGPRs): the set-pooling gradient is routed
      // into the gradient row as soon as it arrives (before the second layer is evaluated), and the raw first-layer
      // output is evaluated again where its backward needs it instead of being carried through the stage.
      u32x4 arq[4], dpq[4];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const uint32_t off = ok ? (uint32_t)p.vpj * 128u + (8u * qq + 4u * h) * 4u : OOB;
        arq[qq] = ld128(AR, off);
        dpq[qq] = ld128(DP, off);
      }
      bf16x8 a1[2];
      asm volatile("" ::: "memory");
      const bf16x8 xp = pack_x(p.x);
      {
        const f32x16 t1 = CH_MFMA(lds_op(s_ops, L_W1F, lane), xp, bias_acc(s_tab[0], T_B6, h));
        act_fold<false>(t1, keep, a1);       // unmasked: dz2 is masked (pack16) before the dW2 product and da1
      }
      // dy2 of the view path (stage 5) + the gradient of the max-pooled set features, which goes to the arg view of
      // each channel; dpooled arrives as leaky'(y*) dpooled (dva_chain_route_stats: the set pooling saw the plain y2)
      f32x16 da2t = unpack_da(p.dlo, p.dhi);
      {
        const int vg = p.ti.v0 + j;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const uint32_t ai[4] = {arq[qq].x, arq[qq].y, arq[qq].z, arq[qq].w};
          const uint32_t di[4] = {dpq[qq].x, dpq[qq].y, dpq[qq].z, dpq[qq].w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            da2t[4 * qq + e] += (int)ai[e] == vg ? __uint_as_float(di[e]) : 0.f;   // lanes without a view: arg reads 0 != vg
        }
      }
      asm volatile("" ::: "memory");
      {
        const f32x16 z2 = mm32_lds(s_ops, OP_W2, lane, a1, zero);
        bn_bwd_apply(z2, da2t, s_tab[1], h, dz);
      }
      pack16(dz, keep, dzp);
      tileN_put_packed(ta, j, h, dzp);
      tileN_put_packed(tb_, j, h, a1);
      const f32x16 da1 = mm32_lds(s_ops, L_W2T, lane, dzp, zero);
      {
        // dy1 = leaky'(y1) da1 with the sign of the folded product (what the forward's activation saw), evaluated
        // again here instead of carried through the stage; zero for lanes without a view (their da1 is)
        const f32x16 t1 = CH_MFMA(lds_op(s_ops, L_W1F, lane), xp, bias_acc(s_tab[0], T_B6, h));
#pragma unroll
        for (int r = 0; r < 16; ++r) dz[r] = da1[r] * dleaky(t1[r]);
      }
      // P = sum_v dy1 [x_hi | x_lo | 1]^T: the first-layer weight gradient AND (BatchNorm-1 backward being linear in
      // z1 = W1 x) the statistics of layer 1: S1 = P[:, 16], sum dy1 z1 = sum_f W1[:, f] (P[:, f] + P[:, 8 + f])
      {
        bf16x8 dy1p[2];
        pack16(dz, 0xffffffffu, dy1p);
        tileN_put_packed(tc, j, h, dy1p);
      }
      {
        const float xs[4] = {p.x.x, p.x.y, p.x.z, p.x.w};      // features 4 h .. 4 h + 3; lanes without a view: zeros
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const uint32_t hi2 = pack_bf16x2(xs[e], xs[e + 1]);
          tileT_put(td, 4 * h + e, 4 * h + e + 1, j, xs[e], xs[e + 1]);
          tileT_put(td, 8 + 4 * h + e, 8 + 4 * h + e + 1, j, xs[e] - __uint_as_float(hi2 << 16),
                    xs[e + 1] - __uint_as_float(hi2 & 0xffff0000u));
        }
        if (h == 0) td[16 * TSB + j] = ok ? (bf16_t)0x3f80 : (bf16_t)0;
      }
      wave_sync();
      accW = wgradN(ta, tb_, lane, accW);        // dW2[n][k] = sum_v dz2[v][n] a1[v][k]
      accS = wgradN_T(tc, td, lane, j, TD_ROWS, h, accS);         // P[n][f] = sum_v dy1[v][n] x[v][f]
      wave_sync();
    }
  });
  flush_matrix_nat(accW, dW, STAGE == 5 ? 2 * D : D, D, false, s_red, true);
  if (MERGED) flush_matrix_nat(accS, Pm, D, D, false, s_red, true);          // dW6
  if (STAGE == 2) flush_matrix_nat(accS, Pm, 20, 17, false, s_red, false);
  else if (MERGED) flush_col_stats(cs1, cs2, stats, s_red);
  else flush_stats<2>(st, stats, s_red);
}

// S2 of layer 2, per-point part: the set-pooling gradient lands on one view per (point, channel) whose z2 is
// zstar: stats += sum_p leaky'(y*) dpooled | the same times z_hat*  over the seen points.  dpy [N, 32] receives
// leaky'(y*) dpooled (0 for unseen points): what stage 2 routes to the arg views (the set pooling took the activation
// of the PLAIN product G2 z2 + B2, so its derivative follows that sign, not the folded product's).
__global__ __launch_bounds__(256) void route_stats_kernel(const float* __restrict__ zstar,
                                                          const float* __restrict__ dpooled,
                                                          const float* __restrict__ bn2,
                                                          const int64_t* __restrict__ ptr, double* __restrict__ stats,
                                                          float* __restrict__ dpy, int64_t N) {
  __shared__ float s_red[2 * D];
  const int q = threadIdx.x & 7, sub = threadIdx.x >> 3;       // 32 points per block iteration, 4 channels per thread
  float g[4], b[4], iv[4], mm[4], s1[4], s2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * q + e;
    g[e] = bn2[2 * D + c] * bn2[D + c];
    b[e] = bn2[3 * D + c] - bn2[c] * g[e];
    iv[e] = bn2[D + c];
    mm[e] = -bn2[c] * bn2[D + c];
    s1[e] = s2[e] = 0.f;
  }
  const int64_t stride = (int64_t)gridDim.x * 32;
  for (int64_t p0 = (int64_t)blockIdx.x * 32 + sub; p0 < N; p0 += 2 * stride) {
    // two points in flight per thread
    const int64_t p1 = p0 + stride;
    const bool ok0 = ptr[p0 + 1] > ptr[p0], ok1 = p1 < N && ptr[p1 + 1] > ptr[p1];
    float4 z0 = make_float4(0.f, 0.f, 0.f, 0.f), d0 = z0, z1 = z0, d1 = z0;
    if (ok0) {
      z0 = *reinterpret_cast<const float4*>(zstar + p0 * D + 4 * q);
      d0 = *reinterpret_cast<const float4*>(dpooled + p0 * D + 4 * q);
    }
    if (ok1) {
      z1 = *reinterpret_cast<const float4*>(zstar + p1 * D + 4 * q);
      d1 = *reinterpret_cast<const float4*>(dpooled + p1 * D + 4 * q);
    }
    const float zz[2][4] = {{z0.x, z0.y, z0.z, z0.w}, {z1.x, z1.y, z1.z, z1.w}};
    const float dd[2][4] = {{d0.x, d0.y, d0.z, d0.w}, {d1.x, d1.y, d1.z, d1.w}};
    float dyv[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dy = dd[u][e] * dleaky(__builtin_fmaf(zz[u][e], g[e], b[e]));   // unseen points: dpooled read as 0
        dyv[u][e] = dy;
        s1[e] += dy;
        s2[e] = __builtin_fmaf(dy, __builtin_fmaf(zz[u][e], iv[e], mm[e]), s2[e]);
      }
    }
    *reinterpret_cast<float4*>(dpy + p0 * D + 4 * q) = make_float4(dyv[0][0], dyv[0][1], dyv[0][2], dyv[0][3]);
    if (p1 < N) *reinterpret_cast<float4*>(dpy + p1 * D + 4 * q) = make_float4(dyv[1][0], dyv[1][1], dyv[1][2], dyv[1][3]);
  }
  if (threadIdx.x < 2 * D) s_red[threadIdx.x] = 0.f;
  __syncthreads();
  // lanes q, q + 8, ... of a wavefront own the same channels
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    for (int off = 8; off < 64; off <<= 1) {
      s1[e] += __shfl_xor(s1[e], off);
      s2[e] += __shfl_xor(s2[e], off);
    }
    if ((threadIdx.x & 63) < 8) {
      atomicAdd(&s_red[4 * q + e], s1[e]);
      atomicAdd(&s_red[D + 4 * q + e], s2[e]);
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * D) atomicAdd(&stats[threadIdx.x], (double)s_red[threadIdx.x]);
}

// ---- host-side arithmetic on the 64-entry statistics, as single launches -----------------------------------------
// stats fp64 [64] = S1 | sum dy z (raw layer output).  do_hat: S2 = invstd (sum dy z - mean S1) in place.
// sm = stats * inv_m (the constants of the next pass), dgamma = S2, dbeta = S1.
__global__ void bn_bwd_consts_kernel(double* __restrict__ stats, const float* __restrict__ bn, double inv_m,
                                     int do_hat, float* __restrict__ sm, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, int C) {
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const double s1 = stats[c];
    double s2 = stats[C + c];
    if (do_hat) {
      s2 = (double)bn[C + c] * (s2 - (double)bn[c] * s1);
      stats[C + c] = s2;
    }
    if (sm) {
      sm[c] = (float)(s1 * inv_m);
      sm[C + c] = (float)(s2 * inv_m);
    }
    if (dgamma) dgamma[c] = (float)s2;
    if (dbeta) dbeta[c] = (float)s1;
  }
}

// dW1 = G1 (P - (S1/M) SX^T - (S2/M) . Q),  Q = sum_v z1_hat x^T = invstd (bf16(W1) XX - mean SX^T) from the moments
// of x_map (mom fp64 [44] = SX [8] | upper triangle of XX, row-major).  One thread per entry, fp64.
// P fp32 [32][20] = sum_v dy1 [x_hi (8) | x_lo (8) | 1 | .]^T from stage 2.
__global__ void dw1_kernel(const float* __restrict__ P, const double* __restrict__ mom, const float* __restrict__ W1,
                           int exact_w1, const float* __restrict__ bn1, const float* __restrict__ sm1,
                           float* __restrict__ dW1) {
  const int i = threadIdx.x >> 3, k = threadIdx.x & 7;
  double acc = 0.0;
#pragma unroll
  for (int l = 0; l < 8; ++l) {
    const int a = l < k ? l : k, b = l < k ? k : l;
    const double xx = mom[8 + a * 8 - a * (a - 1) / 2 + (b - a)];
    acc += (exact_w1 ? (double)W1[i * 8 + l] : (double)bf2f(f2bf(W1[i * 8 + l]))) * xx;
  }
  const double mean = bn1[i], inv = bn1[D + i], gam = bn1[2 * D + i], sx = mom[k];
  const double q = inv * (acc - mean * sx);
  const double pik = (double)P[i * 20 + k] + (double)P[i * 20 + 8 + k];
  dW1[i * 8 + k] = (float)(gam * inv * (pik - (double)sm1[i] * sx - (double)sm1[D + i] * q));
}

// statistics of the BatchNorm-1 backward from P: S1 = sum dy1 = P[:, 16]; sum dy1 z1 with z1 = bf16(W1) x
__global__ void stats1_from_p_kernel(const float* __restrict__ P, const float* __restrict__ W1, int exact_w1,
                                     double* __restrict__ stats) {
  const int n = threadIdx.x;
  if (n >= D) return;
  double s2 = 0.0;
#pragma unroll
  for (int f = 0; f < 8; ++f)
    s2 += (exact_w1 ? (double)W1[n * 8 + f] : (double)bf2f(f2bf(W1[n * 8 + f]))) *
          ((double)P[n * 20 + f] + (double)P[n * 20 + 8 + f]);
  stats[n] = (double)P[n * 20 + 16];
  stats[D + n] = s2;
}

}  // namespace chain
}  // namespace dva

using namespace dva;
using namespace dva::chain;

extern "C" {

static int chain_attn_bwd_impl(const int32_t* rec_pos, const float* scores, const int32_t* view_point, const void* tiles, const int32_t* n_tiles,
                       const void* rows, const int32_t* row_idx, const int64_t* ptr, const float* gate_w,
                       const float* gate_b, const void* grad_out, const void* out, float* grad_scores, void* view_rec,
                       float* grad_gate_wb, int64_t n_points, int64_t n_views, int64_t n_rows, int32_t C, int32_t G,
                       int32_t scaling, float eps, void* stream) {
  if (n_views < 0 || n_points < 0) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!scores || !view_point || !tiles || !n_tiles || !rows || !row_idx || !ptr || !grad_out || !out ||
      !grad_scores || !view_rec || ((gate_w == nullptr) != (gate_b == nullptr)) || (gate_w && !grad_gate_wb))
    return DVA_ERR_INVALID;
  if (n_views * 16 > 0xfffffff0ll || n_rows * C * 2 > 0xfffffff0ll || n_points * C * 2 > 0xfffffff0ll)
    return DVA_ERR_UNSUPPORTED;
  const dim3 grid(chain_grid(C <= 64 ? 4 : 3)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define DVA_ATTN_BWD(LPR_, G_)                                                                                    \
  hipLaunchKernelGGL((attn_bwd_kernel<bf16_t, LPR_, G_>), grid, block, 0, s, scores, view_point,                  \
                     (const int2*)tiles, n_tiles, (const bf16_t*)rows, row_idx, ptr, gate_w, gate_b,              \
                     (const bf16_t*)grad_out, (const bf16_t*)out, grad_scores, (uint32_t*)view_rec, grad_gate_wb, \
                     scaling, eps, n_views, n_points, n_rows, rec_pos)
  const int key = C * 8 + G;
  switch (key) {
    case 32 * 8 + 1: DVA_ATTN_BWD(4, 1); break;
    case 32 * 8 + 2: DVA_ATTN_BWD(4, 2); break;
    case 32 * 8 + 4: DVA_ATTN_BWD(4, 4); break;
    case 64 * 8 + 1: DVA_ATTN_BWD(8, 1); break;
    case 64 * 8 + 2: DVA_ATTN_BWD(8, 2); break;
    case 64 * 8 + 4: DVA_ATTN_BWD(8, 4); break;
    case 128 * 8 + 1: DVA_ATTN_BWD(16, 1); break;
    case 128 * 8 + 2: DVA_ATTN_BWD(16, 2); break;
    case 128 * 8 + 4: DVA_ATTN_BWD(16, 4); break;
    case 256 * 8 + 1: DVA_ATTN_BWD(32, 1); break;
    case 256 * 8 + 2: DVA_ATTN_BWD(32, 2); break;
    case 256 * 8 + 4: DVA_ATTN_BWD(32, 4); break;
    case 512 * 8 + 1: DVA_ATTN_BWD(64, 1); break;
    case 512 * 8 + 2: DVA_ATTN_BWD(64, 2); break;
    case 512 * 8 + 4: DVA_ATTN_BWD(64, 4); break;
    default: return DVA_ERR_UNSUPPORTED;
  }
#undef DVA_ATTN_BWD
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_attn_bwd(const float* scores, const int32_t* view_point, const void* tiles, const int32_t* n_tiles,
                       const void* rows, const int32_t* row_idx, const int64_t* ptr, const float* gate_w,
                       const float* gate_b, const void* grad_out, const void* out, float* grad_scores, void* view_rec,
                       float* grad_gate_wb, int64_t n_points, int64_t n_views, int64_t n_rows, int32_t C, int32_t G,
                       int32_t scaling, float eps, void* stream) {
  return chain_attn_bwd_impl(nullptr, scores, view_point, tiles, n_tiles, rows, row_idx, ptr, gate_w, gate_b, grad_out, out,
                             grad_scores, view_rec, grad_gate_wb, n_points, n_views, n_rows, C, G, scaling, eps, stream);
}

int dva_chain_attn_bwd_planrec(const int32_t* rec_pos, const float* scores, const int32_t* view_point, const void* tiles,
                               const int32_t* n_tiles, const void* rows, const int32_t* row_idx, const int64_t* ptr,
                               const float* gate_w, const float* gate_b, const void* grad_out, const void* out,
                               float* grad_scores, void* view_rec, float* grad_gate_wb, int64_t n_points, int64_t n_views,
                               int64_t n_rows, int32_t C, int32_t G, int32_t scaling, float eps, void* stream) {
  if (!rec_pos && n_views > 0) return DVA_ERR_INVALID;
  return chain_attn_bwd_impl(rec_pos, scores, view_point, tiles, n_tiles, rows, row_idx, ptr, gate_w, gate_b, grad_out, out,
                             grad_scores, view_rec, grad_gate_wb, n_points, n_views, n_rows, C, G, scaling, eps, stream);
}

int dva_chain_attn_bwd_f32(const float* scores, const int32_t* view_point, const void* tiles, const int32_t* n_tiles,
                           const float* rows, const int32_t* row_idx, const int64_t* ptr, const float* gate_w,
                           const float* gate_b, const float* grad_out, const float* out, float* grad_scores,
                           float* view_rec, float* grad_gate_wb, int64_t n_points, int64_t n_views, int64_t n_rows,
                           int32_t C, int32_t G, int32_t scaling, float eps, void* stream) {
  if (n_views < 0 || n_points < 0) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!scores || !view_point || !tiles || !n_tiles || !rows || !row_idx || !ptr || !grad_out || !out ||
      !grad_scores || !view_rec || ((gate_w == nullptr) != (gate_b == nullptr)) || (gate_w && !grad_gate_wb))
    return DVA_ERR_INVALID;
  if (n_views * 32 > 0xfffffff0ll || n_rows * C * 4 > 0xfffffff0ll || n_points * C * 4 > 0xfffffff0ll)
    return DVA_ERR_UNSUPPORTED;
  const dim3 grid(chain_grid(3)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define DVA_ATTN_BWD32(LPR_, G_)                                                                                  \
  hipLaunchKernelGGL((attn_bwd_kernel<float, LPR_, G_>), grid, block, 0, s, scores, view_point,                   \
                     (const int2*)tiles, n_tiles, rows, row_idx, ptr, gate_w, gate_b, grad_out, out, grad_scores, \
                     (uint32_t*)view_rec, grad_gate_wb, scaling, eps, n_views, n_points, n_rows)
  const int key = C * 8 + G;
  switch (key) {
    case 32 * 8 + 1: DVA_ATTN_BWD32(8, 1); break;
    case 32 * 8 + 2: DVA_ATTN_BWD32(8, 2); break;
    case 32 * 8 + 4: DVA_ATTN_BWD32(8, 4); break;
    case 64 * 8 + 1: DVA_ATTN_BWD32(16, 1); break;
    case 64 * 8 + 2: DVA_ATTN_BWD32(16, 2); break;
    case 64 * 8 + 4: DVA_ATTN_BWD32(16, 4); break;
    case 128 * 8 + 1: DVA_ATTN_BWD32(32, 1); break;
    case 128 * 8 + 2: DVA_ATTN_BWD32(32, 2); break;
    case 128 * 8 + 4: DVA_ATTN_BWD32(32, 4); break;
    case 256 * 8 + 1: DVA_ATTN_BWD32(64, 1); break;
    case 256 * 8 + 2: DVA_ATTN_BWD32(64, 2); break;
    case 256 * 8 + 4: DVA_ATTN_BWD32(64, 4); break;
    default: return DVA_ERR_UNSUPPORTED;
  }
#undef DVA_ATTN_BWD32
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_score_stats(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                          const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2,
                          const float* bn5, const float* bn6, const float* grad_scores, double* stats6, float* dWs,
                          float* dbs, int32_t G, int64_t n_views, int64_t n_points, void* stream) {
  if (n_views < 0 || n_points < 0 || G < 1 || G > 4) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!x_map || !view_point || !u || !tiles || !n_tiles || !ops || !bn1 || !bn2 || !bn5 || !bn6 || !grad_scores ||
      !stats6 || !dWs || !dbs)
    return DVA_ERR_INVALID;
  if (n_views * 32 > 0xfffffff0ll || n_points * 128 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((score_stats_kernel<false>), dim3(chain_grid(3)), dim3(256), 0, (hipStream_t)stream, x_map,
                     view_point, u, (const int2*)tiles, n_tiles, (const uint4*)ops, bn1, bn2, bn5, bn6, grad_scores,
                     stats6, dWs, dbs, (int)G, n_views, n_points, (const float*)nullptr, 0.f);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

// The stored-a2 hybrid (round 6): the same pass starting from the layer-2 activation row the forward stored
// (dva_chain_stats_a2(5): bf16 [V, 32], accumulator order) instead of x_map -- 64 instead of 32 bytes per view in, layers 1
// and 2 not evaluated.  Same results bit for bit (the row IS the operand layer 5 consumes).
int dva_chain_score_stats_a2(const void* a2, const int32_t* view_point, const float* u, const void* tiles,
                             const int32_t* n_tiles, const void* ops, const float* bn5, const float* bn6,
                             const float* grad_scores, double* stats6, float* dWs, float* dbs, int32_t G,
                             int64_t n_views, int64_t n_points, void* stream) {
  if (n_views < 0 || n_points < 0 || G < 1 || G > 4) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!a2 || !view_point || !u || !tiles || !n_tiles || !ops || !bn5 || !bn6 || !grad_scores || !stats6 || !dWs ||
      !dbs || ((uintptr_t)a2 & 15))
    return DVA_ERR_INVALID;
  if (n_views * 64 > 0xfffffff0ll || n_points * 128 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((score_stats_kernel<false, true>), dim3(chain_grid(3)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)nullptr, view_point, u, (const int2*)tiles, n_tiles, (const uint4*)ops,
                     (const float*)nullptr, (const float*)nullptr, bn5, bn6, grad_scores, stats6, dWs, dbs, (int)G,
                     n_views, n_points, (const float*)nullptr, 0.f, (const bf16_t*)a2);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

// stage 6 of dva_chain_bwd_layer from the stored a2 row: dW6, dy5 (da_out bf16 [V, 32]), S of layer 5
int dva_chain_bwd_layer6_a2(const void* a2, const int32_t* view_point, const float* u, const void* tiles,
                            const int32_t* n_tiles, const void* ops, const float* bn5, const float* bn6,
                            const float* sm6, const float* grad_scores, void* da_out, float* dW, double* stats,
                            int32_t G, int64_t n_views, int64_t n_points, void* stream) {
  if (n_views < 0 || n_points < 0 || G < 1 || G > 4) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!a2 || !view_point || !u || !tiles || !n_tiles || !ops || !bn5 || !bn6 || !sm6 || !grad_scores || !da_out ||
      !dW || !stats || ((uintptr_t)a2 & 15))
    return DVA_ERR_INVALID;
  if (n_views * 64 > 0xfffffff0ll || n_points * 128 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((layer_bwd_kernel<6, 3, false, false, true>), dim3(chain_grid(3)), dim3(256), 0,
                     (hipStream_t)stream, (const float*)nullptr, view_point, u, (const int2*)tiles, n_tiles,
                     (const uint4*)ops, (const float*)nullptr, (const float*)nullptr, bn5, bn6, (const float*)nullptr,
                     (const float*)nullptr, sm6, grad_scores, (const int32_t*)nullptr, (const float*)nullptr,
                     (const bf16_t*)nullptr, (bf16_t*)da_out, dW, (float*)nullptr, (float*)nullptr, stats, (int)G,
                     n_views, n_points, (const float*)nullptr, 0.f, (const bf16_t*)a2);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

// key layer of QKVBimodalCSRPool as the chain's last layer: grad_compat fp32 [V][4] (G = 1, 2, 4 query-key groups used),
// queries fp32 [N][32] in position order, scale = 1 / sqrt(nc_qk) or 1; dWk [32][32], dbk [32]
int dva_chain_score_stats_keys(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                               const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2,
                               const float* bn5, const float* bn6, const float* grad_compat, const float* queries,
                               double* stats6, float* dWk, float* dbk, int32_t G, float scale, int64_t n_views,
                               int64_t n_points, void* stream) {
  if (n_views < 0 || n_points < 0 || (G != 1 && G != 2 && G != 4)) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!x_map || !view_point || !u || !tiles || !n_tiles || !ops || !bn1 || !bn2 || !bn5 || !bn6 || !grad_compat ||
      !queries || !stats6 || !dWk || !dbk || ((uintptr_t)queries & 15) || ((uintptr_t)grad_compat & 15))
    return DVA_ERR_INVALID;
  if (n_views * 32 > 0xfffffff0ll || n_points * 128 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((score_stats_kernel<true>), dim3(chain_grid(3)), dim3(256), 0, (hipStream_t)stream, x_map,
                     view_point, u, (const int2*)tiles, n_tiles, (const uint4*)ops, bn1, bn2, bn5, bn6, grad_compat,
                     stats6, dWk, dbk, (int)G, n_views, n_points, queries, scale);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_bwd_layer(int32_t stage, const float* x_map, const int32_t* view_point, const float* u,
                        const void* tiles, const int32_t* n_tiles, const void* ops, const float* bn1,
                        const float* bn2, const float* bn5, const float* bn6, const float* sm2, const float* sm5,
                        const float* sm6, const float* grad_scores, const int32_t* arg, const float* dpooled,
                        const void* da_in, void* da_out, float* dW, float* du, float* P,
                        double* stats, int32_t G, int64_t n_views, int64_t n_points, void* stream) {
  if (n_views < 0 || (stage != 6 && stage != 5 && stage != 2) || G < 1 || G > 4) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!x_map || !view_point || !u || !tiles || !n_tiles || !ops || !bn1 || !bn2 || !bn5 || !bn6 || !dW || (stage != 2 && !stats))
    return DVA_ERR_INVALID;
  if (stage == 6 && (!sm6 || !grad_scores || !da_out)) return DVA_ERR_INVALID;
  if (stage == 5 && (!sm5 || !du || !da_in || !da_out)) return DVA_ERR_INVALID;
  if (stage == 2 && (!sm2 || !arg || !dpooled || !P || !da_in)) return DVA_ERR_INVALID;
  if (n_views * 64 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  if (n_views * 32 > 0xfffffff0ll || n_points * 128 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  const dim3 block(256);
  hipStream_t s = (hipStream_t)stream;
#define DVA_LAYER_BWD(ST_, BPC_)                                                                                  \
  hipLaunchKernelGGL((layer_bwd_kernel<ST_, BPC_>), dim3(chain_grid(BPC_)), block, 0, s, x_map, view_point, u,  \
                     (const int2*)tiles, n_tiles, (const uint4*)ops, bn1, bn2, bn5, bn6, sm2, sm5, sm6,          \
                     grad_scores, arg, dpooled, (const bf16_t*)da_in, (bf16_t*)da_out, dW, du, P, stats, G, \
                     n_views, n_points, (const float*)nullptr, 0.f)
  static const int occ5 = tune_int("DVA_STAGE5_OCC", 3);
  if (stage == 6) DVA_LAYER_BWD(6, 3);
  else if (stage == 5 && occ5 == 2) DVA_LAYER_BWD(5, 2);
  else if (stage == 5) DVA_LAYER_BWD(5, 3);
  else DVA_LAYER_BWD(2, 3);
#undef DVA_LAYER_BWD
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

// Merged backward (round 5): score layer + the linear pieces of S5 in one pass (score_l6_kernel), S5 from them
// (l6_consts_kernel), stage 5 from the score gradients (layer_bwd_kernel<5, ., false, true>): no stage 6, no dy5 tensor.
int dva_chain_score_l6_stats(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                             const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2,
                             const float* bn5, const float* bn6, const float* grad_scores, double* stats6, float* dWs,
                             float* dbs, float* acc5, double* vec5, int32_t G, int64_t n_views, int64_t n_points,
                             void* stream) {
  if (n_views < 0 || n_points < 0 || G < 1 || G > 4) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!x_map || !view_point || !u || !tiles || !n_tiles || !ops || !bn1 || !bn2 || !bn5 || !bn6 || !grad_scores ||
      !stats6 || !dWs || !dbs || !acc5 || !vec5)
    return DVA_ERR_INVALID;
  if (n_views * 32 > 0xfffffff0ll || n_points * 128 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(score_l6_kernel, dim3(chain_grid(3)), dim3(256), 0, (hipStream_t)stream, x_map, view_point, u,
                     (const int2*)tiles, n_tiles, (const uint4*)ops, bn1, bn2, bn5, bn6, grad_scores, stats6, dWs, dbs,
                     acc5, vec5, (int)G, n_views, n_points);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_l6_consts(const float* sm6, const float* bn6, const float* W6, const float* acc5, const double* vec5,
                        double* stats5, void* stream) {
  if (!sm6 || !bn6 || !W6 || !acc5 || !vec5 || !stats5) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(l6_consts_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sm6, bn6, W6, acc5, vec5, stats5);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_bwd_layer5_merged(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                                const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2,
                                const float* bn5, const float* bn6, const float* sm5, const float* sm6,
                                const float* grad_scores, void* da_out, float* dW5, float* dW6, float* du, double* stats2,
                                int32_t G, int64_t n_views, int64_t n_points, void* stream) {
  if (n_views < 0 || n_points < 0 || G < 1 || G > 4) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!x_map || !view_point || !u || !tiles || !n_tiles || !ops || !bn1 || !bn2 || !bn5 || !bn6 || !sm5 || !sm6 ||
      !grad_scores || !da_out || !dW5 || !dW6 || !du || !stats2)
    return DVA_ERR_INVALID;
  if (n_views * 64 > 0xfffffff0ll || n_points * 128 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  const dim3 block(256);
  hipStream_t s = (hipStream_t)stream;
  static const int occ = tune_int("DVA_STAGE5M_OCC", 3);
#define DVA_LAYER5M(OCC_)                                                                                        \
  hipLaunchKernelGGL((layer_bwd_kernel<5, OCC_, false, true>), dim3(chain_grid(OCC_)), block, 0, s, x_map,        \
                     view_point, u, (const int2*)tiles, n_tiles, (const uint4*)ops, bn1, bn2, bn5, bn6,           \
                     (const float*)nullptr, sm5, sm6, grad_scores, (const int32_t*)nullptr, (const float*)nullptr, \
                     (const bf16_t*)nullptr, (bf16_t*)da_out, dW5, du, dW6, stats2, (int)G, n_views, n_points,    \
                     (const float*)nullptr, 0.f)
  if (occ == 3) DVA_LAYER5M(3);
  else DVA_LAYER5M(2);
#undef DVA_LAYER5M
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

// stage 6 of the chain backward below the key layer (see dva_chain_score_stats_keys): dy5 rows out, dW6, statistics of layer 5
int dva_chain_bwd_layer6_keys(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                              const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2,
                              const float* bn5, const float* bn6, const float* sm6, const float* grad_compat,
                              const float* queries, void* da_out, float* dW, double* stats, int32_t G, float scale,
                              int64_t n_views, int64_t n_points, void* stream) {
  if (n_views < 0 || n_points < 0 || (G != 1 && G != 2 && G != 4)) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!x_map || !view_point || !u || !tiles || !n_tiles || !ops || !bn1 || !bn2 || !bn5 || !bn6 || !sm6 || !grad_compat ||
      !queries || !da_out || !dW || !stats || ((uintptr_t)queries & 15) || ((uintptr_t)grad_compat & 15))
    return DVA_ERR_INVALID;
  if (n_views * 64 > 0xfffffff0ll || n_points * 128 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((layer_bwd_kernel<6, 3, true>), dim3(chain_grid(3)), dim3(256), 0, (hipStream_t)stream, x_map,
                     view_point, u, (const int2*)tiles, n_tiles, (const uint4*)ops, bn1, bn2, bn5, bn6,
                     (const float*)nullptr, (const float*)nullptr, sm6, grad_compat, (const int32_t*)nullptr,
                     (const float*)nullptr, (const bf16_t*)nullptr, (bf16_t*)da_out, dW, (float*)nullptr,
                     (float*)nullptr, stats, (int)G, n_views, n_points, queries, scale);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_route_stats(const float* zstar, const float* dpooled, const float* bn2, const int64_t* ptr,
                          double* stats, float* dpooled_dy, int64_t n_points, void* stream) {
  if (n_points < 0) return DVA_ERR_INVALID;
  if (n_points == 0) return DVA_OK;
  if (!zstar || !dpooled || !bn2 || !ptr || !stats || !dpooled_dy) return DVA_ERR_INVALID;
  int64_t blocks = (n_points + 63) / 64;
  const int cap = chain_grid(4);
  hipLaunchKernelGGL(route_stats_kernel, dim3((int)(blocks < cap ? blocks : cap)), dim3(256), 0,
                     (hipStream_t)stream, zstar, dpooled, bn2, ptr, stats, dpooled_dy, n_points);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_bn_bwd_consts(double* stats, const float* bn, double inv_m, int32_t do_hat, float* sm, float* dgamma,
                      float* dbeta, int32_t C, void* stream) {
  if (!stats || (do_hat && !bn) || C < 1) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(bn_bwd_consts_kernel, dim3(1), dim3(C <= 64 ? 64 : 256), 0, (hipStream_t)stream, stats, bn,
                     inv_m, (int)do_hat, sm, dgamma, dbeta, (int)C);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_stats1(const float* P, const float* W1, int32_t exact_w1, double* stats, void* stream) {
  if (!P || !W1 || !stats) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(stats1_from_p_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, P, W1, (int)exact_w1, stats);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_dw1(const float* P, const double* mom, const float* W1, int32_t exact_w1, const float* bn1,
                  const float* sm1, float* dW1, void* stream) {
  if (!P || !mom || !W1 || !bn1 || !sm1 || !dW1) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(dw1_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, P, mom, W1, (int)exact_w1, bn1, sm1, dW1);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // extern "C"
