// Recompute chain, per-point set branch of DeepSetFeat (modules/multimodal/pooling.py:660-664):
//   pooled [N, 32] (+ sqrt(1 / (n + 1e-3)) when use_num)  ->  mlp_set = MLP[33, 32, 32]  ->  u = Wc[:, 32:] . s
// u is the per-point half of the concatenation layer that the view kernels add to z5.
// Same geometry as the view chain (chain_common.h) with lane (j, h) = point j of a 32-point tile, and the same
// discipline: every pass re-evaluates the branch from `pooled`, one pass per BatchNorm barrier --
//   dva_chain_set_fwd  stage 1: statistics of s1 | stage 2: statistics of s2 | stage 3: u
//   dva_chain_set_bwd  stage 1: dWc (set half), S of layer s2 | stage 2: dWsb, S of layer s1 |
//                      stage 3: dWsa (incl. the set-size column), dpooled
// N is ~3 % of V: these passes are small, so the products keep fp32-class accuracy with the three-term split
//   x = hi + lo (bf16 each),  W x ~ W_hi x_hi + W_lo x_hi + W_hi x_lo   (dropped term < 2^-16 relative);
// the weight gradients take bf16-rounded operands through transposed LDS tiles like the view passes.
#include "chain_split.h"

namespace dva {
namespace chain {

// operand table of the set branch: per matrix hi (2 k-blocks) | lo (2 k-blocks)
enum { SO_WSA = 0, SO_WSB = 4, SO_WCB = 8, SO_WCBT = 12, SO_WSBT = 16, SO_WSAT = 20, N_SOPS = 24 };

__global__ __launch_bounds__(64) void set_prep_kernel(const float* __restrict__ Wsa, int ldsa,
                                                      const float* __restrict__ Wsb, const float* __restrict__ Wc,
                                                      int ldc, uint4* __restrict__ ops) {
  const int op = blockIdx.x, lane = threadIdx.x, j = lane & 31, h = lane >> 5;
  const int mat = op / 4, lo = (op >> 1) & 1, m = op & 1;
  float w[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int c = chan(8 * m + s, h);
    float v;
    switch (mat) {
      case 0: v = Wsa[j * ldsa + c]; break;             // forward: W[out = j][in = c]
      case 1: v = Wsb[j * D + c]; break;
      case 2: v = Wc[j * ldc + D + c]; break;
      case 3: v = Wc[c * ldc + D + j]; break;           // transposed: W[out = c][in = j]
      case 4: v = Wsb[c * D + j]; break;
      default: v = Wsa[c * ldsa + j]; break;
    }
    const float hi = bf2f(f2bf(v));
    w[s] = lo ? v - hi : hi;
  }
  ops[op * 64 + lane] = __builtin_bit_cast(uint4, pack8(w));
}

// w33 [32]: the set-size column of Wsa (use_num), nullptr otherwise.  DIR = 0 forward, 1 backward.
template <int DIR, int STAGE>
__global__ __launch_bounds__(256, DIR == 0 ? 4 : 3) void set_kernel(
    const float* __restrict__ pooled, const int64_t* __restrict__ ptr, const float* __restrict__ w33,
    const uint4* __restrict__ ops, const float* __restrict__ bn_s1, const float* __restrict__ bn_s2,
    const float* __restrict__ sm_s1, const float* __restrict__ sm_s2, const float* __restrict__ du,
    float* __restrict__ u_out, float* __restrict__ dpooled, float* __restrict__ dW, int ld_dw,
    float* __restrict__ dw33, double* __restrict__ stats, int64_t N) {
  __shared__ __attribute__((aligned(16))) float s_tab[2][TAB_FLOATS];
  __shared__ __attribute__((aligned(16))) float s_w33[D];
  __shared__ __attribute__((aligned(16))) uint4 s_ops[N_SOPS * 64];
  __shared__ __attribute__((aligned(16))) bf16_t s_ta[4][32 * TSB], s_tb[4][32 * TSB];
  __shared__ float s_red[D * D];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < N_SOPS * 64; i += blockDim.x) s_ops[i] = ops[i];
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
  bf16_t* ta = s_ta[wv];
  bf16_t* tb = s_tb[wv];
  const f32x16 zero = {0};
  const int64_t tiles = (N + 31) / 32;
  const int64_t wave = rfl((int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  // (round 5, measured: requesting the rows of the wavefront's NEXT tile before the current one is computed changes
  //  nothing -- 40 / 46 / 49 us forward, 129 / 105 / 99 us backward at 2^20 points either way -- and costs 32 registers,
  //  i.e. the third wavefront per SIMD of the backward stages; SQ counters: the vector unit is busy 0.28 of a wave's
  //  cycles at two wavefronts per SIMD, like the view passes at three.  Grids of 4 (forward) / 3 (backward) blocks per CU
  //  instead of 2 -- the registers allow them -- measured SLOWER on the reference-sized batch (78 -> 97 us forward,
  //  168 -> 191 us backward for 3e5 points: every block pays the 24 KB operand table and the flush): 2 stays)
  for (int64_t t = wave; t < tiles; t += n_waves) {
    const int64_t p = t * 32 + j;
    const bool ok = p < N;
    const uint32_t keep = ok ? 0xffffffffu : 0u, row = (uint32_t)p;
    float x[16], w3[16];
    load_rows16(PL, ok, row, h, x);
    float num = 0.f;
    if (w33 && ok) num = sqrtf(1.f / ((float)(ptr[p + 1] - ptr[p]) + 1e-3f));
    tab16(s_w33, 0, h, w3);
    // ---- forward: s1 = Wsa [pooled | num], a1 = act(BN(s1)), s2 = Wsb a1, a2 = act(BN(s2)), u = WcB a2
    const Split xs = split16(x, keep);
    f32x16 z1 = mm3(s_ops, SO_WSA, lane, xs, zero);
#pragma unroll
    for (int r = 0; r < 16; ++r) z1[r] = ok ? __builtin_fmaf(num, w3[r], z1[r]) : 0.f;
    auto add_stats = [&](const f32x16& z) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[0][r] += z[r];
        st[1][r] = __builtin_fmaf(z[r], z[r], st[1][r]);
      }
    };
    if (DIR == 0 && STAGE == 1) { add_stats(z1); continue; }
    float a1[16], a2[16];
    act16(z1, s_tab[0], h, a1);
    const Split a1s = split16(a1, keep);
    const f32x16 z2 = mm3(s_ops, SO_WSB, lane, a1s, zero);
    if (DIR == 0 && STAGE == 2) { add_stats(z2); continue; }
    act16(z2, s_tab[1], h, a2);
    const Split a2s = split16(a2, keep);
    if (DIR == 0) {
      store_rows16(UO, ok, row, h, mm3(s_ops, SO_WCB, lane, a2s, zero));
      continue;
    }
    // ---- backward
    float d[16], dz[16], unused_st[2][16];
    load_rows16(DU, ok, row, h, d);
    const Split ds = split16(d, keep);
    const f32x16 da2 = mm3(s_ops, SO_WCBT, lane, ds, zero);
    if (STAGE == 1) {
      layer_bwd<true, false>(z2, da2, s_tab[1], h, ok, st, dz);
      tileT_put_f32(ta, j, h, d, ok);
      tileT_put_f32(tb, j, h, a2, ok);
      wave_sync();
      accW = wgrad(ta, tb, j, h, accW);          // dWcB[n][k] = sum_p du[p][n] a2[p][k]
      wave_sync();
      continue;
    }
    layer_bwd<false, true>(z2, da2, s_tab[1], h, ok, unused_st, dz);
#pragma unroll
    for (int r = 0; r < 16; ++r) dz[r] = ok ? dz[r] : 0.f;
    const Split dz2s = split16(dz, keep);
    const f32x16 da1 = mm3(s_ops, SO_WSBT, lane, dz2s, zero);
    if (STAGE == 2) {
      float tmp[16];
      layer_bwd<true, false>(z1, da1, s_tab[0], h, ok, st, tmp);
      tileT_put_f32(ta, j, h, dz, ok);
      tileT_put_f32(tb, j, h, a1, ok);
      wave_sync();
      accW = wgrad(ta, tb, j, h, accW);          // dWsb[n][k] = sum_p dz2[p][n] a1[p][k]
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
    const Split dz1s = split16(dz1, keep);
    store_rows16(DPO, ok, row, h, mm3(s_ops, SO_WSAT, lane, dz1s, zero));
    tileT_put_f32(ta, j, h, dz1, ok);
    tileT_put_f32(tb, j, h, x, ok);
    wave_sync();
    accW = wgrad(ta, tb, j, h, accW);            // dWsa[n][k] = sum_p dz1[p][n] pooled[p][k]
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

}  // namespace chain
}  // namespace dva

using namespace dva;
using namespace dva::chain;

extern "C" {

int dva_chain_set_prep(const float* Wsa, int32_t ld_sa, const float* Wsb, const float* Wc, int32_t ld_c, void* ops,
                       void* stream) {
  if (!Wsa || !Wsb || !Wc || !ops || ld_sa < D || ld_c < 2 * D) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(set_prep_kernel, dim3(N_SOPS), dim3(64), 0, (hipStream_t)stream, Wsa, ld_sa, Wsb, Wc, ld_c,
                     (uint4*)ops);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_set_fwd(int32_t stage, const float* pooled, const int64_t* ptr, const float* w33, const void* ops,
                      const float* bn_s1, const float* bn_s2, float* u, double* stats, int64_t n_points,
                      void* stream) {
  if (n_points < 0 || stage < 1 || stage > 3) return DVA_ERR_INVALID;
  if (n_points == 0) return DVA_OK;
  if (!pooled || !ptr || !ops || (stage >= 2 && !bn_s1) || (stage == 3 && (!bn_s2 || !u)) || (stage < 3 && !stats))
    return DVA_ERR_INVALID;
  if (n_points * 128 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  const int64_t tiles = (n_points + 31) / 32;
  static const int bpc = tune_int("DVA_SET_FWD_BPC", 2);       // blocks per CU of the grid (read once)
  const int cap = chain_grid(bpc);
  // at least 8 tiles per wavefront: every block pays the 24 KB operand table and the flush (3e5 points: 170 -> 149 us
  // for the three backward stages with half the blocks)
  const int64_t want = (tiles + 31) / 32;
  const dim3 grid((int)(want < cap ? want : cap)), block(256);
  hipStream_t s = (hipStream_t)stream;
  const float *b1 = bn_s1, *b2 = bn_s2;
#define DVA_SET_FWD(ST_)                                                                                          \
  hipLaunchKernelGGL((set_kernel<0, ST_>), grid, block, 0, s, pooled, ptr, w33, (const uint4*)ops, b1, b2,      \
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, u, (float*)nullptr,   \
                     (float*)nullptr, 0, (float*)nullptr, stats, n_points)
  if (stage == 1) DVA_SET_FWD(1);
  else if (stage == 2) DVA_SET_FWD(2);
  else DVA_SET_FWD(3);
#undef DVA_SET_FWD
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_set_bwd(int32_t stage, const float* pooled, const int64_t* ptr, const float* w33, const void* ops,
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
  static const int bpc = tune_int("DVA_SET_BWD_BPC", 2);       // blocks per CU of the grid (read once)
  const int cap = chain_grid(bpc);
  const int64_t want = (tiles + 31) / 32;      // at least 8 tiles per wavefront (see dva_chain_set_fwd)
  const dim3 grid((int)(want < cap ? want : cap)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define DVA_SET_BWD(ST_)                                                                                          \
  hipLaunchKernelGGL((set_kernel<1, ST_>), grid, block, 0, s, pooled, ptr, w33, (const uint4*)ops, bn_s1, bn_s2, \
                     sm_s1, sm_s2, du, (float*)nullptr, dpooled, dW, ld_dw, dw33, stats, n_points)
  if (stage == 1) DVA_SET_BWD(1);
  else if (stage == 2) DVA_SET_BWD(2);
  else DVA_SET_BWD(3);
#undef DVA_SET_BWD
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // extern "C"
