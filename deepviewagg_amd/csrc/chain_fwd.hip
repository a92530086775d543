// Recompute chain, forward side (see chain_common.h for the geometry):
//   dva_chain_prep          weight operand table (bf16, MFMA k-slot order) + fp32 copy of the forward operands (the
//                           kernels fold BatchNorm into them: bf16(0.6 gamma invstd W), chain_common.h)
//   dva_chain_bn_consts     BatchNorm constants of one layer: mean | invstd | gamma | beta | shift of the folded product
//   dva_chain_tile_chunks / _count / _offsets / _build    tile table of a CSR pointer array
//   dva_chain_moments       sum x, sum x x^T of the mapping features -> BatchNorm-1 statistics analytically
//   dva_chain_stats2        statistics of layer 2 + per-point extremum of the layer-2 output (set pooling)
//   dva_chain_pooled        pooled set features from the extrema
//   dva_chain_stats         statistics of layer 5 / 6 (train mode only)
//   dva_chain_attn_fwd      the fully fused view kernel: x_map -> DeepSetFeat -> scores -> softmax over the
//                           point's views -> gather of the value rows -> weighted sum -> gate  (no [V, .] tensor
//                           is read or written besides x_map / the view->point index / the row index)
// Reference maths: modules/multimodal/pooling.py:658-669 (DeepSetFeat.forward), :263-315
// (GroupBimodalCSRPool.forward), core/common_modules/base_modules.py:38-48 (MLP block).
#include "chain_common.h"

namespace dva {
namespace chain {

// ------------------------------------------------------------------------------------------------
// weight operands
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void prep_kernel(const float* __restrict__ W1, const float* __restrict__ W2,
                                                  const float* __restrict__ W5, int ld5,
                                                  const float* __restrict__ W6, const float* __restrict__ Ws, int G,
                                                  uint4* __restrict__ ops) {
  const int op = blockIdx.x, lane = threadIdx.x, j = lane & 31, h = lane >> 5;
  float w[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) w[s] = 0.f;
  if (op == OP_W1) {
#pragma unroll
    for (int s = 0; s < 4; ++s) w[s] = w[s + 4] = W1[j * 8 + 4 * h + s];
  } else if (op == OP_WST) {
    if (h == 0) {
#pragma unroll
      for (int s = 0; s < 4; ++s) w[s] = w[s + 4] = s < G ? Ws[s * D + j] : 0.f;
    }
  } else {
    const int m = op >= OP_WKT ? op - OP_WKT : (op - 1) & 1;      // (the two-block operands below OP_WST start at odd indices)
    const int base = op - m;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int c = chan(8 * m + s, h);
      float v = 0.f;
      if (base == OP_W2) v = W2[j * D + c];
      else if (base == OP_W5) v = W5[j * ld5 + c];
      else if (base == OP_W6) v = W6[j * D + c];
      else if (base == OP_WS) v = j < G ? Ws[j * D + c] : 0.f;
      else if (base == OP_W6T) v = W6[c * D + j];
      else if (base == OP_W5T) v = W5[c * ld5 + j];
      else if (base == OP_W2T) v = W2[c * D + j];
      else if (base == OP_WKT) v = G == D ? Ws[c * D + j] : 0.f;
      w[s] = v;
    }
  }
  ops[op * 64 + lane] = __builtin_bit_cast(uint4, pack8(w));
  if (op < N_OPS32) {       // fp32 copy of the forward operands: source of the BatchNorm-folded variants
    float4* w32 = reinterpret_cast<float4*>(ops + N_OPS * 64) + (op * 64 + lane) * 2;
    w32[0] = make_float4(w[0], w[1], w[2], w[3]);
    w32[1] = make_float4(w[4], w[5], w[6], w[7]);
  }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm bookkeeping of one chain layer (dva_bn_finalize semantics) + the fifth table row: the 0.6-scaled
// shift of the layer's product.  Plain layer: 0.6 (beta - mean G).  Folded layer in training (W, suma given): the
// product is a . bf16(0.6 G W)^T, whose batch mean is bf16(0.6 G W) . mean(a), NOT 0.6 G mean(z) (the rounded
// operand is not G times the rounded W), so the shift is 0.6 beta - bf16(0.6 G W) . mean(a): the folded
// pre-activation keeps the exact batch mean beta (a systematic per-channel offset otherwise).
// ------------------------------------------------------------------------------------------------
__global__ void chain_bn_consts_kernel(const double* __restrict__ sums, double m, float* __restrict__ rmean,
                                       float* __restrict__ rvar, int64_t* __restrict__ nbt,
                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                       float momentum, float eps, int training, const float* __restrict__ W, int ldw,
                                       int K, const double* __restrict__ suma, float* __restrict__ bn) {
  const int c = threadIdx.x;
  if (c >= D) return;
  float mean, var;
  if (training) {
    const double mu = sums[c] / m;
    double v = sums[D + c] / m - mu * mu;
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
  const float inv = rsqrtf(var + eps), g = gamma[c] * inv;
  bn[c] = mean;
  bn[D + c] = inv;
  bn[2 * D + c] = gamma[c];
  bn[3 * D + c] = beta[c];
  float shift = 0.6f * (beta[c] - mean * g);
  if (training && W && suma) {
    const float s = 0.6f * g;
    double acc = 0.0;
    for (int k = 0; k < K; ++k) acc += (double)bf2f(f2bf(W[c * ldw + k] * s)) * (suma[k] / m);
    shift = (float)(0.6 * (double)beta[c] - acc);
  }
  bn[4 * D + c] = shift;
  if (training && nbt && c == 0) *nbt += 1;
}

// ------------------------------------------------------------------------------------------------
// tile table: the points [cp[c], cp[c + 1]) of chunk c are tiled greedily
// ------------------------------------------------------------------------------------------------
template <bool WRITE>
__global__ __launch_bounds__(64) void tile_walk_kernel(const int64_t* __restrict__ ptr,
                                                       const int64_t* __restrict__ cp, int n_chunks,
                                                       int32_t* __restrict__ counts,
                                                       const int64_t* __restrict__ offsets, int2* __restrict__ tiles) {
  // one LANE per chunk: the greedy grouping is a serial walk over the points of a chunk (every step depends on the
  // tile opened before), so the parallelism is across chunks -- 64 independent walks per wavefront instead of one
  // walk whose every step is a cross-lane ballot / readlane chain
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= n_chunks) return;
  int64_t p = cp[c];
  const int64_t p_end = cp[c + 1];
  int count = 0;
  int64_t o = WRITE ? offsets[c] : 0;
  auto emit = [&](int64_t v0, int nv, int frag) {
    if (WRITE) tiles[o] = make_int2((int)v0, nv | (frag << 8));
    ++o;
    ++count;
  };
  if (p < p_end) {
    int64_t s = ptr[p];            // first view of the open tile
    int64_t e_prev = s;            // end of the last point taken into the open tile
    // the walk is a chain of dependent loads (every step needs ptr[p + 1]): the next eight pointers are requested
    // together (round 5: 0.13 -> 0.03 ms per step on the reference-sized S3DIS batch, ~175 points per lane)
    while (p < p_end) {
      int64_t eb[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) eb[i] = ptr[p + 1 + i < p_end ? p + 1 + i : p_end];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (p >= p_end) break;
        const int64_t e = eb[i];
        if (e - s > 32) {
          if (e_prev > s) {          // close the open tile (whole points only): the point starts an empty one
            emit(s, (int)(e_prev - s), 0);
            s = e_prev;
          }
          if (e - s > 32) {          // a point with more than 32 views on an empty tile: fragments of 32 views
            const int64_t n = e - s;
            const int nf = (int)((n + 31) / 32);
            for (int f = 0; f < nf; ++f)
              emit(s + 32 * f, f == nf - 1 ? (int)(n - 32 * f) : 32, f == 0 ? 1 : (f == nf - 1 ? 3 : 2));
            s = e;
          }
        }
        e_prev = e;                  // the point is in the open tile (or was emitted as fragments: s == e)
        ++p;
      }
    }
    if (e_prev > s) emit(s, (int)(e_prev - s), 0);
  }
  if (!WRITE) counts[c] = count;
}

// chunk c covers the views [c * step, (c + 1) * step): cp[c] = first point whose views start at or after c * step
__global__ void tile_chunks_kernel(const int64_t* __restrict__ ptr, int64_t N, int64_t step, int n_chunks,
                                   int64_t* __restrict__ cp) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > n_chunks) return;
  if (c == 0 || c == n_chunks) {
    cp[c] = c == 0 ? 0 : N;
    return;
  }
  const int64_t bound = (int64_t)c * step;
  int64_t lo = 0, hi = N;                       // lower bound over ptr[0 .. N)
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (ptr[mid] < bound) lo = mid + 1;
    else hi = mid;
  }
  cp[c] = lo;
}

// exclusive scan of the per-chunk tile counts: block b owns the entries [1024 b, 1024 b + 1024); it first sums
// everything before its segment itself (coalesced, a few thousand integers: no second launch, no scratch buffer)
__global__ __launch_bounds__(1024) void tile_offsets_kernel(const int32_t* __restrict__ counts, int n_chunks,
                                                            int64_t* __restrict__ offsets,
                                                            int32_t* __restrict__ n_tiles) {
  __shared__ int s_sum[1024];
  __shared__ long long s_prefix;
  const int t = threadIdx.x, base = blockIdx.x * 1024;
  long long acc = 0;
  for (int i = t; i < base; i += 1024) acc += counts[i];
  if (t == 0) s_prefix = 0;
  __syncthreads();
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((t & 63) == 0 && acc) atomicAdd((unsigned long long*)&s_prefix, (unsigned long long)acc);
  const int v = base + t < n_chunks ? counts[base + t] : 0;
  s_sum[t] = v;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int u = t >= off ? s_sum[t - off] : 0;
    __syncthreads();
    s_sum[t] += u;
    __syncthreads();
  }
  if (base + t < n_chunks) offsets[base + t] = s_prefix + s_sum[t] - v;
  if (blockIdx.x == gridDim.x - 1 && t == 1023) n_tiles[0] = (int32_t)(s_prefix + s_sum[1023]);
}

// ------------------------------------------------------------------------------------------------
// first and second moments of the mapping features (fp64 sums): BatchNorm-1 statistics are a function of
// them (z1 = W1 x is linear), and so is the Q term of the first layer's weight gradient
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void moments_kernel(const float* __restrict__ x_map, int64_t V,
                                                      double* __restrict__ mom /* 8 + 36 */) {
  __shared__ float s_red[4][44];      // one slot per wavefront, summed in a fixed order (see flush_stats)
  const __amdgpu_buffer_rsrc_t X = make_rsrc(x_map, (uint64_t)V * 32);
  float s1[8], s2[36];
#pragma unroll
  for (int i = 0; i < 8; ++i) s1[i] = 0.f;
#pragma unroll
  for (int i = 0; i < 36; ++i) s2[i] = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += stride) {
    const float4 a = as_f4(ld128(X, (uint32_t)(v * 32))), b = as_f4(ld128(X, (uint32_t)(v * 32 + 16)));
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    int k = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s1[i] += x[i];
#pragma unroll
      for (int jj = i; jj < 8; ++jj, ++k) s2[k] = __builtin_fmaf(x[i], x[jj], s2[k]);
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 44; ++i) {
    float v = i < 8 ? s1[i] : s2[i - 8];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off);
    if (lane == 0) s_red[wv][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 44)
    atomicAdd(&mom[threadIdx.x], (((double)s_red[0][threadIdx.x] + (double)s_red[1][threadIdx.x]) +
                                  (double)s_red[2][threadIdx.x]) + (double)s_red[3][threadIdx.x]);
}

// statistics of z1 = bf16(W1) x from the moments: stats = sum z1 | sum z1^2 (the form dva_bn_finalize takes)
// exact_w1: the fp32 chain (chain_f32.hip) multiplies with W1 itself (hi + lo), the bf16 chain with bf16(W1)
__global__ void stats1_kernel(const double* __restrict__ mom, const float* __restrict__ W1, int exact_w1,
                              double* __restrict__ stats) {
  const int c = threadIdx.x;
  if (c >= D) return;
  double w[8];
  for (int f = 0; f < 8; ++f) w[f] = exact_w1 ? (double)W1[c * 8 + f] : (double)bf2f(f2bf(W1[c * 8 + f]));
  double s = 0, q = 0;
  int k = 0;
  for (int i = 0; i < 8; ++i) {
    s += w[i] * mom[i];
    for (int jj = i; jj < 8; ++jj, ++k) q += (i == jj ? 1.0 : 2.0) * w[i] * w[jj] * mom[8 + k];
  }
  stats[c] = s;
  stats[D + c] = q;
}

// ------------------------------------------------------------------------------------------------
// layer 2: statistics + per-point extremum.  a2 = leaky(G2 z2 + B2) is monotone in z2 per channel with the
// sign of gamma2, so the set pooling max_v a2 is taken on sign(gamma2) * z2 BEFORE the statistics exist.
// ------------------------------------------------------------------------------------------------
constexpr int TZ = 36;  // fp32 tile row stride (floats)
__global__ __launch_bounds__(256, 3) void stats2_kernel(
    const float* __restrict__ x_map, const int32_t* __restrict__ vp, const int2* __restrict__ tiles,
    const int32_t* __restrict__ n_tiles_dev, const uint4* __restrict__ ops, const float* __restrict__ bn1,
    const float* __restrict__ gamma2, double* __restrict__ stats, float* __restrict__ zstar,
    int32_t* __restrict__ arg, int64_t V) {
  __shared__ __attribute__((aligned(16))) float s_tab[TAB_FLOATS];
  __shared__ __attribute__((aligned(16))) float s_tile[4][32 * TZ];
  __shared__ float s_red[STATS_RED_FLOATS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  stage_tab(s_tab, bn1, nullptr);
  __syncthreads();
  const bf16x8 w1 = load_op_fold(ops, OP_W1, lane, bn1);     // layer 1: BatchNorm inside the product
  const WOp w2 = load_wop(ops, OP_W2, lane);
  const uint32_t flip = gamma2[j] < 0.f ? 0x80000000u : 0u;   // walker lane c = j
  const __amdgpu_buffer_rsrc_t X = make_rsrc(x_map, (uint64_t)V * 32), P = make_rsrc(vp, (uint64_t)V * 4);
  float st[3][16];      // sum z2 | sum z2^2 | sum a1
#pragma unroll
  for (int r = 0; r < 16; ++r) st[0][r] = st[1][r] = st[2][r] = 0.f;
  float run_m = -INFINITY;
  int run_a = -1;
  float* tz = s_tile[wv];
  const int n_tiles = n_tiles_dev[0];
  int ta, tb;
  wave_tile_range(tiles, n_tiles, ta, tb);
  struct Pre {
    TileInfo ti;
    float4 x;
    int vpj;
  };
  run_tiles<Pre>(tiles, ta, tb, [&](const TileInfo& ti, int t) {
    Pre p;
    p.ti = ti;
    const bool ok = j < p.ti.nv;
    p.x = as_f4(ld128(X, ok ? (uint32_t)(p.ti.v0 + j) * 32u + 16u * h : OOB));
    p.vpj = (int)ld32(P, ok ? (uint32_t)(p.ti.v0 + j) * 4u : OOB);
    return p;
  }, [&](const Pre& p) {
    const int nv = p.ti.nv;
    const uint32_t keep = j < nv ? 0xffffffffu : 0u;
    const f32x16 zero = {0};
    const f32x16 t1 = CH_MFMA(w1, pack_x(p.x), bias_acc(s_tab, T_B6, h));
    bf16x8 a1[2];
    act_fold<false>(t1, keep, a1, st[2]);
    const f32x16 z2 = mm32(w2, a1, zero);
    if (keep) {      // lanes without a view hold an unmasked a1: their z2 stays out of the sums (and of the walk below)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[0][r] += z2[r];
        st[1][r] = __builtin_fmaf(z2[r], z2[r], st[1][r]);
      }
    }
    // tile -> LDS [view][channel]; lane c then walks the views of its channel in order
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(tz + j * TZ + 8 * q + 4 * h) =
          make_float4(z2[4 * q], z2[4 * q + 1], z2[4 * q + 2], z2[4 * q + 3]);
    // last view of its point, and the point is complete in this tile
    const int nxt = shfl(p.vpj, lane + 1);
    const bool is_end = j < nv && (j == nv - 1 || nxt != p.vpj);
    uint32_t endmask = (uint32_t)__ballot(is_end);
    if (p.ti.frag == 1 || p.ti.frag == 2) endmask = 0;
    wave_sync();
    if (nv == 32 && p.ti.frag == 0 && endmask == 0x80000000u) {
      // ---- one whole point of 32 views (the headline shape): the two half-waves take 16 views each, a
      // comparison tree instead of the serial walk; ties go to the earlier view at every node
      float m[16];
      int a[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        m[i] = __uint_as_float(__float_as_uint(tz[(16 * h + i) * TZ + j]) ^ flip);
        a[i] = i;
      }
#pragma unroll
      for (int w = 1; w < 16; w <<= 1) {
#pragma unroll
        for (int i = 0; i < 16; i += 2 * w) {
          const bool gt = m[i + w] > m[i];
          m[i] = gt ? m[i + w] : m[i];
          a[i] = gt ? a[i + w] : a[i];
        }
      }
      uint32_t m0 = __float_as_uint(m[0]), m1 = m0, a0 = (uint32_t)(p.ti.v0 + 16 * h + a[0]), a1 = a0;
      swap_halves(m0, m1);           // m1 / a1 in the h = 0 lanes = result of the h = 1 lane (views 16 .. 31)
      swap_halves(a0, a1);
      if (h == 0) {
        const bool gt = __uint_as_float(m1) > m[0];
        const float mm = gt ? __uint_as_float(m1) : m[0];
        const int aa = gt ? (int)a1 : p.ti.v0 + a[0];
        const int pt = __builtin_amdgcn_readfirstlane(p.vpj);
        zstar[(int64_t)pt * D + j] = __uint_as_float(__float_as_uint(mm) ^ flip);
        arg[(int64_t)pt * D + j] = aa;
      }
      wave_sync();
      return;
    }
    float xv[32];
#pragma unroll
    for (int v = 0; v < 32; ++v) xv[v] = __uint_as_float(__float_as_uint(tz[v * TZ + j]) ^ flip);
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
  flush_stats<3>(st, stats, s_red);
}

// pooled[p][c] = leaky(G2 z* + B2) for seen points, 0 for unseen ones (segment_csr max convention)
__global__ __launch_bounds__(256) void pooled_kernel(const float* __restrict__ zstar, const float* __restrict__ bn2,
                                                     const int64_t* __restrict__ ptr, float* __restrict__ pooled,
                                                     int64_t N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one float4 per thread
  if (i >= N * 8) return;
  const int64_t p = i >> 3;
  const int c0 = (int)(i & 7) * 4;
  float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ptr[p + 1] > ptr[p]) {
    const float4 z = *reinterpret_cast<const float4*>(zstar + p * D + c0);
    const float zz[4] = {z.x, z.y, z.z, z.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = c0 + e;
      const float g = bn2[2 * D + c] * bn2[D + c];
      o[e] = leaky(__builtin_fmaf(zz[e], g, bn2[3 * D + c] - bn2[c] * g));
    }
    out = make_float4(o[0], o[1], o[2], o[3]);
  }
  *reinterpret_cast<float4*>(pooled + p * D + c0) = out;
}

// ------------------------------------------------------------------------------------------------
// statistics of layer 5 (z5 = W5a a2 + u[point]) or layer 6 (train mode)
// ------------------------------------------------------------------------------------------------
// A2 (round 6, the stored-a2 hybrid: DVA_CHAIN_A2=1): 0 = everything from x_map; 1 (L == 5) = the pass also WRITES the
// layer-2 activation a2 as one bf16 row per view ([V, 32] in accumulator order: the 32 bytes a lane holds are
// contiguous, position 16 h + r = channel chan(r, h)) -- exactly the packed B operand layer 5 consumes, so a pass that
// starts from the row evaluates the same numbers; 2 (L == 6) = the pass READS that row instead of x_map and skips
// layers 1 and 2 (the hi | lo split, three MFMAs, two bias / activation / pack sequences per view).
template <int L, int A2 = 0>
__global__ __launch_bounds__(256, L == 5 ? 4 : 3) void stats_mid_kernel(
    const float* __restrict__ x_map, const int32_t* __restrict__ vp, const float* __restrict__ u,
    const int2* __restrict__ tiles, const int32_t* __restrict__ n_tiles_dev, const uint4* __restrict__ ops,
    const float* __restrict__ bn1, const float* __restrict__ bn2, const float* __restrict__ bn5,
    double* __restrict__ stats, int64_t V, int64_t N, bf16_t* __restrict__ a2buf) {
  static_assert(A2 == 0 || (A2 == 1 && L == 5) || (A2 == 2 && L == 6), "a2 is written by stats5 and read by stats6");
  __shared__ __attribute__((aligned(16))) float s_tab[3][TAB_FLOATS];
  __shared__ float s_red[STATS_RED_FLOATS];
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  if (A2 != 2) {
    stage_tab(s_tab[0], bn1, nullptr);
    stage_tab(s_tab[1], bn2, nullptr);
  }
  if (L == 6) stage_tab(s_tab[2], bn5, nullptr);
  __syncthreads();
  bf16x8 w1;
  WOp w2;
  if (A2 != 2) {
    w1 = load_op_fold(ops, OP_W1, lane, bn1);     // layers 1, 2: BatchNorm inside the product
    w2 = load_wop_fold(ops, OP_W2, lane, bn2);
  }
  const WOp w5 = load_wop(ops, OP_W5, lane);
  WOp w6;
  if (L == 6) w6 = load_wop(ops, OP_W6, lane);
  const __amdgpu_buffer_rsrc_t X = make_rsrc(x_map, A2 == 2 ? 0 : (uint64_t)V * 32), P = make_rsrc(vp, (uint64_t)V * 4),
                               U = make_rsrc(u, (uint64_t)N * 128), A = make_rsrc(a2buf, A2 ? (uint64_t)V * 64 : 0);
  float st[3][16];      // sum z | sum z^2 | (L == 6) sum a5
#pragma unroll
  for (int r = 0; r < 16; ++r) st[0][r] = st[1][r] = st[2][r] = 0.f;
  const int n_tiles = n_tiles_dev[0];
  int ta, tb;
  wave_tile_range(tiles, n_tiles, ta, tb);
  struct Pre {
    TileInfo ti;
    float4 x;
    u32x4 alo, ahi;
    int vpj;
  };
  run_tiles<Pre>(tiles, ta, tb, [&](const TileInfo& ti, int t) {
    Pre p;
    p.ti = ti;
    const bool ok = j < p.ti.nv;
    const uint32_t view = (uint32_t)(p.ti.v0 + j);
    if (A2 == 2) {
      p.alo = ld128(A, ok ? view * 64u + 32u * h : OOB);
      p.ahi = ld128(A, ok ? view * 64u + 32u * h + 16u : OOB);
    } else {
      p.x = as_f4(ld128(X, ok ? view * 32u + 16u * h : OOB));
    }
    p.vpj = (int)ld32(P, ok ? view * 4u : OOB);
    return p;
  }, [&](const Pre& p) {
    const bool ok = j < p.ti.nv;
    const uint32_t keep = ok ? 0xffffffffu : 0u;
    f32x16 uacc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = as_f4(ld128(U, ok ? (uint32_t)p.vpj * 128u + (8u * q + 4u * h) * 4u : OOB));
      uacc[4 * q] = v.x; uacc[4 * q + 1] = v.y; uacc[4 * q + 2] = v.z; uacc[4 * q + 3] = v.w;
    }
    const f32x16 zero = {0};
    bf16x8 a2[2];
    if (A2 == 2) {
      a2[0] = __builtin_bit_cast(bf16x8, p.alo);      // lanes without a view loaded zeros (OOB): the masked operand
      a2[1] = __builtin_bit_cast(bf16x8, p.ahi);
    } else {
      const f32x16 t1 = CH_MFMA(w1, pack_x(p.x), bias_acc(s_tab[0], T_B6, h));
      bf16x8 a1[2];
      act_fold<false>(t1, keep, a1);
      const f32x16 t2 = mm32(w2, a1, bias_acc(s_tab[1], T_B6, h));
      act_fold<false>(t2, keep, a2);         // (the a2 row of a lane without a view is never stored: `ok` below)
      if (A2 == 1) {
        const uint32_t off = ok ? (uint32_t)(p.ti.v0 + j) * 64u + 32u * h : OOB;
        st128(A, off, __builtin_bit_cast(u32x4, a2[0]));
        st128(A, ok ? off + 16u : OOB, __builtin_bit_cast(u32x4, a2[1]));
      }
    }
    f32x16 z = mm32(w5, a2, uacc);
    if (L == 6) {
      bf16x8 a5[2];
      act_pack<false>(z, s_tab[2], h, keep, a5, T_G6, T_B6, st[2]);
      z = mm32(w6, a5, zero);
    }
    if (keep) {      // unmasked operands: the sums take the lanes that own a view
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[0][r] += z[r];
        st[1][r] = __builtin_fmaf(z[r], z[r], st[1][r]);
      }
    }
  });
  flush_stats<3>(st, stats, s_red);      // L == 5: the third row stays zero
}

// ------------------------------------------------------------------------------------------------
// key layer of QKVBimodalCSRPool (reference modules/multimodal/pooling.py:454-547: keys = K(E_map(x_map)), a Linear
// 32 -> nc_qk G = 32 behind DeepSetFeat): the chain evaluated exactly as the fused view kernel evaluates it (layers 1, 2, 6
// with BatchNorm folded into the operand), then one more 32 x 32 product with the rows of W_k (operand OP_WS prepared with
// G = 32) + bias, written as ONE bf16 row per view in accumulator order (position 16 h + r = channel chan(r, h): the 32
// bytes a lane holds are contiguous) -- the layout in which the backward hands gradient rows between its passes, and in
// which dva_chain_score_stats / dva_chain_bwd_layer(6) take d keys back (G = 32).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void keys_kernel(
    const float* __restrict__ x_map, const int32_t* __restrict__ vp, const float* __restrict__ u,
    const int2* __restrict__ tiles, const int32_t* __restrict__ n_tiles_dev, const uint4* __restrict__ ops,
    const float* __restrict__ bn1, const float* __restrict__ bn2, const float* __restrict__ bn5,
    const float* __restrict__ bn6, const float* __restrict__ bk, bf16_t* __restrict__ keys, int64_t V, int64_t N,
    const float* __restrict__ qp, float* __restrict__ compat, int G, float qscale) {
  __shared__ __attribute__((aligned(16))) float s_tab[4][2 * D];      // G | B rows only
  __shared__ __attribute__((aligned(16))) uint4 s_ops[OP_W6T * 64];   // forward operands only
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < OP_W6T * 64; i += blockDim.x) s_ops[i] = ops[i];
  stage_tab_fwd(s_tab[0], bn1);
  stage_tab_fwd(s_tab[1], bn2);
  stage_tab_fwd(s_tab[2], bn5);
  stage_tab_fwd(s_tab[3], bn6);
  __syncthreads();
  fold_ops(s_ops, OP_W1, ops, OP_W1, 1, bn1);
  fold_ops(s_ops, OP_W2, ops, OP_W2, 2, bn2);
  fold_ops(s_ops, OP_W6, ops, OP_W6, 2, bn6);
  __syncthreads();
  const __amdgpu_buffer_rsrc_t X = make_rsrc(x_map, (uint64_t)V * 32), P = make_rsrc(vp, (uint64_t)V * 4),
                               U = make_rsrc(u, (uint64_t)N * 128), KO = make_rsrc(keys, (uint64_t)V * 64),
                               QP = make_rsrc(qp, qp ? (uint64_t)N * 128 : 0),
                               CO = make_rsrc(compat, compat ? (uint64_t)V * 16 : 0);
  f32x16 kb;            // the bias of this lane's 16 key channels
#pragma unroll
  for (int r = 0; r < 16; ++r) kb[r] = bk[chan(r, h)];
  const int n_tiles = n_tiles_dev[0];
  int ta, tb;
  wave_tile_range(tiles, n_tiles, ta, tb);
  struct Pre {
    TileInfo ti;
    float4 x;
    int vpj;
  };
  run_tiles<Pre>(tiles, ta, tb, [&](const TileInfo& ti, int t) {
    Pre p;
    p.ti = ti;
    const bool ok = j < p.ti.nv;
    p.x = as_f4(ld128(X, ok ? (uint32_t)(p.ti.v0 + j) * 32u + 16u * h : OOB));
    p.vpj = (int)ld32(P, ok ? (uint32_t)(p.ti.v0 + j) * 4u : OOB);
    return p;
  }, [&](const Pre& p) {
    const bool ok = j < p.ti.nv;
    f32x16 uacc;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const float4 v = as_f4(ld128(U, ok ? (uint32_t)p.vpj * 128u + (8u * qq + 4u * h) * 4u : OOB));
      uacc[4 * qq] = v.x; uacc[4 * qq + 1] = v.y; uacc[4 * qq + 2] = v.z; uacc[4 * qq + 3] = v.w;
    }
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
    z = mm32_lds(s_ops, OP_WS, lane, a2, kb);
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = z[r];
    const uint32_t off = ok ? (uint32_t)(p.ti.v0 + j) * 64u + 32u * h : OOB;
    const bf16x8 k0 = pack8(&t[0]), k1 = pack8(&t[8]);
    st128(KO, off, __builtin_bit_cast(u32x4, k0));
    st128(KO, ok ? off + 16u : OOB, __builtin_bit_cast(u32x4, k1));
    if (qp) {
      // compatibilities (round 4: no second pass over the key rows): the lane's 16 positions are four quads of one group
      // each (group of position 16 h + r = (r >> 2) G / 4); products of the ROUNDED key (what the backward reads back)
      // with the point's query row, the two half-waves added, [V][4] fp32 out (unused groups 0)
      float ka[8], kc[8], kk[16], s4[4];
      unpack8(k0, ka);
      unpack8(k1, kc);
#pragma unroll
      for (int r = 0; r < 8; ++r) { kk[r] = ka[r]; kk[8 + r] = kc[r]; }
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const float4 q = as_f4(ld128(QP, ok ? (uint32_t)p.vpj * 128u + 64u * h + 16u * qq : OOB));
        s4[qq] = __builtin_fmaf(kk[4 * qq + 3], q.w, __builtin_fmaf(kk[4 * qq + 2], q.z,
                 __builtin_fmaf(kk[4 * qq + 1], q.y, kk[4 * qq] * q.x)));
      }
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        uint32_t a_ = __float_as_uint(s4[qq]), b_ = a_;
        swap_halves(a_, b_);                       // the other half-wave's share of the same view
        s4[qq] += __uint_as_float(h ? a_ : b_);
      }
      float4 c;
      if (G == 4) c = make_float4(s4[0], s4[1], s4[2], s4[3]);
      else if (G == 2) c = make_float4(s4[0] + s4[1], s4[2] + s4[3], 0.f, 0.f);
      else c = make_float4((s4[0] + s4[1]) + (s4[2] + s4[3]), 0.f, 0.f, 0.f);
      c.x *= qscale; c.y *= qscale; c.z *= qscale; c.w *= qscale;
      st128(CO, ok && h == 0 ? (uint32_t)(p.ti.v0 + j) * 16u : OOB, as_u4(c.x, c.y, c.z, c.w));
    }
  });
}

// ------------------------------------------------------------------------------------------------
// the fused view kernel
// ------------------------------------------------------------------------------------------------
// Team layout of the value rows: LPR = C / 8 lanes cover one bf16 row with 16-byte loads, ROWS = 64 / LPR row
// slots per wavefront, slot s handles the KV = 32 / ROWS consecutive views [s KV, (s + 1) KV) of the tile.
// Single-point tiles (the 32-views-per-point headline shape, and the fragments of long points): per-lane
// accumulators + a butterfly over the slots.  Tiles with several points: every slot accumulates runs of views
// of one point; a point that lies inside one slot is stored directly, a point split over slots is summed in a
// [ROWS][C] LDS buffer keyed by the slot it starts in (each slot starts at most one split point) and stored by
// the slot it ends in.
// OCC = waves per SIMD the register budget is cut for: 4 pays when nearly every tile is one point (the single-point
// path fits 128 VGPRs); the several-points path needs the 168 of OCC = 3.
// KEYS (round 4, QKVBimodalCSRPool): the last layer is the KEY layer (operand OP_WS prepared with G = 32, `bs` = its bias
// [32]); the scores are the compatibilities scale * <key, query of the point> per group (queries `qp` fp32 [N][32] in
// position order); the bf16 key rows go to `keys_out` when a backward follows (dQ needs them).
template <int LPR, int G, int OCC, bool KEYS = false>
__global__ __launch_bounds__(256, OCC) void attn_fwd_kernel(
    const float* __restrict__ x_map, const int32_t* __restrict__ vp, const float* __restrict__ u,
    const int2* __restrict__ tiles, const int32_t* __restrict__ n_tiles_dev, const uint4* __restrict__ ops,
    const float* __restrict__ bn1, const float* __restrict__ bn2, const float* __restrict__ bn5,
    const float* __restrict__ bn6, const float* __restrict__ bs, const bf16_t* __restrict__ rows,
    const int32_t* __restrict__ row_idx, const int64_t* __restrict__ ptr, const float* __restrict__ gw,
    const float* __restrict__ gb, bf16_t* __restrict__ out, float* __restrict__ scores_out, int scaling, float eps,
    int64_t V, int64_t N, int64_t R, const float* __restrict__ qp = nullptr, float qscale = 0.f,
    bf16_t* __restrict__ keys_out = nullptr) {
  constexpr int C = LPR * 8, ROWS = 64 / LPR, KV = 32 / ROWS;
  constexpr int KB = KV < 8 ? KV : 8, NB = KV / KB;
  constexpr int NE = G == 1 ? 1 : 2;           // score values per lane on the softmax side
  constexpr bool RS = LPR == 8;                // reduce-scatter epilogue of single-point tiles (C = 64)
  static_assert(LPR % G == 0, "whole 16-byte lanes per channel group");
  __shared__ __attribute__((aligned(16))) float s_tab[4][2 * D];      // G | B rows only
  __shared__ __attribute__((aligned(16))) uint4 s_ops[OP_W6T * 64];   // forward operands only
  __shared__ __attribute__((aligned(16))) float s_ev[4][4 * 32];     // exp(.) per [group][view]
  __shared__ __attribute__((aligned(16))) float s_sc[4][4 * 32];     // gate / (sum + eps) per [group][local point]
  __shared__ __attribute__((aligned(16))) int s_pid[4][32], s_ri[4][32];
  __shared__ __attribute__((aligned(16))) float s_alpha[4][4], s_scg[4][4];
  // several points per tile.  C <= 128 (round 4): the gathered rows of the tile staged in LDS ([view][C] bf16, row
  // pitch C * 2 + 32 bytes), lane l owns the C / 64 channels from l * C / 64 and walks the views in order -- the point
  // boundaries are wave-uniform (a ballot), so the walk is straight-line code with a scalar-tested flush per view.
  // Wider rows: one private row [C] per row slot for the partial sums of points split over slots.
  constexpr bool STAGED = C <= 128 && C >= 64;
  constexpr int CPL = STAGED ? C / 64 : 1, PITCH = C + 16;     // PITCH in bf16 elements
  __shared__ __attribute__((aligned(16))) float s_acc[4][STAGED ? 4 : ROWS * C];
  __shared__ __attribute__((aligned(16))) bf16_t s_rows[4][STAGED ? 32 * PITCH : 8];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < OP_W6T * 64; i += blockDim.x) s_ops[i] = ops[i];
  stage_tab_fwd(s_tab[0], bn1);
  stage_tab_fwd(s_tab[1], bn2);
  stage_tab_fwd(s_tab[2], bn5);
  stage_tab_fwd(s_tab[3], bn6);
  __syncthreads();
  fold_ops(s_ops, OP_W1, ops, OP_W1, 1, bn1);  // layers 1, 2, 6: BatchNorm inside the product (layer 5 adds u[point] first)
  fold_ops(s_ops, OP_W2, ops, OP_W2, 2, bn2);
  fold_ops(s_ops, OP_W6, ops, OP_W6, 2, bn6);
  __syncthreads();
  const __amdgpu_buffer_rsrc_t X = make_rsrc(x_map, (uint64_t)V * 32), P = make_rsrc(vp, (uint64_t)V * 4),
                               U = make_rsrc(u, (uint64_t)N * 128), RI = make_rsrc(row_idx, (uint64_t)V * 4),
                               RW = make_rsrc(rows, (uint64_t)R * C * 2), O = make_rsrc(out, (uint64_t)N * C * 2),
                               SC = make_rsrc(scores_out, scores_out ? (uint64_t)V * 16 : 0),
                               QP = make_rsrc(qp, KEYS ? (uint64_t)N * 128 : 0),
                               KO = make_rsrc(keys_out, KEYS && keys_out ? (uint64_t)V * 64 : 0);
  f32x16 kb = {0};        // KEYS: the bias of this lane's 16 key channels
  if constexpr (KEYS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) kb[r] = bs[chan(r, h)];
  }
  // softmax side: lane (j, h) owns the groups gl[e]
  const bool s_active = G == 4 || h == 0;
  int gl[NE];
  float bias[NE], gwl[NE], gbl[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    gl[e] = (G == 4 ? 2 * h : 0) + e;
    if (gl[e] >= G) gl[e] = G - 1;
    bias[e] = KEYS ? 0.f : bs[gl[e]];
    gwl[e] = gw ? gw[gl[e]] : 0.f;
    gbl[e] = gw ? gb[gl[e]] : 0.f;
  }
  // team side
  const int slot = lane / LPR, q = lane % LPR, sv0 = slot * KV;
  const int tg = q / (LPR / G);
  // reduce-scatter epilogue (LPR == 8): the channel of the chunk this lane ends up with
  const int rs_ch = 4 * (1 - ((lane >> 5) & 1)) + 2 * (1 - ((lane >> 4) & 1)) + (1 - ((lane >> 3) & 1));
  float* ev_t = s_ev[wv];
  float* sc_t = s_sc[wv];
  int* pid_t = s_pid[wv];
  int* ri_t = s_ri[wv];
  float* acc_t = s_acc[wv];
  // carry of a long point (fragments)
  float run_m[NE], run_s[NE], run_acc[8];
#pragma unroll
  for (int e = 0; e < NE; ++e) { run_m[e] = -INFINITY; run_s[e] = 0.f; }
#pragma unroll
  for (int k = 0; k < 8; ++k) run_acc[k] = 0.f;

  const int n_tiles = n_tiles_dev[0];
  int ta, tb;
  wave_tile_range(tiles, n_tiles, ta, tb);
  struct Pre {
    TileInfo ti;
    float4 x;
    int vpj, rij;
  };
  run_tiles<Pre>(tiles, ta, tb, [&](const TileInfo& ti, int t) {
    Pre p;
    p.ti = ti;
    const bool ok = j < p.ti.nv;
    p.x = as_f4(ld128(X, ok ? (uint32_t)(p.ti.v0 + j) * 32u + 16u * h : OOB));
    p.vpj = (int)ld32(P, ok ? (uint32_t)(p.ti.v0 + j) * 4u : OOB);
    p.rij = (int)ld32(RI, ok ? (uint32_t)(p.ti.v0 + j) * 4u : OOB);
    return p;
  }, [&](const Pre& p) {
    const int nv = p.ti.nv, frag = p.ti.frag;
    const bool ok = j < nv;
    // ---- row indices to the team lanes, value rows in flight before the chain starts
    if (h == 0) ri_t[j] = p.rij;     // lanes without a view read 0 (row 0, weight 0)
    f32x16 uacc;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const float4 v = as_f4(ld128(U, ok ? (uint32_t)p.vpj * 128u + (8u * qq + 4u * h) * 4u : OOB));
      uacc[4 * qq] = v.x; uacc[4 * qq + 1] = v.y; uacc[4 * qq + 2] = v.z; uacc[4 * qq + 3] = v.w;
    }
    wave_sync();
    u32x4 xr[2][KB];
    auto issue_rows = [&](int b) {
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) {
        const uint32_t ri = (uint32_t)ri_t[sv0 + b * KB + kk];
        xr[b & 1][kk] = ld128(RW, ri * (uint32_t)(C * 2) + (uint32_t)q * 16u);
      }
    };
    issue_rows(0);
    // ---- DeepSetFeat chain -> scores
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
    float c[NE];
    if constexpr (KEYS) {
      z = mm32_lds(s_ops, OP_WS, lane, a2, kb);
      float t[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = z[r];
      const bf16x8 k0 = pack8(&t[0]), k1 = pack8(&t[8]);
      const uint32_t koff = ok ? (uint32_t)(p.ti.v0 + j) * 64u + 32u * h : OOB;
      st128(KO, koff, __builtin_bit_cast(u32x4, k0));          // zero-sized buffer without a backward: dropped
      st128(KO, ok ? koff + 16u : OOB, __builtin_bit_cast(u32x4, k1));
      float ka[8], kc[8], s4[4];
      unpack8(k0, ka);
      unpack8(k1, kc);
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const float4 q4 = as_f4(ld128(QP, ok ? (uint32_t)p.vpj * 128u + 64u * h + 16u * qq : OOB));
        const float* kq = qq < 2 ? &ka[4 * qq] : &kc[4 * (qq - 2)];
        s4[qq] = __builtin_fmaf(kq[3], q4.w, __builtin_fmaf(kq[2], q4.z, __builtin_fmaf(kq[1], q4.y, kq[0] * q4.x)));
      }
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        uint32_t a_ = __float_as_uint(s4[qq]), b_ = a_;
        swap_halves(a_, b_);
        s4[qq] += __uint_as_float(h ? a_ : b_);
      }
      if constexpr (G == 4) {
        c[0] = (h ? s4[2] : s4[0]) * qscale;
        c[NE - 1] = (h ? s4[3] : s4[1]) * qscale;
      } else if constexpr (G == 2) {
        c[0] = (s4[0] + s4[1]) * qscale;
        c[NE - 1] = (s4[2] + s4[3]) * qscale;
      } else {
        c[0] = ((s4[0] + s4[1]) + (s4[2] + s4[3])) * qscale;
      }
    } else {
    z = mm32_lds(s_ops, OP_WS, lane, a2, zero);
    if constexpr (G == 4) {
      uint32_t A0 = __float_as_uint(z[0]), A2 = __float_as_uint(z[2]);
      uint32_t A1 = __float_as_uint(z[1]), A3 = __float_as_uint(z[3]);
      swap_halves(A0, A2);   // A0: h = 0 -> group 0, h = 1 -> group 2
      swap_halves(A1, A3);
      c[0] = __uint_as_float(A0) + bias[0];
      c[1] = __uint_as_float(A1) + bias[1];
    } else {
#pragma unroll
      for (int e = 0; e < NE; ++e) c[e] = z[e] + bias[e];
    }
    }
    // training: the scores [V, 4] stay for the attention backward (16 bytes per view instead of a chain evaluation);
    // a null pointer makes a zero-sized buffer: the stores are dropped
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
    // the value-row accumulation of one lane (weights from the LDS table of the tile)
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    auto fma_row = [&](const u32x4& r, float w) {
      const uint32_t rw[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[2 * i] = __builtin_fmaf(w, __uint_as_float(rw[i] << 16), acc[2 * i]);
        acc[2 * i + 1] = __builtin_fmaf(w, __uint_as_float(rw[i] & 0xffff0000u), acc[2 * i + 1]);
      }
    };
    if (single) {
      // ================= one point in the tile (32-views-per-point scenes, fragments of long points) ==========
      int n_pt = nv;
      if (frag != 0) n_pt = (int)(ptr[vp0 + 1] - ptr[vp0]);
      const float isn = (scaling ? __builtin_amdgcn_rsqf((float)n_pt) : 1.f) * 1.44269504f;   // x log2(e)
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        float m = half_max(ok ? c[e] : -INFINITY);
        float alpha = 0.f;
        if (frag != 0) {           // online softmax across the fragments
          const float m_new = vmaxf(run_m[e], m);
          alpha = __builtin_amdgcn_exp2f((run_m[e] - m_new) * isn);   // first fragment: run_m = -inf -> 0
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
      wave_sync();
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (b + 1 < NB) issue_rows(b + 1);
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) fma_row(xr[b & 1][kk], ev_t[tg * 32 + sv0 + b * KB + kk]);
      }
      const float sc = s_scg[wv][tg], al = s_alpha[wv][tg];
      const bool done = frag == 0 || frag == 3;
      if (RS) {
        // reduce-scatter over the 8 row slots: each level halves the channels a lane carries (the row swaps
        // exchange and the add reduces); the lane ends with channel rs_ch of its 8-channel chunk
        float v4[4], v2[2], v1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          uint32_t x = __float_as_uint(acc[4 + i]), y = __float_as_uint(acc[i]);
          swap_halves(x, y);                                   // lanes [32, 64) of x <-> lanes [0, 32) of y
          v4[i] = __uint_as_float(x) + __uint_as_float(y);     // h = 0: channel 4 + i, h = 1: channel i
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v4[2 + i]), __float_as_uint(v4[i]), false, false);
          v2[i] = __uint_as_float(r.x) + __uint_as_float(r.y);  // even row: sub-channel 2 + i, odd row: i
        }
        {
          const bool up = (lane >> 3) & 1;
          const float mine = up ? v2[0] : v2[1], send = up ? v2[1] : v2[0];
          v1 = mine + __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(send), 0x128,
                                                                            0xf, 0xf, false));   // row_ror:8
        }
        if (frag != 0) {
          v1 = __builtin_fmaf(run_acc[0], al, v1);
          run_acc[0] = frag == 3 ? 0.f : v1;
        }
        if (done)
          __builtin_amdgcn_raw_buffer_store_b16((short)f2bf(v1 * sc), O,
                                                (int)((uint32_t)vp0 * (uint32_t)(C * 2) + (uint32_t)(q * 8 + rs_ch) * 2u),
                                                0, 0);
      } else {
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] += __shfl_xor(acc[i], off);
        }
        if (frag != 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_fmaf(run_acc[i], al, acc[i]);
            run_acc[i] = frag == 3 ? 0.f : acc[i];
          }
        }
        if (done) {
          const u32x4 o = {pack_bf16x2(acc[0] * sc, acc[1] * sc), pack_bf16x2(acc[2] * sc, acc[3] * sc),
                           pack_bf16x2(acc[4] * sc, acc[5] * sc), pack_bf16x2(acc[6] * sc, acc[7] * sc)};
          st128(O, slot == 0 ? (uint32_t)vp0 * (uint32_t)(C * 2) + (uint32_t)q * 16u : OOB, o);
        }
      }
    } else {
      // ================= several points in the tile ==============================================================
      // Every row slot accumulates runs of consecutive views of one point.  A run whose point lies inside the slot
      // is scaled and stored directly.  A point spread over several slots: every slot but the last writes its
      // partial sum to its private LDS row (plain stores: LDS float atomics cost 0.5 ms on the ragged workload),
      // the slot the point ends in adds the rows of the slots before it and stores.
      const SegInfo sg = seg_setup(p.vpj, j, lane, nv);
      const float isn = (scaling ? __builtin_amdgcn_rsqf((float)(sg.se - sg.ss + 1)) : 1.f) * 1.44269504f;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const float m = seg_total(seg_scan_max(ok ? c[e] : -INFINITY, sg, lane), sg, h);
        const float ev = ok ? __builtin_amdgcn_exp2f((c[e] - m) * isn) : 0.f;
        const float s = seg_total(seg_scan_sum(ev, sg, lane), sg, h);
        const float gt = gw ? tanh_pos(vmaxf(__builtin_fmaf(gwl[e], m, gbl[e]), 0.f)) : 1.f;
        if (s_active) {
          const float sc1 = gt * __builtin_amdgcn_rcpf(s + eps);
          ev_t[gl[e] * 32 + j] = STAGED ? (ok ? ev * sc1 : 0.f) : ev;       // staged walk: the scaled weight
          sc_t[gl[e] * 32 + j] = sc1;
        }
      }
      if constexpr (STAGED) {
        bf16_t* rt = s_rows[wv];
#pragma unroll
        for (int kk = 0; kk < KV; ++kk)
          *reinterpret_cast<u32x4*>(rt + (sv0 + kk) * PITCH + q * 8) = xr[0][kk];
        const uint32_t emx = sg.emask | (nv < 32 ? 0xffffffffu << nv : 0u);
        wave_sync();
        const int c0 = lane * CPL, tgc = c0 / (C / G);
        float av[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) av[i] = 0.f;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          if (8 * ch < nv) {
            const float4 wa = *reinterpret_cast<const float4*>(ev_t + tgc * 32 + 8 * ch);
            const float4 wb = *reinterpret_cast<const float4*>(ev_t + tgc * 32 + 8 * ch + 4);
            const float w8[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
            uint32_t rv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              if (CPL == 1) rv[i] = *reinterpret_cast<const uint16_t*>(rt + (8 * ch + i) * PITCH + c0);
              else rv[i] = *reinterpret_cast<const uint32_t*>(rt + (8 * ch + i) * PITCH + c0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int v = 8 * ch + i;
              if (CPL == 1) {
                av[0] = __builtin_fmaf(w8[i], __uint_as_float(rv[i] << 16), av[0]);
              } else {
                av[0] = __builtin_fmaf(w8[i], __uint_as_float(rv[i] << 16), av[0]);
                av[CPL - 1] = __builtin_fmaf(w8[i], __uint_as_float(rv[i] & 0xffff0000u), av[CPL - 1]);
              }
              if ((emx >> v) & 1u) {           // wave-uniform: view v ends its point
                if (v < nv) {
                  const uint32_t pid = (uint32_t)__builtin_amdgcn_readlane(p.vpj, v);
                  const uint32_t oo = pid * (uint32_t)(C * 2) + (uint32_t)c0 * 2u;
                  if (CPL == 1) __builtin_amdgcn_raw_buffer_store_b16((short)f2bf(av[0]), O, (int)oo, 0, 0);
                  else st32(O, oo, pack_bf16x2(av[0], av[CPL - 1]));
                }
#pragma unroll
                for (int k2 = 0; k2 < CPL; ++k2) av[k2] = 0.f;
              }
            }
          }
        }
      } else {
        if (h == 0) pid_t[j] = p.vpj;
        // where the points of the tile start / end: two wave-uniform masks (views without a point = one-view segments)
        const uint32_t inv = nv < 32 ? 0xffffffffu << nv : 0u;
        const uint32_t smx = sg.smask | inv, emx = sg.emask | inv;
        wave_sync();
        float hp[8];
        bool has_head = false;
        int head_view = 0, head_slot0 = 0;
        float* mine = acc_t + slot * C + q * 8;
  #pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (b + 1 < NB) issue_rows(b + 1);
  #pragma unroll
          for (int kk = 0; kk < KB; ++kk) {
            const int k = b * KB + kk, vt = sv0 + k;
            fma_row(xr[b & 1][kk], ev_t[tg * 32 + vt]);
            const bool last = (k == KV - 1) || ((smx >> (vt + 1)) & 1u);
            if (last) {
              if (vt < nv) {
                const int ssk = 31 - __clz((int)(smx & (0xffffffffu >> (31 - vt))));     // first view of vt's point
                const int sek = vt + __ffs((int)(emx >> vt)) - 1;                        // its last view
                if (sek >= sv0 + KV) {
                  // the point continues in the next slot: park the partial sum
                  *reinterpret_cast<float4*>(mine) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                  *reinterpret_cast<float4*>(mine + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
                } else if (ssk >= sv0) {
                  const float s1 = sc_t[tg * 32 + vt];
                  const u32x4 o = {pack_bf16x2(acc[0] * s1, acc[1] * s1), pack_bf16x2(acc[2] * s1, acc[3] * s1),
                                   pack_bf16x2(acc[4] * s1, acc[5] * s1), pack_bf16x2(acc[6] * s1, acc[7] * s1)};
                  st128(O, (uint32_t)pid_t[vt] * (uint32_t)(C * 2) + (uint32_t)q * 16u, o);
                } else {
                  // the point started in an earlier slot and ends here
                  has_head = true;
                  head_view = vt;
                  head_slot0 = ssk / KV;
  #pragma unroll
                  for (int i = 0; i < 8; ++i) hp[i] = acc[i];
                }
              }
  #pragma unroll
              for (int i = 0; i < 8; ++i) acc[i] = 0.f;
            }
          }
        }
        wave_sync();
        if (has_head) {
  #pragma unroll
          for (int d = 1; d < ROWS; ++d) {
            if (slot - d >= head_slot0) {
              const float* src = acc_t + (slot - d) * C + q * 8;
              const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
              hp[0] += lo.x; hp[1] += lo.y; hp[2] += lo.z; hp[3] += lo.w;
              hp[4] += hi.x; hp[5] += hi.y; hp[6] += hi.z; hp[7] += hi.w;
            }
          }
          const float s1 = sc_t[tg * 32 + head_view];
          const u32x4 o = {pack_bf16x2(hp[0] * s1, hp[1] * s1), pack_bf16x2(hp[2] * s1, hp[3] * s1),
                           pack_bf16x2(hp[4] * s1, hp[5] * s1), pack_bf16x2(hp[6] * s1, hp[7] * s1)};
          st128(O, (uint32_t)pid_t[head_view] * (uint32_t)(C * 2) + (uint32_t)q * 16u, o);
        }
    
      }
    }
    wave_sync();
  });
}

}  // namespace chain
}  // namespace dva

using namespace dva;
using namespace dva::chain;

extern "C" {

int dva_chain_prep(const float* W1, const float* W2, const float* W5, int32_t ld5, const float* W6,
                   const float* Ws, int32_t G, void* ops, void* stream) {
  if (!W1 || !W2 || !W5 || !W6 || !Ws || !ops || G < 1 || (G > 4 && G != D) || ld5 < D) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(prep_kernel, dim3(N_OPS), dim3(64), 0, (hipStream_t)stream, W1, W2, W5, ld5, W6, Ws, G,
                     (uint4*)ops);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_bn_consts(const double* sums, double m, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, const float* gamma, const float* beta, float momentum, float eps,
                        int32_t training, const float* W, int32_t ldw, int32_t K, const double* sum_a, float* bn,
                        void* stream) {
  if (!bn || !gamma || !beta) return DVA_ERR_INVALID;
  if (training ? (!sums || m <= 0.0) : (!running_mean || !running_var)) return DVA_ERR_INVALID;
  if ((running_mean == nullptr) != (running_var == nullptr)) return DVA_ERR_INVALID;
  if (W && (ldw < K || K < 1 || K > D)) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(chain_bn_consts_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, m, running_mean,
                     running_var, num_batches_tracked, gamma, beta, momentum, eps, (int)training, W, (int)ldw, (int)K,
                     sum_a, bn);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_tile_chunks(const int64_t* ptr, int64_t n_points, int64_t views_per_chunk, int32_t n_chunks,
                          int64_t* chunk_points, void* stream) {
  if (n_chunks < 0 || n_points < 0 || views_per_chunk < 1) return DVA_ERR_INVALID;
  if (!ptr || !chunk_points) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(tile_chunks_kernel, dim3((n_chunks + 256) / 256), dim3(256), 0, (hipStream_t)stream, ptr,
                     n_points, views_per_chunk, (int)n_chunks, chunk_points);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_tile_offsets(const int32_t* counts, int32_t n_chunks, int64_t* offsets, int32_t* n_tiles,
                           void* stream) {
  if (n_chunks < 0 || n_chunks > (1 << 20) || !offsets || !n_tiles || (n_chunks && !counts)) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(tile_offsets_kernel, dim3(n_chunks > 0 ? (n_chunks + 1023) / 1024 : 1), dim3(1024), 0,
                     (hipStream_t)stream, counts, (int)n_chunks, offsets, n_tiles);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_tile_count(const int64_t* ptr, const int64_t* chunk_points, int32_t n_chunks, int32_t* counts,
                         void* stream) {
  if (n_chunks < 0) return DVA_ERR_INVALID;
  if (n_chunks == 0) return DVA_OK;
  if (!ptr || !chunk_points || !counts) return DVA_ERR_INVALID;
  hipLaunchKernelGGL((tile_walk_kernel<false>), dim3((n_chunks + 63) / 64), dim3(64), 0, (hipStream_t)stream, ptr,
                     chunk_points, n_chunks, counts, (const int64_t*)nullptr, (int2*)nullptr);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_tile_build(const int64_t* ptr, const int64_t* chunk_points, int32_t n_chunks,
                         const int64_t* offsets, void* tiles, void* stream) {
  if (n_chunks < 0) return DVA_ERR_INVALID;
  if (n_chunks == 0) return DVA_OK;
  if (!ptr || !chunk_points || !offsets || !tiles) return DVA_ERR_INVALID;
  hipLaunchKernelGGL((tile_walk_kernel<true>), dim3((n_chunks + 63) / 64), dim3(64), 0, (hipStream_t)stream, ptr,
                     chunk_points, n_chunks, (int32_t*)nullptr, offsets, (int2*)tiles);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_moments(const float* x_map, int64_t n_views, const float* W1, int32_t exact_w1, double* moments,
                      double* stats1, void* stream) {
  if (n_views < 0 || !moments || !stats1 || !W1) return DVA_ERR_INVALID;
  if (n_views > 0) {
    if (!x_map) return DVA_ERR_INVALID;
    int64_t blocks = (n_views + 256 * 16 - 1) / (256 * 16);       // >= 16 views per thread: every block ends with
    if (blocks < 1) blocks = 1;                                    // 44 fp64 atomics on the same addresses
    const int cap = chain_grid(8);
    hipLaunchKernelGGL(moments_kernel, dim3((int)(blocks < cap ? blocks : cap)), dim3(256), 0, (hipStream_t)stream,
                       x_map, n_views, moments);
    DVA_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(stats1_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, moments, W1, (int)exact_w1, stats1);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_stats2(const float* x_map, const int32_t* view_point, const void* tiles, const int32_t* n_tiles,
                     const void* ops, const float* bn1, const float* gamma2, double* stats, float* zstar,
                     int32_t* arg, int64_t n_views, void* stream) {
  if (n_views < 0) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!x_map || !view_point || !tiles || !n_tiles || !ops || !bn1 || !gamma2 || !stats || !zstar || !arg)
    return DVA_ERR_INVALID;
  if (n_views * 32 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(stats2_kernel, dim3(chain_grid(3)), dim3(256), 0, (hipStream_t)stream, x_map, view_point,
                     (const int2*)tiles, n_tiles, (const uint4*)ops, bn1, gamma2, stats, zstar, arg, n_views);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_pooled(const float* zstar, const float* bn2, const int64_t* ptr, float* pooled, int64_t n_points,
                     void* stream) {
  if (n_points < 0) return DVA_ERR_INVALID;
  if (n_points == 0) return DVA_OK;
  if (!zstar || !bn2 || !ptr || !pooled) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(pooled_kernel, dim3(blocks_for(n_points * 8, 256)), dim3(256), 0, (hipStream_t)stream, zstar,
                     bn2, ptr, pooled, n_points);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

static int chain_stats_impl(int32_t layer, const float* x_map, const int32_t* view_point, const float* u,
                            const void* tiles, const int32_t* n_tiles, const void* ops, const float* bn1,
                            const float* bn2, const float* bn5, double* stats, int64_t n_views, int64_t n_points,
                            void* a2, void* stream) {
  if (n_views < 0 || (layer != 5 && layer != 6)) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  const bool from_a2 = a2 && layer == 6;
  if ((!x_map && !from_a2) || !view_point || !u || !tiles || !n_tiles || !ops || !bn1 || !bn2 || !stats ||
      (layer == 6 && !bn5) || ((uintptr_t)a2 & 15))
    return DVA_ERR_INVALID;
  if (n_views * 32 > 0xfffffff0ll || n_points * 128 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  if (a2 && n_views * 64 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  // layer 5 fits 116 VGPRs: four blocks per CU; layer 6 (150) three
  static const int bpc5 = tune_int("DVA_STATS5_BPC", 4);      // read once
  const dim3 grid(chain_grid(layer == 5 ? bpc5 : 3)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define DVA_STATS_MID(L_, A_)                                                                                         \
  hipLaunchKernelGGL((stats_mid_kernel<L_, A_>), grid, block, 0, s, x_map, view_point, u, (const int2*)tiles, n_tiles, \
                     (const uint4*)ops, bn1, bn2, bn5, stats, n_views, n_points, (bf16_t*)a2)
  if (layer == 5 && a2) DVA_STATS_MID(5, 1);
  else if (layer == 5) DVA_STATS_MID(5, 0);
  else if (a2) DVA_STATS_MID(6, 2);
  else DVA_STATS_MID(6, 0);
#undef DVA_STATS_MID
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_stats(int32_t layer, const float* x_map, const int32_t* view_point, const float* u,
                    const void* tiles, const int32_t* n_tiles, const void* ops, const float* bn1,
                    const float* bn2, const float* bn5, double* stats, int64_t n_views, int64_t n_points,
                    void* stream) {
  return chain_stats_impl(layer, x_map, view_point, u, tiles, n_tiles, ops, bn1, bn2, bn5, stats, n_views, n_points,
                          nullptr, stream);
}

// The stored-a2 hybrid (round 6): layer 5 = dva_chain_stats(5) that also WRITES a2 bf16 [V, 32] (accumulator order);
// layer 6 = dva_chain_stats(6) starting from that row instead of x_map (x_map may be null).
int dva_chain_stats_a2(int32_t layer, const float* x_map, const int32_t* view_point, const float* u,
                       const void* tiles, const int32_t* n_tiles, const void* ops, const float* bn1,
                       const float* bn2, const float* bn5, double* stats, int64_t n_views, int64_t n_points,
                       void* a2, void* stream) {
  if (!a2) return DVA_ERR_INVALID;
  return chain_stats_impl(layer, x_map, view_point, u, tiles, n_tiles, ops, bn1, bn2, bn5, stats, n_views, n_points, a2,
                          stream);
}

static int chain_keys_impl(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                           const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2, const float* bn5,
                           const float* bn6, const float* key_bias, void* keys, const float* queries, float* compat,
                           int32_t G, float scale, int64_t n_views, int64_t n_points, void* stream) {
  if (n_views < 0 || n_points < 0) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!x_map || !view_point || !u || !tiles || !n_tiles || !ops || !bn1 || !bn2 || !bn5 || !bn6 || !key_bias || !keys)
    return DVA_ERR_INVALID;
  if (n_views * 64 > 0xfffffff0ll || n_points * 128 > 0xfffffff0ll) return DVA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(keys_kernel, dim3(chain_grid(3)), dim3(256), 0, (hipStream_t)stream, x_map, view_point, u,
                     (const int2*)tiles, n_tiles, (const uint4*)ops, bn1, bn2, bn5, bn6, key_bias, (bf16_t*)keys, n_views,
                     n_points, queries, compat, (int)G, scale);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_chain_keys(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                   const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2, const float* bn5,
                   const float* bn6, const float* key_bias, void* keys, int64_t n_views, int64_t n_points, void* stream) {
  return chain_keys_impl(x_map, view_point, u, tiles, n_tiles, ops, bn1, bn2, bn5, bn6, key_bias, keys, nullptr, nullptr, 1,
                         0.f, n_views, n_points, stream);
}

// keys + compatibilities in one pass: queries fp32 [N][32] in position order, compat fp32 [V][4] (G = 1, 2, 4 groups used)
int dva_chain_keys_compat(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                          const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2, const float* bn5,
                          const float* bn6, const float* key_bias, void* keys, const float* queries, float* compat,
                          int32_t G, float scale, int64_t n_views, int64_t n_points, void* stream) {
  if (G != 1 && G != 2 && G != 4) return DVA_ERR_INVALID;
  if (n_views > 0 && (!queries || !compat || ((uintptr_t)queries & 15) || ((uintptr_t)compat & 15))) return DVA_ERR_INVALID;
  return chain_keys_impl(x_map, view_point, u, tiles, n_tiles, ops, bn1, bn2, bn5, bn6, key_bias, keys, queries, compat, G,
                         scale, n_views, n_points, stream);
}

int dva_chain_attn_fwd(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                       const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2,
                       const float* bn5, const float* bn6, const float* score_bias, const void* rows,
                       const int32_t* row_idx, const int64_t* ptr, const float* gate_w, const float* gate_b,
                       void* out, float* scores_out, int64_t n_points, int64_t n_views, int64_t n_rows, int32_t C,
                       int32_t G, int32_t scaling, float eps, void* stream) {
  if (n_views < 0 || n_points < 0) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!x_map || !view_point || !u || !tiles || !n_tiles || !ops || !bn1 || !bn2 || !bn5 || !bn6 || !score_bias ||
      !rows || !row_idx || !ptr || !out || ((gate_w == nullptr) != (gate_b == nullptr)))
    return DVA_ERR_INVALID;
  if (n_views * 32 > 0xfffffff0ll || n_points * 128 > 0xfffffff0ll || n_rows * C * 2 > 0xfffffff0ll ||
      n_points * C * 2 > 0xfffffff0ll)
    return DVA_ERR_UNSUPPORTED;
  static const int occ_env = getenv("DVA_ATTN_FWD_OCC") ? atoi(getenv("DVA_ATTN_FWD_OCC")) : 0;   // A/B: 3 or 4
  const bool dense = occ_env ? (occ_env == 4 && C <= 64) : (C <= 64 && n_views >= 24 * n_points);   // mostly one point per tile
  const dim3 grid(chain_grid(dense ? 4 : 3)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define DVA_ATTN_FWD_O(LPR_, G_, OCC_)                                                                          \
  hipLaunchKernelGGL((attn_fwd_kernel<LPR_, G_, OCC_>), grid, block, 0, s, x_map, view_point, u,               \
                     (const int2*)tiles, n_tiles, (const uint4*)ops, bn1, bn2, bn5, bn6, score_bias,            \
                     (const bf16_t*)rows, row_idx, ptr, gate_w, gate_b, (bf16_t*)out, scores_out, scaling, eps, \
                     n_views, n_points, n_rows)
#define DVA_ATTN_FWD(LPR_, G_)                                                                                  \
  do {                                                                                                          \
    if (LPR_ <= 8 && dense) DVA_ATTN_FWD_O(LPR_, G_, (LPR_ <= 8 ? 4 : 3));                                      \
    else DVA_ATTN_FWD_O(LPR_, G_, 3);                                                                           \
  } while (0)
  const int key = C * 8 + G;
  switch (key) {
    case 32 * 8 + 1: DVA_ATTN_FWD(4, 1); break;
    case 32 * 8 + 2: DVA_ATTN_FWD(4, 2); break;
    case 32 * 8 + 4: DVA_ATTN_FWD(4, 4); break;
    case 64 * 8 + 1: DVA_ATTN_FWD(8, 1); break;
    case 64 * 8 + 2: DVA_ATTN_FWD(8, 2); break;
    case 64 * 8 + 4: DVA_ATTN_FWD(8, 4); break;
    case 128 * 8 + 1: DVA_ATTN_FWD(16, 1); break;
    case 128 * 8 + 2: DVA_ATTN_FWD(16, 2); break;
    case 128 * 8 + 4: DVA_ATTN_FWD(16, 4); break;
    case 256 * 8 + 1: DVA_ATTN_FWD(32, 1); break;
    case 256 * 8 + 2: DVA_ATTN_FWD(32, 2); break;
    case 256 * 8 + 4: DVA_ATTN_FWD(32, 4); break;
    case 512 * 8 + 1: DVA_ATTN_FWD(64, 1); break;
    case 512 * 8 + 2: DVA_ATTN_FWD(64, 2); break;
    case 512 * 8 + 4: DVA_ATTN_FWD(64, 4); break;
    default: return DVA_ERR_UNSUPPORTED;
  }
#undef DVA_ATTN_FWD
#undef DVA_ATTN_FWD_O
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

// QKVBimodalCSRPool in ONE view kernel (round 4): the fused view kernel with the KEY layer as the chain's last layer (ops
// prepared with Ws = K.weight, G = 32; key_bias [32]) and the compatibilities scale * <key, queries[point]> per query-key group
// as the scores (queries fp32 [N][32] in position order, 16-byte aligned; G = groups in {1, 2, 4}); keys_out (nullable;
// training) bf16 [V][32] receives the key rows (position order) that dva_qkv_dquery reads back.
int dva_chain_attn_fwd_keys(const float* x_map, const int32_t* view_point, const float* u, const void* tiles,
                            const int32_t* n_tiles, const void* ops, const float* bn1, const float* bn2,
                            const float* bn5, const float* bn6, const float* key_bias, const float* queries, float scale,
                            const void* rows, const int32_t* row_idx, const int64_t* ptr, const float* gate_w,
                            const float* gate_b, void* out, float* scores_out, void* keys_out, int64_t n_points,
                            int64_t n_views, int64_t n_rows, int32_t C, int32_t G, int32_t scaling, float eps,
                            void* stream) {
  if (n_views < 0 || n_points < 0) return DVA_ERR_INVALID;
  if (n_views == 0) return DVA_OK;
  if (!x_map || !view_point || !u || !tiles || !n_tiles || !ops || !bn1 || !bn2 || !bn5 || !bn6 || !key_bias || !queries ||
      !rows || !row_idx || !ptr || !out || ((gate_w == nullptr) != (gate_b == nullptr)) || ((uintptr_t)queries & 15))
    return DVA_ERR_INVALID;
  if (n_views * 64 > 0xfffffff0ll || n_points * 128 > 0xfffffff0ll || n_rows * C * 2 > 0xfffffff0ll ||
      n_points * C * 2 > 0xfffffff0ll)
    return DVA_ERR_UNSUPPORTED;
  const dim3 grid(chain_grid(3)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define DVA_ATTN_FWD_K(LPR_, G_)                                                                                \
  hipLaunchKernelGGL((attn_fwd_kernel<LPR_, G_, 3, true>), grid, block, 0, s, x_map, view_point, u,             \
                     (const int2*)tiles, n_tiles, (const uint4*)ops, bn1, bn2, bn5, bn6, key_bias,              \
                     (const bf16_t*)rows, row_idx, ptr, gate_w, gate_b, (bf16_t*)out, scores_out, scaling, eps, \
                     n_views, n_points, n_rows, queries, scale, (bf16_t*)keys_out)
  const int key = C * 8 + G;
  switch (key) {
    case 32 * 8 + 1: DVA_ATTN_FWD_K(4, 1); break;
    case 32 * 8 + 2: DVA_ATTN_FWD_K(4, 2); break;
    case 32 * 8 + 4: DVA_ATTN_FWD_K(4, 4); break;
    case 64 * 8 + 1: DVA_ATTN_FWD_K(8, 1); break;
    case 64 * 8 + 2: DVA_ATTN_FWD_K(8, 2); break;
    case 64 * 8 + 4: DVA_ATTN_FWD_K(8, 4); break;
    case 128 * 8 + 1: DVA_ATTN_FWD_K(16, 1); break;
    case 128 * 8 + 2: DVA_ATTN_FWD_K(16, 2); break;
    case 128 * 8 + 4: DVA_ATTN_FWD_K(16, 4); break;
    case 256 * 8 + 1: DVA_ATTN_FWD_K(32, 1); break;
    case 256 * 8 + 2: DVA_ATTN_FWD_K(32, 2); break;
    case 256 * 8 + 4: DVA_ATTN_FWD_K(32, 4); break;
    case 512 * 8 + 1: DVA_ATTN_FWD_K(64, 1); break;
    case 512 * 8 + 2: DVA_ATTN_FWD_K(64, 2); break;
    case 512 * 8 + 4: DVA_ATTN_FWD_K(64, 4); break;
    default: return DVA_ERR_UNSUPPORTED;
  }
#undef DVA_ATTN_FWD_K
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // extern "C"
