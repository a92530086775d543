// Multi-view feature gather / scatter for gfx950 (reference: core/multimodal/image.py:1262-1287,
// :105-170, :1871-1885, :1916-1980).
//
// Feature maps are channels-last [B,H,W,C]: the C features of one mapped pixel are one contiguous
// burst, so a group of C*s/16 lanes moves one atom with 16-byte accesses and a wavefront moves
// 64*16 B = 1 KiB per instruction.  The maps themselves (tens of MB) live in L2 / Infinity Cache;
// HBM traffic is the 8-byte packed index in and the [P,C] rows out:
//   nearest fwd bytes = P*(8 + C*s [map read, cache] + C*s [write]);   bwd = P*(8 + C*s) + atomics.
#include "dva_common.h"

namespace dva {

// torch.floor_divide on floats (c10::div_floor_floating): floor of the exact quotient a/b.
__device__ __forceinline__ float floordiv_f32(float a, float b) {
  const float mod = fmodf(a, b);
  float div = __fdiv_rn(__fsub_rn(a, mod), b);
  if (mod != 0.f && ((b < 0.f) != (mod < 0.f))) div = __fsub_rn(div, 1.f);
  if (div != 0.f) {
    float fl = floorf(div);
    if (__fsub_rn(div, fl) > 0.5f) fl = __fadd_rn(fl, 1.f);
    return fl;
  }
  return copysignf(0.f, __fdiv_rn(a, b));
}

template <typename PIX>
__global__ __launch_bounds__(256) void pack_index_kernel(const int64_t* __restrict__ images,
                                                          const int64_t* __restrict__ atom_ptr,
                                                          const PIX* __restrict__ pixels,
                                                          double ratio, int64_t n_views,
                                                          PackedIdx* __restrict__ out) {
  // one thread per view walks its atoms (P == V in exact mode, so usually one atom each)
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n_views;
       v += (int64_t)gridDim.x * blockDim.x) {
    const int32_t img = (int32_t)images[v];
    for (int64_t a = atom_ptr[v]; a < atom_ptr[v + 1]; ++a) {
      PackedIdx pi;
      pi.img = img;
      // image.py:1953-1954: (pix // ratio).long(); torch promotes the integer pixel to fp32 for a
      // Python-float ratio and applies its Python-style floor division (true floor of the exact
      // quotient), restated in floordiv_f32().
      if (ratio == 1.0) {
        pi.x = (int16_t)pixels[2 * a];
        pi.y = (int16_t)pixels[2 * a + 1];
      } else {
        const float rf = (float)ratio;
        pi.x = (int16_t)floordiv_f32((float)pixels[2 * a], rf);
        pi.y = (int16_t)floordiv_f32((float)pixels[2 * a + 1], rf);
      }
      out[a] = pi;
    }
  }
}

// pack_index + row_index in one pass for the lazy gather (no packed index is written or read back)
template <typename PIX>
__global__ __launch_bounds__(256) void mapping_row_index_kernel(const int64_t* __restrict__ images,
                                                                 const int64_t* __restrict__ atom_ptr,
                                                                 const PIX* __restrict__ pixels, double ratio,
                                                                 int64_t n_views, int H, int W,
                                                                 int32_t* __restrict__ row_idx) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n_views;
       v += (int64_t)gridDim.x * blockDim.x) {
    const int32_t img = (int32_t)images[v];
    for (int64_t a = atom_ptr[v]; a < atom_ptr[v + 1]; ++a) {
      int32_t x, y;
      if (ratio == 1.0) {
        x = (int32_t)pixels[2 * a];
        y = (int32_t)pixels[2 * a + 1];
      } else {
        const float rf = (float)ratio;
        x = (int32_t)(int16_t)floordiv_f32((float)pixels[2 * a], rf);
        y = (int32_t)(int16_t)floordiv_f32((float)pixels[2 * a + 1], rf);
      }
      row_idx[a] = (img * H + y) * W + x;
    }
  }
}

// Flat row index (img*H + y)*W + x of every atom into the [B*H*W, C] view of a channels-last map,
// plus (optionally) the number of atoms that land on every row: the weights with which per-view
// batch statistics can be evaluated at feature-map level (DESIGN.md "E_mod hoisting").
__global__ __launch_bounds__(256) void row_index_kernel(const PackedIdx* __restrict__ idx,
                                                         int64_t n_atoms, int H, int W,
                                                         int32_t row_offset,
                                                         int32_t* __restrict__ row_idx,
                                                         int32_t* __restrict__ counts) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n_atoms;
       p += (int64_t)gridDim.x * blockDim.x) {
    const PackedIdx pi = idx[p];
    const int32_t r = (pi.img * H + pi.y) * W + pi.x;
    row_idx[p] = r + row_offset;
    if (counts) atomicAdd(&counts[r], 1);
  }
}

// Nearest gather: pure byte movement in UNIT-byte units (16, 4 or 2).
template <typename UNIT>
__global__ __launch_bounds__(256) void gather_nearest_fwd_kernel(const UNIT* __restrict__ x,
                                                                  const PackedIdx* __restrict__ idx,
                                                                  UNIT* __restrict__ out,
                                                                  int64_t n_atoms, int H, int W,
                                                                  int units_per_row) {
  const int64_t total = n_atoms * (int64_t)units_per_row;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = t / units_per_row;
    const int u = (int)(t - p * units_per_row);
    const PackedIdx pi = idx[p];
    const int64_t pix = ((int64_t)pi.img * H + pi.y) * W + pi.x;
    out[t] = x[pix * units_per_row + u];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_nearest_bwd_kernel(const T* __restrict__ gout,
                                                                  const PackedIdx* __restrict__ idx,
                                                                  float* __restrict__ gx,
                                                                  int64_t n_atoms, int H, int W,
                                                                  int C) {
  const int64_t total = n_atoms * (int64_t)C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = t / C;
    const int c = (int)(t - p * C);
    const PackedIdx pi = idx[p];
    const int64_t pix = ((int64_t)pi.img * H + pi.y) * W + pi.x;
    atomicAdd(&gx[pix * C + c], Elt<T>::ld(gout, t));
  }
}

// anchor of a view = the cell (top, left) of its 2 x 2 tap block on the REPLICATION-PADDED grid of its image,
// (img (H + 1) + top) (W + 1) + left with top in [0, H], left in [0, W]: the four taps are then the padded cells
// (top, left), (top, left + 1), (top + 1, left), (top + 1, left + 1) for every view, borders included -- the structure
// the anchor scatter (dva_anchor_rows_sum / dva_anchor_combine) relies on.  A view whose floor(q + 1) != floor(q) + 1
// (the reference evaluates both floors separately, image.py:142-145: possible when q + 1 rounds up across an integer)
// does not have it and gets the dummy anchor B (H + 1) (W + 1): dva_anchor_fixup adds its taps one by one.
struct Taps {
  int64_t tl, tr, bl, br;  // pixel offsets (in pixels) into [B,H,W]
  float w_tl, w_tr, w_bl, w_br;
  int64_t anchor;          // see above; -1 = no 2 x 2 structure
};

// image.py:138-165 with ReplicationPad2d(1): padded index i -> clamp(i-1, 0, size-1)
__device__ __forceinline__ Taps bilinear_taps(const PackedIdx pi, float cy, float cx, int H, int W) {
  const float qy = __fadd_rn(__fmul_rn(cy, (float)H), 0.5f);
  const float qx = __fadd_rn(__fmul_rn(cx, (float)W), 0.5f);
  const float top = floorf(qy), bottom = floorf(__fadd_rn(qy, 1.f));
  const float left = floorf(qx), right = floorf(__fadd_rn(qx, 1.f));
  Taps t;
  t.w_tl = fabsf(__fmul_rn(__fsub_rn(qy, bottom), __fsub_rn(qx, right)));
  t.w_tr = fabsf(__fmul_rn(__fsub_rn(qy, bottom), __fsub_rn(qx, left)));
  t.w_bl = fabsf(__fmul_rn(__fsub_rn(qy, top), __fsub_rn(qx, right)));
  t.w_br = fabsf(__fmul_rn(__fsub_rn(qy, top), __fsub_rn(qx, left)));
  const int it = min(max((int)top - 1, 0), H - 1), ib = min(max((int)bottom - 1, 0), H - 1);
  const int il = min(max((int)left - 1, 0), W - 1), ir = min(max((int)right - 1, 0), W - 1);
  const int64_t base = (int64_t)pi.img * H;
  t.tl = (base + it) * W + il;
  t.tr = (base + it) * W + ir;
  t.bl = (base + ib) * W + il;
  t.br = (base + ib) * W + ir;
  const bool regular = bottom == top + 1.f && right == left + 1.f && top >= 0.f && left >= 0.f &&
                       top <= (float)H && left <= (float)W;
  t.anchor = regular ? ((int64_t)pi.img * (H + 1) + (int)top) * (W + 1) + (int)left : -1;
  return t;
}

template <typename T>
__global__ __launch_bounds__(256) void gather_bilinear_fwd_kernel(
    const T* __restrict__ x, const PackedIdx* __restrict__ idx, const float* __restrict__ coords,
    T* __restrict__ out, int64_t n_atoms, int H, int W, int C) {
  const int64_t total = n_atoms * (int64_t)C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = t / C;
    const int c = (int)(t - p * C);
    const Taps tp = bilinear_taps(idx[p], coords[2 * p], coords[2 * p + 1], H, W);
    // reference evaluation order: ((w_tl*X_tl + w_tr*X_tr) + w_bl*X_bl) + w_br*X_br, no fma
    float acc = __fmul_rn(tp.w_tl, Elt<T>::ld(x, tp.tl * C + c));
    acc = __fadd_rn(acc, __fmul_rn(tp.w_tr, Elt<T>::ld(x, tp.tr * C + c)));
    acc = __fadd_rn(acc, __fmul_rn(tp.w_bl, Elt<T>::ld(x, tp.bl * C + c)));
    acc = __fadd_rn(acc, __fmul_rn(tp.w_br, Elt<T>::ld(x, tp.br * C + c)));
    Elt<T>::st(out, t, acc);
  }
}

// The same gather with 16-byte accesses: `lpr` = C / VEC lanes (a power of two) cover one atom's row, the taps are
// evaluated once per lane instead of once per element, the arithmetic per element is the scalar kernel's (same
// operation order, no fma: bit-identical output).  The scalar kernel moved 2.8 TB/s at C = 256 (bf16): 2-byte loads,
// a 64-bit division and the tap arithmetic per element.
template <typename T>
struct GVec;
template <>
struct GVec<float> {
  static constexpr int N = 4;
  typedef float4 raw;
  static __device__ __forceinline__ void unpack(const raw& r, float* f) { f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w; }
  static __device__ __forceinline__ raw pack(const float* f) { return make_float4(f[0], f[1], f[2], f[3]); }
};
template <>
struct GVec<bf16_t> {
  static constexpr int N = 8;
  typedef uint4 raw;
  static __device__ __forceinline__ void unpack(const raw& r, float* f) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(w[i] << 16);
      f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ raw pack(const float* f) {
    uint4 r;
    r.x = (uint32_t)f2bf(f[0]) | ((uint32_t)f2bf(f[1]) << 16);
    r.y = (uint32_t)f2bf(f[2]) | ((uint32_t)f2bf(f[3]) << 16);
    r.z = (uint32_t)f2bf(f[4]) | ((uint32_t)f2bf(f[5]) << 16);
    r.w = (uint32_t)f2bf(f[6]) | ((uint32_t)f2bf(f[7]) << 16);
    return r;
  }
};
template <typename T>
__global__ __launch_bounds__(256) void gather_bilinear_fwd_vec_kernel(
    const T* __restrict__ x, const PackedIdx* __restrict__ idx, const float* __restrict__ coords,
    T* __restrict__ out, int64_t n_atoms, int H, int W, int C, int lpr) {
  constexpr int VEC = GVec<T>::N;
  typedef typename GVec<T>::raw raw_t;
  const int lane = threadIdx.x & 63, lane_r = lane & (lpr - 1), slot = lane / lpr, slots = 64 / lpr;
  const int64_t col = (int64_t)lane_r * VEC;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t p0 = wave * slots; p0 < n_atoms; p0 += n_waves * slots) {
    const int64_t p = p0 + slot;
    if (p >= n_atoms) continue;
    const Taps tp = bilinear_taps(idx[p], coords[2 * p], coords[2 * p + 1], H, W);
    const raw_t r0 = *reinterpret_cast<const raw_t*>(x + tp.tl * C + col);
    const raw_t r1 = *reinterpret_cast<const raw_t*>(x + tp.tr * C + col);
    const raw_t r2 = *reinterpret_cast<const raw_t*>(x + tp.bl * C + col);
    const raw_t r3 = *reinterpret_cast<const raw_t*>(x + tp.br * C + col);
    float a[VEC], b[VEC], c[VEC], d[VEC], o[VEC];
    GVec<T>::unpack(r0, a);
    GVec<T>::unpack(r1, b);
    GVec<T>::unpack(r2, c);
    GVec<T>::unpack(r3, d);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float acc = __fmul_rn(tp.w_tl, a[e]);
      acc = __fadd_rn(acc, __fmul_rn(tp.w_tr, b[e]));
      acc = __fadd_rn(acc, __fmul_rn(tp.w_bl, c[e]));
      o[e] = __fadd_rn(acc, __fmul_rn(tp.w_br, d[e]));
    }
    *reinterpret_cast<raw_t*>(out + p * C + col) = GVec<T>::pack(o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_bilinear_bwd_kernel(
    const T* __restrict__ gout, const PackedIdx* __restrict__ idx, const float* __restrict__ coords,
    float* __restrict__ gx, int64_t n_atoms, int H, int W, int C) {
  const int64_t total = n_atoms * (int64_t)C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = t / C;
    const int c = (int)(t - p * C);
    const Taps tp = bilinear_taps(idx[p], coords[2 * p], coords[2 * p + 1], H, W);
    const float g = Elt<T>::ld(gout, t);
    atomicAdd(&gx[tp.tl * C + c], tp.w_tl * g);
    atomicAdd(&gx[tp.tr * C + c], tp.w_tr * g);
    atomicAdd(&gx[tp.bl * C + c], tp.w_bl * g);
    atomicAdd(&gx[tp.br * C + c], tp.w_br * g);
  }
}

// rows[4p + k], weights[4p + k]: the 4 corner rows of the [B*H*W, C] map and their bilinear weights for atom p
// (same taps as the forward), the input of the row-plan backward (dva_gather_rows_sum with weights).
__global__ __launch_bounds__(256) void bilinear_taps_kernel(const PackedIdx* __restrict__ idx,
                                                             const float* __restrict__ coords,
                                                             int64_t n_atoms, int H, int W,
                                                             int32_t* __restrict__ rows,
                                                             float* __restrict__ weights,
                                                             int32_t* __restrict__ anchors, int32_t dummy) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n_atoms;
       p += (int64_t)gridDim.x * blockDim.x) {
    const Taps tp = bilinear_taps(idx[p], coords[2 * p], coords[2 * p + 1], H, W);
    *reinterpret_cast<int4*>(rows + 4 * p) = make_int4((int)tp.tl, (int)tp.tr, (int)tp.bl, (int)tp.br);
    *reinterpret_cast<float4*>(weights + 4 * p) = make_float4(tp.w_tl, tp.w_tr, tp.w_bl, tp.w_br);
    if (anchors) anchors[p] = tp.anchor >= 0 ? (int32_t)tp.anchor : dummy;
  }
}

// The taps of several SETTINGS (feature maps of different sizes) as one gather over the stacked map rows, in a given view
// order (reference core/multimodal/image.py:1549-1588 view_cat_sorting applied to the concatenated [V_s, C] tensors,
// modules/multimodal/modules.py:514-525): view i of the result = view order[i] of the concatenation; its tap rows move by the
// row offset of its setting, its anchor by the anchor offset (the per-setting dummy anchor becomes the common one).
constexpr int TAPS_CAT_MAX = 8;
struct TapsCatArgs {
  const int4* rows[TAPS_CAT_MAX];
  const float4* weights[TAPS_CAT_MAX];
  const int32_t* anchors[TAPS_CAT_MAX];
  int64_t v_end[TAPS_CAT_MAX];       // exclusive prefix ends of the settings' view ranges in the concatenation
  int32_t row_off[TAPS_CAT_MAX], anchor_off[TAPS_CAT_MAX], n_anchor[TAPS_CAT_MAX];
  int32_t n, dummy;
};
__global__ __launch_bounds__(256) void taps_cat_kernel(TapsCatArgs a, const int64_t* __restrict__ order, int64_t V,
                                                       int4* __restrict__ rows_out, float4* __restrict__ w_out,
                                                       int32_t* __restrict__ anchors_out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t src = order ? order[i] : i;
    int s = 0;
    int64_t base = 0;
#pragma unroll
    for (int k = 0; k < TAPS_CAT_MAX - 1; ++k)
      if (k + 1 < a.n && src >= a.v_end[k]) { s = k + 1; base = a.v_end[k]; }
    const int64_t l = src - base;
    int4 r = a.rows[s][l];
    const int32_t ro = a.row_off[s];
    r.x += ro; r.y += ro; r.z += ro; r.w += ro;
    rows_out[i] = r;
    w_out[i] = a.weights[s][l];
    const int32_t an = a.anchors[s][l];
    anchors_out[i] = an == a.n_anchor[s] ? a.dummy : an + a.anchor_off[s];
  }
}

// dY[r][c] (fp32, written) from the per-anchor sums S[a][k][c] of dva_anchor_rows_sum: row (b, y, x) collects the padded
// cells that replicate it -- (py, px) with clamp(py - 1) = y, clamp(px - 1) = x -- and a padded cell collects tap k of
// the anchors it is tap k of: S0[py][px] + S1[py][px - 1] + S2[py - 1][px] + S3[py - 1][px - 1].
__global__ __launch_bounds__(256) void anchor_combine_kernel(const float* __restrict__ S, float* __restrict__ dY,
                                                              int B, int H, int W, int C) {
  const int64_t total = (int64_t)B * H * W * (C / 4);
  const int H1 = H + 1, W1 = W + 1, cq = C / 4;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(t % cq);
    const int64_t r = t / cq;
    const int x = (int)(r % W), y = (int)((r / W) % H), b = (int)(r / ((int64_t)W * H));
    const int py0 = y == 0 ? 0 : y + 1, py1 = y == H - 1 ? H + 1 : y + 1;
    const int px0 = x == 0 ? 0 : x + 1, px1 = x == W - 1 ? W + 1 : x + 1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add = [&](int t_, int l_, int k) {
      if (t_ < 0 || t_ > H || l_ < 0 || l_ > W) return;
      const float4 v = *reinterpret_cast<const float4*>(S + ((((int64_t)b * H1 + t_) * W1 + l_) * 4 + k) * C + 4 * c4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    };
    for (int py = py0; py <= py1; ++py) {
      for (int px = px0; px <= px1; ++px) {
        add(py, px, 0);
        add(py, px - 1, 1);
        add(py - 1, px, 2);
        add(py - 1, px - 1, 3);
      }
    }
    *reinterpret_cast<float4*>(dY + r * C + 4 * c4) = acc;
  }
}

// views without the 2 x 2 structure (dummy anchor): their four taps, one by one (fp32 atomics: there are none on real
// data, see bilinear_taps)
// z != nullptr: the rows are dy_a in position order and get the BatchNorm_a backward of dva_anchor_rows_sum_bn first
template <typename T>
__global__ __launch_bounds__(256) void anchor_fixup_kernel(const T* __restrict__ grad, const int32_t* __restrict__ rows,
                                                            const float* __restrict__ weights,
                                                            const int32_t* __restrict__ anchors, int32_t dummy,
                                                            float* __restrict__ dY, int64_t n_atoms, int C,
                                                            const T* __restrict__ z, const float* __restrict__ bn,
                                                            const float* __restrict__ sm) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n_atoms; p += (int64_t)gridDim.x * blockDim.x) {
    if (anchors[p] != dummy) continue;
    for (int k = 0; k < 4; ++k) {
      const float w = weights[4 * p + k];
      const int64_t r = rows[4 * p + k];
      for (int c = 0; c < C; ++c) {
        float g = Elt<T>::ld(grad, p * C + c);
        if (z) {
          const int rr = c & 15, hh = (c >> 4) & 1, ch = (c & ~31) + (rr & 3) + 8 * (rr >> 2) + 4 * hh;
          const float mean = bn[ch], inv = bn[C + ch], gg = bn[2 * C + ch] * inv, s1 = sm[ch], s2 = sm[C + ch];
          g = fmaf(-(gg * inv * s2), Elt<T>::ld(z, p * C + c), fmaf(gg, g, -(gg * (s1 - mean * inv * s2))));
        }
        atomicAdd(&dY[r * C + c], w * g);
      }
    }
  }
}

// ---- nearest gather fused with the atomic max pool of a NON-exact mapping (several pixels per view) --------------------
// Reference: x_mod = features[idx] ([P, C], core/multimodal/image.py:1262-1287) followed by BimodalCSRPool('max') over the
// atoms of each view (modules/multimodal/pooling.py:14-71 through modules.py:400-407): out[v][c] = max over the atoms a of
// view v of rows[row_idx[a]][c], 0 for a view without atoms; ties -> the first atom (as dva_segment_csr_fwd).  No [P, C]
// tensor: one thread per (view, 8- or 4-channel group) walks the view's atoms (the map rows come out of the cache
// hierarchy), arg uint16 [V][C] = offset of the winning atom inside the view (0xffff: none) for the backward.
template <typename T>
__global__ __launch_bounds__(256) void gather_segment_max_fwd_kernel(const T* __restrict__ rows,
                                                                      const int32_t* __restrict__ row_idx,
                                                                      const int64_t* __restrict__ atom_ptr,
                                                                      T* __restrict__ out, uint16_t* __restrict__ arg,
                                                                      int64_t V, int C) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int cpr = C / VEC;
  const int64_t total = V * (int64_t)cpr;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = t / cpr;
    const int c0 = (int)(t - v * cpr) * VEC;
    const int64_t beg = atom_ptr[v], end = atom_ptr[v + 1];
    float acc[VEC];
    uint16_t best[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      acc[k] = 0.f;
      best[k] = 0xffffu;
    }
    for (int64_t a = beg; a < end; ++a) {
      const uint4 raw = *reinterpret_cast<const uint4*>(rows + (int64_t)row_idx[a] * C + c0);
      float f[VEC];
      if (sizeof(T) == 4) {
        f[0] = __uint_as_float(raw.x); f[1] = __uint_as_float(raw.y);
        f[2] = __uint_as_float(raw.z); f[VEC - 1] = __uint_as_float(raw.w);
      } else {
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f[(2 * e) % VEC] = __uint_as_float(w[e] << 16);
          f[(2 * e + 1) % VEC] = __uint_as_float(w[e] & 0xffff0000u);
        }
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if (a == beg || f[k] > acc[k]) {
          acc[k] = f[k];
          best[k] = (uint16_t)(a - beg);
        }
      }
    }
    if (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(out + v * C + c0) = make_float4(acc[0], acc[1], acc[2], acc[VEC - 1]);
      *reinterpret_cast<uint2*>(arg + v * C + c0) =
          make_uint2(best[0] | ((uint32_t)best[1] << 16), best[2] | ((uint32_t)best[VEC - 1] << 16));
    } else {
      *reinterpret_cast<uint4*>(out + v * C + c0) =
          make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4 % VEC], acc[5 % VEC]),
                     pack_bf16x2(acc[6 % VEC], acc[7 % VEC]));
      *reinterpret_cast<uint4*>(arg + v * C + c0) =
          make_uint4(best[0] | ((uint32_t)best[1] << 16), best[2] | ((uint32_t)best[3] << 16),
                     best[4 % VEC] | ((uint32_t)best[5 % VEC] << 16), best[6 % VEC] | ((uint32_t)best[7 % VEC] << 16));
    }
  }
}

// backward: grad_rows[row_idx[beg + arg[v][c]]][c] += grad_out[v][c] (fp32 atomics into the map-sized gradient, caller-zeroed)
template <typename T>
__global__ __launch_bounds__(256) void gather_segment_max_bwd_kernel(const T* __restrict__ gout,
                                                                      const uint16_t* __restrict__ arg,
                                                                      const int32_t* __restrict__ row_idx,
                                                                      const int64_t* __restrict__ atom_ptr,
                                                                      float* __restrict__ grows, int64_t V, int C) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int cpr = C / VEC;
  const int64_t total = V * (int64_t)cpr;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = t / cpr;
    const int c0 = (int)(t - v * cpr) * VEC;
    const int64_t beg = atom_ptr[v];
    if (atom_ptr[v + 1] <= beg) continue;
    float g[VEC];
    uint16_t k16[VEC];
    const uint4 raw = *reinterpret_cast<const uint4*>(gout + v * C + c0);
    if (sizeof(T) == 4) {
      g[0] = __uint_as_float(raw.x); g[1] = __uint_as_float(raw.y);
      g[2] = __uint_as_float(raw.z); g[VEC - 1] = __uint_as_float(raw.w);
      const uint2 a2 = *reinterpret_cast<const uint2*>(arg + v * C + c0);
      k16[0] = (uint16_t)a2.x; k16[1] = (uint16_t)(a2.x >> 16); k16[2] = (uint16_t)a2.y; k16[VEC - 1] = (uint16_t)(a2.y >> 16);
    } else {
      const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
      const uint4 a4 = *reinterpret_cast<const uint4*>(arg + v * C + c0);
      const uint32_t aw[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        g[(2 * e) % VEC] = __uint_as_float(w[e] << 16);
        g[(2 * e + 1) % VEC] = __uint_as_float(w[e] & 0xffff0000u);
        k16[(2 * e) % VEC] = (uint16_t)aw[e];
        k16[(2 * e + 1) % VEC] = (uint16_t)(aw[e] >> 16);
      }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      if (k16[k] == 0xffffu) continue;
      const int64_t r = row_idx[beg + k16[k]];
      atomicAdd(&grows[r * C + c0 + k], g[k]);
    }
  }
}


// the same backward WITHOUT atomics, over the row plan of the atoms (perm = atoms sorted by map row, row_ptr [R + 1]): one
// wavefront per map row, lpr lanes per 16-byte column group, the atoms of the row spread over 64 / lpr slots; atom a of view
// v = view_of_atom[a] contributes grad_out[v][c] to the channels whose arg offset is a - atom_ptr[v].  Deterministic;
// per atom 16 bytes of indices + the view's arg row and gradient row (one 2 C + C s burst each).
template <typename T>
__global__ __launch_bounds__(256) void gather_segment_max_plan_bwd_kernel(
    const T* __restrict__ gout, const uint16_t* __restrict__ arg, const int64_t* __restrict__ atom_ptr,
    const int32_t* __restrict__ perm, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ view_of_atom,
    float* __restrict__ grows, int64_t R, int C, int lpr) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int U = 2;
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
      int64_t v[U];
      uint32_t koff[U];
      uint4 graw[U], araw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * slots + slot;
        const bool ok = i < end;
        const int a = perm[ok ? i : beg];
        v[u] = view_of_atom[a];
        koff[u] = ok ? (uint32_t)((int64_t)a - atom_ptr[v[u]]) : 0xfffeu;      // 0xfffe never is a stored offset
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        graw[u] = *reinterpret_cast<const uint4*>(gout + v[u] * C + col);
        if (sizeof(T) == 4) {
          const uint2 a2 = *reinterpret_cast<const uint2*>(arg + v[u] * C + col);
          araw[u] = make_uint4(a2.x, a2.y, 0u, 0u);
        } else {
          araw[u] = *reinterpret_cast<const uint4*>(arg + v[u] * C + col);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t gw[4] = {graw[u].x, graw[u].y, graw[u].z, graw[u].w};
        const uint32_t aw[4] = {araw[u].x, araw[u].y, araw[u].z, araw[u].w};
        if (sizeof(T) == 4) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t ak = (aw[k >> 1] >> (16 * (k & 1))) & 0xffffu;
            acc[k % VEC] += ak == koff[u] ? __uint_as_float(gw[k]) : 0.f;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[(2 * e) % VEC] += (aw[e] & 0xffffu) == koff[u] ? __uint_as_float(gw[e] << 16) : 0.f;
            acc[(2 * e + 1) % VEC] += (aw[e] >> 16) == koff[u] ? __uint_as_float(gw[e] & 0xffff0000u) : 0.f;
          }
        }
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
        *reinterpret_cast<float4*>(dst + k) = make_float4(acc[k], acc[k + 1], acc[(k + 2) % VEC], acc[(k + 3) % VEC]);
    }
  }
}

static inline int grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  const int64_t cap = 256 * 32;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace dva

using namespace dva;

extern "C" {

int dva_pack_gather_index(const int64_t* images, const int64_t* atom_ptr, const void* pixels,
                          int32_t pix_bytes, double ratio, int64_t n_views, int64_t n_atoms,
                          void* packed_idx, void* stream) {
  if (n_views < 0 || n_atoms < 0 || !(ratio >= 1.0)) return DVA_ERR_INVALID;
  if (n_views == 0 || n_atoms == 0) return DVA_OK;
  if (!images || !atom_ptr || !pixels || !packed_idx) return DVA_ERR_INVALID;
  const int grid = grid_for(n_views);
  hipStream_t s = (hipStream_t)stream;
  PackedIdx* out = (PackedIdx*)packed_idx;
  switch (pix_bytes) {
    case 2:
      hipLaunchKernelGGL((pack_index_kernel<int16_t>), dim3(grid), dim3(256), 0, s, images, atom_ptr,
                         (const int16_t*)pixels, ratio, n_views, out);
      break;
    case 4:
      hipLaunchKernelGGL((pack_index_kernel<int32_t>), dim3(grid), dim3(256), 0, s, images, atom_ptr,
                         (const int32_t*)pixels, ratio, n_views, out);
      break;
    case 8:
      hipLaunchKernelGGL((pack_index_kernel<int64_t>), dim3(grid), dim3(256), 0, s, images, atom_ptr,
                         (const int64_t*)pixels, ratio, n_views, out);
      break;
    default:
      return DVA_ERR_INVALID;
  }
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_mapping_row_index(const int64_t* images, const int64_t* atom_ptr, const void* pixels, int32_t pix_bytes,
                          double ratio, int64_t n_views, int64_t n_atoms, int32_t B, int32_t H, int32_t W,
                          int32_t* row_idx, void* stream) {
  if (n_views < 0 || n_atoms < 0 || !(ratio >= 1.0) || B < 0 || H < 0 || W < 0) return DVA_ERR_INVALID;
  if ((int64_t)B * H * W > 0x7fffffffLL || H > 32767 || W > 32767) return DVA_ERR_UNSUPPORTED;
  if (n_views == 0 || n_atoms == 0) return DVA_OK;
  if (!images || !atom_ptr || !pixels || !row_idx) return DVA_ERR_INVALID;
  const int grid = grid_for(n_views);
  hipStream_t s = (hipStream_t)stream;
  switch (pix_bytes) {
    case 2:
      hipLaunchKernelGGL((mapping_row_index_kernel<int16_t>), dim3(grid), dim3(256), 0, s, images, atom_ptr,
                         (const int16_t*)pixels, ratio, n_views, H, W, row_idx);
      break;
    case 4:
      hipLaunchKernelGGL((mapping_row_index_kernel<int32_t>), dim3(grid), dim3(256), 0, s, images, atom_ptr,
                         (const int32_t*)pixels, ratio, n_views, H, W, row_idx);
      break;
    case 8:
      hipLaunchKernelGGL((mapping_row_index_kernel<int64_t>), dim3(grid), dim3(256), 0, s, images, atom_ptr,
                         (const int64_t*)pixels, ratio, n_views, H, W, row_idx);
      break;
    default:
      return DVA_ERR_INVALID;
  }
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_gather_row_index(const void* packed_idx, int64_t n_atoms, int32_t B, int32_t H, int32_t W,
                         int32_t row_offset, int32_t* row_idx, int32_t* counts, void* stream) {
  if (n_atoms < 0 || B < 0 || H < 0 || W < 0) return DVA_ERR_INVALID;
  if ((int64_t)B * H * W + row_offset > 0x7fffffffLL) return DVA_ERR_UNSUPPORTED;
  if (n_atoms == 0) return DVA_OK;
  if (!packed_idx || !row_idx) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(row_index_kernel, dim3(grid_for(n_atoms)), dim3(256), 0, (hipStream_t)stream,
                     (const PackedIdx*)packed_idx, n_atoms, H, W, row_offset, row_idx, counts);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

static int check_map(int64_t n_atoms, int B, int H, int W, int C, int dtype) {
  if (n_atoms < 0 || B < 0 || H < 0 || W < 0 || C < 0) return DVA_ERR_INVALID;
  if (dtype != DVA_F32 && dtype != DVA_BF16) return DVA_ERR_INVALID;
  if (H > 32767 || W > 32767) return DVA_ERR_UNSUPPORTED;
  return DVA_OK;
}

int dva_gather_nearest_fwd(const void* x, const void* packed_idx, void* out, int64_t n_atoms,
                           int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream) {
  int rc = check_map(n_atoms, B, H, W, C, dtype);
  if (rc) return rc;
  if (n_atoms == 0 || C == 0) return DVA_OK;
  if (!x || !packed_idx || !out) return DVA_ERR_INVALID;
  const int64_t row_bytes = (int64_t)C * (dtype == DVA_F32 ? 4 : 2);
  hipStream_t s = (hipStream_t)stream;
  const PackedIdx* idx = (const PackedIdx*)packed_idx;
  const bool al16 = ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0);
  if (row_bytes % 16 == 0 && al16) {
    const int u = (int)(row_bytes / 16);
    hipLaunchKernelGGL((gather_nearest_fwd_kernel<uint4>), dim3(grid_for(n_atoms * u)), dim3(256), 0,
                       s, (const uint4*)x, idx, (uint4*)out, n_atoms, H, W, u);
  } else if (row_bytes % 4 == 0) {
    const int u = (int)(row_bytes / 4);
    hipLaunchKernelGGL((gather_nearest_fwd_kernel<uint32_t>), dim3(grid_for(n_atoms * u)), dim3(256),
                       0, s, (const uint32_t*)x, idx, (uint32_t*)out, n_atoms, H, W, u);
  } else {
    const int u = (int)(row_bytes / 2);
    hipLaunchKernelGGL((gather_nearest_fwd_kernel<uint16_t>), dim3(grid_for(n_atoms * u)), dim3(256),
                       0, s, (const uint16_t*)x, idx, (uint16_t*)out, n_atoms, H, W, u);
  }
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_gather_nearest_bwd(const void* grad_out, const void* packed_idx, float* grad_x,
                           int64_t n_atoms, int32_t B, int32_t H, int32_t W, int32_t C,
                           int32_t dtype, void* stream) {
  int rc = check_map(n_atoms, B, H, W, C, dtype);
  if (rc) return rc;
  if (n_atoms == 0 || C == 0) return DVA_OK;
  if (!grad_out || !packed_idx || !grad_x) return DVA_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  const PackedIdx* idx = (const PackedIdx*)packed_idx;
  const int grid = grid_for(n_atoms * (int64_t)C);
  if (dtype == DVA_F32)
    hipLaunchKernelGGL((gather_nearest_bwd_kernel<float>), dim3(grid), dim3(256), 0, s,
                       (const float*)grad_out, idx, grad_x, n_atoms, H, W, C);
  else
    hipLaunchKernelGGL((gather_nearest_bwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, s,
                       (const bf16_t*)grad_out, idx, grad_x, n_atoms, H, W, C);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_gather_bilinear_fwd(const void* x, const void* packed_idx, const float* coords, void* out,
                            int64_t n_atoms, int32_t B, int32_t H, int32_t W, int32_t C,
                            int32_t dtype, void* stream) {
  int rc = check_map(n_atoms, B, H, W, C, dtype);
  if (rc) return rc;
  if (n_atoms == 0 || C == 0) return DVA_OK;
  if (!x || !packed_idx || !coords || !out) return DVA_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  const PackedIdx* idx = (const PackedIdx*)packed_idx;
  const int vec = dtype == DVA_F32 ? 4 : 8, lpr = C / vec;
  const bool vec_ok = (C % vec) == 0 && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0 &&
                      ((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0;
  if (vec_ok) {
    const int slots = 64 / lpr;
    const int gridv = grid_for(((n_atoms + slots - 1) / slots) * 64);
    if (dtype == DVA_F32)
      hipLaunchKernelGGL((gather_bilinear_fwd_vec_kernel<float>), dim3(gridv), dim3(256), 0, s, (const float*)x, idx,
                         coords, (float*)out, n_atoms, H, W, C, lpr);
    else
      hipLaunchKernelGGL((gather_bilinear_fwd_vec_kernel<bf16_t>), dim3(gridv), dim3(256), 0, s, (const bf16_t*)x, idx,
                         coords, (bf16_t*)out, n_atoms, H, W, C, lpr);
    DVA_CHECK_LAUNCH();
    return DVA_OK;
  }
  const int grid = grid_for(n_atoms * (int64_t)C);
  if (dtype == DVA_F32)
    hipLaunchKernelGGL((gather_bilinear_fwd_kernel<float>), dim3(grid), dim3(256), 0, s,
                       (const float*)x, idx, coords, (float*)out, n_atoms, H, W, C);
  else
    hipLaunchKernelGGL((gather_bilinear_fwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, s,
                       (const bf16_t*)x, idx, coords, (bf16_t*)out, n_atoms, H, W, C);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_gather_bilinear_bwd(const void* grad_out, const void* packed_idx, const float* coords,
                            float* grad_x, int64_t n_atoms, int32_t B, int32_t H, int32_t W,
                            int32_t C, int32_t dtype, void* stream) {
  int rc = check_map(n_atoms, B, H, W, C, dtype);
  if (rc) return rc;
  if (n_atoms == 0 || C == 0) return DVA_OK;
  if (!grad_out || !packed_idx || !coords || !grad_x) return DVA_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  const PackedIdx* idx = (const PackedIdx*)packed_idx;
  const int grid = grid_for(n_atoms * (int64_t)C);
  if (dtype == DVA_F32)
    hipLaunchKernelGGL((gather_bilinear_bwd_kernel<float>), dim3(grid), dim3(256), 0, s,
                       (const float*)grad_out, idx, coords, grad_x, n_atoms, H, W, C);
  else
    hipLaunchKernelGGL((gather_bilinear_bwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, s,
                       (const bf16_t*)grad_out, idx, coords, grad_x, n_atoms, H, W, C);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_gather_bilinear_taps(const void* packed_idx, const float* coords, int64_t n_atoms, int32_t B,
                             int32_t H, int32_t W, int32_t* rows, float* weights, void* stream) {
  return dva_gather_bilinear_taps_anchor(packed_idx, coords, n_atoms, B, H, W, rows, weights, nullptr, stream);
}

int dva_gather_bilinear_taps_anchor(const void* packed_idx, const float* coords, int64_t n_atoms, int32_t B,
                                    int32_t H, int32_t W, int32_t* rows, float* weights, int32_t* anchors,
                                    void* stream) {
  if (n_atoms < 0 || B < 0 || H <= 0 || W <= 0) return DVA_ERR_INVALID;
  if ((int64_t)B * (H + 1) * (W + 1) >= 0x7fffffffLL || n_atoms > 0x1fffffffLL) return DVA_ERR_UNSUPPORTED;
  if (n_atoms == 0) return DVA_OK;
  if (!packed_idx || !coords || !rows || !weights) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(bilinear_taps_kernel, dim3(grid_for(n_atoms)), dim3(256), 0, (hipStream_t)stream,
                     (const PackedIdx*)packed_idx, coords, n_atoms, H, W, rows, weights, anchors,
                     (int32_t)((int64_t)B * (H + 1) * (W + 1)));
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_bilinear_taps_cat(int32_t n_settings, const void* const* tap_rows, const void* const* tap_weights,
                          const void* const* anchors, const int64_t* n_views, const int64_t* n_rows,
                          const int64_t* n_anchors, const int64_t* order, int64_t n_views_total, int32_t* rows_out,
                          float* weights_out, int32_t* anchors_out, void* stream) {
  if (n_settings < 1 || n_views_total < 0 || !tap_rows || !tap_weights || !anchors || !n_views || !n_rows || !n_anchors)
    return DVA_ERR_INVALID;
  if (n_settings > TAPS_CAT_MAX) return DVA_ERR_UNSUPPORTED;
  TapsCatArgs a;
  int64_t v = 0, r = 0, an = 0;
  for (int s = 0; s < TAPS_CAT_MAX; ++s) {
    const bool on = s < n_settings;
    if (on && (n_views[s] < 0 || n_rows[s] < 0 || n_anchors[s] < 0)) return DVA_ERR_INVALID;
    if (on && n_views[s] > 0 && (!tap_rows[s] || !tap_weights[s] || !anchors[s])) return DVA_ERR_INVALID;
    a.rows[s] = on ? (const int4*)tap_rows[s] : nullptr;
    a.weights[s] = on ? (const float4*)tap_weights[s] : nullptr;
    a.anchors[s] = on ? (const int32_t*)anchors[s] : nullptr;
    a.row_off[s] = (int32_t)r;
    a.anchor_off[s] = (int32_t)an;
    a.n_anchor[s] = on ? (int32_t)n_anchors[s] : 0;
    if (on) { v += n_views[s]; r += n_rows[s]; an += n_anchors[s]; }
    a.v_end[s] = v;
  }
  if (v != n_views_total) return DVA_ERR_INVALID;
  if (r >= 0x7fffffffLL || an >= 0x7fffffffLL) return DVA_ERR_UNSUPPORTED;
  if (n_views_total == 0) return DVA_OK;
  if (!rows_out || !weights_out || !anchors_out) return DVA_ERR_INVALID;
  a.n = n_settings;
  a.dummy = (int32_t)an;
  hipLaunchKernelGGL(taps_cat_kernel, dim3(grid_for(n_views_total)), dim3(256), 0, (hipStream_t)stream, a, order,
                     n_views_total, (int4*)rows_out, (float4*)weights_out, anchors_out);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_anchor_combine(const float* S, float* grad_rows, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4)) return DVA_ERR_INVALID;
  if (B == 0) return DVA_OK;
  if (!S || !grad_rows) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(anchor_combine_kernel, dim3(grid_for((int64_t)B * H * W * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, S, grad_rows, (int)B, (int)H, (int)W, (int)C);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_anchor_fixup(const void* grad, const int32_t* rows, const float* weights, const int32_t* anchors,
                     float* grad_rows, int64_t n_atoms, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype,
                     void* stream) {
  if (n_atoms < 0 || C <= 0 || (dtype != DVA_F32 && dtype != DVA_BF16)) return DVA_ERR_INVALID;
  if (n_atoms == 0) return DVA_OK;
  if (!grad || !rows || !weights || !anchors || !grad_rows) return DVA_ERR_INVALID;
  const int32_t dummy = (int32_t)((int64_t)B * (H + 1) * (W + 1));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DVA_F32)
    hipLaunchKernelGGL((anchor_fixup_kernel<float>), dim3(grid_for(n_atoms)), dim3(256), 0, s, (const float*)grad, rows,
                       weights, anchors, dummy, grad_rows, n_atoms, (int)C, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr);
  else
    hipLaunchKernelGGL((anchor_fixup_kernel<bf16_t>), dim3(grid_for(n_atoms)), dim3(256), 0, s, (const bf16_t*)grad,
                       rows, weights, anchors, dummy, grad_rows, n_atoms, (int)C, (const bf16_t*)nullptr,
                       (const float*)nullptr, (const float*)nullptr);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_anchor_fixup_bn(const void* dy_a, const void* z_a, const float* bn_a, const float* sm_a, const int32_t* rows,
                        const float* weights, const int32_t* anchors, float* grad_rows, int64_t n_atoms, int32_t B,
                        int32_t H, int32_t W, int32_t C, void* stream) {
  if (n_atoms < 0 || C <= 0 || (C % 32)) return DVA_ERR_INVALID;
  if (n_atoms == 0) return DVA_OK;
  if (!dy_a || !z_a || !bn_a || !sm_a || !rows || !weights || !anchors || !grad_rows) return DVA_ERR_INVALID;
  const int32_t dummy = (int32_t)((int64_t)B * (H + 1) * (W + 1));
  hipLaunchKernelGGL((anchor_fixup_kernel<bf16_t>), dim3(grid_for(n_atoms)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy_a, rows, weights, anchors, dummy, grad_rows, n_atoms, (int)C, (const bf16_t*)z_a,
                     bn_a, sm_a);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_gather_segment_max_fwd(const void* rows, const int32_t* row_idx, const int64_t* atom_ptr, void* out, void* arg,
                               int64_t n_views, int64_t n_atoms, int64_t n_rows, int32_t C, int32_t dtype, void* stream) {
  if (n_views < 0 || n_atoms < 0 || n_rows < 0 || C < 0 || (dtype != DVA_F32 && dtype != DVA_BF16)) return DVA_ERR_INVALID;
  if (n_views == 0 || C == 0) return DVA_OK;
  if (!rows || !atom_ptr || !out || !arg || (n_atoms > 0 && !row_idx)) return DVA_ERR_INVALID;
  const int vec = dtype == DVA_F32 ? 4 : 8;
  if (C % vec || ((uintptr_t)rows & 15) || ((uintptr_t)out & 15) || ((uintptr_t)arg & 15)) return DVA_ERR_UNSUPPORTED;
  const int64_t total = n_views * (C / vec);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DVA_F32)
    hipLaunchKernelGGL((gather_segment_max_fwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, (const float*)rows,
                       row_idx, atom_ptr, (float*)out, (uint16_t*)arg, n_views, (int)C);
  else
    hipLaunchKernelGGL((gather_segment_max_fwd_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)rows,
                       row_idx, atom_ptr, (bf16_t*)out, (uint16_t*)arg, n_views, (int)C);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_gather_segment_max_bwd(const void* grad_out, const void* arg, const int32_t* row_idx, const int64_t* atom_ptr,
                               const int32_t* perm, const int32_t* row_ptr, const int32_t* view_of_atom, float* grad_rows,
                               int64_t n_views, int64_t n_atoms, int64_t n_rows, int32_t C, int32_t dtype, void* stream) {
  if (n_views < 0 || n_atoms < 0 || n_rows < 0 || C < 0 || (dtype != DVA_F32 && dtype != DVA_BF16)) return DVA_ERR_INVALID;
  if (n_rows == 0 || C == 0) return DVA_OK;
  if (!grad_rows || !atom_ptr) return DVA_ERR_INVALID;
  const int vec = dtype == DVA_F32 ? 4 : 8;
  if (C % vec || ((uintptr_t)grad_out & 15) || ((uintptr_t)arg & 15) || ((uintptr_t)grad_rows & 15)) return DVA_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (perm || row_ptr || view_of_atom) {
    // deterministic: segmented reduction over the row plan of the atoms (grad_rows is written, not accumulated)
    if (!row_ptr || (n_atoms > 0 && (!perm || !view_of_atom || !grad_out || !arg))) return DVA_ERR_INVALID;
    if (n_atoms > 0x7fffffffLL) return DVA_ERR_UNSUPPORTED;
    const int groups = C / vec;
    int lpr = 1;
    while (lpr < groups) lpr <<= 1;
    if (lpr != groups || lpr > 64) return DVA_ERR_UNSUPPORTED;      // whole power-of-two lane teams per row
    int64_t blocks = (n_rows + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (dtype == DVA_F32)
      hipLaunchKernelGGL((gather_segment_max_plan_bwd_kernel<float>), dim3((int)blocks), dim3(256), 0, s,
                         (const float*)grad_out, (const uint16_t*)arg, atom_ptr, perm, row_ptr, view_of_atom, grad_rows,
                         n_rows, (int)C, lpr);
    else
      hipLaunchKernelGGL((gather_segment_max_plan_bwd_kernel<bf16_t>), dim3((int)blocks), dim3(256), 0, s,
                         (const bf16_t*)grad_out, (const uint16_t*)arg, atom_ptr, perm, row_ptr, view_of_atom, grad_rows,
                         n_rows, (int)C, lpr);
    DVA_CHECK_LAUNCH();
    return DVA_OK;
  }
  if (n_views == 0 || n_atoms == 0) return DVA_OK;
  if (!grad_out || !arg || !row_idx) return DVA_ERR_INVALID;
  const int64_t total = n_views * (C / vec);
  if (dtype == DVA_F32)
    hipLaunchKernelGGL((gather_segment_max_bwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, (const float*)grad_out,
                       (const uint16_t*)arg, row_idx, atom_ptr, grad_rows, n_views, (int)C);
  else
    hipLaunchKernelGGL((gather_segment_max_bwd_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)grad_out,
                       (const uint16_t*)arg, row_idx, atom_ptr, grad_rows, n_views, (int)C);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // extern "C"
