// Point -> pixel mapping build for gfx950 (reference: core/multimodal/visibility.py, the CPU/numba
// path: camera_projection_cpu :478-538, *_splat_cpu :630-953, visibility_from_splatting_cpu
// :1073-1195, postprocess_features :1548-1582).
//
// One image per call, candidates in CALLER ORDER (tie-breaks depend on it):
//   project_kernel   range cull + projection + FoV/mask cull per candidate (float-width contract of
//                    float32 geometry, float64 pixel arithmetic, correctly rounded
//                    float32 atan2/acos through float64)
//   exclusive scan + compact_kernel   order-preserving compaction -> local index j = list position
//   splat_kernel     bounding box of every survivor's footprint (float64, round-half-even)
//   zbuffer_kernel   one WAVEFRONT per point, lanes sweep the box pixels, 64-bit atomicMin of
//                    (depth_bits << 32 | j): min depth wins, ties -> smallest j == the reference's
//                    strict '<' in list order (:1157)
//   exact mode       seen-flag pass, then atomicMax of j on the centre pixel: the largest seen j
//                    survives == the reference's ascending overwrite (:1184-1187)
//   flags + scan + emit_kernel   winners in (x-major, y) order (:1190-1195)
// HBM-bound integer/atomic work; the 12-16 B/pixel maps (2048x1024 -> 33 MB) stay in L2/MALL.
#include <cstring>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "dva_common.h"

namespace dva {

// Column of element t of a box of height d (t / d) without an integer division: (t + 0.5) * fl(1 / d) truncated is exact for
// t < 2^22 (the quotient lies at least 0.5 / d from an integer, the two roundings move it by less than (t + 0.5) 2^-23 / d);
// larger boxes divide.  A 32-bit integer division is ~20 instructions per pixel test, this is 4.
struct BoxDiv {
  float rinv;
  int d;
  bool fast;
  __device__ __forceinline__ BoxDiv(int d_, int area) : rinv(1.0f / (float)d_), d(d_), fast(area < (1 << 22)) {}
  __device__ __forceinline__ int col(int t) const { return fast ? (int)(((float)t + 0.5f) * rinv) : t / d; }
};

struct MapCounters {
  int32_t m;  // candidates surviving projection
  int32_t pad;
};

__device__ __forceinline__ double np_mod(double a, double b) {
  double m = fmod(a, b);
  if (m != 0.0 && ((b < 0) != (m < 0))) m += b;
  return m;
}

__device__ __forceinline__ void fisheye_project(const float p0, const float p1, const float p2,
                                                const float* fe, double* x, double* y, double* z) {
  const float xi = fe[0], k1 = fe[1], k2 = fe[2], g1 = fe[3], g2 = fe[4], u0 = fe[5], v0 = fe[6];
  const float norm = sqrtf((p0 * p0 + p1 * p1) + p2 * p2);
  const float den = norm + 1e-4f;
  float fx = p0 / den, fy = p1 / den;
  const float fz = p2 / den;
  fx = fx / (fz + xi);
  fy = fy / (fz + xi);
  const float r2 = fx * fx + fy * fy;
  const float r4 = r2 * r2;
  const float poly = (1.0f + k1 * r2) + k2 * r4;
  *x = (double)((g1 * poly) * fx + u0);
  *y = (double)((g2 * poly) * fy + v0);
  *z = (double)((norm * p2) / fabsf(p2 + 1e-4f));
}

__device__ __forceinline__ void to_camera(const dva_camera& c, float q0, float q1, float q2, float* p) {
  const float* R = c.rot;
  if (c.model == DVA_CAM_PINHOLE_SCANNET) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
      p[j] = fmaf(R[3 * j + 2], q2, fmaf(R[3 * j + 1], q1, R[3 * j] * q0)) + c.trans[j];
  } else {
    const float d0 = q0 - c.trans[0], d1 = q1 - c.trans[1], d2 = q2 - c.trans[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) p[j] = fmaf(R[6 + j], d2, fmaf(R[3 + j], d1, R[j] * d0));
  }
}

// one candidate point: range cull, projection, FoV / mask cull
__device__ __forceinline__ int project_one(const dva_camera& c, const uint8_t* __restrict__ mask, float q0, float q1,
                                           float q2, float* dist_out, double* x_out, double* y_out) {
  const int W = c.img_w, H = c.img_h;
  {
    const float d0 = q0 - c.img_xyz[0], d1 = q1 - c.img_xyz[1], d2 = q2 - c.img_xyz[2];
    const float dd = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
    int keep = (c.r_min_d < (double)dd && (double)dd < c.r_max_d) ? 1 : 0;
    double x = 0.0, y = 0.0, z = 1.0;
    if (keep) {
      if (c.model == DVA_CAM_EQUIRECT) {
        const float* R = c.rot;
        float v[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) v[j] = fmaf(d2, R[3 * j + 2], fmaf(d1, R[3 * j + 1], d0 * R[3 * j]));
        const float t = (float)atan2((double)v[1], (double)v[0]);
        const float p = (float)acos((double)(v[2] / dd));
        x = np_mod((double)(W - 1) * (1.0 - (double)t / M_PI) / 2.0, (double)W);
        y = np_mod((double)((float)(H - 1) * p) / M_PI, (double)H);
        if (isnan(x)) x = 0.0;
        if (isnan(y)) y = 0.0;
      } else {
        float p[3];
        to_camera(c, q0, q1, q2, p);
        if (c.model == DVA_CAM_FISHEYE_KITTI) {
          fisheye_project(p[0], p[1], p[2], c.fisheye, &x, &y, &z);
        } else {
          x = (double)((p[0] * c.fx) / p[2] + c.mx);
          y = (double)((p[1] * c.fy) / p[2] + c.my);
          z = (double)p[2];
        }
      }
      keep = (0.0 <= x && x < (double)W) && ((double)c.crop_top <= y && y < (double)(H - c.crop_bottom)) &&
             (0.0 < z);
      if (keep && mask) {
        const uint32_t xi = (uint32_t)floor(x), yi = (uint32_t)floor(y);
        keep = mask[(size_t)xi * H + yi] ? 1 : 0;
      }
    }
    *dist_out = dd;
    *x_out = x;
    *y_out = y;
    return keep;
  }
}

__global__ __launch_bounds__(256) void project_kernel(const float* __restrict__ xyz, int64_t n,
                                                       const dva_camera c,
                                                       const uint8_t* __restrict__ mask,
                                                       int32_t* __restrict__ flag,
                                                       float* __restrict__ dist_u,
                                                       double* __restrict__ xp_u,
                                                       double* __restrict__ yp_u) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float dd;
    double x, y;
    flag[i] = project_one(c, mask, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], &dd, &x, &y);
    dist_u[i] = dd;
    xp_u[i] = x;
    yp_u[i] = y;
  }
}

__global__ __launch_bounds__(256) void compact_kernel(int64_t n, const int32_t* __restrict__ flag,
                                                       const int32_t* __restrict__ pos,
                                                       const float* __restrict__ dist_u,
                                                       const double* __restrict__ xp_u,
                                                       const double* __restrict__ yp_u,
                                                       int32_t* __restrict__ idx1,
                                                       float* __restrict__ dist,
                                                       double* __restrict__ xp,
                                                       double* __restrict__ yp,
                                                       MapCounters* __restrict__ cnt) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (flag[i]) {
      const int32_t j = pos[i];
      idx1[j] = (int32_t)i;
      dist[j] = dist_u[i];
      xp[j] = xp_u[i];
      yp[j] = yp_u[i];
    }
    if (i == n - 1) cnt->m = pos[i] + flag[i];
  }
}

__device__ __forceinline__ int32_t clampi(int32_t v, int32_t lo, int32_t hi) {
  return v < lo ? lo : (v > hi ? hi : v);
}

// boxes in CROPPED coordinates (y shifted by -crop_top, :1138)
__device__ __forceinline__ int4 splat_one(const dva_camera& c, const float* __restrict__ xyz, int64_t i, float dist_j,
                                          double xp_j, double yp_j) {
  const int W = c.img_w, H = c.img_h;
  const double logd = log(c.d_swell);
  {
    double wx, wy;
    if (c.model == DVA_CAM_FISHEYE_KITTI) {
      const float q0 = xyz[3 * i], q1 = xyz[3 * i + 1], q2 = xyz[3 * i + 2];
      const float da = sqrtf((q0 * q0 + q1 * q1) + q2 * q2);
      const double swell = 1.0 + c.k_swell * exp((double)(-da) / logd);
      const float q2o = q2 + (float)(swell * c.voxel / 2.0);
      float p[3];
      double x2, y2, z2;
      to_camera(c, q0 + 0.0f, q1 + 0.0f, q2o, p);
      fisheye_project(p[0], p[1], p[2], c.fisheye, &x2, &y2, &z2);
      const double ex = xp_j - x2, ey = yp_j - y2;
      wx = wy = 2.0 * sqrt(ex * ex + ey * ey);
    } else {
      const double a = (1.0 + c.k_swell * exp((double)(-dist_j) / logd)) * c.voxel / (double)dist_j;
      if (c.model == DVA_CAM_EQUIRECT) {
        wy = a * (double)H / M_PI;
        wx = (a * (double)W / (2.0 * M_PI)) / (sin((M_PI / (double)H) * yp_j) + 0.001);
      } else {
        wx = a * (double)c.fx;
        wy = a * (double)c.fy;
      }
    }
    const int32_t xa = (int32_t)(float)rint(xp_j - wx / 2.0);
    const int32_t xb = (int32_t)(float)rint(xp_j + wx / 2.0 + 1.0);
    const int32_t ya = (int32_t)(float)rint(yp_j - wy / 2.0);
    const int32_t yb = (int32_t)(float)rint(yp_j + wy / 2.0 + 1.0);
    const int32_t y_min = c.crop_top, y_max = H - c.crop_bottom;
    int4 s;
    s.x = clampi(xa, 0, W - 1);
    s.y = clampi(xb, 1, W);
    s.z = clampi(ya, y_min, y_max - 1) - c.crop_top;
    s.w = clampi(yb, y_min + 1, y_max) - c.crop_top;
    return s;
  }
}

__global__ __launch_bounds__(256) void splat_kernel(const float* __restrict__ xyz,
                                                     const int32_t* __restrict__ idx1,
                                                     const float* __restrict__ dist,
                                                     const double* __restrict__ xp,
                                                     const double* __restrict__ yp, const dva_camera c,
                                                     const MapCounters* __restrict__ cnt,
                                                     int4* __restrict__ splat) {
  const int m = cnt->m;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x)
    splat[j] = splat_one(c, xyz, idx1[j], dist[j], xp[j], yp[j]);
}

// One wavefront per point; lanes sweep the box column-major like the map ([x][y], y fastest).
__global__ __launch_bounds__(256) void zbuffer_kernel(const int4* __restrict__ splat,
                                                       const float* __restrict__ dist,
                                                       const MapCounters* __restrict__ cnt,
                                                       unsigned long long* __restrict__ zbuf, int Hc) {
  const int m = cnt->m;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * blockDim.x) >> 6;
  for (int j = wave; j < m; j += n_waves) {
    const int4 s = splat[j];
    const int bw = s.y - s.x, bh = s.w - s.z;
    const int area = bw * bh;
    const unsigned long long key = ((unsigned long long)__float_as_uint(dist[j]) << 32) | (unsigned)j;
    const BoxDiv bd(bh, area);
    for (int t = lane; t < area; t += 64) {
      const int bx = bd.col(t), by = t - bx * bh;
      unsigned long long* cell = zbuf + (size_t)(s.x + bx) * Hc + (s.z + by);
      if (key < *cell) atomicMin(cell, key);  // cheap pre-test; atomicMin decides
    }
  }
}

__global__ __launch_bounds__(256) void seen_kernel(const unsigned long long* __restrict__ zbuf,
                                                    int64_t npix, uint8_t* __restrict__ seen) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < npix;
       k += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long z = zbuf[k];
    if (z != ~0ull) seen[(uint32_t)z] = 1;
  }
}

__global__ __launch_bounds__(256) void resplat_kernel(const uint8_t* __restrict__ seen,
                                                       const double* __restrict__ xp,
                                                       const double* __restrict__ yp,
                                                       const MapCounters* __restrict__ cnt,
                                                       int32_t* __restrict__ pixmap, int Hc,
                                                       int crop_top) {
  const int m = cnt->m;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x) {
    if (!seen[j]) continue;
    const int x = (int)xp[j], y = (int)yp[j] - crop_top;  // astype(np.int32): truncation (:1177-1178)
    atomicMax(&pixmap[(size_t)x * Hc + y], j);
  }
}

__global__ __launch_bounds__(256) void winners_kernel(const unsigned long long* __restrict__ zbuf,
                                                       int64_t npix, int32_t* __restrict__ pixmap) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < npix;
       k += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long z = zbuf[k];
    pixmap[k] = (z == ~0ull) ? -1 : (int32_t)(uint32_t)z;
  }
}

__global__ __launch_bounds__(256) void pixflag_kernel(const int32_t* __restrict__ pixmap, int64_t npix,
                                                       int32_t* __restrict__ pixflag) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < npix;
       k += (int64_t)gridDim.x * blockDim.x)
    pixflag[k] = pixmap[k] >= 0 ? 1 : 0;
}

__global__ __launch_bounds__(256) void emit_kernel(
    const int32_t* __restrict__ pixmap, const int32_t* __restrict__ pixflag,
    const int32_t* __restrict__ pixpos, int64_t npix, int Hc, int crop_top,
    const int32_t* __restrict__ idx1, const float* __restrict__ dist, const double* __restrict__ xp,
    const double* __restrict__ yp, int64_t* __restrict__ idx, int64_t* __restrict__ x_pix,
    int64_t* __restrict__ y_pix, float* __restrict__ depth, double* __restrict__ x_proj,
    double* __restrict__ y_proj, int64_t* __restrict__ n_out) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < npix;
       k += (int64_t)gridDim.x * blockDim.x) {
    if (pixflag[k]) {
      const int32_t o = pixpos[k], j = pixmap[k];
      idx[o] = idx1[j];
      x_pix[o] = k / Hc;
      y_pix[o] = k % Hc + crop_top;
      depth[o] = dist[j];
      x_proj[o] = xp[j];
      y_proj[o] = yp[j];
    }
    if (k == npix - 1) *n_out = (int64_t)pixpos[k] + pixflag[k];
  }
}

__global__ __launch_bounds__(256) void features_kernel(
    const float* __restrict__ xyz, const int64_t* __restrict__ idx, const float* __restrict__ depth,
    const double* __restrict__ y_proj, const float* __restrict__ lin, const float* __restrict__ pla,
    const float* __restrict__ sca, const float* __restrict__ nrm, const dva_camera c, int64_t q,
    int ncol, float* __restrict__ out) {
  const float rmin = (float)c.r_min_d;
  const float den = (float)(c.r_max_d + 1e-4);
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < q;
       k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx[k];
    float* o = out + k * ncol;
    int col = 0;
    o[col++] = (depth[k] - rmin) / den;
    if (lin) o[col++] = lin[i];
    if (pla) o[col++] = pla[i];
    if (sca) o[col++] = sca[i];
    if (nrm) {
      const float dd = depth[k] + 1e-4f;
      const float u0 = (xyz[3 * i] - c.img_xyz[0]) / dd, u1 = (xyz[3 * i + 1] - c.img_xyz[1]) / dd,
                  u2 = (xyz[3 * i + 2] - c.img_xyz[2]) / dd;
      o[col++] = fabsf((u0 * nrm[3 * i] + u1 * nrm[3 * i + 1]) + u2 * nrm[3 * i + 2]);
    }
    o[col++] = (float)(y_proj[k] / (double)c.img_h);
  }
}

// ------------------------------------------------------------------------------------------------
// Batched build: B images of ONE setting (same projection size, crops, camera model, splatting parameters; a pose per
// image) against the same candidate cloud in one set of launches.  Every per-image array gains an image axis laid
// out image-major, so that ONE scan compacts the survivors of all images (image b owns [moff[b], moff[b + 1])) and ONE
// scan orders the output rows (image-major, inside an image x-major then y like the single-image build).  The
// survivor index j that breaks depth ties is the GLOBAL position in the compacted list: monotone in the local
// position inside an image, and keys of different images never meet (each image has its own z-buffer plane).
// The only host round trip is the total row count, once per batch.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void project_batch_kernel(const float* __restrict__ xyz, int64_t n,
                                                             const dva_camera* __restrict__ cams,
                                                             const uint8_t* __restrict__ mask,
                                                             int32_t* __restrict__ flag, float* __restrict__ dist_u,
                                                             double* __restrict__ xp_u, double* __restrict__ yp_u) {
  __shared__ dva_camera c;
  if (threadIdx.x == 0) c = cams[blockIdx.y];
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.y * n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float dd;
    double x, y;
    flag[base + i] = project_one(c, mask, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], &dd, &x, &y);
    dist_u[base + i] = dd;
    xp_u[base + i] = x;
    yp_u[base + i] = y;
  }
}

__global__ __launch_bounds__(256) void compact_batch_kernel(int64_t n, int B, const int32_t* __restrict__ flag,
                                                             const int32_t* __restrict__ pos,
                                                             const float* __restrict__ dist_u,
                                                             const double* __restrict__ xp_u,
                                                             const double* __restrict__ yp_u,
                                                             int32_t* __restrict__ idx1, int32_t* __restrict__ simg,
                                                             float* __restrict__ dist, double* __restrict__ xp,
                                                             double* __restrict__ yp, MapCounters* __restrict__ cnt) {
  const int64_t total = n * B;        // <= 2^31 - 1 (entry precondition): a 32-bit division
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    if (flag[i]) {
      const int32_t j = pos[i];
      const int b = (int)((uint32_t)i / (uint32_t)n);
      idx1[j] = (int32_t)(i - (int64_t)b * n);
      simg[j] = b;
      dist[j] = dist_u[i];
      xp[j] = xp_u[i];
      yp[j] = yp_u[i];
    }
    if (i == total - 1) cnt->m = pos[i] + flag[i];
  }
}

__global__ __launch_bounds__(256) void splat_batch_kernel(const float* __restrict__ xyz,
                                                           const int32_t* __restrict__ idx1,
                                                           const int32_t* __restrict__ simg,
                                                           const float* __restrict__ dist,
                                                           const double* __restrict__ xp,
                                                           const double* __restrict__ yp,
                                                           const dva_camera* __restrict__ cams,
                                                           const MapCounters* __restrict__ cnt,
                                                           int4* __restrict__ splat) {
  const int m = cnt->m;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x)
    splat[j] = splat_one(cams[simg[j]], xyz, idx1[j], dist[j], xp[j], yp[j]);
}

// Footprints range from ~16 pixels (far points: the bulk) to several hundred (near points).  A wavefront takes FOUR
// consecutive survivors at a time: if all four boxes are small (<= 128 pixels) each gets a quarter of the wavefront
// (16 lanes, <= 8 sweeps) and the four are swept concurrently; otherwise the whole wavefront sweeps them one after
// the other (the single-image kernel's scheme).  Measured on the S3DIS setting (23 pixels per box on average):
// a wavefront per point keeps 36 % of the lanes busy.
__global__ __launch_bounds__(256) void zbuffer_batch_kernel(const int4* __restrict__ splat,
                                                             const int32_t* __restrict__ simg,
                                                             const float* __restrict__ dist,
                                                             const MapCounters* __restrict__ cnt,
                                                             unsigned long long* __restrict__ zbuf, int Hc,
                                                             int64_t npix) {
  const int m = cnt->m;
  const int lane = threadIdx.x & 63, quarter = lane >> 4, ql = lane & 15;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * blockDim.x) >> 6;
  for (int j0 = wave * 4; j0 < m; j0 += n_waves * 4) {
    // lane group `quarter` looks at survivor j0 + quarter
    const int jq = j0 + quarter;
    int4 sq = make_int4(0, 0, 0, 0);
    if (jq < m) sq = splat[jq];
    const int area_q = (sq.y - sq.x) * (sq.w - sq.z);
    int amax = area_q;
    amax = max(amax, __shfl_xor(amax, 16));
    amax = max(amax, __shfl_xor(amax, 32));
    if (amax <= 128) {
      if (jq < m) {
        const int bh = sq.w - sq.z;
        const unsigned long long key = ((unsigned long long)__float_as_uint(dist[jq]) << 32) | (unsigned)jq;
        unsigned long long* plane = zbuf + (size_t)simg[jq] * (size_t)npix;
        const BoxDiv bd(bh, area_q);
        for (int t = ql; t < area_q; t += 16) {
          const int bx = bd.col(t), by = t - bx * bh;
          unsigned long long* cell = plane + (size_t)(sq.x + bx) * Hc + (sq.z + by);
          if (key < *cell) atomicMin(cell, key);
        }
      }
    } else {
      for (int u = 0; u < 4 && j0 + u < m; ++u) {
        const int j = j0 + u;
        const int4 s = splat[j];
        const int bh = s.w - s.z;
        const int area = (s.y - s.x) * bh;
        const unsigned long long key = ((unsigned long long)__float_as_uint(dist[j]) << 32) | (unsigned)j;
        unsigned long long* plane = zbuf + (size_t)simg[j] * (size_t)npix;
        const BoxDiv bd(bh, area);
        for (int t = lane; t < area; t += 64) {
          const int bx = bd.col(t), by = t - bx * bh;
          unsigned long long* cell = plane + (size_t)(s.x + bx) * Hc + (s.z + by);
          if (key < *cell) atomicMin(cell, key);
        }
      }
    }
  }
}

// ---- tiled z-buffer of the batched build ------------------------------------------------------------------------------
// 147 M pixel tests per batch of 32 images through 64-bit L2 atomics on a map-sized plane took 1.8 of the 3.9 ms of a batch
// (with or without a load in front of the atomic: scattered atomics run at 5 - 20 G requests/s, whatever they do).  Here
// the survivors are binned into 32 x 32-pixel screen tiles first (count, scan, fill: ~1.3 list entries per survivor; the
// tile counters of a block's chunk of survivors live in LDS, so the global atomics are one per (block, tile) instead of
// one per entry), and one workgroup per tile keeps the tile's z-buffer in LDS (8 KiB, ds_min_u64), walks the tile's list
// (16-byte entries {survivor, depth bits, box}: read once, coalesced) and writes the winners out once: the map-sized
// 64-bit plane is neither cleared, nor written, nor read again.  min over the keys (depth bits | survivor index) is
// order independent, so the winners are those of the atomic z-buffer, bit for bit.
//   * a survivor covering more than ZT_BIG tiles goes to a per-image list that every tile of the image clips against;
//   * whatever does not fit (tile lists beyond their capacity of 4 entries per candidate, more than ZT_BIGCAP large
//     boxes per image) falls back to the global atomic plane, which the tile kernel then merges at start (n_fb > 0).
// (ZT_CHUNK: 16384 survivors per block left 1.5 blocks per CU at the S3DIS setting -- 2048: fill pass 212 -> 163 us per batch)
constexpr int ZT = 32, ZT_BIG = 16, ZT_BIGCAP = 4096, ZT_CHUNK = 2048, ZT_BLOCK = 1024;
// ctl: int32 [B + 1] = large boxes per image | n_fb (fallback survivors)

__device__ __forceinline__ int4 tile_entry(int j, float d, const int4& s) {
  return make_int4(j, (int)__float_as_uint(d), s.x | (s.y << 16), s.z | (s.w << 16));
}

// FILL = false: tile_count += entries per tile.  FILL = true: the entries themselves (tile_off = exclusive scan of
// tile_count).  One block per chunk of ZT_CHUNK consecutive survivors.  The compaction is image-major (simg is
// non-decreasing), so a chunk mostly touches the tiles of the image of its first survivor and the next one: their
// counters are in LDS (hist / base [n_img * T], n_img <= 2); survivors of any other image use the global counters.
template <bool FILL>
__global__ __launch_bounds__(ZT_BLOCK) void tile_bin_kernel(
    const int4* __restrict__ splat, const int32_t* __restrict__ simg, const float* __restrict__ dist,
    const MapCounters* __restrict__ cnt, int Tx, int Ty, int B, int n_img, int32_t* __restrict__ tile_count,
    const int32_t* __restrict__ tile_off, int32_t* __restrict__ tile_cursor, int4* __restrict__ list, int64_t cap,
    int32_t* __restrict__ ctl, int4* __restrict__ big_list, int32_t* __restrict__ fb_list) {
  extern __shared__ int32_t sm[];
  const int m = cnt->m, T = Tx * Ty, HN = n_img * T;
  int32_t* hist = sm;
  int32_t* base = sm + HN;      // FILL only
  for (int64_t c0 = (int64_t)blockIdx.x * ZT_CHUNK; c0 < m; c0 += (int64_t)gridDim.x * ZT_CHUNK) {
    const int c1 = (int)(c0 + ZT_CHUNK < m ? c0 + ZT_CHUNK : m);
    const int b_lo = simg[c0];
    for (int i = threadIdx.x; i < HN; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    // pass 1: entries per tile of this chunk
    for (int j = (int)c0 + threadIdx.x; j < c1; j += blockDim.x) {
      const int4 s = splat[j];
      if (s.y <= s.x || s.w <= s.z) continue;
      const int b = simg[j];
      const int tx0 = s.x / ZT, tx1 = (s.y - 1) / ZT, ty0 = s.z / ZT, ty1 = (s.w - 1) / ZT;
      if ((tx1 - tx0 + 1) * (ty1 - ty0 + 1) > ZT_BIG) continue;
      const bool local = (unsigned)(b - b_lo) < (unsigned)n_img;
      for (int tx = tx0; tx <= tx1; ++tx)
        for (int ty = ty0; ty <= ty1; ++ty) {
          if (local) atomicAdd(&hist[(b - b_lo) * T + tx * Ty + ty], 1);
          else if (!FILL) atomicAdd(&tile_count[(b * Tx + tx) * Ty + ty], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HN; i += blockDim.x) {
      const int c = hist[i];
      if (c) {
        if (!FILL) atomicAdd(&tile_count[b_lo * T + i], c);
        else base[i] = atomicAdd(&tile_cursor[b_lo * T + i], c);
      }
      hist[i] = 0;
    }
    __syncthreads();
    if (FILL) {
      // pass 2: the entries
      for (int j = (int)c0 + threadIdx.x; j < c1; j += blockDim.x) {
        const int4 s = splat[j];
        if (s.y <= s.x || s.w <= s.z) continue;
        const int b = simg[j];
        const int4 ent = tile_entry(j, dist[j], s);
        const int tx0 = s.x / ZT, tx1 = (s.y - 1) / ZT, ty0 = s.z / ZT, ty1 = (s.w - 1) / ZT;
        bool fb = false;
        if ((tx1 - tx0 + 1) * (ty1 - ty0 + 1) > ZT_BIG) {
          const int slot = atomicAdd(&ctl[b], 1);
          if (slot < ZT_BIGCAP) big_list[(int64_t)b * ZT_BIGCAP + slot] = ent;
          else fb = true;
        } else {
          const bool local = (unsigned)(b - b_lo) < (unsigned)n_img;
          for (int tx = tx0; tx <= tx1; ++tx)
            for (int ty = ty0; ty <= ty1; ++ty) {
              const int g = (b * Tx + tx) * Ty + ty;
              int slot;
              if (local) {
                const int i = (b - b_lo) * T + tx * Ty + ty;
                slot = base[i] + atomicAdd(&hist[i], 1);
              } else {
                slot = atomicAdd(&tile_cursor[g], 1);
              }
              const int64_t pos = (int64_t)tile_off[g] + slot;
              if (pos >= 0 && pos < cap) list[pos] = ent;      // (pos < 0: the int32 scan of the counts wrapped)
              else fb = true;
            }
        }
        if (fb) fb_list[atomicAdd(&ctl[B], 1)] = j;
      }
      __syncthreads();
    }
  }
}

// the atomic plane is cleared only when a survivor fell back to it
__global__ __launch_bounds__(256) void zbuffer_cond_clear_kernel(const int32_t* __restrict__ ctl, int B,
                                                                  unsigned long long* __restrict__ zbuf, int64_t n) {
  if (ctl[B] <= 0) return;
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
    zbuf[k] = ~0ull;
}

// fallback survivors: the atomic plane (one wavefront per survivor)
__global__ __launch_bounds__(256) void zbuffer_fallback_kernel(const int32_t* __restrict__ fb_list,
                                                                const int32_t* __restrict__ ctl, int B,
                                                                const int4* __restrict__ splat,
                                                                const int32_t* __restrict__ simg,
                                                                const float* __restrict__ dist,
                                                                unsigned long long* __restrict__ zbuf, int Hc,
                                                                int64_t npix) {
  const int n_fb = ctl[B];
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  for (int e = wave; e < n_fb; e += n_waves) {
    const int j = fb_list[e];
    const int4 s = splat[j];
    const int bh = s.w - s.z, area = (s.y - s.x) * bh;
    const unsigned long long key = ((unsigned long long)__float_as_uint(dist[j]) << 32) | (unsigned)j;
    unsigned long long* plane = zbuf + (size_t)simg[j] * (size_t)npix;
    const BoxDiv bd(bh, area);
    for (int t = lane; t < area; t += 64) {
      const int bx = bd.col(t), by = t - bx * bh;
      atomicMin(plane + (size_t)(s.x + bx) * Hc + (s.z + by), key);
    }
  }
}

// one workgroup per (image, tile): winners of the tile.  exact: seen[winner] = 1 and the tile of pixmap cleared to -1
// (the re-splat writes it next); otherwise pixmap = winner index or -1.
__global__ __launch_bounds__(256) void tile_raster_kernel(
    const int32_t* __restrict__ tile_off, const int32_t* __restrict__ tile_count, const int4* __restrict__ list,
    int64_t cap, const int32_t* __restrict__ ctl, const int4* __restrict__ big_list,
    const unsigned long long* __restrict__ zbuf, uint8_t* __restrict__ seen, int32_t* __restrict__ pixmap, int W,
    int Hc, int Tx, int Ty, int B, int exact) {
  // column pitch ZP = 40 entries (8 bytes each): the 16 lanes that rasterise one box touch up to four columns of <= 8 rows, which
  // land on different LDS banks (a pitch of 32 put every column of a box on the same banks: a 5-way conflict per ds_min_u64)
  constexpr int ZP = 40;
  __shared__ unsigned long long z[ZT * ZP];
  __shared__ int4 ent[256];
  const int tile = blockIdx.x, b = tile / (Tx * Ty), r = tile - b * Tx * Ty, tx = r / Ty, ty = r - tx * Ty;
  const int x0 = tx * ZT, y0 = ty * ZT;
  const int64_t npix = (int64_t)W * Hc;
  // the tile's list: its first 256 entries are requested before anything else (round 4: a tile holds ~130 entries, so this
  // one load was the exposed latency of the block)
  const int64_t off = tile_off[tile];
  int64_t len = tile_count[tile];
  if (off < 0) len = 0;
  else if (off + len > cap) len = cap > off ? cap - off : 0;
  int4 first = make_int4(0, 0, 0, 0);
  if ((int64_t)threadIdx.x < len) first = list[off + threadIdx.x];
  const bool merge = ctl[B] > 0;
  int nb = ctl[b];
  if (nb > ZT_BIGCAP) nb = ZT_BIGCAP;
  for (int i = threadIdx.x; i < ZT * ZT; i += blockDim.x) {
    const int gx = x0 + (i >> 5), gy = y0 + (i & 31);
    z[(i >> 5) * ZP + (i & 31)] = (merge && gx < W && gy < Hc) ? zbuf[(size_t)b * npix + (size_t)gx * Hc + gy] : ~0ull;
  }
  const int grp = threadIdx.x >> 4, gl = threadIdx.x & 15;
  auto walk = [&](const int4* src, int64_t len_, bool have_first) {
    for (int64_t e0 = 0; e0 < len_; e0 += 256) {
      const int n = (int)(len_ - e0 < 256 ? len_ - e0 : 256);
      __syncthreads();                       // z initialised / the previous batch of entries consumed
      if ((int)threadIdx.x < n) ent[threadIdx.x] = (have_first && e0 == 0) ? first : src[e0 + threadIdx.x];
      __syncthreads();
      for (int e = grp; e < n; e += 16) {
        const int4 en = ent[e];
        const int sx0 = en.z & 0xffff, sx1 = (int)((uint32_t)en.z >> 16), sy0 = en.w & 0xffff,
                  sy1 = (int)((uint32_t)en.w >> 16);
        const int cx0 = max(sx0, x0), cx1 = min(sx1, x0 + ZT), cy0 = max(sy0, y0), cy1 = min(sy1, y0 + ZT);
        const int w = cx1 - cx0, hh = cy1 - cy0;
        if (w <= 0 || hh <= 0) continue;
        const unsigned long long key = ((unsigned long long)(uint32_t)en.y << 32) | (uint32_t)en.x;
        const int area = w * hh;                     // <= 1024: the multiply form of t / hh is exact even with the 1-ulp
        const float rinv = __builtin_amdgcn_rcpf((float)hh);       // reciprocal (see BoxDiv), 24-bit products suffice
        const int base = (cx0 - x0) * ZP + (cy0 - y0);
        for (int t = gl; t < area; t += 16) {
          const int bx = (int)(((float)t + 0.5f) * rinv), by = t - (int)__umul24((uint32_t)bx, (uint32_t)hh);
          atomicMin(&z[base + bx * ZP + by], key);
        }
      }
    }
  };
  walk(list + off, len, true);
  walk(big_list + (int64_t)b * ZT_BIGCAP, nb, false);
  __syncthreads();
  for (int i = threadIdx.x; i < ZT * ZT; i += blockDim.x) {
    const int gx = x0 + (i >> 5), gy = y0 + (i & 31);
    if (gx >= W || gy >= Hc) continue;
    const int zi = (i >> 5) * ZP + (i & 31);
    const unsigned long long k = z[zi];
    const int32_t win = k == ~0ull ? -1 : (int32_t)(uint32_t)k;
    if (exact) {
      pixmap[(size_t)b * npix + (size_t)gx * Hc + gy] = -1;
      // a winner covers ~20 pixels of the tile: only the pixels whose upper / left neighbour in the tile has another
      // winner mark it (round 4: one scattered byte store per visible region instead of one per pixel)
      if (win >= 0 && !(((i & 31) && (int32_t)(uint32_t)z[zi - 1] == win && z[zi - 1] != ~0ull) ||
                        ((i >> 5) && (int32_t)(uint32_t)z[zi - ZP] == win && z[zi - ZP] != ~0ull)))
        seen[win] = 1;
    } else {
      pixmap[(size_t)b * npix + (size_t)gx * Hc + gy] = win;
    }
  }
}

__global__ __launch_bounds__(256) void resplat_batch_kernel(const uint8_t* __restrict__ seen,
                                                             const int32_t* __restrict__ simg,
                                                             const double* __restrict__ xp,
                                                             const double* __restrict__ yp,
                                                             const MapCounters* __restrict__ cnt,
                                                             int32_t* __restrict__ pixmap, int Hc, int crop_top,
                                                             int64_t npix) {
  const int m = cnt->m;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x) {
    if (!seen[j]) continue;
    const int x = (int)xp[j], y = (int)yp[j] - crop_top;
    atomicMax(&pixmap[(size_t)simg[j] * (size_t)npix + (size_t)x * Hc + y], j);
  }
}

// Output order of the batched build without a map-sized scan: mapped pixels (pixmap >= 0) are counted per block of
// EB_PIX consecutive pixels, the (few) block counts are scanned, and the emit kernel re-derives the position of a pixel
// inside its block from ballots: the plane is read twice (2 x 4 bytes per pixel) instead of read / written / read twice.
constexpr int EB_PER_THREAD = 8, EB_PIX = 256 * EB_PER_THREAD;

__global__ __launch_bounds__(256) void pix_block_count_kernel(const int32_t* __restrict__ pixmap, int64_t total,
                                                               int32_t* __restrict__ block_count) {
  __shared__ int s_c[4];
  const int64_t k0 = (int64_t)blockIdx.x * EB_PIX;
  int c = 0;
#pragma unroll
  for (int i = 0; i < EB_PER_THREAD; ++i) {
    const int64_t k = k0 + (int64_t)i * 256 + threadIdx.x;
    c += (k < total && pixmap[k] >= 0) ? 1 : 0;
  }
  for (int off = 1; off < 64; off <<= 1) c += __shfl_xor(c, off);
  if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}

__global__ __launch_bounds__(256) void emit_blocks_kernel(
    const int32_t* __restrict__ pixmap, const int32_t* __restrict__ block_off, int64_t n_blocks, int64_t npix, int B,
    int Hc, int crop_top, const int32_t* __restrict__ idx1, const float* __restrict__ dist,
    const double* __restrict__ xp, const double* __restrict__ yp, int64_t* __restrict__ idx,
    int64_t* __restrict__ x_pix, int64_t* __restrict__ y_pix, float* __restrict__ depth, double* __restrict__ x_proj,
    double* __restrict__ y_proj, int64_t* __restrict__ row_ptr, int64_t* __restrict__ n_out) {
  __shared__ int s_w[EB_PER_THREAD][4];
  const int64_t total = npix * B;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t k0 = (int64_t)blockIdx.x * EB_PIX;
  // round 4: the eight pixel loads of a thread are issued together and the block synchronises once (the loop of rounds
  // 2-3 had a dependent load -> ballot -> barrier -> gather chain per pass: 353 us per batch of 32 images, latency bound)
  int32_t jv[EB_PER_THREAD];
  unsigned long long bal[EB_PER_THREAD];
#pragma unroll
  for (int i = 0; i < EB_PER_THREAD; ++i) {
    const int64_t k = k0 + (int64_t)i * 256 + threadIdx.x;
    jv[i] = k < total ? pixmap[k] : -1;
  }
#pragma unroll
  for (int i = 0; i < EB_PER_THREAD; ++i) {
    bal[i] = __ballot(jv[i] >= 0);
    if (lane == 0) s_w[i][wv] = __popcll(bal[i]);
  }
  __syncthreads();
  int run = block_off[blockIdx.x];             // rows before the pixels of this pass
  const int b0 = (int)(k0 / npix);
#pragma unroll
  for (int i = 0; i < EB_PER_THREAD; ++i) {
    const int64_t k = k0 + (int64_t)i * 256 + threadIdx.x;
    const int32_t j = jv[i];
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      before += w < wv ? s_w[i][w] : 0;
      all += s_w[i][w];
    }
    const int o = run + before + __popcll(bal[i] & ((1ull << lane) - 1ull));
    if (k < total) {
      // image / pixel of k: one 64-bit division per block (b0), then at most a few subtractions -- not a 64-bit division
      // per pixel; the 32-bit column division only for mapped pixels (npix B <= 2^31 - 1 is the entry's precondition)
      int b = b0;
      int64_t kk = k - (int64_t)b0 * npix;
      while (kk >= npix) {
        kk -= npix;
        ++b;
      }
      if (kk == 0) row_ptr[b] = o;                 // rows of the images before b
      if (j >= 0) {
        const uint32_t xc = (uint32_t)kk / (uint32_t)Hc;
        idx[o] = idx1[j];
        x_pix[o] = xc;
        y_pix[o] = (int64_t)((uint32_t)kk - xc * (uint32_t)Hc) + crop_top;
        depth[o] = dist[j];
        x_proj[o] = xp[j];
        y_proj[o] = yp[j];
      }
      if (k == total - 1) {
        const int64_t q = (int64_t)o + (j >= 0 ? 1 : 0);
        row_ptr[B] = q;
        *n_out = q;
      }
    }
    run += all;
  }
}

// mapping features of the rows of a batched build: the camera of row k is the image whose row range holds k
__global__ __launch_bounds__(256) void features_batch_kernel(
    const float* __restrict__ xyz, const int64_t* __restrict__ idx, const float* __restrict__ depth,
    const double* __restrict__ y_proj, const float* __restrict__ lin, const float* __restrict__ pla,
    const float* __restrict__ sca, const float* __restrict__ nrm, const dva_camera* __restrict__ cams,
    const int64_t* __restrict__ row_ptr, int B, int64_t q, int ncol, float* __restrict__ out,
    int32_t* __restrict__ row_image) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < q; k += (int64_t)gridDim.x * blockDim.x) {
    int lo = 0, hi = B;                                  // largest b with row_ptr[b] <= k
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (row_ptr[mid] <= k) lo = mid;
      else hi = mid;
    }
    const dva_camera& c = cams[lo];
    if (row_image) row_image[k] = lo;
    const float rmin = (float)c.r_min_d;
    const float den = (float)(c.r_max_d + 1e-4);
    const int64_t i = idx[k];
    float* o = out + k * ncol;
    int col = 0;
    o[col++] = (depth[k] - rmin) / den;
    if (lin) o[col++] = lin[i];
    if (pla) o[col++] = pla[i];
    if (sca) o[col++] = sca[i];
    if (nrm) {
      const float dd = depth[k] + 1e-4f;
      const float u0 = (xyz[3 * i] - c.img_xyz[0]) / dd, u1 = (xyz[3 * i + 1] - c.img_xyz[1]) / dd,
                  u2 = (xyz[3 * i + 2] - c.img_xyz[2]) / dd;
      o[col++] = fabsf((u0 * nrm[3 * i] + u1 * nrm[3 * i + 1]) + u2 * nrm[3 * i + 2]);
    }
    o[col++] = (float)(y_proj[k] / (double)c.img_h);
  }
}

static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct VisLayout {
  size_t flag, pos, dist_u, xp_u, yp_u, idx1, dist, xp, yp, splat, seen, cnt, zbuf, pixmap, pixflag,
      pixpos, temp, temp_bytes, total;
};

static int vis_layout(const dva_camera* c, int64_t n, VisLayout* L) {
  const int64_t Hc = (int64_t)c->img_h - c->crop_top - c->crop_bottom;
  if (c->img_w <= 0 || Hc <= 0) return DVA_ERR_INVALID;
  const size_t npix = (size_t)c->img_w * (size_t)Hc;
  const size_t big = (size_t)n > npix ? (size_t)n : npix;
  size_t scan_tmp = 0;
  int32_t* nul = nullptr;
  if (rocprim::exclusive_scan(nullptr, scan_tmp, nul, nul, 0, big, rocprim::plus<int32_t>(),
                              (hipStream_t)0) != hipSuccess)
    return DVA_ERR_LAUNCH;
  size_t o = 0;
  L->flag = o;    o += al((size_t)n * 4);
  L->pos = o;     o += al((size_t)n * 4);
  L->dist_u = o;  o += al((size_t)n * 4);
  L->xp_u = o;    o += al((size_t)n * 8);
  L->yp_u = o;    o += al((size_t)n * 8);
  L->idx1 = o;    o += al((size_t)n * 4);
  L->dist = o;    o += al((size_t)n * 4);
  L->xp = o;      o += al((size_t)n * 8);
  L->yp = o;      o += al((size_t)n * 8);
  L->splat = o;   o += al((size_t)n * 16);
  L->seen = o;    o += al((size_t)n);
  L->cnt = o;     o += al(sizeof(MapCounters));
  L->zbuf = o;    o += al(npix * 8);
  L->pixmap = o;  o += al(npix * 4);
  L->pixflag = o; o += al(npix * 4);
  L->pixpos = o;  o += al(npix * 4);
  L->temp = o;
  L->temp_bytes = scan_tmp;
  o += al(scan_tmp);
  L->total = o;
  return DVA_OK;
}

static inline int grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

struct VisBatchLayout {
  size_t flag, pos, dist_u, xp_u, yp_u, idx1, simg, dist, xp, yp, splat, seen, cnt, zbuf, pixmap, pixflag, pixpos,
      temp, temp_bytes, total;
  // tiled z-buffer: tile_count | tile_cursor | ctl are one zero-filled region
  size_t tile_count, tile_cursor, ctl, tile_zero_bytes, tile_off, list, big_list, fb_list;
  int64_t list_cap;
  int Tx, Ty;
};

static int vis_batch_layout(const dva_camera* c, int64_t n, int64_t B, VisBatchLayout* L) {
  const int64_t Hc = (int64_t)c->img_h - c->crop_top - c->crop_bottom;
  if (c->img_w <= 0 || Hc <= 0 || B < 1) return DVA_ERR_INVALID;
  const size_t npix = (size_t)c->img_w * (size_t)Hc * (size_t)B, nc = (size_t)n * (size_t)B;
  const size_t big = nc > npix ? nc : npix;
  size_t scan_tmp = 0;
  int32_t* nul = nullptr;
  if (rocprim::exclusive_scan(nullptr, scan_tmp, nul, nul, 0, big, rocprim::plus<int32_t>(), (hipStream_t)0) !=
      hipSuccess)
    return DVA_ERR_LAUNCH;
  size_t o = 0;
  L->flag = o;    o += al(nc * 4);
  L->pos = o;     o += al(nc * 4);
  L->dist_u = o;  o += al(nc * 4);
  L->xp_u = o;    o += al(nc * 8);
  L->yp_u = o;    o += al(nc * 8);
  L->idx1 = o;    o += al(nc * 4);
  L->simg = o;    o += al(nc * 4);
  L->dist = o;    o += al(nc * 4);
  L->xp = o;      o += al(nc * 8);
  L->yp = o;      o += al(nc * 8);
  L->splat = o;   o += al(nc * 16);
  L->seen = o;    o += al(nc);
  L->cnt = o;     o += al(sizeof(MapCounters));
  L->zbuf = o;    o += al(npix * 8);
  L->pixmap = o;  o += al(npix * 4);
  L->pixflag = o; o += al(npix * 4);
  L->pixpos = o;  o += al(npix * 4);
  L->temp = o;
  L->temp_bytes = scan_tmp;
  o += al(scan_tmp);
  L->Tx = (int)((c->img_w + ZT - 1) / ZT);
  L->Ty = (int)((Hc + ZT - 1) / ZT);
  const size_t nt = (size_t)L->Tx * (size_t)L->Ty * (size_t)B;
  L->tile_count = o;  o += al(nt * 4);
  L->tile_cursor = o; o += al(nt * 4);
  L->ctl = o;         o += al((size_t)(B + 1) * 4);
  L->tile_zero_bytes = o - L->tile_count;
  L->tile_off = o;    o += al(nt * 4);
  L->list_cap = (int64_t)nc * 4;
  L->list = o;        o += al((size_t)L->list_cap * 16);
  L->big_list = o;    o += al((size_t)B * ZT_BIGCAP * 16);
  L->fb_list = o;     o += al(nc * 4);
  L->total = o;
  return DVA_OK;
}

// survivors of the projection to the caller's arrays (dva_camera_projection): int64 indices, count
__global__ __launch_bounds__(256) void projection_out_kernel(int64_t n, const int32_t* __restrict__ flag,
                                                              const int32_t* __restrict__ pos,
                                                              const float* __restrict__ dist_u,
                                                              const double* __restrict__ xp_u,
                                                              const double* __restrict__ yp_u,
                                                              int64_t* __restrict__ idx, float* __restrict__ dist,
                                                              double* __restrict__ xp, double* __restrict__ yp,
                                                              int64_t* __restrict__ n_out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (flag[i]) {
      const int32_t j = pos[i];
      idx[j] = i;
      dist[j] = dist_u[i];
      xp[j] = xp_u[i];
      yp[j] = yp_u[i];
    }
    if (i == n - 1) *n_out = (int64_t)pos[i] + flag[i];
  }
}

}  // namespace dva

using namespace dva;

extern "C" {

// camera_projection alone (reference core/multimodal/visibility.py:478-538): range / field-of-view / crop / mask cull and
// the float projection of the survivors, in candidate order -- what the visibility models other than the splatting one
// start from (DepthBasedVisibility, BiasuttiVisibility: visibility.py:1356-1496).  Workspace: dva_visibility_workspace_bytes.
int dva_camera_projection(const float* xyz, int64_t n, const dva_camera* cam, const uint8_t* mask, int64_t* idx,
                          float* depth, double* x_proj, double* y_proj, int64_t* n_out_dev, void* workspace,
                          int64_t workspace_bytes, void* stream) {
  if (!cam || n < 0 || !n_out_dev) return DVA_ERR_INVALID;
  if (cam->model < DVA_CAM_EQUIRECT || cam->model > DVA_CAM_FISHEYE_KITTI) return DVA_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    if (hipMemsetAsync(n_out_dev, 0, sizeof(int64_t), s) != hipSuccess) return DVA_ERR_LAUNCH;
    return DVA_OK;
  }
  if (n > 0x7fffffff) return DVA_ERR_UNSUPPORTED;
  if (!xyz || !idx || !depth || !x_proj || !y_proj || !workspace) return DVA_ERR_INVALID;
  VisLayout L;
  int rc = vis_layout(cam, n, &L);
  if (rc) return rc;
  if ((int64_t)L.total > workspace_bytes) return DVA_ERR_INVALID;
  char* ws = (char*)workspace;
  int32_t* flag = (int32_t*)(ws + L.flag);
  int32_t* pos = (int32_t*)(ws + L.pos);
  float* dist_u = (float*)(ws + L.dist_u);
  double* xp_u = (double*)(ws + L.xp_u);
  double* yp_u = (double*)(ws + L.yp_u);
  const dva_camera c = *cam;
  hipLaunchKernelGGL(project_kernel, dim3(grid_for(n)), dim3(256), 0, s, xyz, n, c, mask, flag, dist_u, xp_u, yp_u);
  size_t tmp = L.temp_bytes;
  if (rocprim::exclusive_scan(ws + L.temp, tmp, flag, pos, 0, (size_t)n, rocprim::plus<int32_t>(), s) != hipSuccess)
    return DVA_ERR_LAUNCH;
  hipLaunchKernelGGL(projection_out_kernel, dim3(grid_for(n)), dim3(256), 0, s, n, flag, pos, dist_u, xp_u, yp_u, idx,
                     depth, x_proj, y_proj, n_out_dev);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int64_t dva_visibility_workspace_bytes(const dva_camera* cam, int64_t n) {
  if (!cam || n < 0) return DVA_ERR_INVALID;
  VisLayout L;
  int rc = vis_layout(cam, n > 0 ? n : 1, &L);
  if (rc) return rc;
  return (int64_t)L.total;
}

int dva_visibility(const float* xyz, int64_t n, const dva_camera* cam, const uint8_t* mask,
                   int64_t* idx, int64_t* x_pix, int64_t* y_pix, float* depth, double* x_proj,
                   double* y_proj, int64_t* n_out_dev, void* workspace, int64_t workspace_bytes,
                   void* stream) {
  if (!cam || n < 0 || !n_out_dev) return DVA_ERR_INVALID;
  if (cam->model < DVA_CAM_EQUIRECT || cam->model > DVA_CAM_FISHEYE_KITTI) return DVA_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    if (hipMemsetAsync(n_out_dev, 0, sizeof(int64_t), s) != hipSuccess) return DVA_ERR_LAUNCH;
    return DVA_OK;
  }
  if (n > 0x7fffffff) return DVA_ERR_UNSUPPORTED;
  if (!xyz || !idx || !x_pix || !y_pix || !depth || !x_proj || !y_proj || !workspace)
    return DVA_ERR_INVALID;
  VisLayout L;
  int rc = vis_layout(cam, n, &L);
  if (rc) return rc;
  if ((int64_t)L.total > workspace_bytes) return DVA_ERR_INVALID;
  char* ws = (char*)workspace;
  const int Hc = cam->img_h - cam->crop_top - cam->crop_bottom;
  const int64_t npix = (int64_t)cam->img_w * Hc;
  int32_t* flag = (int32_t*)(ws + L.flag);
  int32_t* pos = (int32_t*)(ws + L.pos);
  float* dist_u = (float*)(ws + L.dist_u);
  double* xp_u = (double*)(ws + L.xp_u);
  double* yp_u = (double*)(ws + L.yp_u);
  int32_t* idx1 = (int32_t*)(ws + L.idx1);
  float* dist = (float*)(ws + L.dist);
  double* xp = (double*)(ws + L.xp);
  double* yp = (double*)(ws + L.yp);
  int4* splat = (int4*)(ws + L.splat);
  uint8_t* seen = (uint8_t*)(ws + L.seen);
  MapCounters* cnt = (MapCounters*)(ws + L.cnt);
  unsigned long long* zbuf = (unsigned long long*)(ws + L.zbuf);
  int32_t* pixmap = (int32_t*)(ws + L.pixmap);
  int32_t* pixflag = (int32_t*)(ws + L.pixflag);
  int32_t* pixpos = (int32_t*)(ws + L.pixpos);

  const dva_camera c = *cam;
  hipLaunchKernelGGL(project_kernel, dim3(grid_for(n)), dim3(256), 0, s, xyz, n, c, mask, flag, dist_u,
                     xp_u, yp_u);
  size_t tmp = L.temp_bytes;
  if (rocprim::exclusive_scan(ws + L.temp, tmp, flag, pos, 0, (size_t)n, rocprim::plus<int32_t>(), s) !=
      hipSuccess)
    return DVA_ERR_LAUNCH;
  hipLaunchKernelGGL(compact_kernel, dim3(grid_for(n)), dim3(256), 0, s, n, flag, pos, dist_u, xp_u, yp_u,
                     idx1, dist, xp, yp, cnt);
  hipLaunchKernelGGL(splat_kernel, dim3(grid_for(n)), dim3(256), 0, s, xyz, idx1, dist, xp, yp, c, cnt,
                     splat);
  if (hipMemsetAsync(zbuf, 0xFF, (size_t)npix * 8, s) != hipSuccess) return DVA_ERR_LAUNCH;
  {
    int64_t blocks = (n + 3) / 4;  // 4 wavefronts (points) per block
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(zbuffer_kernel, dim3((int)blocks), dim3(256), 0, s, splat, dist, cnt, zbuf, Hc);
  }
  if (c.exact) {
    if (hipMemsetAsync(seen, 0, (size_t)n, s) != hipSuccess) return DVA_ERR_LAUNCH;
    if (hipMemsetAsync(pixmap, 0xFF, (size_t)npix * 4, s) != hipSuccess) return DVA_ERR_LAUNCH;
    hipLaunchKernelGGL(seen_kernel, dim3(grid_for(npix)), dim3(256), 0, s, zbuf, npix, seen);
    hipLaunchKernelGGL(resplat_kernel, dim3(grid_for(n)), dim3(256), 0, s, seen, xp, yp, cnt, pixmap, Hc,
                       c.crop_top);
  } else {
    hipLaunchKernelGGL(winners_kernel, dim3(grid_for(npix)), dim3(256), 0, s, zbuf, npix, pixmap);
  }
  hipLaunchKernelGGL(pixflag_kernel, dim3(grid_for(npix)), dim3(256), 0, s, pixmap, npix, pixflag);
  tmp = L.temp_bytes;
  if (rocprim::exclusive_scan(ws + L.temp, tmp, pixflag, pixpos, 0, (size_t)npix,
                              rocprim::plus<int32_t>(), s) != hipSuccess)
    return DVA_ERR_LAUNCH;
  hipLaunchKernelGGL(emit_kernel, dim3(grid_for(npix)), dim3(256), 0, s, pixmap, pixflag, pixpos, npix, Hc,
                     c.crop_top, idx1, dist, xp, yp, idx, x_pix, y_pix, depth, x_proj, y_proj, n_out_dev);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int64_t dva_visibility_batch_workspace_bytes(const dva_camera* cam, int64_t n, int32_t n_images) {
  if (!cam || n < 0 || n_images < 1) return DVA_ERR_INVALID;
  VisBatchLayout L;
  int rc = vis_batch_layout(cam, n > 0 ? n : 1, n_images, &L);
  if (rc) return rc;
  return (int64_t)L.total;
}

int dva_visibility_batch(const float* xyz, int64_t n, const dva_camera* cam0, const dva_camera* cams_dev,
                         int32_t n_images, const uint8_t* mask, int64_t* idx, int64_t* x_pix, int64_t* y_pix,
                         float* depth, double* x_proj, double* y_proj, int64_t* row_ptr, int64_t* n_out_dev,
                         void* workspace, int64_t workspace_bytes, void* stream) {
  if (!cam0 || !cams_dev || n < 0 || n_images < 1 || !n_out_dev || !row_ptr) return DVA_ERR_INVALID;
  if (cam0->model < DVA_CAM_EQUIRECT || cam0->model > DVA_CAM_FISHEYE_KITTI) return DVA_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  const int B = n_images;
  if (n == 0) {
    if (hipMemsetAsync(n_out_dev, 0, sizeof(int64_t), s) != hipSuccess) return DVA_ERR_LAUNCH;
    if (hipMemsetAsync(row_ptr, 0, sizeof(int64_t) * (B + 1), s) != hipSuccess) return DVA_ERR_LAUNCH;
    return DVA_OK;
  }
  const int Hc = cam0->img_h - cam0->crop_top - cam0->crop_bottom;
  const int64_t npix = (int64_t)cam0->img_w * Hc;
  if (n * B > 0x7fffffffLL || npix * B > 0x7fffffffLL) return DVA_ERR_UNSUPPORTED;
  if (!xyz || !idx || !x_pix || !y_pix || !depth || !x_proj || !y_proj || !workspace) return DVA_ERR_INVALID;
  VisBatchLayout L;
  int rc = vis_batch_layout(cam0, n, B, &L);
  if (rc) return rc;
  if ((int64_t)L.total > workspace_bytes) return DVA_ERR_INVALID;
  char* ws = (char*)workspace;
  int32_t* flag = (int32_t*)(ws + L.flag);
  int32_t* pos = (int32_t*)(ws + L.pos);
  float* dist_u = (float*)(ws + L.dist_u);
  double* xp_u = (double*)(ws + L.xp_u);
  double* yp_u = (double*)(ws + L.yp_u);
  int32_t* idx1 = (int32_t*)(ws + L.idx1);
  int32_t* simg = (int32_t*)(ws + L.simg);
  float* dist = (float*)(ws + L.dist);
  double* xp = (double*)(ws + L.xp);
  double* yp = (double*)(ws + L.yp);
  int4* splat = (int4*)(ws + L.splat);
  uint8_t* seen = (uint8_t*)(ws + L.seen);
  MapCounters* cnt = (MapCounters*)(ws + L.cnt);
  unsigned long long* zbuf = (unsigned long long*)(ws + L.zbuf);
  int32_t* pixmap = (int32_t*)(ws + L.pixmap);
  int32_t* pixflag = (int32_t*)(ws + L.pixflag);
  int32_t* pixpos = (int32_t*)(ws + L.pixpos);
  const int64_t nc = n * B, npt = npix * B;

  {
    int gx = grid_for(n);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(project_batch_kernel, dim3(gx, B), dim3(256), 0, s, xyz, n, cams_dev, mask, flag, dist_u, xp_u,
                       yp_u);
  }
  size_t tmp = L.temp_bytes;
  if (rocprim::exclusive_scan(ws + L.temp, tmp, flag, pos, 0, (size_t)nc, rocprim::plus<int32_t>(), s) != hipSuccess)
    return DVA_ERR_LAUNCH;
  hipLaunchKernelGGL(compact_batch_kernel, dim3(grid_for(nc)), dim3(256), 0, s, n, B, flag, pos, dist_u, xp_u, yp_u,
                     idx1, simg, dist, xp, yp, cnt);
  hipLaunchKernelGGL(splat_batch_kernel, dim3(grid_for(nc)), dim3(256), 0, s, xyz, idx1, simg, dist, xp, yp, cams_dev,
                     cnt, splat);
  static const int tiled = tune_int("DVA_ZBUF_TILED", 1);
  // (box corners are packed into 16 bits; tile_off is an int32 scan of up to ZT_BIG entries per candidate: beyond
  //  2^31 - 1 possible entries the whole batch takes the atomic plane)
  if (tiled && cam0->img_w < 65536 && Hc < 65536 && nc * ZT_BIG <= 0x7fffffffLL) {
    int32_t* tile_count = (int32_t*)(ws + L.tile_count);
    int32_t* tile_cursor = (int32_t*)(ws + L.tile_cursor);
    int32_t* ctl = (int32_t*)(ws + L.ctl);
    int32_t* tile_off = (int32_t*)(ws + L.tile_off);
    int4* list = (int4*)(ws + L.list);
    int4* big_list = (int4*)(ws + L.big_list);
    int32_t* fb_list = (int32_t*)(ws + L.fb_list);
    const size_t nt = (size_t)L.Tx * (size_t)L.Ty * (size_t)B;
    const int T = L.Tx * L.Ty;
    const int n_img = T <= 4096 ? 2 : (T <= 8192 ? 1 : 0);        // images whose tile counters fit the block's LDS
    const size_t lds = (size_t)n_img * T * 4;
    int bin_blocks = (int)((nc + ZT_CHUNK - 1) / ZT_CHUNK);
    if (bin_blocks > 4096) bin_blocks = 4096;
    if (hipMemsetAsync(ws + L.tile_count, 0, L.tile_zero_bytes, s) != hipSuccess) return DVA_ERR_LAUNCH;
    hipLaunchKernelGGL((tile_bin_kernel<false>), dim3(bin_blocks), dim3(ZT_BLOCK), lds, s, splat, simg, dist, cnt, L.Tx,
                       L.Ty, B, n_img, tile_count, tile_off, tile_cursor, list, L.list_cap, ctl, big_list, fb_list);
    tmp = L.temp_bytes;
    if (rocprim::exclusive_scan(ws + L.temp, tmp, tile_count, tile_off, 0, nt, rocprim::plus<int32_t>(), s) !=
        hipSuccess)
      return DVA_ERR_LAUNCH;
    hipLaunchKernelGGL((tile_bin_kernel<true>), dim3(bin_blocks), dim3(ZT_BLOCK), 2 * lds, s, splat, simg, dist, cnt,
                       L.Tx, L.Ty, B, n_img, tile_count, tile_off, tile_cursor, list, L.list_cap, ctl, big_list,
                       fb_list);
    hipLaunchKernelGGL(zbuffer_cond_clear_kernel, dim3(grid_for(npt)), dim3(256), 0, s, ctl, B, zbuf, npt);
    hipLaunchKernelGGL(zbuffer_fallback_kernel, dim3(256), dim3(256), 0, s, fb_list, ctl, B, splat, simg, dist, zbuf,
                       Hc, npix);
    if (cam0->exact && hipMemsetAsync(seen, 0, (size_t)nc, s) != hipSuccess) return DVA_ERR_LAUNCH;
    hipLaunchKernelGGL(tile_raster_kernel, dim3((unsigned)nt), dim3(256), 0, s, tile_off, tile_count, list, L.list_cap,
                       ctl, big_list, zbuf, seen, pixmap, (int)cam0->img_w, Hc, L.Tx, L.Ty, B, (int)cam0->exact);
    if (cam0->exact)
      hipLaunchKernelGGL(resplat_batch_kernel, dim3(grid_for(nc)), dim3(256), 0, s, seen, simg, xp, yp, cnt, pixmap,
                         Hc, cam0->crop_top, npix);
  } else {
    if (hipMemsetAsync(zbuf, 0xFF, (size_t)npt * 8, s) != hipSuccess) return DVA_ERR_LAUNCH;
    {
      int64_t blocks = (nc + 15) / 16;          // 4 wavefronts x 4 survivors per block iteration
      if (blocks > 256 * 32) blocks = 256 * 32;
      hipLaunchKernelGGL(zbuffer_batch_kernel, dim3((int)blocks), dim3(256), 0, s, splat, simg, dist, cnt, zbuf, Hc,
                         npix);
    }
    if (cam0->exact) {
      if (hipMemsetAsync(seen, 0, (size_t)nc, s) != hipSuccess) return DVA_ERR_LAUNCH;
      if (hipMemsetAsync(pixmap, 0xFF, (size_t)npt * 4, s) != hipSuccess) return DVA_ERR_LAUNCH;
      hipLaunchKernelGGL(seen_kernel, dim3(grid_for(npt)), dim3(256), 0, s, zbuf, npt, seen);
      hipLaunchKernelGGL(resplat_batch_kernel, dim3(grid_for(nc)), dim3(256), 0, s, seen, simg, xp, yp, cnt, pixmap, Hc,
                         cam0->crop_top, npix);
    } else {
      hipLaunchKernelGGL(winners_kernel, dim3(grid_for(npt)), dim3(256), 0, s, zbuf, npt, pixmap);
    }
}
  // output position of every mapped pixel: per-block counts, a scan over the blocks, positions inside a block from
  // ballots (the map-sized exclusive scan of round 3a: one write + one read of a B x map-sized plane more)
  {
    const int64_t n_blocks = (npt + EB_PIX - 1) / EB_PIX;
    int32_t* block_count = pixflag;                       // [n_blocks] (the flag plane is otherwise unused here)
    int32_t* block_off = pixpos;
    hipLaunchKernelGGL(pix_block_count_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, pixmap, npt, block_count);
    tmp = L.temp_bytes;
    if (rocprim::exclusive_scan(ws + L.temp, tmp, block_count, block_off, 0, (size_t)n_blocks,
                                rocprim::plus<int32_t>(), s) != hipSuccess)
      return DVA_ERR_LAUNCH;
    hipLaunchKernelGGL(emit_blocks_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, pixmap, block_off, n_blocks, npix,
                       B, Hc, cam0->crop_top, idx1, dist, xp, yp, idx, x_pix, y_pix, depth, x_proj, y_proj, row_ptr,
                       n_out_dev);
  }
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_mapping_features_batch(const float* xyz, const int64_t* idx, const float* depth, const double* y_proj,
                               const float* linearity, const float* planarity, const float* scattering,
                               const float* normals, const dva_camera* cams_dev, const int64_t* row_ptr,
                               int32_t n_images, int64_t q, float* features, int32_t* row_image, int32_t* n_cols,
                               void* stream) {
  if (!cams_dev || !row_ptr || q < 0 || n_images < 1) return DVA_ERR_INVALID;
  const int ncol = 2 + (linearity != nullptr) + (planarity != nullptr) + (scattering != nullptr) +
                   (normals != nullptr);
  if (n_cols) *n_cols = ncol;
  if (q == 0) return DVA_OK;
  if (!xyz || !idx || !depth || !y_proj || !features) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(features_batch_kernel, dim3(grid_for(q)), dim3(256), 0, (hipStream_t)stream, xyz, idx, depth,
                     y_proj, linearity, planarity, scattering, normals, cams_dev, row_ptr, (int)n_images, q, ncol,
                     features, row_image);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

int dva_mapping_features(const float* xyz, const int64_t* idx, const float* depth,
                         const double* y_proj, const float* linearity, const float* planarity,
                         const float* scattering, const float* normals, const dva_camera* cam,
                         int64_t q, float* features, int32_t* n_cols, void* stream) {
  if (!cam || q < 0) return DVA_ERR_INVALID;
  const int ncol = 2 + (linearity != nullptr) + (planarity != nullptr) + (scattering != nullptr) +
                   (normals != nullptr);
  if (n_cols) *n_cols = ncol;
  if (q == 0) return DVA_OK;
  if (!xyz || !idx || !depth || !y_proj || !features) return DVA_ERR_INVALID;
  hipLaunchKernelGGL(features_kernel, dim3(grid_for(q)), dim3(256), 0, (hipStream_t)stream, xyz, idx,
                     depth, y_proj, linearity, planarity, scattering, normals, *cam, q, ncol, features);
  DVA_CHECK_LAUNCH();
  return DVA_OK;
}

}  // extern "C"
