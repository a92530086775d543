/*
 * CPU ORACLE (test infrastructure, NOT product code): plain-C restatement of the reference's
 * point->pixel mapping build, the CPU/numba path the authors designate as the one to trust
 * (README.md:122-123).  Every function cites the reference lines it follows (paths relative to
 * /root/reference/torch_points3d/core/multimodal/visibility.py unless stated).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Pinned against the reference itself by tests/test_oracle_mapping.py, which compares it with
 * tests/golden/vis_*.npz — outputs of the reference's own Python source (oracle/gen_golden.py).
 *
 * Float-width contract: float32 inputs stay float32 until they meet a float64 scalar
 * (np.pi, r_min, r_max, voxel, k_swell, np.log(d_swell)), exactly as the golden run was made.
 * float32 transcendentals of the projection (atan2f / acosf) are taken correctly rounded, i.e.
 * computed in float64 and rounded once; BLAS 3x3 products are taken as an FMA chain over k.
 * Build with -ffp-contract=off: every fused operation below is an explicit fmaf().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DVO_CAM_EQUIRECT 0
#define DVO_CAM_PINHOLE_SCANNET 1
#define DVO_CAM_PINHOLE_KITTI 2
#define DVO_CAM_FISHEYE_KITTI 3

/* same layout as struct dva_camera of include/dva.h */
typedef struct dvo_camera {
  int32_t model;
  int32_t img_w, img_h;
  int32_t crop_top, crop_bottom;
  float r_min, r_max; /* informative; the comparisons use r_min_d / r_max_d */
  float img_xyz[3];
  float rot[9];   /* row-major 3x3 used by the projection (see dva.h) */
  float trans[3];
  float fx, fy, mx, my;
  float fisheye[7];
  double r_min_d, r_max_d;
  double voxel, k_swell, d_swell;
  int32_t exact;
} dvo_camera;

static float cr_atan2f(float y, float x) { return (float)atan2((double)y, (double)x); }
static float cr_acosf(float x) { return (float)acos((double)x); }

/* np.remainder for float64 (used by `% width`, :175-176) */
static double np_mod(double a, double b) {
  double m = fmod(a, b);
  if (m != 0.0 && ((b < 0) != (m < 0))) m += b;
  return m;
}

/* fisheye_projection_cpu :288-339 (float32 until the final cast) */
static void fisheye_project(const float p[3], const float* fe, double* x, double* y, double* z) {
  const float xi = fe[0], k1 = fe[1], k2 = fe[2], g1 = fe[3], g2 = fe[4], u0 = fe[5], v0 = fe[6];
  const float norm = sqrtf((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]); /* norm_cpu :129-137 */
  const float den = norm + 1e-4f;
  float fx = p[0] / den, fy = p[1] / den;
  const float fz = p[2] / den;
  fx = fx / (fz + xi);
  fy = fy / (fz + xi);
  const float r2 = fx * fx + fy * fy;
  const float r4 = r2 * r2;
  const float poly = (1.0f + k1 * r2) + k2 * r4;
  *x = (double)((g1 * poly) * fx + u0);
  *y = (double)((g2 * poly) * fy + v0);
  *z = (double)((norm * p[2]) / fabsf(p[2] + 1e-4f));
}

/* camera-frame coordinates of a world point for the pinhole / fisheye models (:232-242, :304-308) */
static void to_camera(const dvo_camera* c, const float* q, float p[3]) {
  const float* R = c->rot;
  if (c->model == DVO_CAM_PINHOLE_SCANNET) {
    /* p = R @ xyz.T + T with (R, T) = inv(extrinsic) */
    for (int j = 0; j < 3; ++j)
      p[j] = fmaf(R[3 * j + 2], q[2], fmaf(R[3 * j + 1], q[1], R[3 * j] * q[0])) + c->trans[j];
  } else {
    /* p = R.T @ (xyz - T).T */
    const float d0 = q[0] - c->trans[0], d1 = q[1] - c->trans[1], d2 = q[2] - c->trans[2];
    for (int j = 0; j < 3; ++j) p[j] = fmaf(R[6 + j], d2, fmaf(R[3 + j], d1, R[j] * d0));
  }
}

/*
 * camera_projection_cpu :478-538.  Outputs the in-range, in-FoV candidates in input order:
 * idx1[m] (index into xyz), dist[m] (f32), xp[m], yp[m] (f64).  Returns m.
 */
int64_t dvo_camera_projection(const float* xyz, int64_t n, const dvo_camera* c, const uint8_t* mask,
                              int64_t* idx1, float* dist, double* xp, double* yp) {
  const int W = c->img_w, H = c->img_h;
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    const float* q = xyz + 3 * i;
    /* dist = norm_cpu(xyz - img_xyz) :509 */
    const float d0 = q[0] - c->img_xyz[0], d1 = q[1] - c->img_xyz[1], d2 = q[2] - c->img_xyz[2];
    const float dd = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
    /* r_min < dist < r_max :510, compared in float64 (numba promotion) */
    if (!(c->r_min_d < (double)dd && (double)dd < c->r_max_d)) continue;
    double x, y, z;
    if (c->model == DVO_CAM_EQUIRECT) {
      /* equirectangular_projection_cpu :150-182 */
      const float* R = c->rot;
      float v[3];
      for (int j = 0; j < 3; ++j) v[j] = fmaf(d2, R[3 * j + 2], fmaf(d1, R[3 * j + 1], d0 * R[3 * j]));
      const float t = cr_atan2f(v[1], v[0]);
      const float p = cr_acosf(v[2] / dd);
      x = np_mod((double)(W - 1) * (1.0 - (double)t / M_PI) / 2.0, (double)W);
      /* (height - 1) * p is a python-int * float32-array product: float32 under NumPy 2 (golden run) */
      y = np_mod((double)((float)(H - 1) * p) / M_PI, (double)H);
      if (isnan(x)) x = 0.0;
      if (isnan(y)) y = 0.0;
      z = 1.0;
    } else {
      float p[3];
      to_camera(c, q, p);
      if (c->model == DVO_CAM_FISHEYE_KITTI) {
        fisheye_project(p, c->fisheye, &x, &y, &z);
      } else {
        /* pinhole_projection_cpu :246-252 (float32, then cast) */
        x = (double)((p[0] * c->fx) / p[2] + c->mx);
        y = (double)((p[1] * c->fy) / p[2] + c->my);
        z = (double)p[2];
      }
    }
    /* field_of_view_cpu :395-435 */
    if (!(0.0 <= x && x < (double)W)) continue;
    if (!((double)c->crop_top <= y && y < (double)(H - c->crop_bottom))) continue;
    if (!(0.0 < z)) continue;
    if (mask) {
      const uint32_t xi = (uint32_t)floor(x), yi = (uint32_t)floor(y);
      if (!mask[(size_t)xi * H + yi]) continue;
    }
    idx1[m] = i;
    dist[m] = dd;
    xp[m] = x;
    yp[m] = y;
    ++m;
  }
  return m;
}

static int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/*
 * Splat boxes: equirectangular_splat_cpu :630-704, pinhole_splat_cpu :761-827, fisheye_splat_cpu
 * :876-953.  xyz_sel = absolute coordinates of the m candidates (fisheye only).  splat[m][4] =
 * (x_a, x_b, y_a, y_b) in NON-cropped coordinates.
 */
void dvo_splat(const double* xp, const double* yp, const float* dist, const float* xyz_sel, int64_t m,
               const dvo_camera* c, int32_t* splat) {
  const int W = c->img_w, H = c->img_h;
  const double logd = log(c->d_swell);
  for (int64_t i = 0; i < m; ++i) {
    double wx, wy;
    if (c->model == DVO_CAM_FISHEYE_KITTI) {
      const float* q = xyz_sel + 3 * i;
      const float da = sqrtf((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]); /* norm of ABSOLUTE xyz :900 */
      const double swell = 1.0 + c->k_swell * exp((double)(-da) / logd);
      float q2[3] = {q[0] + 0.0f, q[1] + 0.0f, q[2] + (float)(swell * c->voxel / 2.0)};
      float p[3];
      double x2, y2, z2;
      to_camera(c, q2, p);
      fisheye_project(p, c->fisheye, &x2, &y2, &z2);
      const double ex = xp[i] - x2, ey = yp[i] - y2;
      wx = wy = 2.0 * sqrt(ex * ex + ey * ey);
    } else {
      const double a = (1.0 + c->k_swell * exp((double)(-dist[i]) / logd)) * c->voxel / (double)dist[i];
      if (c->model == DVO_CAM_EQUIRECT) {
        wy = a * (double)H / M_PI;
        wx = (a * (double)W / (2.0 * M_PI)) / (sin((M_PI / (double)H) * yp[i]) + 0.001);
      } else {
        wx = a * (double)c->fx;
        wy = a * (double)c->fy;
      }
    }
    /* np.round (half to even) into a float32 array, then astype(int32) :676-681 */
    int32_t xa = (int32_t)(float)rint(xp[i] - wx / 2.0);
    int32_t xb = (int32_t)(float)rint(xp[i] + wx / 2.0 + 1.0);
    int32_t ya = (int32_t)(float)rint(yp[i] - wy / 2.0);
    int32_t yb = (int32_t)(float)rint(yp[i] + wy / 2.0 + 1.0);
    const int32_t y_min = c->crop_top, y_max = H - c->crop_bottom;
    splat[4 * i + 0] = clampi(xa, 0, W - 1);
    splat[4 * i + 1] = clampi(xb, 1, W);
    splat[4 * i + 2] = clampi(ya, y_min, y_max - 1);
    splat[4 * i + 3] = clampi(yb, y_min + 1, y_max);
  }
}

/*
 * visibility_from_splatting_cpu :1073-1195: z-buffer with strict '<' in list order (:1147-1162),
 * exact re-splat where the highest seen index wins a shared centre pixel (:1168-1187), output in
 * (x-major, y) order (:1190-1195).  Returns q; idx2[q] are LOCAL indices into the m candidates.
 */
int64_t dvo_zbuffer(const double* xp, const double* yp, const float* dist, const int32_t* splat, int64_t m,
                    const dvo_camera* c, int64_t* idx2, int64_t* x_pix, int64_t* y_pix) {
  const int W = c->img_w, Hc = c->img_h - c->crop_top - c->crop_bottom;
  const size_t npix = (size_t)W * Hc;
  float* depth = (float*)malloc(npix * sizeof(float));
  int64_t* imap = (int64_t*)malloc(npix * sizeof(int64_t));
  float dmax = dist[0];
  for (int64_t i = 1; i < m; ++i) dmax = dist[i] > dmax ? dist[i] : dmax;
  const float init = (float)((double)dmax + 1.0 + 1.0); /* d_max + 1, then + 1 (:1135-1137) */
  for (size_t k = 0; k < npix; ++k) { depth[k] = init; imap[k] = -1; }
  for (int64_t i = 0; i < m; ++i) {
    const int32_t xa = splat[4 * i], xb = splat[4 * i + 1];
    const int32_t ya = splat[4 * i + 2] - c->crop_top, yb = splat[4 * i + 3] - c->crop_top;
    for (int32_t x = xa; x < xb; ++x)
      for (int32_t y = ya; y < yb; ++y) {
        const size_t k = (size_t)x * Hc + y;
        if (dist[i] < depth[k]) { depth[k] = dist[i]; imap[k] = i; }
      }
  }
  if (c->exact) {
    uint8_t* seen = (uint8_t*)calloc((size_t)m, 1);
    for (size_t k = 0; k < npix; ++k) if (imap[k] >= 0) seen[imap[k]] = 1;
    for (size_t k = 0; k < npix; ++k) imap[k] = -1;
    for (int64_t i = 0; i < m; ++i) { /* ascending i: the last (largest) one stays */
      if (!seen[i]) continue;
      const int32_t x = (int32_t)xp[i], y = (int32_t)yp[i] - c->crop_top;
      imap[(size_t)x * Hc + y] = i;
    }
    free(seen);
  }
  int64_t q = 0;
  for (int32_t x = 0; x < W; ++x)
    for (int32_t y = 0; y < Hc; ++y) {
      const int64_t i = imap[(size_t)x * Hc + y];
      if (i < 0) continue;
      idx2[q] = i; x_pix[q] = x; y_pix[q] = y + c->crop_top; ++q;
    }
  free(depth);
  free(imap);
  return q;
}

/*
 * VisibilityModel.__call__ :1699-1757 (projection -> visibility -> gather by idx_1[idx_2]).
 * All output arrays have capacity n.  Returns q (0 when nothing projects, :1721-1729).
 */
int64_t dvo_visibility(const float* xyz, int64_t n, const dvo_camera* c, const uint8_t* mask, int64_t* idx,
                       int64_t* x_pix, int64_t* y_pix, float* depth, double* x_proj, double* y_proj) {
  if (n <= 0) return 0;
  int64_t* idx1 = (int64_t*)malloc((size_t)n * sizeof(int64_t));
  float* dist = (float*)malloc((size_t)n * sizeof(float));
  double* xp = (double*)malloc((size_t)n * sizeof(double));
  double* yp = (double*)malloc((size_t)n * sizeof(double));
  const int64_t m = dvo_camera_projection(xyz, n, c, mask, idx1, dist, xp, yp);
  int64_t q = 0;
  if (m > 0) {
    float* sel = (float*)malloc((size_t)m * 3 * sizeof(float));
    for (int64_t i = 0; i < m; ++i) memcpy(sel + 3 * i, xyz + 3 * idx1[i], 3 * sizeof(float));
    int32_t* splat = (int32_t*)malloc((size_t)m * 4 * sizeof(int32_t));
    const size_t npix = (size_t)c->img_w * (size_t)(c->img_h - c->crop_top - c->crop_bottom);
    int64_t* idx2 = (int64_t*)malloc(((size_t)m > npix ? (size_t)m : npix) * sizeof(int64_t));
    dvo_splat(xp, yp, dist, sel, m, c, splat);
    /* a projection map never holds more winners than candidates in exact mode; in dense mode q can
       exceed m, so the caller sizes the outputs by max(n, W*Hc) */
    q = dvo_zbuffer(xp, yp, dist, splat, m, c, idx2, x_pix, y_pix);
    for (int64_t k = 0; k < q; ++k) {
      const int64_t i = idx2[k];
      idx[k] = idx1[i]; depth[k] = dist[i]; x_proj[k] = xp[i]; y_proj[k] = yp[i];
    }
    free(sel); free(splat); free(idx2);
  }
  free(idx1); free(dist); free(xp); free(yp);
  return q;
}

/*
 * postprocess_features :1548-1582 (+ normalize_dist_cuda :1503-1518, orientation_cuda :1521-1545) for
 * the q mapped points.  Nullable inputs drop their column.  Returns the number of columns.
 */
int32_t dvo_mapping_features(const float* xyz, const int64_t* idx, const float* depth, const double* y_proj,
                             const float* lin, const float* pla, const float* sca, const float* nrm,
                             const dvo_camera* c, int64_t q, float* out) {
  const int32_t ncol = 1 + (lin != 0) + (pla != 0) + (sca != 0) + (nrm != 0) + 1;
  const float rmin = (float)c->r_min_d;                 /* tensor - python float -> float32 scalar */
  const float den = (float)(c->r_max_d + 1e-4);         /* (d_max + 1e-4) in double, then cast */
  for (int64_t k = 0; k < q; ++k) {
    const int64_t i = idx[k];
    float* o = out + (size_t)k * ncol;
    int32_t col = 0;
    o[col++] = (depth[k] - rmin) / den;
    if (lin) o[col++] = lin[i];
    if (pla) o[col++] = pla[i];
    if (sca) o[col++] = sca[i];
    if (nrm) {
      const float dd = depth[k] + 1e-4f;
      const float u0 = (xyz[3 * i] - c->img_xyz[0]) / dd, u1 = (xyz[3 * i + 1] - c->img_xyz[1]) / dd,
                  u2 = (xyz[3 * i + 2] - c->img_xyz[2]) / dd;
      o[col++] = fabsf((u0 * nrm[3 * i] + u1 * nrm[3 * i + 1]) + u2 * nrm[3 * i + 2]);
    }
    o[col++] = (float)(y_proj[k] / (double)c->img_h);
  }
  return ncol;
}

/* Total number of inner z-buffer iterations (sum of box areas): the work unit of the CPU baseline. */
int64_t dvo_splat_area(const int32_t* splat, int64_t m) {
  int64_t a = 0;
  for (int64_t i = 0; i < m; ++i)
    a += (int64_t)(splat[4 * i + 1] - splat[4 * i]) * (splat[4 * i + 3] - splat[4 * i + 2]);
  return a;
}
