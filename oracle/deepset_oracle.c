/* TEST INFRASTRUCTURE ONLY (oracle): plain C + OpenMP restatement of the mapping-feature encoder of the view pooling --
 * DeepSetFeat followed by the score layer -- forward and backward in train mode, on the host cores.  Imported / linked
 * only by tests/ and bench.py's cpu_baseline leg -- never by the product path.
 *
 * Reference maths (paths under /root/reference/torch_points3d/):
 *   MLP block = Linear(bias=False) -> FastBatchNorm1d -> LeakyReLU(0.2)      core/common_modules/base_modules.py:38-48
 *   DeepSetFeat.forward: mlp_elt_1 -> max over the views of a point (+ sqrt(1 / (n + 1e-3)) with use_num) -> mlp_set ->
 *     gather back to the views -> concatenation -> mlp_elt_2              modules/multimodal/pooling.py:604-673
 *   E_score = Linear(32, G, bias=True)                                        modules/multimodal/pooling.py:282
 *   segment max with empty segments -> 0, gradient to the first maximal row  torch_scatter segment_csr semantics
 * Pinned by tests/test_oracle_deepset_c.py against oracle/pooling_oracle.py (itself pinned on the reference's golden
 * vectors).  The orchestration (which block feeds which) is in oracle/deepset_oracle.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DO 32 /* output width of every block */

int oracle_deepset_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* z [M][32] = x [M][K] W^T (W [32][K]); train-mode batch statistics (biased variance, fp64 sums);
 * a = leaky(gamma (z - mean) invstd + beta).  mean, invstd: [32] out. */
/* W [32][K] -> Wt [K][32]: the 32 outputs of a row are the contiguous (vectorised) dimension of every inner loop */
static float* transpose_w(const float* W, int K) {
  float* Wt = (float*)malloc(sizeof(float) * (size_t)K * DO);
  for (int o = 0; o < DO; ++o)
    for (int k = 0; k < K; ++k) Wt[k * DO + o] = W[(int64_t)o * K + k];
  return Wt;
}
#define CHUNK 256 /* rows per statistics chunk: fp32 partial sums inside a chunk, fp64 across chunks */

void oracle_block_fwd(const float* x, int64_t M, int K, const float* W, const float* gamma, const float* beta, float eps,
                      float* z, float* a, float* mean, float* invstd) {
  double s1[DO] = {0}, s2[DO] = {0};
  float* Wt = transpose_w(W, K);
#pragma omp parallel
  {
    double t1[DO] = {0}, t2[DO] = {0};
#pragma omp for schedule(static)
    for (int64_t r0 = 0; r0 < M; r0 += CHUNK) {
      float c1[DO] = {0}, c2[DO] = {0};
      const int64_t r1 = r0 + CHUNK < M ? r0 + CHUNK : M;
      for (int64_t r = r0; r < r1; ++r) {
        const float* xr = x + r * K;
        float acc[DO] = {0};
        for (int k = 0; k < K; ++k) {
          const float xk = xr[k];
          const float* w = Wt + k * DO;
#pragma omp simd
          for (int o = 0; o < DO; ++o) acc[o] = fmaf(xk, w[o], acc[o]);
        }
        float* zr = z + r * DO;
#pragma omp simd
        for (int o = 0; o < DO; ++o) {
          zr[o] = acc[o];
          c1[o] += acc[o];
          c2[o] = fmaf(acc[o], acc[o], c2[o]);
        }
      }
      for (int o = 0; o < DO; ++o) { t1[o] += c1[o]; t2[o] += c2[o]; }
    }
#pragma omp critical
    for (int o = 0; o < DO; ++o) { s1[o] += t1[o]; s2[o] += t2[o]; }
  }
  free(Wt);
  float g[DO], b[DO];
  for (int o = 0; o < DO; ++o) {
    const double mu = M > 0 ? s1[o] / (double)M : 0.0;
    double var = M > 0 ? s2[o] / (double)M - mu * mu : 0.0;
    if (var < 0) var = 0;
    mean[o] = (float)mu;
    invstd[o] = (float)(1.0 / sqrt(var + (double)eps));
    g[o] = gamma[o] * invstd[o];
    b[o] = beta[o] - mean[o] * g[o];
  }
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < M; ++r) {
#pragma omp simd
    for (int o = 0; o < DO; ++o) {
      const float y = fmaf(z[r * DO + o], g[o], b[o]);
      a[r * DO + o] = y > 0.f ? y : 0.2f * y;
    }
  }
}

/* backward of one block: da [M][32] -> dx [M][K] (nullable), dW [32][K], dgamma, dbeta [32] */
void oracle_block_bwd(const float* x, const float* z, const float* da, int64_t M, int K, const float* W,
                      const float* gamma, const float* beta, const float* mean, const float* invstd, float* dx,
                      float* dW, float* dgamma, float* dbeta) {
  double s1[DO] = {0}, s2[DO] = {0};
  float g[DO], b[DO];
  for (int o = 0; o < DO; ++o) {
    g[o] = gamma[o] * invstd[o];
    b[o] = beta[o] - mean[o] * g[o];
  }
#pragma omp parallel
  {
    double t1[DO] = {0}, t2[DO] = {0};
#pragma omp for schedule(static)
    for (int64_t r0 = 0; r0 < M; r0 += CHUNK) {
      float c1[DO] = {0}, c2[DO] = {0};
      const int64_t r1 = r0 + CHUNK < M ? r0 + CHUNK : M;
      for (int64_t r = r0; r < r1; ++r) {
#pragma omp simd
        for (int o = 0; o < DO; ++o) {
          const float zz = z[r * DO + o];
          const float y = fmaf(zz, g[o], b[o]);
          const float dy = y > 0.f ? da[r * DO + o] : 0.2f * da[r * DO + o];
          c1[o] += dy;
          c2[o] = fmaf(dy, (zz - mean[o]) * invstd[o], c2[o]);
        }
      }
      for (int o = 0; o < DO; ++o) { t1[o] += c1[o]; t2[o] += c2[o]; }
    }
#pragma omp critical
    for (int o = 0; o < DO; ++o) { s1[o] += t1[o]; s2[o] += t2[o]; }
  }
  float k1[DO], k2[DO];
  for (int o = 0; o < DO; ++o) {
    dbeta[o] = (float)s1[o];
    dgamma[o] = (float)s2[o];
    k1[o] = M > 0 ? (float)(s1[o] / (double)M) : 0.f;
    k2[o] = M > 0 ? (float)(s2[o] / (double)M) : 0.f;
  }
  const int nt = oracle_deepset_num_threads();
  float* Wt = transpose_w(W, K);
  double* part = (double*)calloc((size_t)nt * DO * K, sizeof(double));      /* [thread][k][o] */
#pragma omp parallel
  {
#ifdef _OPENMP
    double* mine = part + (size_t)omp_get_thread_num() * DO * K;
#else
    double* mine = part;
#endif
    float* cw = (float*)malloc(sizeof(float) * (size_t)K * DO);              /* fp32 partial of one chunk */
#pragma omp for schedule(static)
    for (int64_t r0 = 0; r0 < M; r0 += CHUNK) {
      memset(cw, 0, sizeof(float) * (size_t)K * DO);
      const int64_t r1 = r0 + CHUNK < M ? r0 + CHUNK : M;
      for (int64_t r = r0; r < r1; ++r) {
        const float* xr = x + r * K;
        float dz[DO];
#pragma omp simd
        for (int o = 0; o < DO; ++o) {
          const float zz = z[r * DO + o];
          const float y = fmaf(zz, g[o], b[o]);
          const float dy = y > 0.f ? da[r * DO + o] : 0.2f * da[r * DO + o];
          const float zh = (zz - mean[o]) * invstd[o];
          dz[o] = g[o] * (dy - k1[o] - zh * k2[o]);
        }
        for (int k = 0; k < K; ++k) {
          const float xk = xr[k];
          float* c = cw + k * DO;
          const float* w = Wt + k * DO;
          float acc = 0.f;
#pragma omp simd reduction(+ : acc)
          for (int o = 0; o < DO; ++o) {
            c[o] = fmaf(dz[o], xk, c[o]);
            acc = fmaf(dz[o], w[o], acc);
          }
          if (dx) dx[r * K + k] = acc;
        }
      }
      for (int i = 0; i < K * DO; ++i) mine[i] += cw[i];
    }
    free(cw);
  }
  for (int o = 0; o < DO; ++o)
    for (int k = 0; k < K; ++k) {
      double acc = 0.0;
      for (int t = 0; t < nt; ++t) acc += part[(size_t)t * DO * K + k * DO + o];
      dW[(int64_t)o * K + k] = (float)acc;
    }
  free(part);
  free(Wt);
}

/* pooled [N][32] = max over the views of each point (0 for points without views), arg int64 [N][32] = first maximal view */
void oracle_segmax_fwd(const float* a, const int64_t* csr, int64_t N, float* pooled, int64_t* arg) {
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t p = 0; p < N; ++p) {
    for (int o = 0; o < DO; ++o) {
      float m = 0.f;
      int64_t am = -1;
      for (int64_t v = csr[p]; v < csr[p + 1]; ++v) {
        const float t = a[v * DO + o];
        if (am < 0 || t > m) { m = t; am = v; }
      }
      pooled[p * DO + o] = m;
      arg[p * DO + o] = am;
    }
  }
}
/* da [V][32] += dpooled routed to the arg views */
void oracle_segmax_bwd(const float* dpooled, const int64_t* arg, int64_t N, float* da) {
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t p = 0; p < N; ++p)
    for (int o = 0; o < DO; ++o) {
      const int64_t v = arg[p * DO + o];
      if (v >= 0) da[v * DO + o] += dpooled[p * DO + o];      /* a point's views belong to it alone: no race */
    }
}
/* cat [V][64] = [a2[v] | s[point(v)]] and its backward (da2 [V][32] = left half, ds [N][32] = segment sum of the right) */
void oracle_concat_fwd(const float* a2, const float* s, const int64_t* csr, int64_t N, float* cat) {
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t p = 0; p < N; ++p)
    for (int64_t v = csr[p]; v < csr[p + 1]; ++v) {
      memcpy(cat + v * 2 * DO, a2 + v * DO, sizeof(float) * DO);
      memcpy(cat + v * 2 * DO + DO, s + p * DO, sizeof(float) * DO);
    }
}
void oracle_concat_bwd(const float* dcat, const int64_t* csr, int64_t N, float* da2, float* ds) {
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t p = 0; p < N; ++p) {
    float acc[DO] = {0};
    for (int64_t v = csr[p]; v < csr[p + 1]; ++v) {
      memcpy(da2 + v * DO, dcat + v * 2 * DO, sizeof(float) * DO);
      for (int o = 0; o < DO; ++o) acc[o] += dcat[v * 2 * DO + DO + o];
    }
    memcpy(ds + p * DO, acc, sizeof(float) * DO);
  }
}
/* scores [V][G] = a6 Ws^T + bs;  backward: da6 = dc Ws, dWs [G][32], dbs [G] */
void oracle_score_fwd(const float* a6, int64_t V, int G, const float* Ws, const float* bs, float* scores) {
#pragma omp parallel for schedule(static)
  for (int64_t v = 0; v < V; ++v)
    for (int g = 0; g < G; ++g) {
      float acc = bs[g];
      for (int o = 0; o < DO; ++o) acc = fmaf(a6[v * DO + o], Ws[g * DO + o], acc);
      scores[v * G + g] = acc;
    }
}
void oracle_score_bwd(const float* a6, const float* dc, int64_t V, int G, const float* Ws, float* da6, float* dWs,
                      float* dbs) {
  const int nt = oracle_deepset_num_threads();
  double* part = (double*)calloc((size_t)nt * (G * DO + G), sizeof(double));
#pragma omp parallel
  {
#ifdef _OPENMP
    double* mine = part + (size_t)omp_get_thread_num() * (G * DO + G);
#else
    double* mine = part;
#endif
#pragma omp for schedule(static)
    for (int64_t v = 0; v < V; ++v) {
      for (int o = 0; o < DO; ++o) {
        float acc = 0.f;
        for (int g = 0; g < G; ++g) acc = fmaf(dc[v * G + g], Ws[g * DO + o], acc);
        da6[v * DO + o] = acc;
      }
      for (int g = 0; g < G; ++g) {
        const float d = dc[v * G + g];
        for (int o = 0; o < DO; ++o) mine[g * DO + o] += (double)d * a6[v * DO + o];
        mine[G * DO + g] += d;
      }
    }
  }
  for (int i = 0; i < G * DO + G; ++i) {
    double acc = 0.0;
    for (int t = 0; t < nt; ++t) acc += part[(size_t)t * (G * DO + G) + i];
    if (i < G * DO) dWs[i] = (float)acc;
    else dbs[i - G * DO] = (float)acc;
  }
  free(part);
}

/* ---- E_mod on the MAP rows (the nearest gather of an exact mapping commutes with the row-wise MLP: DESIGN.md "E_mod
 * hoisting"; reference E_mod = MLP([in_mod, out_mod, out_mod]) applied to the gathered [V, C] rows, pooling.py:245,275)
 * and the fusion concat (modules/multimodal/fusion.py:33-53).  A map row that cnt[r] views read counts cnt[r] times in
 * the train-mode batch statistics, so the result equals the per-view evaluation the reference runs (pinned by
 * tests/test_oracle_deepset_c.py against the per-view PyTorch restatement).  Generic widths K -> O (O <= 512). */
#define WMAX 512
void oracle_wblock_fwd(const float* x, const float* cnt, int64_t M, int K, int O, const float* W, const float* gamma,
                       const float* beta, float eps, float* z, float* a, float* mean, float* invstd) {
  double s0 = 0.0, s1[WMAX] = {0}, s2[WMAX] = {0};
#pragma omp parallel
  {
    double t0 = 0.0, t1[WMAX] = {0}, t2[WMAX] = {0};
#pragma omp for schedule(static)
    for (int64_t r = 0; r < M; ++r) {
      const float* xr = x + r * K;
      float* zr = z + r * O;
      const double c = cnt[r];
      for (int o = 0; o < O; ++o) {
        const float* w = W + (int64_t)o * K;
        float acc = 0.f;
#pragma omp simd reduction(+ : acc)
        for (int k = 0; k < K; ++k) acc += xr[k] * w[k];
        zr[o] = acc;
        t1[o] += c * acc;
        t2[o] += c * (double)acc * acc;
      }
      t0 += c;
    }
#pragma omp critical
    {
      s0 += t0;
      for (int o = 0; o < O; ++o) { s1[o] += t1[o]; s2[o] += t2[o]; }
    }
  }
  float g[WMAX], b[WMAX];
  for (int o = 0; o < O; ++o) {
    const double mu = s0 > 0 ? s1[o] / s0 : 0.0;
    double var = s0 > 0 ? s2[o] / s0 - mu * mu : 0.0;
    if (var < 0) var = 0;
    mean[o] = (float)mu;
    invstd[o] = (float)(1.0 / sqrt(var + (double)eps));
    g[o] = gamma[o] * invstd[o];
    b[o] = beta[o] - mean[o] * g[o];
  }
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < M; ++r)
    for (int o = 0; o < O; ++o) {
      const float y = fmaf(z[r * O + o], g[o], b[o]);
      a[r * O + o] = y > 0.f ? y : 0.2f * y;
    }
}

/* da [M][O] = dL/da of the map rows (already summed over the views of a row) -> dx [M][K] (nullable), dW [O][K], dgamma,
 * dbeta.  With mu, sigma weighted by cnt:  dz[r] = g (dy[r] - cnt[r]/V S1 - cnt[r]/V zh[r] S2),  S1 = sum dy, S2 = sum dy zh */
void oracle_wblock_bwd(const float* x, const float* z, const float* da, const float* cnt, int64_t M, int K, int O,
                       const float* W, const float* gamma, const float* beta, const float* mean, const float* invstd,
                       float* dx, float* dW, float* dgamma, float* dbeta) {
  double s0 = 0.0, s1[WMAX] = {0}, s2[WMAX] = {0};
  float g[WMAX], b[WMAX];
  for (int o = 0; o < O; ++o) {
    g[o] = gamma[o] * invstd[o];
    b[o] = beta[o] - mean[o] * g[o];
  }
#pragma omp parallel
  {
    double t0 = 0.0, t1[WMAX] = {0}, t2[WMAX] = {0};
#pragma omp for schedule(static)
    for (int64_t r = 0; r < M; ++r) {
      for (int o = 0; o < O; ++o) {
        const float zz = z[r * O + o];
        const float y = fmaf(zz, g[o], b[o]);
        const float dy = y > 0.f ? da[r * O + o] : 0.2f * da[r * O + o];
        t1[o] += dy;
        t2[o] += (double)dy * ((zz - mean[o]) * invstd[o]);
      }
      t0 += cnt[r];
    }
#pragma omp critical
    {
      s0 += t0;
      for (int o = 0; o < O; ++o) { s1[o] += t1[o]; s2[o] += t2[o]; }
    }
  }
  float k1[WMAX], k2[WMAX];
  for (int o = 0; o < O; ++o) {
    dbeta[o] = (float)s1[o];
    dgamma[o] = (float)s2[o];
    k1[o] = s0 > 0 ? (float)(s1[o] / s0) : 0.f;
    k2[o] = s0 > 0 ? (float)(s2[o] / s0) : 0.f;
  }
  const int nt = oracle_deepset_num_threads();
  double* part = (double*)calloc((size_t)nt * O * K, sizeof(double));        /* [thread][o][k] */
#pragma omp parallel
  {
#ifdef _OPENMP
    double* mine = part + (size_t)omp_get_thread_num() * O * K;
#else
    double* mine = part;
#endif
    float dz[WMAX];
#pragma omp for schedule(static)
    for (int64_t r = 0; r < M; ++r) {
      const float* xr = x + r * K;
      const float c = cnt[r];
      for (int o = 0; o < O; ++o) {
        const float zz = z[r * O + o];
        const float y = fmaf(zz, g[o], b[o]);
        const float dy = y > 0.f ? da[r * O + o] : 0.2f * da[r * O + o];
        const float zh = (zz - mean[o]) * invstd[o];
        dz[o] = g[o] * (dy - c * k1[o] - c * zh * k2[o]);
        double* m = mine + (size_t)o * K;
        for (int k = 0; k < K; ++k) m[k] += (double)dz[o] * xr[k];
      }
      if (dx)
        for (int k = 0; k < K; ++k) {
          float acc = 0.f;
          for (int o = 0; o < O; ++o) acc = fmaf(dz[o], W[(int64_t)o * K + k], acc);
          dx[r * K + k] = acc;
        }
    }
  }
  for (int i = 0; i < O * K; ++i) {
    double acc = 0.0;
    for (int t = 0; t < nt; ++t) acc += part[(size_t)t * O * K + i];
    dW[i] = (float)acc;
  }
  free(part);
}

/* BimodalFusion('concatenation'): out [N][A + C] = [x_3d | x_pool]; backward = the two column blocks of the gradient */
void oracle_fusion_concat_fwd(const float* x3d, const float* xpool, int64_t N, int A, int C, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < N; ++p) {
    memcpy(out + p * (A + C), x3d + p * A, sizeof(float) * (size_t)A);
    memcpy(out + p * (A + C) + A, xpool + p * C, sizeof(float) * (size_t)C);
  }
}
void oracle_fusion_concat_bwd(const float* dout, int64_t N, int A, int C, float* dx3d, float* dxpool) {
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < N; ++p) {
    memcpy(dx3d + p * A, dout + p * (A + C), sizeof(float) * (size_t)A);
    memcpy(dxpool + p * C, dout + p * (A + C) + A, sizeof(float) * (size_t)C);
  }
}
