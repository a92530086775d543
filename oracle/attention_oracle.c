/* TEST INFRASTRUCTURE ONLY (oracle): plain C + OpenMP restatement of the view gather + attention pooling tail of
 * GroupBimodalCSRPool, forward and backward, on the host cores.  Imported / linked only by tests/, smoke() and
 * bench.py's cpu_baseline leg -- never by the product path.
 *
 * Reference maths (paths under /root/reference/torch_points3d/):
 *   nearest gather of the mapped features            core/multimodal/image.py:1262-1287 (x[idx] of the row matrix)
 *   segment_softmax_csr (centre on the group max, optional 1/sqrt(n) AFTER centring, exp, / (sum + eps))
 *                                                    modules/multimodal/pooling.py:758-810
 *   expand_group_feat (channel c belongs to group g; the first C mod G groups are one channel larger)
 *                                                    modules/multimodal/pooling.py:737-755
 *   x_pool = segment_csr(x_mod * A, 'sum')           modules/multimodal/pooling.py:284-291
 *   gating = tanh(relu(w * max_v compat + b))        modules/multimodal/pooling.py:293-300, Gating :690-715
 * Pinned by tests/test_oracle_attention_c.py against the PyTorch restatement (itself pinned on the reference's golden
 * vectors) to 1e-5.
 *
 * One thread team over the points; a point's views are contiguous (CSR).  The rows gradient is a scatter-add over
 * random rows: every thread accumulates into the shared fp32 buffer with atomics (what index_add does on the CPU).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline int group_of(int c, int C, int G) {
  /* group sizes: floor(C / G), the first C - G floor(C / G) groups one more (pooling.py:737-745) */
  const int base = C / G, extra = C - base * G;
  const int big = extra * (base + 1);
  return c < big ? c / (base + 1) : extra + (c - big) / base;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* out [N][C], att [V][G], gate [N][G] (1 when gate_w == NULL), amax int32 [N][G] (view index of the group max,
 * first maximal view; -1 for points without views) */
void oracle_gather_attention_fwd(const float* rows, const int32_t* row_idx, const float* compat, const int64_t* csr,
                                 const float* gate_w, const float* gate_b, int64_t N, int64_t V, int C, int G,
                                 int scaling, float eps, float* out, float* att, float* gate, int32_t* amax) {
  (void)V;
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t p = 0; p < N; ++p) {
    const int64_t b = csr[p], e = csr[p + 1];
    float* o = out + p * C;
    memset(o, 0, sizeof(float) * (size_t)C);
    for (int g = 0; g < G; ++g) {
      gate[p * G + g] = 0.f;          /* empty group: max = 0 (torch_scatter), tanh(relu(b)) below if seen */
      amax[p * G + g] = -1;
    }
    if (e <= b) {
      if (gate_w)
        for (int g = 0; g < G; ++g) gate[p * G + g] = tanhf(fmaxf(gate_w[g] * 0.f + gate_b[g], 0.f));
      else
        for (int g = 0; g < G; ++g) gate[p * G + g] = 1.f;
      continue;
    }
    const float inv = scaling ? 1.f / sqrtf((float)(e - b)) : 1.f;
    for (int g = 0; g < G; ++g) {
      float m = -INFINITY;
      int64_t am = -1;
      for (int64_t v = b; v < e; ++v)
        if (compat[v * G + g] > m) { m = compat[v * G + g]; am = v; }
      float s = 0.f;
      for (int64_t v = b; v < e; ++v) {
        const float ex = expf((compat[v * G + g] - m) * inv);
        att[v * G + g] = ex;
        s += ex;
      }
      const float r = 1.f / (s + eps);
      for (int64_t v = b; v < e; ++v) att[v * G + g] *= r;
      amax[p * G + g] = (int32_t)am;
      gate[p * G + g] = gate_w ? tanhf(fmaxf(gate_w[g] * m + gate_b[g], 0.f)) : 1.f;
    }
    for (int64_t v = b; v < e; ++v) {
      const float* x = rows + (int64_t)row_idx[v] * C;
      for (int c = 0; c < C; ++c) o[c] += x[c] * att[v * G + group_of(c, C, G)];
    }
    for (int c = 0; c < C; ++c) o[c] *= gate[p * G + group_of(c, C, G)];
  }
}

/* grad_rows [R][C] (zeroed here), grad_compat [V][G], grad_gw / grad_gb [G] (nullable, accumulated from 0) */
void oracle_gather_attention_bwd(const float* grad_out, const float* rows, const int32_t* row_idx, const float* compat,
                                 const int64_t* csr, const float* gate_w, const float* gate_b, const float* att,
                                 const float* gate, const int32_t* amax, int64_t N, int64_t V, int64_t R, int C, int G,
                                 int scaling, float eps, float* grad_rows, float* grad_compat, float* grad_gw,
                                 float* grad_gb) {
  (void)eps;
  (void)gate_b;
  (void)compat;
  memset(grad_rows, 0, sizeof(float) * (size_t)(R * C));
  memset(grad_compat, 0, sizeof(float) * (size_t)(V * G));
  double gw_acc[64] = {0}, gb_acc[64] = {0};
#pragma omp parallel
  {
    double lw[64] = {0}, lb[64] = {0};
    float* q = (float*)malloc(sizeof(float) * 64);
#pragma omp for schedule(dynamic, 256)
    for (int64_t p = 0; p < N; ++p) {
      const int64_t b = csr[p], e = csr[p + 1];
      if (e <= b) continue;
      const float* go = grad_out + p * C;
      const float inv = scaling ? 1.f / sqrtf((float)(e - b)) : 1.f;
      /* x_pool (before the gate) per group: E_g = sum_v att q_v,g, q_v,g = sum_{c in g} go[c] x_v[c] */
      float Eg[64];
      for (int g = 0; g < G; ++g) Eg[g] = 0.f;
      for (int64_t v = b; v < e; ++v) {
        const float* x = rows + (int64_t)row_idx[v] * C;
        float* gr = grad_rows + (int64_t)row_idx[v] * C;
        for (int g = 0; g < G; ++g) q[g] = 0.f;
        for (int c = 0; c < C; ++c) {
          const int g = group_of(c, C, G);
          q[g] += go[c] * x[c];
          const float w = att[v * G + g] * gate[p * G + g];
#pragma omp atomic
          gr[c] += go[c] * w;
        }
        for (int g = 0; g < G; ++g) {
          grad_compat[v * G + g] = q[g];              /* parked: turned into the softmax gradient below */
          Eg[g] += att[v * G + g] * q[g];
        }
      }
      for (int g = 0; g < G; ++g) {
        const float gt = gate[p * G + g];
        /* softmax backward on the scaled, centred scores: dC_v = att_v (q_v - E) gate inv; the centring on the max
         * cancels (sum_v dC_v = 0 up to eps) */
        for (int64_t v = b; v < e; ++v)
          grad_compat[v * G + g] = att[v * G + g] * (grad_compat[v * G + g] - Eg[g]) * gt * inv;
        if (gate_w) {
          /* out_c = pool_c * gate_g: d gate = sum_{c in g} go[c] pool[c] = E_g; tanh(relu(.)) backward */
          const float dpre = gt > 0.f ? Eg[g] * (1.f - gt * gt) : 0.f;
          const int64_t am = amax[p * G + g];
          const float m = compat[am * G + g];
          grad_compat[am * G + g] += dpre * gate_w[g];
          lw[g] += (double)dpre * m;
          lb[g] += dpre;
        }
      }
    }
    free(q);
#pragma omp critical
    for (int g = 0; g < G; ++g) { gw_acc[g] += lw[g]; gb_acc[g] += lb[g]; }
  }
  if (grad_gw) for (int g = 0; g < G; ++g) grad_gw[g] = (float)gw_acc[g];
  if (grad_gb) for (int g = 0; g < G; ++g) grad_gb[g] = (float)gb_acc[g];
}
