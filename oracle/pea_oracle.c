/*
 * pea_oracle.c — CPU restatement of the reference's embedding -> affinity path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (pixel-embedded-affinity_amd/) may
 * link, import or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * Parity pinning: the reference (weih527/Pixel-Embedded-Affinity) ships no tests or golden
 * vectors for this path.  The oracle is pinned against golden vectors
 * produced by importing the reference's own Python functions (tests/golden/make_golden.py,
 * fixtures committed under tests/golden/), see tests/test_oracle.py.
 *
 * The code follows the reference step by step rather than the fused form the GPU kernels use:
 *   step 1  ehat = F.normalize(e, p=2, dim=1)                cvppp/loss/loss_embedding_mse.py:20
 *                                                            ac34/loss/loss_embedding_mse.py:8,173
 *   step 2  per offset: shifted product summed over channels cvppp/...mse.py:7-10 (torch.roll)
 *                                                            ac34/...mse.py:143-151 (cropped slices)
 *   step 3  WeightedMSE on (a*m, t*m, w)                     cvppp/...mse.py:15, loss/loss.py:112-119 (all three trees)
 *   step 4  loss = sum_i lambda_i L_i ; affs[:, i] = a_i     cvppp/...mse.py:35-47, ac34/...mse.py:178-194
 * Backward = the closed form of SURVEY.md section 8a (what torch.autograd computes for steps 1-4);
 * tests/test_oracle.py checks it against the reference's autograd gradients in the fixtures.
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/pea.h"

typedef struct {
  long Z, Y, X, S; /* S = Z*Y*X */
} Geo;

static Geo geo_of(const PeaDesc *d) {
  Geo g;
  g.Z = d->dims[0];
  g.Y = d->dims[1];
  g.X = d->dims[2];
  g.S = g.Z * g.Y * g.X;
  return g;
}

/* batch stride in elements of a [B,K,Z,Y,X] operand (0 = dense) */
static inline size_t bs(int64_t stride, const PeaDesc *d, const Geo *g) {
  return stride ? (size_t)stride : (size_t)d->K * (size_t)g->S;
}

/* neighbour of (z,y,x) under offset o and the descriptor's border mode; -1 if outside (crop) */
static inline long neighbour(const PeaDesc *d, const Geo *g, long z, long y, long x, const int32_t *o,
                             int sign) {
  long zz = z + sign * o[0], yy = y + sign * o[1], xx = x + sign * o[2];
  if (d->border == PEA_BORDER_CIRCULAR) {
    zz %= g->Z; if (zz < 0) zz += g->Z;
    yy %= g->Y; if (yy < 0) yy += g->Y;
    xx %= g->X; if (xx < 0) xx += g->X;
  } else if (d->border == PEA_BORDER_REPLICATE) { /* shift_tensor: ReplicationPad3d then slice = clamp the index */
    zz = zz < 0 ? 0 : (zz >= g->Z ? g->Z - 1 : zz);
    yy = yy < 0 ? 0 : (yy >= g->Y ? g->Y - 1 : yy);
    xx = xx < 0 ? 0 : (xx >= g->X ? g->X - 1 : xx);
  } else {
    if (zz < 0 || zz >= g->Z || yy < 0 || yy >= g->Y || xx < 0 || xx >= g->X) return -1;
  }
  return (zz * g->Y + yy) * g->X + xx;
}

/* WeightedMSE's norm_term for channel i (loss.py:113-115) */
static double norm_term(const PeaDesc *d, const Geo *g, int i) {
  if (d->norm == PEA_NORM_BX) return (double)d->B * (double)g->X;
  if (d->norm == PEA_NORM_FULL) return (double)d->B * (double)g->S;
  double n = d->B;
  for (int a = 0; a < 3; ++a) n *= (double)(d->dims[a] - labs((long)d->offsets[i][a]));
  return n;
}

/* step 1: F.normalize(p=2, dim=1, eps) into a fresh buffer */
static float *normalize_new(const PeaDesc *d, const Geo *g, const float *e) {
  float *out = (float *)malloc(sizeof(float) * (size_t)d->B * d->D * g->S);
  if (!out) return NULL;
#pragma omp parallel for collapse(2) schedule(static)
  for (long b = 0; b < d->B; ++b)
    for (long p = 0; p < g->S; ++p) {
      const float *src = e + (size_t)b * d->D * g->S + p;
      float *dst = out + (size_t)b * d->D * g->S + p;
      float ss = 0.f;
      for (long c = 0; c < d->D; ++c) ss += src[c * g->S] * src[c * g->S];
      float n = sqrtf(ss);
      if (n < d->eps) n = d->eps;
      for (long c = 0; c < d->D; ++c) dst[c * g->S] = src[c * g->S] / n;
    }
  return out;
}

static int check(const PeaDesc *d) {
  if (!d) return PEA_E_NULL;
  if (d->abi != PEA_ABI_VERSION || d->K < 1 || d->K > PEA_MAX_K || d->B < 1 || d->D < 1) return PEA_E_DESC;
  if (d->dims[0] < 1 || d->dims[1] < 1 || d->dims[2] < 1) return PEA_E_DESC;
  if (d->dtype != PEA_F32 || d->D > 256) return PEA_E_UNSUPPORTED; /* oracle is f32-in only */
  return PEA_OK;
}

/*
 * Forward.  affs nullable; loss_out[1+K] doubles = {loss, L_0..L_{K-1}} (nullable, then target /
 * weight are not read: this is the inference form embedding2affs / inf_embedding_loss_norm*).
 */
int pea_oracle_fwd(const PeaDesc *d, const float *e, const float *e_other, const float *target,
                   const float *weight, const uint8_t *mask, float *affs, double *loss_out) {
  int rc = check(d);
  if (rc) return rc;
  if (!e) return PEA_E_NULL;
  Geo g = geo_of(d);
  float *eh = normalize_new(d, &g, e);
  float *oh = e_other ? normalize_new(d, &g, e_other) : eh;
  if (!eh || !oh) return PEA_E_WORKSPACE;
  double total = 0.0;
  for (int i = 0; i < d->K; ++i) {
    double acc = 0.0;
#pragma omp parallel for collapse(3) schedule(static) reduction(+ : acc)
    for (long b = 0; b < d->B; ++b)
      for (long z = 0; z < g.Z; ++z)
        for (long y = 0; y < g.Y; ++y)
          for (long x = 0; x < g.X; ++x) {
            long p = (z * g.Y + y) * g.X + x;
            long q = neighbour(d, &g, z, y, x, d->offsets[i], +1);
            size_t ki = ((size_t)b * d->K + i) * g.S + p;
            size_t in = (size_t)i * g.S + p;
            float a = 0.f;
            if (q >= 0) {
              const float *ep = eh + (size_t)b * d->D * g.S + p;
              const float *eq = oh + (size_t)b * d->D * g.S + q;
              for (long c = 0; c < d->D; ++c) a += ep[c * g.S] * eq[c * g.S];
            }
            if (affs) affs[ki] = (d->flags & PEA_FLAG_RELU_AFFS) ? (a > 0.f ? a : 0.f) : a;
            if (loss_out && q >= 0) {
              float m = mask ? (float)mask[bs(d->mask_bstride, d, &g) * b + in] : 1.f;
              float r = a * m - target[bs(d->target_bstride, d, &g) * b + in] * m;
              acc += (double)weight[bs(d->weight_bstride, d, &g) * b + in] * (double)r * (double)r;
            }
          }
    if (loss_out) {
      double Li = acc / norm_term(d, &g, i);
      loss_out[1 + i] = Li;
      total += (double)d->lambda[i] * Li;
    }
  }
  if (loss_out) loss_out[0] = total;
  if (oh != eh) free(oh);
  free(eh);
  return PEA_OK;
}

/* g_i(p) = dloss * lambda_i * 2 w m (a m - t m) / N_i, zero where the neighbour is cropped away */
static inline float gval(const PeaDesc *d, const Geo *g, const float *eh, const float *oh,
                         const float *target, const float *weight, const uint8_t *mask, long b, int i,
                         long p, long q, float scale) {
  const float *ep = eh + (size_t)b * d->D * g->S + p;
  const float *eq = oh + (size_t)b * d->D * g->S + q;
  float a = 0.f;
  for (long c = 0; c < d->D; ++c) a += ep[c * g->S] * eq[c * g->S];
  size_t in = (size_t)i * g->S + p;
  float m = mask ? (float)mask[bs(d->mask_bstride, d, g) * b + in] : 1.f;
  return scale * weight[bs(d->weight_bstride, d, g) * b + in] * m *
         (a * m - target[bs(d->target_bstride, d, g) * b + in] * m);
}

/*
 * Backward: de = dloss * dloss/de, optionally de_other (NULL = second operand detached).
 * Without e_other (self loss) both roles of e contribute to de.
 */
int pea_oracle_bwd(const PeaDesc *d, const float *e, const float *e_other, const float *target,
                   const float *weight, const uint8_t *mask, float dloss, float *de, float *de_other) {
  int rc = check(d);
  if (rc) return rc;
  if (!e || !target || !weight || !de) return PEA_E_NULL;
  Geo g = geo_of(d);
  float *eh = normalize_new(d, &g, e);
  float *oh = e_other ? normalize_new(d, &g, e_other) : eh;
  if (!eh || !oh) return PEA_E_WORKSPACE;
  float scale[PEA_MAX_K];
  for (int i = 0; i < d->K; ++i) scale[i] = (float)(2.0 * dloss * d->lambda[i] / norm_term(d, &g, i));
  const int self = (e_other == NULL);

  if (d->border == PEA_BORDER_REPLICATE) {
    /* A clamped index is not invertible (several p share one neighbour), so this border is restated in SCATTER form,
     * sequentially: every pair (p, q = clamp(p + o_i)) adds g * ehat_other(q) to the first operand's G at p and
     * g * ehat(p) to the second operand's G at q; then both go through F.normalize's Jacobian. */
    const size_t n = (size_t)d->B * d->D * g.S;
    float *G1 = (float *)calloc(n, sizeof(float));
    float *G2 = (float *)calloc(n, sizeof(float));
    if (!G1 || !G2) return PEA_E_WORKSPACE;
    for (long b = 0; b < d->B; ++b)
      for (int i = 0; i < d->K; ++i)
        for (long z = 0; z < g.Z; ++z)
          for (long y = 0; y < g.Y; ++y)
            for (long x = 0; x < g.X; ++x) {
              long p = (z * g.Y + y) * g.X + x;
              long q = neighbour(d, &g, z, y, x, d->offsets[i], +1);
              float gi = gval(d, &g, eh, oh, target, weight, mask, b, i, p, q, scale[i]);
              for (long c = 0; c < d->D; ++c) {
                G1[((size_t)b * d->D + c) * g.S + p] += gi * oh[((size_t)b * d->D + c) * g.S + q];
                G2[((size_t)b * d->D + c) * g.S + q] += gi * eh[((size_t)b * d->D + c) * g.S + p];
              }
            }
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1 && (self || !de_other)) break;
      const float *src = pass == 0 ? e : e_other;
      const float *sh = pass == 0 ? eh : oh;
      float *dst = pass == 0 ? de : de_other;
      for (long b = 0; b < d->B; ++b)
        for (long p = 0; p < g.S; ++p) {
          float ss = 0.f, dot = 0.f, Gc[256];
          for (long c = 0; c < d->D; ++c) {
            size_t o = ((size_t)b * d->D + c) * g.S + p;
            Gc[c] = pass == 0 ? (self ? G1[o] + G2[o] : G1[o]) : G2[o];
            ss += src[o] * src[o];
            dot += sh[o] * Gc[c];
          }
          float nn = sqrtf(ss);
          for (long c = 0; c < d->D; ++c) {
            size_t o = ((size_t)b * d->D + c) * g.S + p;
            dst[o] = nn < d->eps ? Gc[c] / d->eps : (Gc[c] - sh[o] * dot) / nn;
          }
        }
    }
    free(G1); free(G2);
    if (oh != eh) free(oh);
    free(eh);
    return PEA_OK;
  }

  for (int pass = 0; pass < 2; ++pass) {
    /* pass 0: gradient w.r.t. e (first operand); pass 1: w.r.t. e_other */
    if (pass == 1 && (self || !de_other)) break;
    const float *src = pass == 0 ? e : e_other;
    const float *sh = pass == 0 ? eh : oh;
    float *dst = pass == 0 ? de : de_other;
#pragma omp parallel for collapse(3) schedule(static)
    for (long b = 0; b < d->B; ++b)
      for (long z = 0; z < g.Z; ++z)
        for (long y = 0; y < g.Y; ++y)
          for (long x = 0; x < g.X; ++x) {
            long p = (z * g.Y + y) * g.X + x;
            float G[256];
            for (long c = 0; c < d->D; ++c) G[c] = 0.f;
            for (int i = 0; i < d->K; ++i) {
              if (pass == 0) {
                /* first-operand role: g_i(p) * ehat_other(p + o_i) */
                long q = neighbour(d, &g, z, y, x, d->offsets[i], +1);
                if (q >= 0) {
                  float gi = gval(d, &g, eh, oh, target, weight, mask, b, i, p, q, scale[i]);
                  const float *eq = oh + (size_t)b * d->D * g.S + q;
                  for (long c = 0; c < d->D; ++c) G[c] += gi * eq[c * g.S];
                }
              }
              if (pass == 1 || self) {
                /* second-operand role: g_i(p - o_i) * ehat(p - o_i) */
                long q = neighbour(d, &g, z, y, x, d->offsets[i], -1);
                if (q >= 0) {
                  float gi = gval(d, &g, eh, oh, target, weight, mask, b, i, q, p, scale[i]);
                  const float *eq = eh + (size_t)b * d->D * g.S + q;
                  for (long c = 0; c < d->D; ++c) G[c] += gi * eq[c * g.S];
                }
              }
            }
            /* through F.normalize: de = (G - ehat <ehat, G>) / n ; for n < eps: de = G / eps */
            const float *sp = src + (size_t)b * d->D * g.S + p;
            const float *hp = sh + (size_t)b * d->D * g.S + p;
            float ss = 0.f, dot = 0.f;
            for (long c = 0; c < d->D; ++c) {
              ss += sp[c * g.S] * sp[c * g.S];
              dot += hp[c * g.S] * G[c];
            }
            float n = sqrtf(ss);
            float *dp = dst + (size_t)b * d->D * g.S + p;
            if (n < d->eps) {
              for (long c = 0; c < d->D; ++c) dp[c * g.S] = G[c] / d->eps;
            } else {
              for (long c = 0; c < d->D; ++c) dp[c * g.S] = (G[c] - hp[c * g.S] * dot) / n;
            }
          }
  }
  if (oh != eh) free(oh);
  free(eh);
  return PEA_OK;
}

int pea_oracle_version(void) { return PEA_ABI_VERSION; }

/* OpenMP threads of the loops above (rows of every (b, z) plane are shared out: collapse(3)); returns the previous maximum.
 * bench.py's cpu_baseline times the oracle with 1 thread and with all host cores (the stronger CPU line). */
#ifdef _OPENMP
#include <omp.h>
int pea_oracle_set_threads(int n) {
  const int old = omp_get_max_threads();
  if (n > 0) omp_set_num_threads(n);
  return old;
}
#else
int pea_oracle_set_threads(int n) { (void)n; return 1; }
#endif
