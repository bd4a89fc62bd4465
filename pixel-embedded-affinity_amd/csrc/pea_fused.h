// pea_fused.h -- training step in ONE launch: affinities, class-balanced MSE loss AND d loss / d e.
//
// The two-launch path (k_fwd_tiled_v, then k_bwd_tiled) stages the embeddings twice and hands g = d loss / d affs
// from one kernel to the other through HBM (4K B/px written, 4K B/px read, plus the neighbour re-reads).  Here the
// backward's gather loop computes what it needs on the spot.  For pixel p and offset o_i the two terms of
//     G(p) = sum_i [ g_i(p) ehat(p + o_i)  +  g_i(p - o_i) ehat(p - o_i) ]
// need g_i at p and at p - o_i, and
//     g_i(q) = (2 lambda_i / N_i) w_i(q) m_i(q) (a_i(q) m_i(q) - t_i(q) m_i(q)),    a_i(q) = <ehat(q), ehat(q + o_i)>:
//   role A (q = p):        a_i(p)       = <ehat(p), ehat(p + o_i)>  -- the neighbour vector the term multiplies anyway
//   role B (q = p - o_i):  a_i(p - o_i) = <ehat(p - o_i), ehat(p)>  -- again <own, neighbour>: no halo of dot products
// so every neighbour vector read from LDS is used twice (dot product, then axpy) and target / weight / mask are
// sampled at p and at p - o_i.  Role A also yields the outputs of the forward: affs and the loss partials.
// HBM traffic per pixel: 4D (e) + 9K (t, w, m) + 4K (affs) + 4D (de) = 8D + 13K  (12D + 22K for the two launches,
// SURVEY.md section 8d).  The gradient is produced for dloss = 1 unless a device scalar is given; autograd scales it.
// Same tile / LDS structure as k_bwd_tiled (pea_tiled.h).
#pragma once
#include "pea_tiled.h"

namespace pea {

template <typename T, int D_T, int TH, int TW, int PLQ, bool CROP, bool ROLE_B>
__global__ __launch_bounds__(TH* TW, 4) void k_fused_tiled(const KParams P, const TParams Q, const T* __restrict__ xt,
                                                           const T* __restrict__ nbt, const float* __restrict__ target,
                                                           const float* __restrict__ weight,
                                                           const uint8_t* __restrict__ mask, float* __restrict__ affs,
                                                           float* __restrict__ partials, const float* __restrict__ dloss,
                                                           T* __restrict__ dx) {
  typedef Lds<D_T, PLQ> L;
  constexpr int NT = TH * TW, NW = NT / 64;
  constexpr int NR = ROLE_B ? 2 : 1;
  constexpr int KN = 3;  // near offsets per chunk (x NR roles of target / weight / mask samples held in registers; 4 made the allocator spill)
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  float* s_part = (float*)(lds + L::kBytes);  // [NW][K] loss partials per wave
  const int tile = tile_id(Q);
  if (tile >= Q.ntiles) return;
  const int plane = tile / Q.tiles_per_plane;
  const int rem = tile - plane * Q.tiles_per_plane;
  const int ty = rem / Q.tiles_x;
  const int y0 = ty * TH, x0 = (rem - ty * Q.tiles_x) * TW;
  const int b = plane / P.Z, z = plane - b * P.Z;
  const size_t S = (size_t)P.S;
  const unsigned YX = (unsigned)(P.Y * P.X);
  const rsrc_t xB = mkbuf(xt + (size_t)b * D_T * S), nB = mkbuf(nbt + (size_t)b * D_T * S);
  const rsrc_t dB = mkbuf(dx + (size_t)b * D_T * S);
  const rsrc_t tB = mkbuf(target + (size_t)b * P.tbs), wB = mkbuf(weight + (size_t)b * P.wbs);
  const rsrc_t mB = mkbuf(mask ? mask + (size_t)b * P.mbs : nullptr);
  const rsrc_t aB = mkbuf(affs ? affs + (size_t)b * P.K * S : nullptr);
  const bool has_m = mask != nullptr, has_a = affs != nullptr;
  const bool relu = P.flags & PEA_FLAG_RELU_AFFS;
  const unsigned ecs = (unsigned)P.S * (unsigned)sizeof(T);
  const unsigned ezo = (unsigned)z * YX * (unsigned)sizeof(T);
  const unsigned kcs = (unsigned)P.S * 4u, kzo = (unsigned)z * YX * 4u, S32 = (unsigned)P.S;
  const float dl = dloss ? dloss[0] : 1.f;

  int ly, lx;
  lane_pixel<TW>(ly, lx);
  const int py = y0 + ly, px = x0 + lx;
  const bool live = py < P.Y && px < P.X;
  const unsigned po = (unsigned)(py * P.X + px);
  const unsigned pe = live ? po * (unsigned)sizeof(T) : kOOB;
  const unsigned pb = live ? po * 4u : kOOB;
  const int pr = (ly + Q.hy0) * Q.RW + lx + Q.hx0;
  const int wave = threadIdx.x >> 6;

  // target / weight / mask of (near entry k0 + u, role): role A samples p, role B samples q = p - o (wrapped /
  // cropped).  A sample that does not exist reads out of range: w = 0, so the pair contributes nothing.
  float tv[KN][NR], wv[KN][NR], mv[KN][NR];
#define PEA_FUSED_LOAD_TWM1(u, k)                                                                            \
  {                                                                                                          \
    _Pragma("unroll") for (int r = 0; r < NR; ++r) {                                                         \
      const OffEnt en_ = Q.near[min((k), Q.n_near - 1)];                                                     \
      bool oky_ = true, okx_ = true;                                                                         \
      unsigned q_ = po;                                                                                      \
      if (r == 1) {                                                                                          \
        const int yy_ = wrap1<CROP>(py - ent_oy(en_), P.Y, oky_);                                            \
        const int xx_ = wrap1<CROP>(px - ent_ox(en_), P.X, okx_);                                            \
        q_ = (unsigned)(yy_ * P.X + xx_);                                                                    \
      }                                                                                                      \
      const bool ok_ = live && oky_ && okx_ && ((k) < Q.n_near);                                             \
      const unsigned so_ = kzo + (unsigned)en_.i * kcs;                                                      \
      tv[u][r] = bl32(tB, ok_ ? q_ * 4u : kOOB, so_);                                                        \
      wv[u][r] = bl32(wB, ok_ ? q_ * 4u : kOOB, so_);                                                        \
      mv[u][r] = has_m ? bl8(mB, ok_ ? q_ : kOOB, (kzo >> 2) + (unsigned)en_.i * S32) : 1.f;                 \
    }                                                                                                        \
  }
#define PEA_FUSED_LOAD_TWM(k0) { _Pragma("unroll") for (int u = 0; u < KN; ++u) PEA_FUSED_LOAD_TWM1(u, (k0) + u) }

  // (1) own raw pixel and the samples of the first near chunk: in flight during staging
  float xh[D_T];
#pragma unroll
  for (int c = 0; c < D_T; ++c) xh[c] = bl_emb<T>(xB, pe, ezo + c * ecs);
  if (Q.n_near > 0) PEA_FUSED_LOAD_TWM(0)

  // (2) stage the neighbour tensor's region, normalised
  stage_region<T, D_T, PLQ, NT, CROP>(P, Q, nB, ezo, ecs, y0, x0, lds);

  // (3) far (offset, role) pairs, one at a time ahead: pair j = (far offset j / NR, role j % NR)
  const int n_farp = Q.n_far * NR;
  float fv[D_T], ft = 0.f, fw = 0.f, fm = 0.f;
  bool fok = false;
#define PEA_FUSED_LOAD_FAR(j)                                                                                 \
  {                                                                                                           \
    const OffEnt fe_ = Q.far[(j) / NR];                                                                       \
    const int sg_ = ((j) % NR) == 0 ? 1 : -1;                                                                 \
    bool okz_, oky_, okx_;                                                                                    \
    const int zz_ = wrap1<CROP>(z + sg_ * fe_.d, P.Z, okz_);                                                  \
    const int yy_ = wrap1<CROP>(py + sg_ * ent_oy(fe_), P.Y, oky_);                                           \
    const int xx_ = wrap1<CROP>(px + sg_ * ent_ox(fe_), P.X, okx_);                                           \
    fok = live && okz_ && oky_ && okx_;                                                                       \
    const unsigned zc_ = (unsigned)(CROP ? min(max(zz_, 0), P.Z - 1) : zz_);                                  \
    const unsigned qo_ = (unsigned)(yy_ * P.X + xx_);                                                         \
    const unsigned vo_ = fok ? qo_ * (unsigned)sizeof(T) : kOOB;                                              \
    _Pragma("unroll") for (int c = 0; c < D_T; ++c) fv[c] = bl_emb<T>(nB, vo_, zc_ * YX * (unsigned)sizeof(T) + c * ecs); \
    /* role A samples p, role B samples the neighbour q = p - o itself */                                     \
    const unsigned so4_ = (sg_ > 0 ? kzo : zc_ * YX * 4u) + (unsigned)fe_.i * kcs;                            \
    const unsigned sp_ = fok ? (sg_ > 0 ? po : qo_) : 0x20000000u;                                            \
    ft = bl32(tB, sp_ * 4u, so4_);                                                                            \
    fw = bl32(wB, sp_ * 4u, so4_);                                                                            \
    fm = has_m ? bl8(mB, fok ? sp_ : kOOB, (so4_ >> 2)) : 1.f;                                                \
  }

  if (n_farp > 0) PEA_FUSED_LOAD_FAR(0)

  float G[D_T];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < D_T; ++c) {
    ss = fmaf(xh[c], xh[c], ss);
    G[c] = 0.f;
  }
  const bool tiny = ss < P.eps * P.eps;
  const float invp = rnorm(ss, Q.inv_eps);
#pragma unroll
  for (int c = 0; c < D_T; ++c) xh[c] *= invp;
  lds_barrier();

  // one pair: a = <own, nbhat>; role A also writes affs and the loss partial; returns the coefficient of nbhat in G
#define PEA_FUSED_PAIR(ROLE, ent, a_in, exists, t_, w_, m_, coef)                                             \
  {                                                                                                           \
    float a_ = (a_in);                                                                                        \
    if ((ROLE) == 0) {                                                                                        \
      a_ = (exists) ? a_ : 0.f;                                                                               \
      if (has_a) bs32<true>(aB, relu ? fmaxf(a_, 0.f) : a_, pb, kzo + (unsigned)(ent).i * kcs);              \
      const float rr_ = a_ * (m_) - (t_) * (m_);                                                              \
      const float wr_ = (exists) ? (w_) * rr_ : 0.f;                                                          \
      const float red_ = wave_sum63(wr_ * rr_);                                                               \
      if ((threadIdx.x & 63) == 63) s_part[wave * P.K + (ent).i] = red_;                                      \
      coef = (ent).gscale * wr_ * (m_);                                                                       \
    } else {                                                                                                  \
      const float rr_ = a_ * (m_) - (t_) * (m_);                                                              \
      coef = (ent).gscale * (w_) * rr_ * (m_);                                                                \
    }                                                                                                         \
  }
#define PEA_FUSED_FAR(j)                                                                                      \
  {                                                                                                           \
    const OffEnt fe_ = Q.far[(j) / NR];                                                                       \
    float sq_ = 0.f, dot_ = 0.f;                                                                              \
    _Pragma("unroll") for (int c = 0; c < D_T; ++c) {                                                         \
      sq_ = fmaf(fv[c], fv[c], sq_);                                                                          \
      dot_ = fmaf(xh[c], fv[c], dot_);                                                                        \
    }                                                                                                         \
    const float rn_ = rnorm(sq_, Q.inv_eps);                                                                  \
    float cf_;                                                                                                \
    PEA_FUSED_PAIR((j) % NR, fe_, dot_ * rn_, fok, ft, fw, fm, cf_)                                           \
    cf_ *= rn_;                                                                                               \
    _Pragma("unroll") for (int c = 0; c < D_T; ++c) G[c] = fmaf(cf_, fv[c], G[c]);                            \
  }

  // ---- near pairs: neighbour vectors from LDS ---------------------------------------------------------
  int jf = 0;  // next far pair to consume (its loads were issued one step earlier)
  for (int k0 = 0; k0 < Q.n_near; k0 += KN) {
#pragma unroll
    for (int u = 0; u < KN; ++u) {
      if (k0 + u < Q.n_near) {  // uniform
        const OffEnt en = Q.near[k0 + u];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          float v[D_T];
          lds_pixel<D_T, PLQ>(lds, pr + (r == 0 ? en.d : -en.d), v);
          float a = 0.f;
#pragma unroll
          for (int c = 0; c < D_T; ++c) a = fmaf(xh[c], v[c], a);
          bool exists = live;
          if (CROP && r == 0)
            exists = exists && (unsigned)(py + ent_oy(en)) < (unsigned)P.Y && (unsigned)(px + ent_ox(en)) < (unsigned)P.X;
          float cf;
          PEA_FUSED_PAIR(r, en, a, exists, tv[u][r], wv[u][r], mv[u][r], cf)
#pragma unroll
          for (int c = 0; c < D_T; ++c) G[c] = fmaf(cf, v[c], G[c]);
          asm volatile("" ::: "memory");  // one neighbour vector live at a time (see k_bwd_tiled)
        }
        // rolling prefetch: slot u is free again, request the samples of the entry KN steps ahead
        if (k0 + KN + u < Q.n_near) PEA_FUSED_LOAD_TWM1(u, k0 + KN + u)
      }
    }
    // a far pair per near chunk: its round trip hides under the LDS-served pairs
    if (jf < n_farp) {
      PEA_FUSED_FAR(jf)
      ++jf;
      if (jf < n_farp) PEA_FUSED_LOAD_FAR(jf)
    }
  }
  while (jf < n_farp) {
    PEA_FUSED_FAR(jf)
    ++jf;
    if (jf < n_farp) PEA_FUSED_LOAD_FAR(jf)
  }
#undef PEA_FUSED_LOAD_TWM
#undef PEA_FUSED_LOAD_TWM1
#undef PEA_FUSED_LOAD_FAR
#undef PEA_FUSED_PAIR
#undef PEA_FUSED_FAR

  float proj = 0.f;
#pragma unroll
  for (int c = 0; c < D_T; ++c) proj = fmaf(xh[c], G[c], proj);
  if (tiny) proj = 0.f;  // clamp_min branch of F.normalize: d ehat / d e = I / eps
  const float sc = dl * invp;
#pragma unroll
  for (int c = 0; c < D_T; ++c) bs_emb<T, true>(dB, (G[c] - xh[c] * proj) * sc, pe, ezo + c * ecs);

  lds_barrier();
  if (threadIdx.x < P.K) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += s_part[w * P.K + threadIdx.x];
    partials[(size_t)threadIdx.x * Q.ntiles + tile] = v;
  }
}

}  // namespace pea
