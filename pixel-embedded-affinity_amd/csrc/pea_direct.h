// pea_direct.h -- the direct (global-memory) kernels.
//
// The direct kernels are the general fallback (any D in the forward, any offsets that no LDS tile can
// hold): one lane = one pixel, D-loop in registers, every neighbour vector read from global memory
// (coalesced row reads; the L2 norm of the neighbour is accumulated while its channels stream in).
#pragma once
#include "pea_loss.h"

namespace pea {

// ------------------------------------------------------------------------------------------------
// forward (direct form): affs, (TRAIN) per-workgroup loss partials and g = d loss / d affs
//   D_T > 0: channels unrolled, own pixel kept in registers;  D_T == 0: generic D, own pixel re-read (L1)
// ------------------------------------------------------------------------------------------------
template <typename T, int D_T, bool TRAIN>
__global__ __launch_bounds__(kBlock) void k_fwd_direct(const KParams P, const T* __restrict__ e,
                                                       const T* __restrict__ eo,
                                                       const float* __restrict__ target,
                                                       const float* __restrict__ weight,
                                                       const uint8_t* __restrict__ mask,
                                                       float* __restrict__ affs, float* __restrict__ gout,
                                                       LossState* __restrict__ st) {
  extern __shared__ float s_acc[];  // [K][kBlock], TRAIN only
  const int tile = logical_tile(P);
  if (tile >= P.tiles) return;  // whole workgroup exits together (tile is uniform)
  const int b = tile / P.chunks;
  const int p = (tile - b * P.chunks) * kBlock + threadIdx.x;
  const bool live = p < P.S;
  const int D = D_T ? D_T : P.D;
  const size_t S = (size_t)P.S;
  const T* eb = e + (size_t)b * D * S;
  const T* ob = eo + (size_t)b * D * S;
  const size_t kb = (size_t)b * P.K * S;

  int x = 0, y = 0, z = 0;
  float ec[D_T ? D_T : 1];
  float inv_p = 0.f;
  if (live) {
    const int yx = P.Y * P.X;
    z = p / yx;
    const int r = p - z * yx;
    y = r / P.X;
    x = r - y * P.X;
    float ss = 0.f;
    if (D_T) {
#pragma unroll
      for (int c = 0; c < D_T; ++c) {
        ec[c] = ld(eb, c * S + p);
        ss = fmaf(ec[c], ec[c], ss);
      }
    } else {
      for (int c = 0; c < D; ++c) {
        const float v = ld(eb, c * S + p);
        ss = fmaf(v, v, ss);
      }
    }
    inv_p = inv_norm(ss, P.eps);
  }

  for (int i = 0; i < P.K; ++i) {
    float contrib = 0.f;
    if (live) {
      const int q = neighbour(P, z, y, x, P.off[i][0], P.off[i][1], P.off[i][2]);
      float a = 0.f;
      if (q >= 0) {
        float dot = 0.f, sq = 0.f;
        if (D_T) {
#pragma unroll
          for (int c = 0; c < D_T; ++c) {
            const float v = ld(ob, c * S + q);
            dot = fmaf(ec[c], v, dot);
            sq = fmaf(v, v, sq);
          }
        } else {
          for (int c = 0; c < D; ++c) {
            const float v = ld(ob, c * S + q);
            dot = fmaf(ld(eb, c * S + p), v, dot);
            sq = fmaf(v, v, sq);
          }
        }
        a = dot * inv_p * inv_norm(sq, P.eps);
      }
      const size_t in = (size_t)i * S + p;
      if (affs) affs[kb + in] = act_affs(a, P.flags & kActMask);
      if (TRAIN) {
        float g = 0.f;
        if (q >= 0) {
          const float m = mask ? (float)mask[(size_t)b * P.mbs + in] : 1.f;
          const float r = a * m - target[(size_t)b * P.tbs + in] * m;
          const float wr = weight[(size_t)b * P.wbs + in] * r;
          contrib = wr * r;
          g = P.gscale[i] * wr * m;
        }
        if (gout) gout[kb + in] = g;
      }
    }
    if (TRAIN) s_acc[i * kBlock + threadIdx.x] = contrib;
  }

  if (TRAIN) {
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = wave; i < P.K; i += kBlock / 64) {
      const float* row = s_acc + i * kBlock;
      float v = (row[lane] + row[lane + 64]) + (row[lane + 128] + row[lane + 192]);
      v = wave_sum(v);
      if (lane == 0) loss_accumulate(st, tile, i, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward (direct gather form).  x = the tensor being differentiated, g = d loss / d affs [B,K,S].
//   ROLE_A: x is the first operand:  G(p) += g_i(p)       * nhat(p + o_i),  n = nbA (second operand)
//   ROLE_B: x is the second operand: G(p) += g_i(p - o_i) * nhat(p - o_i),  n = nbB (first operand)
//   self loss: both roles, nbA = nbB = x.
//   dx(p) = dloss * (G - xhat <xhat, G>) / n(p)        (G / eps when |x(p)| < eps)
// ------------------------------------------------------------------------------------------------
template <typename T, int D_T, bool ROLE_A, bool ROLE_B>
__global__ __launch_bounds__(kBlock) void k_bwd_direct(const KParams P, const T* __restrict__ xt,
                                                       const T* __restrict__ nbA, const T* __restrict__ nbB,
                                                       const float* __restrict__ gin,
                                                       const float* __restrict__ dloss, T* __restrict__ dx) {
  static_assert(D_T > 0, "backward is specialised on D");
  const int tile = logical_tile(P);
  if (tile >= P.tiles) return;
  const int b = tile / P.chunks;
  const int p = (tile - b * P.chunks) * kBlock + threadIdx.x;
  if (p >= P.S) return;
  const size_t S = (size_t)P.S;
  const T* xb = xt + (size_t)b * D_T * S;
  const float* gb = gin + (size_t)b * P.K * S;
  const float dl = dloss ? dloss[0] : 1.f;

  const int yx = P.Y * P.X;
  const int z = p / yx;
  const int r0 = p - z * yx;
  const int y = r0 / P.X;
  const int x = r0 - y * P.X;

  float xc[D_T], G[D_T];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < D_T; ++c) {
    xc[c] = ld(xb, c * S + p);
    ss = fmaf(xc[c], xc[c], ss);
    G[c] = 0.f;
  }
  const float nrm = sqrtf(ss);
  const float inv_p = 1.0f / fmaxf(nrm, P.eps);

  for (int i = 0; i < P.K; ++i) {
    const int oz = P.off[i][0], oy = P.off[i][1], ox = P.off[i][2];
#pragma unroll
    for (int role = 0; role < 2; ++role) {
      if (role == 0 ? !ROLE_A : !ROLE_B) continue;
      const int sg = role == 0 ? 1 : -1;
      const T* nb = (role == 0 ? nbA : nbB) + (size_t)b * D_T * S;
      if (role == 1 && P.border == PEA_BORDER_REPLICATE) {
        // the clamp is not invertible: gather from EVERY first-operand pixel p' whose clamped neighbour is this pixel
        int z0, z1, y0, y1, x0, x1;
        clamp_preimage(z, oz, P.Z, z0, z1);
        clamp_preimage(y, oy, P.Y, y0, y1);
        clamp_preimage(x, ox, P.X, x0, x1);
        for (int zz = z0; zz <= z1; ++zz)
          for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx) {
              const int q2 = (zz * P.Y + yy) * P.X + xx;
              float v2[D_T], sq2 = 0.f;
#pragma unroll
              for (int c = 0; c < D_T; ++c) {
                v2[c] = ld(nb, c * S + q2);
                sq2 = fmaf(v2[c], v2[c], sq2);
              }
              const float g2 = gb[(size_t)i * S + q2] * inv_norm(sq2, P.eps);
#pragma unroll
              for (int c = 0; c < D_T; ++c) G[c] = fmaf(g2, v2[c], G[c]);
            }
        continue;
      }
      const int q = neighbour(P, z, y, x, sg * oz, sg * oy, sg * ox);
      if (q < 0) continue;
      float v[D_T], sq = 0.f;
#pragma unroll
      for (int c = 0; c < D_T; ++c) {
        v[c] = ld(nb, c * S + q);
        sq = fmaf(v[c], v[c], sq);
      }
      // the loss term lives at the first operand's pixel: p for role A, the neighbour for role B
      const float g = gb[(size_t)i * S + (role == 0 ? p : q)] * inv_norm(sq, P.eps);
#pragma unroll
      for (int c = 0; c < D_T; ++c) G[c] = fmaf(g, v[c], G[c]);
    }
  }

  float proj = 0.f;
#pragma unroll
  for (int c = 0; c < D_T; ++c) proj = fmaf(xc[c] * inv_p, G[c], proj);
  if (nrm < P.eps) proj = 0.f;  // clamp_min branch of F.normalize: d ehat / d e = I / eps
  T* db = dx + (size_t)b * D_T * S;
  const float sc = dl * inv_p;
#pragma unroll
  for (int c = 0; c < D_T; ++c) st(db, c * S + p, (G[c] - xc[c] * inv_p * proj) * sc);
}

// ------------------------------------------------------------------------------------------------
// backward for ANY embedding width (MODEL.emd is a free yaml key in the reference: scripts_cvppp/config/cvppp.yaml:7,
// loss_embedding_mse.py:18-47 takes what it gets).  Runtime D, nothing D-sized in registers: the 2K pair coefficients
// g * 1/|e(q)| are parked in LDS ([2K][256 lanes]); G_c is then formed channel by channel, written raw into dx while
// <ehat, G> accumulates, and the projection is applied in a last sweep over the lane's own D outputs.  Three passes over
// the neighbours: a correct fallback, not a fast path (the specialised kernels cover D = 4, 8, 16, 32, 64).
// CIRCULAR / CROP_ZERO borders (REPLICATE keeps the specialised kernels).
// ------------------------------------------------------------------------------------------------
template <typename T, bool ROLE_A, bool ROLE_B>
__global__ __launch_bounds__(kBlock) void k_bwd_direct_anyd(const KParams P, const T* __restrict__ xt, const T* __restrict__ nbA,
                                                            const T* __restrict__ nbB, const float* __restrict__ gin,
                                                            const float* __restrict__ dloss, T* __restrict__ dx) {
  extern __shared__ float s_coef[];  // [2K][kBlock]
  const int tile = logical_tile(P);
  if (tile >= P.tiles) return;
  const int b = tile / P.chunks;
  const int p = (tile - b * P.chunks) * kBlock + threadIdx.x;
  if (p >= P.S) return;
  const int D = P.D;
  const size_t S = (size_t)P.S;
  const T* xb = xt + (size_t)b * D * S;
  const float* gb = gin + (size_t)b * P.K * S;
  const float dl = dloss ? dloss[0] : 1.f;
  const int yx = P.Y * P.X;
  const int z = p / yx;
  const int r0 = p - z * yx;
  const int y = r0 / P.X;
  const int x = r0 - y * P.X;

  float ss = 0.f;
  for (int c = 0; c < D; ++c) {
    const float v = ld(xb, c * S + p);
    ss = fmaf(v, v, ss);
  }
  const float nrm = sqrtf(ss);
  const float inv_p = 1.0f / fmaxf(nrm, P.eps);

  // (1) coefficient of every (offset, role) pair; 0 where the pair does not exist
  for (int i = 0; i < P.K; ++i) {
#pragma unroll
    for (int role = 0; role < 2; ++role) {
      float coef = 0.f;
      if (role == 0 ? ROLE_A : ROLE_B) {
        const int sg = role == 0 ? 1 : -1;
        const T* nb = (role == 0 ? nbA : nbB) + (size_t)b * D * S;
        const int q = neighbour(P, z, y, x, sg * P.off[i][0], sg * P.off[i][1], sg * P.off[i][2]);
        if (q >= 0) {
          float sq = 0.f;
          for (int c = 0; c < D; ++c) {
            const float v = ld(nb, c * S + q);
            sq = fmaf(v, v, sq);
          }
          coef = gb[(size_t)i * S + (role == 0 ? p : q)] * inv_norm(sq, P.eps);
        }
      }
      s_coef[(2 * i + role) * kBlock + threadIdx.x] = coef;
    }
  }
  // (2) G_c, raw, into dx; <ehat, G>
  T* db = dx + (size_t)b * D * S;
  float proj = 0.f;
  for (int c = 0; c < D; ++c) {
    float Gc = 0.f;
    for (int i = 0; i < P.K; ++i) {
#pragma unroll
      for (int role = 0; role < 2; ++role) {
        if (role == 0 ? !ROLE_A : !ROLE_B) continue;
        const float coef = s_coef[(2 * i + role) * kBlock + threadIdx.x];
        if (coef == 0.f) continue;
        const int sg = role == 0 ? 1 : -1;
        const T* nb = (role == 0 ? nbA : nbB) + (size_t)b * D * S;
        const int q = neighbour(P, z, y, x, sg * P.off[i][0], sg * P.off[i][1], sg * P.off[i][2]);
        Gc = fmaf(coef, ld(nb, c * S + q), Gc);
      }
    }
    proj = fmaf(ld(xb, c * S + p) * inv_p, Gc, proj);
    if (sizeof(T) == 4) st(db, c * S + p, Gc);
    // (f16 storage: G is recomputed in the last sweep instead of being rounded through dx)
  }
  if (nrm < P.eps) proj = 0.f;  // clamp_min branch of F.normalize: d ehat / d e = I / eps
  const float sc = dl * inv_p;
  // (3) projection
  for (int c = 0; c < D; ++c) {
    float Gc;
    if (sizeof(T) == 4) {
      Gc = ld(db, c * S + p);
    } else {  // recompute (no f32 scratch per pixel): one more pass over the neighbours
      Gc = 0.f;
      for (int i = 0; i < P.K; ++i) {
#pragma unroll
        for (int role = 0; role < 2; ++role) {
          if (role == 0 ? !ROLE_A : !ROLE_B) continue;
          const float coef = s_coef[(2 * i + role) * kBlock + threadIdx.x];
          if (coef == 0.f) continue;
          const int sg = role == 0 ? 1 : -1;
          const T* nb = (role == 0 ? nbA : nbB) + (size_t)b * D * S;
          const int q = neighbour(P, z, y, x, sg * P.off[i][0], sg * P.off[i][1], sg * P.off[i][2]);
          Gc = fmaf(coef, ld(nb, c * S + q), Gc);
        }
      }
    }
    st(db, c * S + p, (Gc - ld(xb, c * S + p) * inv_p * proj) * sc);
  }
}

}  // namespace pea
