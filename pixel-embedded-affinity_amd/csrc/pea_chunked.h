// pea_chunked.h -- tiled forward with the channels going through LDS in chunks: D = 64 as two chunks of 32 (too wide for
// one region), D = 32 as two chunks of 16 (a 64-byte region pixel: two workgroups per CU).  Included by pea_hip.hip only.
//
// <ehat(p), ehat(q)> = <e(p), e(q)> / (|e(p)| |e(q)|) separates over channels once the norms are taken out: every chunk is
// staged RAW (the D = 32 region geometry: 16x32 tile, 128 bytes of LDS per region pixel), each lane adds the chunk's
// share of its K raw dot products, and the lane that stages a region pixel accumulates that pixel's sum of squares in a
// register (the same lane stages the same pixels in every chunk) and writes 1 / norm into a small LDS array with the last
// chunk.  The far offsets go the same way: per chunk DC channels from global memory, raw dot product and sum of squares
// accumulated.  One read of e per pixel (plus halo), like the D = 16 / 32 kernels; the direct kernel this replaces for
// D = 64 read every neighbour vector from L2 / HBM once per offset (1.29 ms for inference at B=8 x 64 x 544^2).
#pragma once
#include "pea_tiled.h"

namespace pea {

constexpr int kChN = 12;  // max near offsets (register accumulators)
constexpr int kChF = 4;   // max far offsets

// raw staging of one channel chunk; rss[it] accumulates the sum of squares of the it-th region pixel this lane stages
template <typename T, int DC, int PLQ, int NT, int MAXR, bool CROP>
__device__ __forceinline__ void stage_region_raw(const KParams& P, const TParams& Q, rsrc_t eb, unsigned zo, unsigned cs,
                                                 int y0, int x0, char* __restrict__ lds, float* rss, float* __restrict__ s_inv,
                                                 bool last) {
  typedef Lds<DC, PLQ> L;
  int idx = threadIdx.x;
  int r = (int)(((float)idx + 0.5f) * Q.inv_rw);
  int c = idx - r * Q.RW;
#pragma unroll
  for (int it = 0; it < MAXR; ++it) {
    if (idx < Q.R) {
      bool oky, okx;
      const int gy = wrap1<CROP>(y0 - Q.hy0 + r, P.Y, oky);
      const int gx = wrap1<CROP>(x0 - Q.hx0 + c, P.X, okx);
      const unsigned vo = (oky && okx) ? (unsigned)(gy * P.X + gx) * (unsigned)sizeof(T) : kOOB;  // outside => zeros
      float v[DC];
      float ss = 0.f;
#pragma unroll
      for (int ch = 0; ch < DC; ++ch) {
        v[ch] = bl_emb<T>(eb, vo, zo + ch * cs);
        ss = fmaf(v[ch], v[ch], ss);
      }
      rss[it] += ss;
      char* dst = lds + idx * 16;
#pragma unroll
      for (int q = 0; q < L::S; ++q) {
        f4 t;
        t.x = v[4 * q]; t.y = v[4 * q + 1]; t.z = v[4 * q + 2]; t.w = v[4 * q + 3];
        *(f4*)(dst + q * L::kPlaneB) = t;
      }
      if (last) s_inv[idx] = rnorm(rss[it], Q.inv_eps);
    }
    idx += NT;
    r += Q.dr;
    c += Q.dc;
    if (c >= Q.RW) { c -= Q.RW; r += 1; }
  }
}

template <typename T, int D_T, int DC, int TH, int TW, int PLQ, bool CROP, bool TRAIN, bool SELF>
__global__ __launch_bounds__(TH* TW, (DC > 16 ? 2 : 4)) void k_fwd_tiled_chunked(const KParams P, const TParams Q, const T* __restrict__ e,
                                                                 const T* __restrict__ eo, const float* __restrict__ target,
                                                                 const float* __restrict__ weight,
                                                                 const uint8_t* __restrict__ mask, float* __restrict__ affs,
                                                                 float* __restrict__ gout, LossState* __restrict__ st) {
  typedef Lds<DC, PLQ> L;
  static_assert(D_T % DC == 0, "whole chunks");
  constexpr int NT = TH * TW, NW = NT / 64, NCH = D_T / DC, MAXR = (PLQ + NT - 1) / NT;
  constexpr int KN = 8;
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  float* s_inv = (float*)(lds + L::kBytes);  // [PLQ] 1 / norm of the region pixels (all D_T channels)
  float* s_part = s_inv + PLQ;               // [NW][K]
  const int tile = tile_id(Q);
  if (tile >= Q.ntiles) return;
  const int plane = tile / Q.tiles_per_plane;
  const int rem = tile - plane * Q.tiles_per_plane;
  const int ty = rem / Q.tiles_x;
  const int y0 = ty * TH, x0 = (rem - ty * Q.tiles_x) * TW;
  const int b = plane / P.Z, z = plane - b * P.Z;
  const size_t S = (size_t)P.S;
  const unsigned YX = (unsigned)(P.Y * P.X);
  const rsrc_t eB = mkbuf(e + (size_t)b * D_T * S), oB = mkbuf(eo + (size_t)b * D_T * S);
  FwdU U;
  U.aB = mkbuf(affs ? affs + (size_t)b * P.K * S : nullptr);
  U.gB = mkbuf(gout ? gout + (size_t)b * P.K * S : nullptr);
  U.tB = mkbuf(target + (size_t)b * P.tbs);
  U.wB = mkbuf(weight + (size_t)b * P.wbs);
  U.mB = mkbuf(mask ? mask + (size_t)b * P.mbs : nullptr);
  U.ks = 0xffffffffu;  // no second resource here (plan_tiles keeps >= 2 GiB blocks away from this kernel)
  U.aB1 = U.aB; U.gB1 = U.gB; U.tB1 = U.tB; U.wB1 = U.wB;
  U.kcs = (unsigned)P.S * 4u;
  U.kzo = (unsigned)z * YX * 4u;
  U.S32 = (unsigned)P.S;
  U.has_a = affs != nullptr; U.has_g = gout != nullptr; U.has_m = mask != nullptr;
  U.af = P.flags & kActMask;
  const unsigned ecs = (unsigned)P.S * (unsigned)sizeof(T);  // embedding channel stride, bytes
  const unsigned ezo = (unsigned)z * YX * (unsigned)sizeof(T);

  int ly, lx;
  lane_pixel<TW>(ly, lx);
  const int py = y0 + ly, px = x0 + lx;
  const bool live = py < P.Y && px < P.X;
  const unsigned po = (unsigned)(py * P.X + px);
  const unsigned pb = live ? po * 4u : kOOB;
  const unsigned pm = live ? po : kOOB;
  const unsigned pe = live ? po * (unsigned)sizeof(T) : kOOB;
  const int pr = (ly + Q.hy0) * Q.RW + lx + Q.hx0;

  // (target / weight / mask are requested in the epilogue, not up front as in k_fwd_tiled: holding them across the
  // chunk loop made the register allocator spill, and this toolchain's spill stores are exposed to the same
  // store-data hazard as bs128 -- a randomised sweep caught a wrong g in one f16 instantiation with 25 spilled VGPRs)
  Twm<KN> sa;
  Twm<2> sf;

  // far neighbours: address and validity once
  unsigned fvo[kChF], fzo[kChF];
  bool fok[kChF];
#pragma unroll
  for (int k = 0; k < kChF; ++k) {
    fok[k] = false; fvo[k] = kOOB; fzo[k] = 0;
    if (k < Q.n_far) {
      const OffEnt fe = Q.far[k];
      bool okz, oky, okx;
      const int zz = wrap1<CROP>(z + fe.d, P.Z, okz);
      const int yy = wrap1<CROP>(py + ent_oy(fe), P.Y, oky);
      const int xx = wrap1<CROP>(px + ent_ox(fe), P.X, okx);
      fok[k] = live && okz && oky && okx;
      fzo[k] = (unsigned)(CROP ? min(max(zz, 0), P.Z - 1) : zz) * YX * (unsigned)sizeof(T);
      fvo[k] = fok[k] ? (unsigned)(yy * P.X + xx) * (unsigned)sizeof(T) : kOOB;
    }
  }

  float anear[kChN], fdot[kChF], fsq[kChF], rss[MAXR];
#pragma unroll
  for (int k = 0; k < kChN; ++k) anear[k] = 0.f;
#pragma unroll
  for (int k = 0; k < kChF; ++k) { fdot[k] = 0.f; fsq[k] = 0.f; }
#pragma unroll
  for (int it = 0; it < MAXR; ++it) rss[it] = 0.f;
  float own_ss = 0.f;

#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const unsigned cho = (unsigned)(ch * DC) * ecs;  // byte offset of the chunk's first channel plane
    if (ch > 0) lds_barrier();                       // every lane is done with the previous chunk's region
    stage_region_raw<T, DC, PLQ, NT, MAXR, CROP>(P, Q, oB, ezo + cho, ecs, y0, x0, lds, rss, s_inv, ch == NCH - 1);
    float ownc[DC], fvA[DC], fvB[DC];
    if (!SELF) {
#pragma unroll
      for (int c = 0; c < DC; ++c) ownc[c] = bl_emb<T>(eB, pe, ezo + cho + c * ecs);
    }
#define PEA_CH_LOAD_FAR(fv, k) \
  { _Pragma("unroll") for (int c = 0; c < DC; ++c) fv[c] = bl_emb<T>(oB, fvo[k], fzo[k] + cho + c * ecs); }
#define PEA_CH_ACC_FAR(fv, k)                                                  \
  {                                                                            \
    _Pragma("unroll") for (int c = 0; c < DC; ++c) {                           \
      fdot[k] = fmaf(ownc[c], fv[c], fdot[k]);                                 \
      fsq[k] = fmaf(fv[c], fv[c], fsq[k]);                                     \
    }                                                                          \
  }
    if (Q.n_far > 0) PEA_CH_LOAD_FAR(fvA, 0)  // in flight across the barrier and the near-offset work
    lds_barrier();
    if (SELF) {
      lds_pixel<DC, PLQ>(lds, pr, ownc);
    } else {
#pragma unroll
      for (int c = 0; c < DC; ++c) own_ss = fmaf(ownc[c], ownc[c], own_ss);
    }
#pragma unroll
    for (int k = 0; k < kChN; ++k) {
      if (k < Q.n_near) {  // uniform
        float v[DC];
        lds_pixel<DC, PLQ>(lds, pr + Q.near[k].d, v);
#pragma unroll
        for (int c = 0; c < DC; ++c) anear[k] = fmaf(ownc[c], v[c], anear[k]);
      }
    }
    if (Q.n_far > 1) PEA_CH_LOAD_FAR(fvB, 1)
    if (Q.n_far > 0) PEA_CH_ACC_FAR(fvA, 0)
    if (Q.n_far > 2) PEA_CH_LOAD_FAR(fvA, 2)
    if (Q.n_far > 1) PEA_CH_ACC_FAR(fvB, 1)
    if (Q.n_far > 3) PEA_CH_LOAD_FAR(fvB, 3)
    if (Q.n_far > 2) PEA_CH_ACC_FAR(fvA, 2)
    if (Q.n_far > 3) PEA_CH_ACC_FAR(fvB, 3)
#undef PEA_CH_LOAD_FAR
#undef PEA_CH_ACC_FAR
  }

  // ---- epilogue: the norms come in now (s_inv was completed before the last chunk's barrier)
  const float inv_own = SELF ? s_inv[pr] : rnorm(own_ss, Q.inv_eps);
#pragma unroll
  for (int k0 = 0; k0 < kChN; k0 += KN) {
    if (k0 < Q.n_near) {
      if (TRAIN) fwd_load_twm<KN>(sa, U, Q.near, k0, Q.n_near, pb, pm);
#pragma unroll
      for (int u = 0; u < KN; ++u) {
        if (k0 + u < kChN && k0 + u < Q.n_near) {  // uniform
          const OffEnt en = Q.near[k0 + u];
          float a = anear[k0 + u] * inv_own * s_inv[pr + en.d];
          bool valid = live;
          if (CROP) {
            const bool inside = (unsigned)(py + ent_oy(en)) < (unsigned)P.Y && (unsigned)(px + ent_ox(en)) < (unsigned)P.X;
            a = inside ? a : 0.f;
            valid = valid && inside;
          }
          fwd_finish<TRAIN>(U, P.K, s_part, en, a, valid, sa.t[u], sa.w[u], sa.m[u], pb);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kChF; k += 2) {
    if (k < Q.n_far) {
      if (TRAIN) fwd_load_twm<2>(sf, U, Q.far, k, Q.n_far, pb, pm);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (k + u < Q.n_far) {
          const float a = fok[k + u] ? fdot[k + u] * inv_own * rnorm(fsq[k + u], Q.inv_eps) : 0.f;
          fwd_finish<TRAIN>(U, P.K, s_part, Q.far[k + u], a, fok[k + u], sf.t[u], sf.w[u], sf.m[u], pb);
        }
      }
    }
  }

  if (TRAIN) {
    lds_barrier();
    if (threadIdx.x < P.K) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += s_part[w * P.K + threadIdx.x];
      loss_accumulate(st, tile, threadIdx.x, v);
    }
  }
}

}  // namespace pea
