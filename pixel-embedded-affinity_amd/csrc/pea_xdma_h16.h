// pea_xdma_h16.h -- the LDS-DMA cross kernels (pea_xdma.h) for f16 STORAGE of the embedding (BASELINE.json configs[4]: D = 64 in
// half precision; e and d loss / d e are __half, all arithmetic f32, target / weight / affs / g / 1/norm stay f32).
//
// Round 2 left f16 storage on the box kernels (tiled / chunked forward, direct backward: 0.93 ms at B=8 x 64 x 544^2 against 0.65 ms
// for the same shape stored in f32): an f16 plane in LDS costs MORE instructions per channel in the gather (ds_read_u16 has no
// two-plane form, v_fma_mix is not packed).  So the gather stays exactly the f32 one, and the half precision ends at the staging:
//   * the channel planes arrive RAW in f16 by buffer_load_dwordx4 ... lds -- 8 pixels per lane, ONE wave instruction per plane
//     and wave instead of two, half the bytes from HBM and, what bounds the backward, half the bytes from L2 into the CU --
//     into a ring of three chunk buffers of half-size planes;
//   * before a chunk is gathered the workgroup converts it (ds_read_b64 -> 4 x v_cvt_f32_f16 -> ds_write_b128, at most two quads
//     per lane and plane) into ONE f32 working buffer in which the two channels of a pixel sit side by side: the gather is one
//     ds_read_b64 per pair (256 B/clk) instead of the planar ds_read2st64_b32 (128 B/clk).
// Two barriers per chunk instead of one (converted / consumed); the second workgroup of the CU fills them.  LDS: backward
// 2 x 13 KB + 6 x 6.5 KB = 64 KB (two workgroups per CU), forward 2 x 7.5 KB + 6 x 3.75 KB = 37.5 KB (three).
// HW (the default since the end of round 3, PEA_H16_HW=0 switches back): the working buffer itself stays in f16 -- [pixel][2 channels]
// halves written by an interleave step without any conversion.  The forward's per-pair work over the chunk's two channels
// (dot += <own, v>, ssq += <v, v>) is then v_dot2_f32_f16: scalar accumulators, 48 VGPRs, 30 KB, four workgroups per CU, 193 -> 167 us
// at B=8 x 64 x 544^2; the backward takes the halves with v_fma_mix_f32 (305 -> 300 us).  Products of two halves are exact in f32 and
// the accumulation is f32 either way; f16 denormals are not flushed (tests/test_gpu_parity.py::test_f16_denormal_embeddings).
// Self loss / inference, 2D, X % 8 == 0 (an 8-pixel DMA item never straddles a row end), axis-aligned stencils, D in {16, 32, 64}.
#pragma once
#include <type_traits>

#include "pea_xdma_pf.h"

namespace pea {

typedef _Float16 h4_t __attribute__((ext_vector_type(4)));

// geometry of this lane's DMA items: up to two QUADS (4 pixels: the f32 1 / norm plane, and the conversion) and one OCT
// (8 pixels: the f16 channel planes).  Same region order as pea_xdma.h (VF rows of TW pixels, then strip rows of SW pixels).
// (Plain locals, not a struct: with the operands of the LDS-DMA builtin taken from members of a local struct, ROCm 7.2's host pass
//  silently emits no stub for the kernel -- an undefined symbol at load time.  Found by bisection.)
template <int TH, int TW, bool CROP>
__device__ __forceinline__ void x_items(const KParams& P, const XParams& C, int y0, int x0, int wave, int lane, unsigned (&vo)[2],
                                        bool (&act)[2], int (&qq)[2], unsigned& vo8, bool& act8) {
  constexpr int NT = TH * TW;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int q = (s * (NT / 64) + wave) * 64 + lane;
    qq[s] = q;
    int gy, gx;
    if (q < C.QV) {
      gy = y0 - C.hy0 + q / (TW / 4);
      gx = x0 + 4 * (q % (TW / 4));
    } else {
      const int k = q - C.QV;
      const int sh = C.SW == 64 ? 4 : 3;
      const int cc = 4 * (k & ((1 << sh) - 1));
      gy = y0 + (k >> sh);
      gx = cc < C.split ? x0 + TW + cc : x0 - C.SW + cc;
    }
    act[s] = q < C.QA;
    bool oky, okx;
    gy = wrap1<CROP>(gy, P.Y, oky);
    gx = wrap1<CROP>(gx, P.X, okx);
    vo[s] = (act[s] && oky && okx) ? (unsigned)(gy * P.X + gx) * 4u : kOOB;
  }
  const int o = wave * 64 + lane;
  int gy, gx;
  if (o < (C.QV >> 1)) {
    gy = y0 - C.hy0 + o / (TW / 8);
    gx = x0 + 8 * (o % (TW / 8));
  } else {
    const int k = o - (C.QV >> 1);
    const int sh = C.SW == 64 ? 3 : 2;
    const int cc = 8 * (k & ((1 << sh) - 1));
    gy = y0 + (k >> sh);
    gx = cc < C.split ? x0 + TW + cc : x0 - C.SW + cc;
  }
  act8 = o < (C.QA >> 1);
  bool oky, okx;
  gy = wrap1<CROP>(gy, P.Y, oky);
  gx = wrap1<CROP>(gx, P.X, okx);
  vo8 = (act8 && oky && okx) ? (unsigned)(gy * P.X + gx) * 2u : kOOB;
}

// wait until only the youngest chunk's DMA (npc wave instructions of this wave: 2 or 0) may be in flight, then the barrier
#define PEA_HWAIT1(npc)                                                                          \
  {                                                                                              \
    if ((npc) == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");     \
    else if ((npc) == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                \
  }
#define PEA_HWAIT0() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

// f16 chunk `rbuf` of the ring -> the f32 working buffer, the two channels of a pixel SIDE BY SIDE (8 bytes per region pixel).
// The layout is free here (this step writes it, not the DMA), and it is what makes the gather cheap: one ds_read_b64 per
// (offset, role) pair delivers both channels at 256 B/clk, where the planar form needs ds_read2st64_b32 at 128 B/clk -- and the
// stamps (profiles/microbench/stamp_bwd.hip) say the gather phases are bound by exactly that pipe.
template <int PS, int NT>
__device__ __forceinline__ void convert_chunk(char* W, const char* R, int rbuf, int qa) {
  constexpr int PH = PS / 2;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int q = s * NT + (int)threadIdx.x;  // = (s * (NT / 64) + wave) * 64 + lane: the quads this lane moved by DMA as well
    if (q < qa) {
      const h4_t h0 = *(const h4_t*)(R + (rbuf * 2) * PH + q * 8);
      const h4_t h1 = *(const h4_t*)(R + (rbuf * 2 + 1) * PH + q * 8);
      f4 a, b;
      a.x = (float)h0.x; a.y = (float)h1.x; a.z = (float)h0.y; a.w = (float)h1.y;
      b.x = (float)h0.z; b.y = (float)h1.z; b.z = (float)h0.w; b.w = (float)h1.w;
      *(f4*)(W + q * 32) = a;
      *(f4*)(W + q * 32 + 16) = b;
    }
  }
}

// f16 chunk `rbuf` of the ring -> a HALF-precision working buffer, the two channels of a pixel side by side (4 bytes per region
// pixel): no conversion at all, half the bytes to write and a quarter of the bytes to gather -- the forward's per-pair work is
// `dot += <own, v>`, `ssq += <v, v>` over the two channels, which is exactly v_dot2_f32_f16 (f16 products are exact in f32, the
// accumulation is f32): one ds_read_b32 + two v_dot2 per pair, scalar accumulators (half the registers of the packed-f32 form).
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
template <int PS, int NT>
__device__ __forceinline__ void interleave_chunk(char* W, const char* R, int rbuf, int qa) {
  constexpr int PH = PS / 2;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int q = s * NT + (int)threadIdx.x;
    if (q < qa) {
      const h4_t h0 = *(const h4_t*)(R + (rbuf * 2) * PH + q * 8);
      const h4_t h1 = *(const h4_t*)(R + (rbuf * 2 + 1) * PH + q * 8);
      typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
      *(h8_t*)(W + q * 16) = (h8_t){h0.x, h1.x, h0.y, h1.y, h0.z, h1.z, h0.w, h1.w};
    }
  }
}

// d e_c = (G_c - ehat_c <ehat, G>) * dl / n with the contraction spelled out: k_bwd_xdma_h<PF> and k_bwd_xdma_hqs (pea_xdma_hq.h) then
// round alike and agree bit for bit (left to the compiler, 44 of 958 464 f16 gradients differed by one ulp between the two)
__device__ __forceinline__ float pf_finish(float acc, float o, float proj, float inv_own, float dl) {
  return __builtin_fmaf(-o, proj, acc) * inv_own * dl;
}

// ------------------------------------------------------------------------------------------------------------------
// backward, self loss (both roles), f16 e / de
// ------------------------------------------------------------------------------------------------------------------
// PF: the projection first (pea_xdma_pf.h): `affs` = the raw cosine map of the forward; a chunk then finishes its two channels
// (stored at once, in f16), there is no G array and no second read of the own pixel; WPE = 6 with the small planes.
// HW: the working buffer stays in f16 (interleave_chunk: no conversion, half the LDS bytes written, a ds_read_b32 per pair instead of
// a ds_read_b64); the FMAs take the halves directly (v_fma_mix_f32: f16 operand, f32 coefficient and accumulator -- the same arithmetic)
// OTHER: the cross loss with a detached second operand (ema_embedding_loss behind convert_consistency_flip's detach), role A: xt / invp
// are the SECOND operand and its 1 / norm plane (staged: plan_xdma mode 2), the own pixel comes from the OWN tile staged beside each
// chunk by wave 0 (f16: 1 KB per channel, no halo), `own_inv` is the own operand's signed 1 / norm plane.  Projection first only.
template <int D_T, int TH, int TW, int PSU, bool CROP, int XP = kXP, bool PF = false, int WPE = 4, bool HW = false, bool OTHER = false>
__global__ __launch_bounds__(TH* TW, WPE) void k_bwd_xdma_h(const KParams P, const XParams C, const __half* __restrict__ xt,
                                                             const float* __restrict__ invp, const float* __restrict__ gin,
                                                             const float* __restrict__ affs, const float* __restrict__ dloss,
                                                             __half* __restrict__ dx, const __half* __restrict__ own,
                                                             const float* __restrict__ own_inv) {
  constexpr int NT = TH * TW, PS = PSU * 256, PH = PS / 2, NP = D_T / 2;
  static_assert(TW == 32 && D_T % 2 == 0 && PS % 512 == 0, "lane mapping / channel pairs / half planes in whole 256-byte units");
  static_assert(!OTHER || (PF && HW), "the role-A instantiation: projection first, f16 working buffer");
  constexpr int OWNR = 5 * PS;  // OTHER: three buffers x two channels x 1 KB of own tile behind the working planes and the ring
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  char* const W = lds;            // two f32 working planes (the 1 / norm plane sits in the second one first)
  char* const R = lds + 2 * PS;   // ring: 3 buffers x 2 f16 planes
  int tile, b, z, y0, x0;
  if (!xdma_tile<TH, TW>(C, P, tile, b, z, y0, x0)) return;
  const size_t S = (size_t)P.S;
  const unsigned YX = (unsigned)(P.Y * P.X);
  const rsrc_t xB = mkbuf(xt + (size_t)b * D_T * S), dB = mkbuf(dx + (size_t)b * D_T * S);
  const rsrc_t gB = mkbuf(gin + (size_t)b * P.K * S), iB = mkbuf(invp + (size_t)b * S);
  const unsigned hcs = (unsigned)P.S * 2u, hzo = (unsigned)z * YX * 2u;  // e / de (f16): channel stride, plane offset
  const unsigned fcs = (unsigned)P.S * 4u, fzo = (unsigned)z * YX * 4u;  // g, 1 / norm (f32)
  const float dl = dloss ? dloss[0] : 1.f;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
  const int py = y0 + ly, px = x0 + lx;
  const bool live = py < P.Y && px < P.X;
  const unsigned po = (unsigned)(py * P.X + px);
  const unsigned ph = live ? po * 2u : kOOB;

  unsigned vo[2], vo8;
  bool act[2], act8;
  int qq[2];
  x_items<TH, TW, CROP>(P, C, y0, x0, wave, lane, vo, act, qq, vo8, act8);
  const int wbase = wave * 1024;
  const bool ownw = OTHER && wave == 0;  // uniform: the wave that moves the own tile (64 octs = 16 rows of 32 pixels)
  const rsrc_t oB = mkbuf(OTHER ? own + (size_t)b * D_T * S : nullptr);
  unsigned ownvo = kOOB;
  if (OTHER) {
    const int oy = y0 + (lane >> 2), ox = x0 + 8 * (lane & 3);
    ownvo = (oy < P.Y && ox < P.X) ? (unsigned)(oy * P.X + ox) * 2u : kOOB;
  }
  const int npc = 2 * (__builtin_amdgcn_ballot_w64(act8) != 0) + (ownw ? 2 : 0);
#define PEA_HDMA16(rbuf, ch)                                                                                                  \
  {                                                                                                                         \
    if (act8) {                                                                                                             \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(R + ((rbuf) * 2) * PH + wbase), 16, vo8, hzo + (unsigned)(ch) * hcs, 0, 0);        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(R + ((rbuf) * 2 + 1) * PH + wbase), 16, vo8, hzo + (unsigned)((ch) + 1) * hcs, 0, 0); \
    }                                                                                                                       \
    if (ownw) {                                                                                                             \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(oB, (lds_ptr_t)(lds + OWNR + ((rbuf) * 2) * 1024), 16, ownvo, hzo + (unsigned)(ch) * hcs, 0, 0);        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(oB, (lds_ptr_t)(lds + OWNR + ((rbuf) * 2 + 1) * 1024), 16, ownvo, hzo + (unsigned)((ch) + 1) * hcs, 0, 0); \
    }                                                                                                                       \
  }
  // the 1 / norm plane (f32) -> the second working plane
  if (act[0]) __builtin_amdgcn_raw_ptr_buffer_load_lds(iB, (lds_ptr_t)(W + PS + wbase), 16, vo[0], fzo, 0, 0);
  if (act[1]) __builtin_amdgcn_raw_ptr_buffer_load_lds(iB, (lds_ptr_t)(W + PS + wbase + (NT / 64) * 1024), 16, vo[1], fzo, 0, 0);
  PEA_HDMA16(0, 0)

  // ---- g of every pair (role A at p, role B at p - o) and the LDS slot of every neighbour (pea_xdma.h k_bwd_xdma)
  const rsrc_t aB = mkbuf(PF ? affs + (size_t)b * P.K * S : nullptr);
  float proj = 0.f;
  const unsigned pg = live ? po * 4u : 0xC0000000u;
  static_assert(XP % 2 == 0, "coefficients are kept two to a register pair");
  f2 cx2[XP / 2], cy2[XP / 2];  // pair k in half (k & 1) of element k / 2
  int ax[XP], ay[XP];
  const int vown = ((C.hy0 + ly) * TW + lx) * 4;   // byte offset of the own pixel in a PLANE (the 1 / norm plane); x 2 in W
  const int hrow = (C.QV * 4 + ly * C.SW) * 4;
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    const int go = C.xgo[k];
    const int t = px + go;
    const bool out = (unsigned)t >= (unsigned)P.X;
    const int fix = go > 0 ? -P.X : P.X;
    const unsigned o = CROP ? (out ? kOOB : pg + (unsigned)(go * 4)) : pg + (unsigned)((out ? go + fix : go) * 4);
    const float gk = bl32(gB, k < C.npx ? o : kOOB, fzo + (unsigned)C.xgi[k] * fcs);
    cx2[k / 2][k & 1] = gk;
    if (PF) proj = fmaf(gk, bl32(aB, k < C.npx ? o : kOOB, fzo + (unsigned)C.xgi[k] * fcs), proj);
    const int d = C.xd[k], c = lx + d;
    ax[k] = (unsigned)c < (unsigned)TW ? vown + d * 4 : hrow + (c & C.xm[k]) * 4;
  }
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    const int go = C.ygo[k];
    const int t = py + go;
    const bool out = (unsigned)t >= (unsigned)P.Y;
    const int fix = go > 0 ? -P.Y : P.Y;
    const unsigned o = CROP ? (out ? kOOB : pg + (unsigned)(go * P.X * 4)) : pg + (unsigned)((out ? go + fix : go) * P.X * 4);
    const float gk = bl32(gB, k < C.npy ? o : kOOB, fzo + (unsigned)C.ygi[k] * fcs);
    cy2[k / 2][k & 1] = gk;
    if (PF) proj = fmaf(gk, bl32(aB, k < C.npy ? o : kOOB, fzo + (unsigned)C.ygi[k] * fcs), proj);
    ay[k] = vown + C.yd[k] * TW * 4;
  }
  float invo_g = 0.f;
  if (OTHER) invo_g = bl32(mkbuf(own_inv + (size_t)b * S), live ? po * 4u : kOOB, fzo);
  if (NP > 1) PEA_HDMA16(1, 2)
  // the 1 / norm plane, chunk 0 and g have landed (chunk 1 may still fly)
  if (NP > 1) PEA_HWAIT1(npc)
  else PEA_HWAIT0()
  const float invo = OTHER ? invo_g : *(const float*)(W + PS + vown);
  const float inv_own = fabsf(invo);
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    cx2[k / 2][k & 1] *= fabsf(*(const float*)(W + PS + ax[k]));
    cy2[k / 2][k & 1] *= fabsf(*(const float*)(W + PS + ay[k]));
    if (k & 1) asm volatile("" : "+v"(cx2[k / 2]), "+v"(cy2[k / 2]));
  }
  lds_barrier();  // the 1 / norm plane is dead: the working planes may be written
  if (NP > 2) PEA_HDMA16(2, 4)

  constexpr bool KEEP = D_T <= 16 && !PF;
  f2 G[PF ? 1 : NP], eh[KEEP ? NP : 1];
  if (PF) {
    if (invo < 0.f) proj = 0.f;  // clamp branch of F.normalize
    asm volatile("" : "+v"(proj));
  }
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    f2 o, acc = {0.f, 0.f};
    if constexpr (HW) {
      // (PEA_ABL_H_*: diagnostic builds of profiles/microbench/abl_bwd_h16.hip only -- timing with a phase compiled out)
#ifndef PEA_ABL_H_NOILV
      interleave_chunk<PS, NT>(W, R, ps % 3, C.QA);
      lds_barrier();  // the working buffer holds chunk ps
#endif
      h2_t oh = *(const h2_t*)(W + vown);
      if (OTHER) {  // the own pixel from the own tile ([channel][512 pixels] halves)
        oh.x = *(const _Float16*)(lds + OWNR + ((ps % 3) * 2) * 1024 + (int)threadIdx.x * 2);
        oh.y = *(const _Float16*)(lds + OWNR + ((ps % 3) * 2 + 1) * 1024 + (int)threadIdx.x * 2);
      }
      o = (f2){(float)oh.x, (float)oh.y} * inv_own;
#ifndef PEA_ABL_H_NOGATHER
#pragma unroll
      for (int k = 0; k < XP; ++k) {
        const h2_t v = *(const h2_t*)(W + ax[k]);
        const float c = (k & 1) ? cx2[k / 2].y : cx2[k / 2].x;
        acc.x = __builtin_fmaf((float)v.x, c, acc.x);
        acc.y = __builtin_fmaf((float)v.y, c, acc.y);
        if (k % 5 == 4) asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int k = 0; k < XP; ++k) {
        const h2_t v = *(const h2_t*)(W + ay[k]);
        const float c = (k & 1) ? cy2[k / 2].y : cy2[k / 2].x;
        acc.x = __builtin_fmaf((float)v.x, c, acc.x);
        acc.y = __builtin_fmaf((float)v.y, c, acc.y);
        if (k % 5 == 4) asm volatile("" ::: "memory");
      }
#else
      acc = (f2){cx2[ps % (XP / 2)].x, cy2[ps % (XP / 2)].y};
#endif
    } else {
    convert_chunk<PS, NT>(W, R, ps % 3, C.QA);
    lds_barrier();  // the working buffer holds chunk ps
    o = *(const f2*)(W + 2 * vown);
    o = o * inv_own;
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      const f2 v = *(const f2*)(W + 2 * ax[k]);
      acc = (k & 1) ? pk_fma_c<true>(cx2[k / 2], v, acc) : pk_fma_c<false>(cx2[k / 2], v, acc);
      if (k % 5 == 4) asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      const f2 v = *(const f2*)(W + 2 * ay[k]);
      acc = (k & 1) ? pk_fma_c<true>(cy2[k / 2], v, acc) : pk_fma_c<false>(cy2[k / 2], v, acc);
      if (k % 5 == 4) asm volatile("" ::: "memory");
    }
    }
    if (KEEP) {
      eh[ps] = o;
      asm volatile("" : "+v"(eh[ps]));
    }
    float sx = 0.f, sy = 0.f;
    if (PF) {  // this chunk's two channels are final; stored BEHIND the hand-off below: a store issued just before a counted wait is
               // still in flight when the wait is reached, and stores cannot be counted on (pea_xdma_pf.h) -- behind it, it has a
               // whole chunk's time to retire before the next one
      sx = pf_finish(acc.x, o.x, proj, inv_own, dl);
      sy = pf_finish(acc.y, o.y, proj, inv_own, dl);
      asm volatile("" : "+v"(sx), "+v"(sy));
    } else {
      if (!KEEP) {
        proj = fmaf(o.x, acc.x, fmaf(o.y, acc.y, proj));
        asm volatile("" : "+v"(proj));
      }
      asm volatile("" : "+v"(acc));
      G[ps] = acc;
    }
    if (ps + 1 < NP) {
      // everyone is done with the working buffer and with ring buffer ps % 3; chunk ps + 1 has landed (chunk ps + 2 and, PF, the
      // stores of the last chunks are not counted on, pea_xdma_pf.h)
      // (loads only: a store may retire before an older load, pea_xdma_pf.h)
      // (and a wave that issues no DMA -- npc == 0 -- waits for nothing: vmcnt(0) would make it drain its stores at every chunk)
      const int nd = ps + 2 < NP ? 1 : 0;
      if (PF && npc == 0) lds_barrier();
      else pf_wait(nd * npc);
#ifndef PEA_ABL_H_NODMA
      if (ps + 3 < NP) PEA_HDMA16(ps % 3, 2 * ps + 6)
#endif
    }
#ifndef PEA_ABL_H_NOSTORE
    if (PF) {
      bs_emb<__half, true>(dB, sx, ph, hzo + (unsigned)(2 * ps) * hcs);
      bs_emb<__half, true>(dB, sy, ph, hzo + (unsigned)(2 * ps + 1) * hcs);
    }
#endif
  }
#undef PEA_HDMA16

  if constexpr (!PF) {
    if (KEEP) {
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) proj = fmaf(eh[ps].x, G[ps].x, fmaf(eh[ps].y, G[ps].y, proj));
    }
    if (invo < 0.f) proj = 0.f;  // clamp branch of F.normalize
    const float pn = proj * inv_own, sc = dl * inv_own;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      float ex, ey;
      if (KEEP) { ex = eh[ps].x * proj; ey = eh[ps].y * proj; }
      else {
        ex = bl_emb<__half>(xB, ph, hzo + (unsigned)(2 * ps) * hcs) * pn;
        ey = bl_emb<__half>(xB, ph, hzo + (unsigned)(2 * ps + 1) * hcs) * pn;
      }
      bs_emb<__half, true>(dB, (G[ps].x - ex) * sc, ph, hzo + (unsigned)(2 * ps) * hcs);
      bs_emb<__half, true>(dB, (G[ps].y - ey) * sc, ph, hzo + (unsigned)(2 * ps + 1) * hcs);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// forward (self loss with TRAIN, or inference), f16 e; writes the f32 1 / norm plane for the backward.  Epilogue as k_fwd_xdma.
// ------------------------------------------------------------------------------------------------------------------
// HW: the working buffer stays in f16 (interleave_chunk) and the gather runs on v_dot2_f32_f16
// NXP: offsets the gather walks (kXP = 10; 8 for tables with no more, e.g. BASELINE configs[4]'s offsets[:8] -- an unused slot costs
// its LDS read and its two dot products all the same)
// OTHER: the cross loss a_i(p) = <ehat(p), ehat_other(p + o_i)>: `e` is the SECOND operand (staged), the own pixel comes from the own
// tile of `own` staged beside each chunk by wave 0; both 1 / norm planes are written (inv_out: own, inv_other_out: the second operand's)
template <int D_T, int TH, int TW, int PSU, bool CROP, bool TRAIN, int WPE, bool HW = false, int NXP = kXP, bool OTHER = false>
__global__ __launch_bounds__(TH* TW, WPE) void k_fwd_xdma_h(const KParams P, const XParams C, const __half* __restrict__ e,
                                                             const float* __restrict__ target, const float* __restrict__ weight,
                                                             const uint8_t* __restrict__ mask, float* __restrict__ affs,
                                                             float* __restrict__ gout, LossState* __restrict__ st,
                                                             float* __restrict__ inv_out, const __half* __restrict__ own,
                                                             float* __restrict__ inv_other_out) {
  static_assert(!OTHER || (HW && TRAIN), "the cross-loss instantiation: f16 working buffer, training");
  constexpr int NT = TH * TW, PS = PSU * 256, PH = PS / 2, NP = D_T / 2, TP = NT, QP = TP / 4, NSL = QP / 64;
  constexpr int KMAX = kXP;
  constexpr int ITEMS = (KMAX * QP + NT - 1) / NT;
  static_assert(TW == 32 && D_T % 2 == 0 && QP % 64 == 0 && PS % 512 == 0, "lane mapping / channel pairs");
  constexpr int WP = HW ? 1 : 2;  // planes' worth of working buffer (HW: 4 bytes per region pixel)
  static_assert(KMAX * TP * 4 + KMAX * NSL * 4 <= (WP + 3) * PS, "the parked dot products fit the dead planes");
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  char* const W = lds;
  char* const R = lds + WP * PS;
  float* sA = (float*)lds;                          // [K][TP] dot products, over the dead planes
  float* s_part = (float*)(lds + KMAX * TP * 4);    // [K][NSL]
  int tile, b, z, y0, x0;
  if (!xdma_tile<TH, TW>(C, P, tile, b, z, y0, x0)) return;
  const size_t S = (size_t)P.S;
  const unsigned YX = (unsigned)(P.Y * P.X);
  const rsrc_t xB = mkbuf(e + (size_t)b * D_T * S);
  const rsrc_t aB = mkbuf(affs ? affs + (size_t)b * P.K * S : nullptr), gB = mkbuf(gout ? gout + (size_t)b * P.K * S : nullptr);
  const rsrc_t tB = mkbuf(TRAIN ? target + (size_t)b * P.tbs : nullptr), wB = mkbuf(TRAIN ? weight + (size_t)b * P.wbs : nullptr);
  const rsrc_t mB = mkbuf(mask ? mask + (size_t)b * P.mbs : nullptr);
  const rsrc_t iB = mkbuf(inv_out ? inv_out + (size_t)b * S : nullptr);
  const unsigned hcs = (unsigned)P.S * 2u, hzo = (unsigned)z * YX * 2u;
  const unsigned ecs = (unsigned)P.S * 4u, ezo = (unsigned)z * YX * 4u;  // the f32 tensors
  const bool has_a = affs != nullptr, has_g = gout != nullptr, has_m = mask != nullptr;
  const unsigned af = P.flags & kActMask;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;

  unsigned vo[2], vo8;
  bool act[2], act8;
  int qq[2];
  x_items<TH, TW, CROP>(P, C, y0, x0, wave, lane, vo, act, qq, vo8, act8);
  const int wbase = wave * 1024;
  constexpr int OWNR = (WP + 3) * PS;  // OTHER: three buffers x two channels x 1 KB of own tile behind the ring
  const bool ownw = OTHER && wave == 0;  // uniform: the wave that moves the own tile
  const rsrc_t oB = mkbuf(OTHER ? own + (size_t)b * D_T * S : nullptr);
  unsigned ownvo = kOOB;
  if (OTHER) {
    const int oy = y0 + (lane >> 2), ox = x0 + 8 * (lane & 3);
    ownvo = (oy < P.Y && ox < P.X) ? (unsigned)(oy * P.X + ox) * 2u : kOOB;
  }
  const int npc = 2 * (__builtin_amdgcn_ballot_w64(act8) != 0) + (ownw ? 2 : 0);
// (ONE statement: the call sites are `if (..) PEA_HDMA16(..)` -- as two statements the own tile's DMA escaped the condition, was requested
//  for chunks that do not exist and read past the end of the tensor: harmless bytes, and a memory fault once the tensor ended near a
//  mapping's end.  Found at 400 x 400; tests/test_gpu_fullsize2.py runs the full size.)
#define PEA_HDMA16(rbuf, ch)                                                                                                  \
  {                                                                                                                         \
    if (act8) {                                                                                                             \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(R + ((rbuf) * 2) * PH + wbase), 16, vo8, hzo + (unsigned)(ch) * hcs, 0, 0);        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(R + ((rbuf) * 2 + 1) * PH + wbase), 16, vo8, hzo + (unsigned)((ch) + 1) * hcs, 0, 0); \
    }                                                                                                                       \
    if (ownw) {                                                                                                             \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(oB, (lds_ptr_t)(lds + OWNR + ((rbuf) * 2) * 1024), 16, ownvo, hzo + (unsigned)(ch) * hcs, 0, 0);        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(oB, (lds_ptr_t)(lds + OWNR + ((rbuf) * 2 + 1) * 1024), 16, ownvo, hzo + (unsigned)((ch) + 1) * hcs, 0, 0); \
    }                                                                                                                       \
  }
  PEA_HDMA16(0, 0)
  if (NP > 1) PEA_HDMA16(1, 2)

  int an[NXP];
  const int vown = ((C.hy0 + ly) * TW + lx) * 4;
  const int hrow = (C.QV * 4 + ly * C.SW) * 4;
#pragma unroll
  for (int k = 0; k < NXP; ++k) {
    const int d = C.fd[k], c = lx + d;
    const int a_x = (unsigned)c < (unsigned)TW ? vown + d * 4 : hrow + (c & C.fm[k]) * 4;
    an[k] = C.fax[k] ? a_x : vown + d * TW * 4;
  }
  if (NP > 1) PEA_HWAIT1(npc)
  else PEA_HWAIT0()
  if (NP > 2) PEA_HDMA16(2, 4)

  // HW: scalar accumulators (the two channels of a chunk are summed by v_dot2); else packed over the two channels
  typedef typename std::conditional<HW, float, f2>::type acc_t;
  acc_t dot[NXP], ssq[NXP], oss;
  float css = 0.f;  // OTHER: the second operand's own sum of squares (its 1 / norm goes to the backward)
  if constexpr (HW) oss = 0.f; else oss = (f2){0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NXP; ++k) {
    if constexpr (HW) { dot[k] = 0.f; ssq[k] = 0.f; } else { dot[k] = (f2){0.f, 0.f}; ssq[k] = (f2){0.f, 0.f}; }
  }
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    if constexpr (HW) {
      interleave_chunk<PS, NT>(W, R, ps % 3, C.QA);
      lds_barrier();
      h2_t o = *(const h2_t*)(W + vown);
      if (OTHER) {
        css = __builtin_amdgcn_fdot2(o, o, css, false);
        o.x = *(const _Float16*)(lds + OWNR + ((ps % 3) * 2) * 1024 + (int)threadIdx.x * 2);
        o.y = *(const _Float16*)(lds + OWNR + ((ps % 3) * 2 + 1) * 1024 + (int)threadIdx.x * 2);
      }
      oss = __builtin_amdgcn_fdot2(o, o, oss, false);
#pragma unroll
      for (int k = 0; k < NXP; ++k) {
        const h2_t v = *(const h2_t*)(W + an[k]);
        dot[k] = __builtin_amdgcn_fdot2(o, v, dot[k], false);
        ssq[k] = __builtin_amdgcn_fdot2(v, v, ssq[k], false);
        if (k % 5 == 4) asm volatile("" ::: "memory");
      }
    } else {
    convert_chunk<PS, NT>(W, R, ps % 3, C.QA);
    lds_barrier();
    const f2 o = *(const f2*)(W + 2 * vown);
    oss = __builtin_elementwise_fma(o, o, oss);
#pragma unroll
    for (int k = 0; k < NXP; ++k) {
      const f2 v = *(const f2*)(W + 2 * an[k]);
      dot[k] = __builtin_elementwise_fma(o, v, dot[k]);
      ssq[k] = __builtin_elementwise_fma(v, v, ssq[k]);
      if (k % 5 == 4) asm volatile("" ::: "memory");
    }
    }
#pragma unroll
    for (int k = 0; k < NXP; ++k) asm volatile("" : "+v"(dot[k]), "+v"(ssq[k]));
    asm volatile("" : "+v"(oss));
    if (OTHER) asm volatile("" : "+v"(css));
    if (ps + 1 < NP) {
      if (ps + 2 < NP) PEA_HWAIT1(npc)
      else PEA_HWAIT0()
      if (ps + 3 < NP) PEA_HDMA16(ps % 3, 2 * ps + 6)
    }
  }
#undef PEA_HDMA16

  // ---- normalise; the lane's own 1 / norm for the backward.  (The pixel's coordinates are derived again from an opaque copy of
  //      the lane id: kept from the top they would be two registers more across the channel loop.)
  int tid_ = (int)threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int py = y0 + (tid_ >> 5), px = x0 + (tid_ & 31);
  const bool live = py < P.Y && px < P.X;
  const unsigned pe = live ? (unsigned)(py * P.X + px) * 4u : kOOB;
  float osum;
  if constexpr (HW) osum = oss; else osum = oss.x + oss.y;
  const float inv_eps = 1.0f / P.eps;
  const float inv_own = rnorm(osum, inv_eps);
  if (inv_out) bs32(iB, osum < P.eps * P.eps ? -inv_own : inv_own, pe, ezo);
  if (OTHER && inv_other_out) {
    const float inv_c = rnorm(css, inv_eps);
    bs32(mkbuf(inv_other_out + (size_t)b * S), css < P.eps * P.eps ? -inv_c : inv_c, pe, ezo);
  }
  lds_barrier();  // every lane is done with the working planes: sA goes over them
#pragma unroll
  for (int k = 0; k < NXP; ++k) {
    if (k < C.nf) {
      float dk, sk;
      if constexpr (HW) { dk = dot[k]; sk = ssq[k]; } else { dk = dot[k].x + dot[k].y; sk = ssq[k].x + ssq[k].y; }
      float a = dk * inv_own * rnorm(sk, inv_eps);
      if (CROP) {
        const int q = (C.fax[k] ? px : py) + C.fd[k];
        a = (unsigned)q < (unsigned)(C.fax[k] ? P.X : P.Y) ? a : 0.f;
      }
      sA[C.fi[k] * TP + (int)threadIdx.x] = a;
    }
  }
  // ---- the epilogue's operands: item = (offset, quad of 4 x-adjacent tile pixels)
  bool ion[ITEMS];
  unsigned ivo[ITEMS];
  int iqd[ITEMS], igy[ITEMS], igx[ITEMS], isl[ITEMS];
  f4 t4[ITEMS], w4[ITEMS];
  unsigned m4[ITEMS];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int tt = it * NT + tid_;
    const int sl = __builtin_amdgcn_readfirstlane(tt / QP);
    ion[it] = sl < P.K;
    isl[it] = min(sl, P.K - 1);
    const int qd = tt - (tt / QP) * QP;
    iqd[it] = qd;
    const int l4 = qd * 4;
    igy[it] = y0 + l4 / TW;
    igx[it] = x0 + l4 % TW;
    const bool lv = ion[it] && igy[it] < P.Y && igx[it] < P.X;
    ivo[it] = lv ? (unsigned)(igy[it] * P.X + igx[it]) * 4u : kOOB;
  }
  if (TRAIN) {
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const unsigned so = ezo + (unsigned)isl[it] * ecs;
      t4[it] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(tB, ivo[it], so, kAuxNT));
      w4[it] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(wB, ivo[it], so, kAuxNT));
      m4[it] = has_m ? __builtin_amdgcn_raw_buffer_load_b32(mB, ivo[it] == kOOB ? kOOB : ivo[it] >> 2,
                                                           (ezo >> 2) + (unsigned)isl[it] * (unsigned)P.S, kAuxNT)
                     : 0x01010101u;
    }
  }
  lds_barrier();

#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    if (!ion[it]) continue;  // wave-uniform
    const int sl = isl[it];
    const f4 a4 = *(const f4*)(sA + sl * TP + iqd[it] * 4);
    const unsigned so = ezo + (unsigned)sl * ecs;
    if (has_a) {
      f4 o = a4;
      if (af) { o.x = act_affs(o.x, af); o.y = act_affs(o.y, af); o.z = act_affs(o.z, af); o.w = act_affs(o.w, af); }
      bs128<true>(aB, o, ivo[it], so);
    }
    if (TRAIN) {
      float acc = 0.f;
      f4 g4;
      const float gs = C.gs[sl];
      const int ax_ = C.oax[sl], od_ = C.od[sl];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float m = (float)((m4[it] >> (8 * j)) & 0xffu);
        const float r = a4[j] * m - t4[it][j] * m;
        float wr = w4[it][j] * r;
        if (CROP) {
          const int q = (ax_ == 1 ? igx[it] + j : igy[it]) + od_;
          wr = (unsigned)q < (unsigned)(ax_ == 1 ? P.X : P.Y) ? wr : 0.f;
        }
        g4[j] = gs * wr * m;
        acc = fmaf(wr, r, acc);
      }
      if (has_g) bs128<false>(gB, g4, ivo[it], so);
      const float red = wave_sum63(acc);
      if ((tid_ & 63) == 63) s_part[sl * NSL + (iqd[it] >> 6)] = red;
    }
  }
  if (TRAIN) {
    lds_barrier();
    if (wave == 0 && (int)threadIdx.x < P.K) {
      float v = 0.f;
#pragma unroll
      for (int s = 0; s < NSL; ++s) v += s_part[threadIdx.x * NSL + s];
      loss_accumulate(st, tile, threadIdx.x, v);
    }
  }
}

#undef PEA_HWAIT1
#undef PEA_HWAIT0

}  // namespace pea
