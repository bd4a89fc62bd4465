// pea_xdma_w3.h -- the D = 16 self-loss backward of axis-aligned in-plane stencils with THREE workgroups per CU.
// Included by pea_k_xdma.hip (and by profiles/microbench/bwd_w3.hip).
//
// Why: k_bwd_xdma (pea_xdma.h) moves its bytes at 5.0 TB/s where the forward, with the same read : write mix, moves its own at
// 5.9 -- and the difference between the two kernels is what a CU has in flight: the forward runs three workgroups of 80 VGPRs over
// 45 KB of ring each, the backward two of 104 VGPRs over 78 KB (six 13 KB planes: the two-sided +-27 cross).  Every experiment
// that kept two workgroups (walks, skew, persistent tiles, a start stagger, 16-byte g loads and stores -- profiles/r5_bwd_vec.txt --)
// left the time where it was.  This kernel fits a third workgroup:
//   * LDS: a ring of TWO two-channel buffers (4 planes, 52 KB) instead of three; chunk ps + 2 is requested when the barrier behind the
//     gather of chunk ps has freed its buffer, and has the gather of chunk ps + 1 to land (the other two workgroups cover the rest);
//   * VGPRs (<= 80): the lane keeps G (16) but NOT its own normalised pixel (16): <ehat, G> is accumulated chunk by chunk while the
//     raw own pixel is in LDS anyway, and the own pixel is read once more at the end -- by QUADS, four planes of the wave's sixteen
//     quads per dwordx4 instruction -- for (G - ehat <ehat, G>) / n.  The pair coefficients sit two to a register pair
//     (v_pk_fma_f32 op_sel picks the half: pk_fma_c), the y neighbours' LDS addresses are the own slot plus a scalar;
//   * vector-memory instructions: g arrives in 5 dwordx4 loads per wave (instead of 20 dword loads), the gradient leaves in 4 dwordx4
//     stores (instead of 16), both exchanged with the lane = pixel layout of the gather through a wave-private LDS slot.
// Arithmetic and its order are those of k_bwd_xdma<16, ..>: the two kernels agree bit for bit (tests/test_gpu_parity.py).
#pragma once
#include "pea_xdma.h"

namespace pea {

#ifndef PEA_W3_EQ_EARLY
#define PEA_W3_EQ_EARLY 0
#endif

template <int TH, int TW, int PSU, bool CROP, int AUXS = kAuxNT>
__global__ __launch_bounds__(TH* TW, 6) void k_bwd_xdma_w3(const KParams P, const XParams C, const float* __restrict__ xt,
                                                            const float* __restrict__ invp, const float* __restrict__ gin,
                                                            const float* __restrict__ dloss, float* __restrict__ dx) {
  constexpr int D_T = 16, XP = kXP, NT = TH * TW, PS = PSU * 256, NP = D_T / 2, NQG = (2 * XP) / 4;
  static_assert(TH == 16 && TW == 32 && (2 * XP) % 4 == 0 && PS >= 8 * 1024 && 2 * PS >= 8 * 2048, "a wave = two tile rows = 16 quads");
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  int tile, b, z, y0, x0;
  if (!xdma_tile<TH, TW>(C, P, tile, b, z, y0, x0)) return;
  const size_t S = (size_t)P.S;
  const unsigned YX = (unsigned)(P.Y * P.X);
  const rsrc_t xB = mkbuf(xt + (size_t)b * D_T * S), dB = mkbuf(dx + (size_t)b * D_T * S);
  const rsrc_t gB = mkbuf(gin + (size_t)b * P.K * S), iB = mkbuf(invp + (size_t)b * S);
  const unsigned ecs = (unsigned)P.S * 4u, ezo = (unsigned)z * YX * 4u;
  const float dl = dloss ? dloss[0] : 1.f;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
  const int py = y0 + ly, px = x0 + lx;
  const bool live = py < P.Y && px < P.X;
  const unsigned po4 = (unsigned)(py * P.X + px) * 4u;

  // ---- the (up to) two quads this lane moves per plane (k_bwd_xdma's staging geometry)
  unsigned vo[2];
  bool act[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int q = (s * (NT / 64) + wave) * 64 + lane;
    int gy, gx;
    if (q < C.QV) {
      gy = y0 - C.hy0 + (q >> 3);
      gx = x0 + 4 * (q & 7);
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
  const int wbase = wave * 1024, w1 = wbase + (NT / 64) * 1024;
#define PEA_W3DMA(rsrc, plane_byte, so)                                                                                           \
  {                                                                                                                              \
    if (act[0]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + (plane_byte) + wbase), 16, vo[0], so, 0, 0);     \
    if (act[1]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + (plane_byte) + w1), 16, vo[1], so, 0, 0);        \
  }
  // the 1 / norm plane into plane 2, chunk 0 into planes 0 and 1
  PEA_W3DMA(iB, 2 * PS, ezo)
  PEA_W3DMA(xB, 0, ezo)
  PEA_W3DMA(xB, PS, ezo + ecs)

  // ---- LDS slot of every pair's neighbour: x pairs per lane (inside the tile row or in its strip), y pairs = own slot + a scalar
  int ax[XP];
  const int vown = ((C.hy0 + ly) * TW + lx) * 4;
  const int hrow = (C.QV * 4 + ly * C.SW) * 4;
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    const int d = C.xd[k], c = lx + d;
    ax[k] = (unsigned)c < (unsigned)TW ? vown + d * 4 : hrow + (c & C.xm[k]) * 4;
  }

  // ---- g of every pair: quads (pea_xdma.h, VEC), or one dword per lane in the tile columns whose x displacements leave the image
  float cx[XP], cy[XP];
  f4 gq[NQG];
  const bool vec_tile = x0 >= C.vhx && x0 + TW + C.vhx <= P.X;  // uniform
  const int j4 = lane >> 4, qq = lane & 15;
  const int qy = y0 + 2 * wave + (qq >> 3), qx = x0 + 4 * (qq & 7);
  const unsigned qo = (qy < P.Y && qx < P.X) ? (unsigned)(qy * P.X + qx) * 4u : kOOB;  // the lane's quad of the wave's two rows
  if (vec_tile) {
    const i4 gW = mkbuf_words(gin + (size_t)b * P.K * S);
#pragma unroll
    for (int i = 0; i < NQG; ++i) {
      const int go = (int)((C.vq_go[i] >> (8 * j4)) & 0xffu) - 128, gi = (int)((C.vq_gi[i] >> (8 * j4)) & 0xffu);
      const unsigned xm_ = (4 * i < XP ? 1u : 0u) | (4 * i + 1 < XP ? 2u : 0u) | (4 * i + 2 < XP ? 4u : 0u) | (4 * i + 3 < XP ? 8u : 0u);
      const bool isx = (xm_ >> j4) & 1u;
      bool ok;
      const int ty = wrap1<CROP>(qy + (isx ? 0 : go), P.Y, ok), tx = qx + (isx ? go : 0);
      const unsigned v = (qo != kOOB && gi != 0xff && ok) ? (unsigned)gi * ecs + (unsigned)(ty * P.X + tx) * 4u : kOOB;
      gq[i] = asm_load_b128(gW, v, ezo);
    }
  } else {
    const unsigned pg = live ? po4 : 0xC0000000u;  // dead lanes: stays out of range when a small displacement is added
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      const int go = C.xgo[k];
      const bool out = (unsigned)(px + go) >= (unsigned)P.X;
      const int fix = go > 0 ? -P.X : P.X;
      const unsigned o = CROP ? (out ? kOOB : pg + (unsigned)(go * 4)) : pg + (unsigned)((out ? go + fix : go) * 4);
      cx[k] = bl32(gB, k < C.npx ? o : kOOB, ezo + (unsigned)C.xgi[k] * ecs);
    }
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      const int go = C.ygo[k];
      const bool out = (unsigned)(py + go) >= (unsigned)P.Y;
      const int fix = go > 0 ? -P.Y : P.Y;
      const unsigned o = CROP ? (out ? kOOB : pg + (unsigned)(go * P.X * 4)) : pg + (unsigned)((out ? go + fix : go) * P.X * 4);
      cy[k] = bl32(gB, k < C.npy ? o : kOOB, ezo + (unsigned)C.ygi[k] * ecs);
    }
#pragma unroll
    for (int k = 0; k < XP; ++k) asm volatile("" : "+v"(cx[k]), "+v"(cy[k]));  // the compiler's wait for them stays in this branch
  }
  // 1 / norm, chunk 0 and g have landed -- nothing else is in flight; every wave's share of them too
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (vec_tile) {
    // quads -> pixels through the wave's own 1 KB of plane 3 (free until chunk 1 is requested; the LDS instructions of one wave
    // execute in order, so the slot is reused without a wait)
#pragma unroll
    for (int i = 0; i < NQG; ++i) asm volatile("" : "+v"(gq[i]));
    char* const gs = lds + 3 * PS + wave * 1024;
#pragma unroll
    for (int i = 0; i < NQG; ++i) {
      *(f4*)(gs + lane * 16) = gq[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = *(const float*)(gs + j * 256 + lane * 4);
        if (4 * i + j < XP) cx[4 * i + j] = v;
        else cy[4 * i + j - XP] = v;
      }
    }
  }
  // coefficient of a pair = g * 1 / |e(q)|, two pairs to a register pair
  const float invo = *(const float*)(lds + 2 * PS + vown);
  const float inv_own = fabsf(invo);
  f2 cpx[XP / 2], cpy[XP / 2];
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    cx[k] *= fabsf(*(const float*)(lds + 2 * PS + ax[k]));
    cy[k] *= fabsf(*(const float*)(lds + 2 * PS + vown + C.yd[k] * TW * 4));
    if (k & 1) {
      cpx[k / 2] = (f2){cx[k - 1], cx[k]};
      cpy[k / 2] = (f2){cy[k - 1], cy[k]};
      asm volatile("" : "+v"(cpx[k / 2]), "+v"(cpy[k / 2]));  // computed HERE (pea_xdma.h: the scheduler otherwise sinks it all)
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the 1 / norm plane and the g slots are dead: buffer 1 may be filled
  PEA_W3DMA(xB, 2 * PS, ezo + 2u * ecs)
  PEA_W3DMA(xB, 3 * PS, ezo + 3u * ecs)

  f2 G[NP];
  float proj = 0.f;
  f4 eq[4];  // the own pixel again, by quads: requested behind the last barrier, used after the last gather
  const i4 eW = mkbuf_words(xt + (size_t)b * D_T * S);
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int bo = (ps & 1) * 2 * PS;
    f2 o;
    o.x = *(const float*)(lds + bo + vown);
    o.y = *(const float*)(lds + bo + PS + vown);
    o = o * inv_own;
    f2 acc = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      f2 v;
      v.x = *(const float*)(lds + bo + ax[k]);
      v.y = *(const float*)(lds + bo + PS + ax[k]);
      acc = (k & 1) ? pk_fma_c<true>(cpx[k / 2], v, acc) : pk_fma_c<false>(cpx[k / 2], v, acc);
      if (k % 5 == 4) asm volatile("" ::: "memory");  // bound the ds_read hoisting
    }
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      const int ay = vown + C.yd[k] * TW * 4;
      f2 v;
      v.x = *(const float*)(lds + bo + ay);
      v.y = *(const float*)(lds + bo + PS + ay);
      acc = (k & 1) ? pk_fma_c<true>(cpy[k / 2], v, acc) : pk_fma_c<false>(cpy[k / 2], v, acc);
      if (k % 5 == 4) asm volatile("" ::: "memory");
    }
    proj = fmaf(o.x, acc.x, fmaf(o.y, acc.y, proj));
    asm volatile("" : "+v"(acc), "+v"(proj));  // the chunk's sums exist before its barrier
    G[ps] = acc;
    if (ps + 1 < NP) {
      // chunk ps + 1 -- all that is in flight -- has landed; everyone is done with buffer ps & 1
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (ps + 2 < NP) {
        PEA_W3DMA(xB, bo, ezo + (unsigned)(2 * ps + 4) * ecs)
        PEA_W3DMA(xB, bo + PS, ezo + (unsigned)(2 * ps + 5) * ecs)
      } else if (PEA_W3_EQ_EARLY) {
#pragma unroll
        for (int r = 0; r < 4; ++r) eq[r] = asm_load_b128(eW, qo == kOOB ? kOOB : qo + (unsigned)(4 * r + j4) * ecs, ezo);
      }
    }
  }
#undef PEA_W3DMA
  if (!PEA_W3_EQ_EARLY) {
#pragma unroll
    for (int r = 0; r < 4; ++r) eq[r] = asm_load_b128(eW, qo == kOOB ? kOOB : qo + (unsigned)(4 * r + j4) * ecs, ezo);
  }
  if (invo < 0.f) proj = 0.f;  // clamp branch of F.normalize
  const float sc = dl * inv_own;
  // pixels <-> quads through 2 KB of the wave's own in buffer 0 (dead: the last chunk sits in buffer 1 and every wave is past the
  // barrier in front of it), eight channels at a time: [channel][64 pixels]
  static_assert((NP - 1) % 2 == 1, "the last chunk sits in buffer 1");
  char* const sc_ = lds + wave * 2048;
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(eq[0]), "+v"(eq[1]), "+v"(eq[2]), "+v"(eq[3])::"memory");
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    *(f4*)(sc_ + j4 * 256 + qq * 16) = eq[2 * r];
    *(f4*)(sc_ + (4 + j4) * 256 + qq * 16) = eq[2 * r + 1];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float eh = *(const float*)(sc_ + c * 256 + lane * 4) * inv_own;
      const int ch = 8 * r + c;
      const float gv = (ch & 1) ? G[ch / 2].y : G[ch / 2].x;
      *(float*)(sc_ + c * 256 + lane * 4) = __builtin_fmaf(-eh, proj, gv) * sc;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f4 v = *(const f4*)(sc_ + (4 * h + j4) * 256 + qq * 16);
      bs128<AUXS == kAuxNT>(dB, v, qo == kOOB ? kOOB : qo + (unsigned)(8 * r + 4 * h + j4) * ecs, ezo);
    }
  }
}

}  // namespace pea
