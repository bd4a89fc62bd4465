// pea_xdma_pf.h -- the cross backward with the projection known FIRST (round 3).
//
//   d e(p) = dl / n(p) * ( G(p) - ehat(p) <ehat(p), G(p)> ),     G(p) = sum over (offset, role) pairs of c_k ehat(q_k)
//
// k_bwd_xdma (pea_xdma.h) learns <ehat, G> only after the last channel chunk, so it keeps G (D registers) -- and for D > 16 has no
// registers left for ehat and reads the own pixel AGAIN from global memory at the end: at D = 64 that is a second pass over e
// (605 MB at the bench shape; e does not fit the Infinity Cache), 64 one-dword loads and 64 one-dword stores per lane in the tail.
// The ablation that found it (profiles/microbench/stamp_bwd.hip): with BOTH the gather and the chunk DMA compiled out the D = 64
// backward still takes 282 of 430 us.
//
// But  <ehat(p), ehat(q_k)>  is the affinity of the pair -- a_i(p) for role A, a_i(p - o_i) for role B -- which the forward wrote:
//
//   <ehat(p), G(p)> = sum_i [ g_i(p) a_i(p) + g_i(p - o_i) a_i(p - o_i) ]
//
// needs no embedding channel at all: 2K more loads next to the 2K g loads of the prologue (same addresses, the affs tensor).
// With the projection in hand a chunk FINISHES its two channels: (acc - ehat_c proj) * dl / n is stored right away.  No G array, no
// kept ehat, no second read of e; ~70 VGPRs instead of 104-126, so stencils whose cross fits 7 KB planes (reach <= 11: BBBC039V1,
// the K = 8 table of configs[4], the deep-supervision scales) run THREE workgroups per CU; every width takes 10 pairs per axis.
// Needs the RAW cosine map (no activation flag on the forward's affs output): pea_affinity_bwd_ex2(.., affs, ..).
// Self loss, in-plane stencils (no z offsets), f32 storage here; f16 storage: k_bwd_xdma_h<.., PF> (pea_xdma_h16.h).
#pragma once
#include "pea_xdma.h"

namespace pea {

// s_waitcnt vmcnt(n) lgkmcnt(0); s_barrier for an n that is a constant only after the chunk loop is unrolled (the immediate of
// s_waitcnt has to be a literal: a switch the optimiser folds).  LOADS retire in order: n = the DMA instructions of the chunks
// after the one that has to have landed.  (Rounds 3 counted the gradient stores issued since as well; a store may retire before
// an older load, so that count was unsound -- round 4, pea_zmarch.h zm_bwd_wait.)
__device__ __forceinline__ void pf_wait(int n) {
#define PEA_PFW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
  switch (n) {
    PEA_PFW(0) PEA_PFW(1) PEA_PFW(2) PEA_PFW(3) PEA_PFW(4) PEA_PFW(5) PEA_PFW(6) PEA_PFW(7) PEA_PFW(8) PEA_PFW(9) PEA_PFW(10)
    PEA_PFW(11) PEA_PFW(12) PEA_PFW(13) PEA_PFW(14) PEA_PFW(15) PEA_PFW(16) PEA_PFW(17) PEA_PFW(18) PEA_PFW(19) PEA_PFW(20)
    PEA_PFW(21) PEA_PFW(22) PEA_PFW(23) PEA_PFW(24) PEA_PFW(25) PEA_PFW(26) PEA_PFW(27) PEA_PFW(28) PEA_PFW(29) PEA_PFW(30)
    PEA_PFW(31) PEA_PFW(32)
    default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
  }
#undef PEA_PFW
}

// RB: buffers of the ring (3: two chunks in flight beside the one being gathered; more where the planes are small -- a store has
// to be acknowledged RB - 1 chunk periods after it was issued, or the next hand-off waits for it)
template <int D_T, int TH, int TW, int PSU, bool CROP, int WPE, int RB = 3>
__global__ __launch_bounds__(TH* TW, WPE) void k_bwd_xdma_pf(const KParams P, const XParams C, const float* __restrict__ xt,
                                                              const float* __restrict__ invp, const float* __restrict__ gin,
                                                              const float* __restrict__ affs, const float* __restrict__ dloss,
                                                              float* __restrict__ dx) {
  constexpr int NT = TH * TW, PS = PSU * 256, NP = D_T / 2, XP = kXP, IP = 2 * (RB - 1);  // IP: the plane the 1 / norm plane sits in first
  static_assert(TW == 32 && D_T % 2 == 0 && RB >= 3 && NP >= RB, "lane mapping / channel pairs / ring");
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  int tile, b, z, y0, x0;
  if (!xdma_tile<TH, TW>(C, P, tile, b, z, y0, x0)) return;
  const size_t S = (size_t)P.S;
  const unsigned YX = (unsigned)(P.Y * P.X);
  const rsrc_t xB = mkbuf(xt + (size_t)b * D_T * S), dB = mkbuf(dx + (size_t)b * D_T * S);
  const rsrc_t gB = mkbuf(gin + (size_t)b * P.K * S), iB = mkbuf(invp + (size_t)b * S), aB = mkbuf(affs + (size_t)b * P.K * S);
  const unsigned ecs = (unsigned)P.S * 4u, ezo = (unsigned)z * YX * 4u;
  const float dl = dloss ? dloss[0] : 1.f;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
  const int py = y0 + ly, px = x0 + lx;
  const bool live = py < P.Y && px < P.X;
  const unsigned po4 = (unsigned)(py * P.X + px) * 4u;
  const unsigned pe = live ? po4 : kOOB;

  // ---- the (up to) two quads this lane moves per plane (pea_xdma.h)
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
  const int npc = 2 * ((__builtin_amdgcn_ballot_w64(act[0]) != 0) + (__builtin_amdgcn_ballot_w64(act[1]) != 0));
#define PEA_PFDMA(rsrc, plane_byte, so)                                                                                          \
  {                                                                                                                              \
    if (act[0]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + (plane_byte) + wbase), 16, vo[0], so, 0, 0);    \
    if (act[1]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + (plane_byte) + w1), 16, vo[1], so, 0, 0);       \
  }
  PEA_PFDMA(iB, IP * PS, ezo)
  PEA_PFDMA(xB, 0, ezo)
  PEA_PFDMA(xB, PS, ezo + ecs)

  // ---- g and the affinity of every pair (role A at p, role B at p - o): the coefficients, and the projection
  const unsigned pg = live ? po4 : 0xC0000000u;
  float cx[XP], cy[XP];
  int ax[XP], ay[XP];
  float proj = 0.f;
  const int vown = ((C.hy0 + ly) * TW + lx) * 4;
  const int hrow = (C.QV * 4 + ly * C.SW) * 4;
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    const int go = C.xgo[k];
    const int t = px + go;
    const bool out = (unsigned)t >= (unsigned)P.X;
    const int fix = go > 0 ? -P.X : P.X;
    const unsigned o = CROP ? (out ? kOOB : pg + (unsigned)(go * 4)) : pg + (unsigned)((out ? go + fix : go) * 4);
    const unsigned so = ezo + (unsigned)C.xgi[k] * ecs;
    cx[k] = bl32(gB, k < C.npx ? o : kOOB, so);
    proj = fmaf(cx[k], bl32(aB, k < C.npx ? o : kOOB, so), proj);
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
    const unsigned so = ezo + (unsigned)C.ygi[k] * ecs;
    cy[k] = bl32(gB, k < C.npy ? o : kOOB, so);
    proj = fmaf(cy[k], bl32(aB, k < C.npy ? o : kOOB, so), proj);
    ay[k] = vown + C.yd[k] * TW * 4;
  }
#pragma unroll
  for (int c = 1; c <= RB - 2; ++c) {  // chunks 1 .. RB - 2 into their buffers (the last buffer holds the 1 / norm plane for now)
    PEA_PFDMA(xB, 2 * c * PS, ezo + (unsigned)(2 * c) * ecs)
    PEA_PFDMA(xB, (2 * c + 1) * PS, ezo + (unsigned)(2 * c + 1) * ecs)
  }
  // inv, chunk 0, g and affs have landed (the DMA instructions of chunks 1 .. RB - 2 may still fly)
  pf_wait((RB - 2) * npc);

  const float invo = *(const float*)(lds + IP * PS + vown);
  const float inv_own = fabsf(invo);
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    cx[k] *= fabsf(*(const float*)(lds + IP * PS + ax[k]));
    cy[k] *= fabsf(*(const float*)(lds + IP * PS + ay[k]));
    asm volatile("" : "+v"(cx[k]), "+v"(cy[k]));
  }
  if (invo < 0.f) proj = 0.f;  // clamp branch of F.normalize: d ehat / d e = I / eps
  asm volatile("" : "+v"(proj));
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the inv plane is dead: the last buffer may be filled
  PEA_PFDMA(xB, IP * PS, ezo + (unsigned)(2 * (RB - 1)) * ecs)
  PEA_PFDMA(xB, (IP + 1) * PS, ezo + (unsigned)(2 * (RB - 1) + 1) * ecs)
  const float sc = dl * inv_own;

#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int bo = (ps % RB) * 2 * PS;
    f2 o;
    o.x = *(const float*)(lds + bo + vown);
    o.y = *(const float*)(lds + bo + PS + vown);
    f2 acc = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      f2 v;
      v.x = *(const float*)(lds + bo + ax[k]);
      v.y = *(const float*)(lds + bo + PS + ax[k]);
      acc = __builtin_elementwise_fma((f2){cx[k], cx[k]}, v, acc);
      if (k % 5 == 4) asm volatile("" ::: "memory");  // bound the ds_read hoisting
    }
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      f2 v;
      v.x = *(const float*)(lds + bo + ay[k]);
      v.y = *(const float*)(lds + bo + PS + ay[k]);
      acc = __builtin_elementwise_fma((f2){cy[k], cy[k]}, v, acc);
      if (k % 5 == 4) asm volatile("" ::: "memory");
    }
    // this chunk's two channels are final: (G - ehat <ehat, G>) dl / n
    const float pq = proj * inv_own;
    float vx = (acc.x - o.x * pq) * sc, vy = (acc.y - o.y * pq) * sc;
    asm volatile("" : "+v"(vx), "+v"(vy));  // (stored behind the hand-off: there the stores have a chunk's time to retire)
    if (ps + 1 < NP) {
      // chunk ps + 1 has landed; younger than it: the DMA of chunks ps + 2 .. ps + RB - 1 (those that exist) and the stores of the
      // last min(ps + 1, RB - 1) chunks; everyone is done with buffer ps % RB
      // (the stores in between are NOT counted: they sit on the same counter but may retire before an older load -- counted in,
      //  they let the wait pass with part of the awaited chunk in flight; pea_zmarch.h zm_bwd_wait has the case that showed it)
      // A wave that issues no DMA (npc == 0: the region has fewer blocks than the workgroup has waves) has nothing to wait for --
      // vmcnt(0) would make it drain its own stores at every chunk and hold everybody's barrier.
      const int nd = (ps + RB - 1 < NP ? ps + RB - 1 : NP - 1) - (ps + 1);
      if (npc == 0) lds_barrier();
      else pf_wait(nd * npc);
      if (ps + RB < NP) {
        PEA_PFDMA(xB, bo, ezo + (unsigned)(2 * (ps + RB)) * ecs)
        PEA_PFDMA(xB, bo + PS, ezo + (unsigned)(2 * (ps + RB) + 1) * ecs)
      }
    }
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vx), dB, pe, ezo + (unsigned)(2 * ps) * ecs, kAuxNT);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vy), dB, pe, ezo + (unsigned)(2 * ps + 1) * ecs, kAuxNT);
  }
#undef PEA_PFDMA
}

// ------------------------------------------------------------------------------------------------------------------
// The cross loss with a DETACHED second operand (ema_embedding_loss, scripts_cvppp/loss/loss_embedding_mse.py:79-95 behind
// convert_consistency_flip's detach) at D > 16: role A only,
//     G(p) = sum_i g_i(p) ehat_other(p + o_i),     <ehat(p), G(p)> = sum_i g_i(p) a_i(p)      (every pair, exactly)
// so the projection-first form needs nothing but the K values of g and of the raw map at the own pixel.  xt / invp are the SECOND
// operand and its 1 / norm plane (staged: the one-sided cross of plan_xdma mode 2); the own pixel -- needed raw, two channels per
// chunk, to finish them -- is staged as the own TILE (no halo: 2 KB per channel, the first two waves move it) beside each chunk;
// own_inv: the own operand's signed 1 / norm plane.  No accumulate form (the caller adds two buffers).
// ------------------------------------------------------------------------------------------------------------------
template <int D_T, int TH, int TW, int PSU, bool CROP, int WPE, int RB = 3>
__global__ __launch_bounds__(TH* TW, WPE) void k_bwd_xdma_pfo(const KParams P, const XParams C, const float* __restrict__ xt,
                                                               const float* __restrict__ invp, const float* __restrict__ own,
                                                               const float* __restrict__ own_inv, const float* __restrict__ gin,
                                                               const float* __restrict__ affs, const float* __restrict__ dloss,
                                                               float* __restrict__ dx) {
  constexpr int NT = TH * TW, PS = PSU * 256, NP = D_T / 2, XP = kXP, IP = 2 * (RB - 1), OWNB = 2 * RB * PS;
  static_assert(TW == 32 && D_T % 2 == 0 && RB >= 3 && NP >= RB && NT == 512, "lane mapping / channel pairs / ring");
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  int tile, b, z, y0, x0;
  if (!xdma_tile<TH, TW>(C, P, tile, b, z, y0, x0)) return;
  const size_t S = (size_t)P.S;
  const unsigned YX = (unsigned)(P.Y * P.X);
  const rsrc_t xB = mkbuf(xt + (size_t)b * D_T * S), dB = mkbuf(dx + (size_t)b * D_T * S), oB = mkbuf(own + (size_t)b * D_T * S);
  const rsrc_t gB = mkbuf(gin + (size_t)b * P.K * S), iB = mkbuf(invp + (size_t)b * S), aB = mkbuf(affs + (size_t)b * P.K * S);
  const rsrc_t oiB = mkbuf(own_inv + (size_t)b * S);
  const unsigned ecs = (unsigned)P.S * 4u, ezo = (unsigned)z * YX * 4u;
  const float dl = dloss ? dloss[0] : 1.f;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
  const int py = y0 + ly, px = x0 + lx;
  const bool live = py < P.Y && px < P.X;
  const unsigned pe = live ? (unsigned)(py * P.X + px) * 4u : kOOB;

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
  const bool ownw = wave < NT / 256;  // uniform: the waves that move the own tile (quads 0 .. 127)
  unsigned ownvo = kOOB;
  {
    const int q = (int)threadIdx.x, qy = y0 + (q >> 3), qx = x0 + 4 * (q & 7);
    ownvo = (q < NT / 4 && qy < P.Y && qx < P.X) ? (unsigned)(qy * P.X + qx) * 4u : kOOB;
  }
  const int npc = 2 * ((__builtin_amdgcn_ballot_w64(act[0]) != 0) + (__builtin_amdgcn_ballot_w64(act[1]) != 0)) + (ownw ? 2 : 0);
#define PEA_PFDMA(rsrc, plane_byte, so)                                                                                          \
  {                                                                                                                              \
    if (act[0]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + (plane_byte) + wbase), 16, vo[0], so, 0, 0);    \
    if (act[1]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + (plane_byte) + w1), 16, vo[1], so, 0, 0);       \
  }
// chunk c: the second operand's cross into ring buffer c % RB, the own tile into its 4 KB slot
#define PEA_PFCHUNK(c)                                                                                                            \
  {                                                                                                                              \
    PEA_PFDMA(xB, (2 * ((c) % RB)) * PS, ezo + (unsigned)(2 * (c)) * ecs)                                                        \
    PEA_PFDMA(xB, (2 * ((c) % RB) + 1) * PS, ezo + (unsigned)(2 * (c) + 1) * ecs)                                                \
    if (ownw) {                                                                                                                  \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(oB, (lds_ptr_t)(lds + OWNB + (2 * ((c) % RB)) * 2048 + wave * 1024), 16, ownvo,    \
                                               ezo + (unsigned)(2 * (c)) * ecs, 0, 0);                                            \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(oB, (lds_ptr_t)(lds + OWNB + (2 * ((c) % RB) + 1) * 2048 + wave * 1024), 16, ownvo, \
                                               ezo + (unsigned)(2 * (c) + 1) * ecs, 0, 0);                                        \
    }                                                                                                                            \
  }
  // the second operand's 1 / norm plane sits in the LAST buffer's first plane until the coefficients are done; chunk 0 .. RB - 2
  PEA_PFDMA(iB, IP * PS, ezo)
  PEA_PFCHUNK(0)

  // ---- g and the raw affinity at the own pixel: the coefficients, and the projection
  float cx[XP], cy[XP];
  int ax[XP], ay[XP];
  float proj = 0.f;
  const int vown = ((C.hy0 + ly) * TW + lx) * 4;
  const int hrow = (C.QV * 4 + ly * C.SW) * 4;
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    const unsigned so = ezo + (unsigned)C.xgi[k] * ecs;
    cx[k] = bl32(gB, k < C.npx ? pe : kOOB, so);
    proj = fmaf(cx[k], bl32(aB, k < C.npx ? pe : kOOB, so), proj);
    const int d = C.xd[k], c = lx + d;
    ax[k] = (unsigned)c < (unsigned)TW ? vown + d * 4 : hrow + (c & C.xm[k]) * 4;
  }
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    const unsigned so = ezo + (unsigned)C.ygi[k] * ecs;
    cy[k] = bl32(gB, k < C.npy ? pe : kOOB, so);
    proj = fmaf(cy[k], bl32(aB, k < C.npy ? pe : kOOB, so), proj);
    ay[k] = vown + C.yd[k] * TW * 4;
  }
  const float invo = bl32(oiB, pe, ezo);
#pragma unroll
  for (int c = 1; c <= RB - 2; ++c) PEA_PFCHUNK(c)
  pf_wait((RB - 2) * npc);  // inv, chunk 0, g, affs and the own 1 / norm have landed (chunks 1 .. RB - 2 may still fly)

  const float inv_own = fabsf(invo);
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    cx[k] *= fabsf(*(const float*)(lds + IP * PS + ax[k]));
    cy[k] *= fabsf(*(const float*)(lds + IP * PS + ay[k]));
    asm volatile("" : "+v"(cx[k]), "+v"(cy[k]));
  }
  if (invo < 0.f) proj = 0.f;  // clamp branch of F.normalize
  asm volatile("" : "+v"(proj));
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the inv plane is dead: the last buffer may be filled
  PEA_PFCHUNK(RB - 1)
  const float sc = dl * inv_own;
  const float pq = proj * inv_own;

#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int bo = (ps % RB) * 2 * PS;
    f2 o;  // the own pixel, raw
    o.x = *(const float*)(lds + OWNB + (2 * (ps % RB)) * 2048 + (int)threadIdx.x * 4);
    o.y = *(const float*)(lds + OWNB + (2 * (ps % RB) + 1) * 2048 + (int)threadIdx.x * 4);
    f2 acc = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      f2 v;
      v.x = *(const float*)(lds + bo + ax[k]);
      v.y = *(const float*)(lds + bo + PS + ax[k]);
      acc = __builtin_elementwise_fma((f2){cx[k], cx[k]}, v, acc);
      if (k % 5 == 4) asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      f2 v;
      v.x = *(const float*)(lds + bo + ay[k]);
      v.y = *(const float*)(lds + bo + PS + ay[k]);
      acc = __builtin_elementwise_fma((f2){cy[k], cy[k]}, v, acc);
      if (k % 5 == 4) asm volatile("" ::: "memory");
    }
    float vx = __builtin_fmaf(-o.x, pq, acc.x) * sc, vy = __builtin_fmaf(-o.y, pq, acc.y) * sc;
    asm volatile("" : "+v"(vx), "+v"(vy));
    if (ps + 1 < NP) {
      const int nd = (ps + RB - 1 < NP ? ps + RB - 1 : NP - 1) - (ps + 1);  // chunks requested behind the one that has to have landed
      if (npc == 0) lds_barrier();
      else pf_wait(nd * npc);
      if (ps + RB < NP) PEA_PFCHUNK(ps + RB)
    }
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vx), dB, pe, ezo + (unsigned)(2 * ps) * ecs, kAuxNT);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vy), dB, pe, ezo + (unsigned)(2 * ps + 1) * ecs, kAuxNT);
  }
#undef PEA_PFDMA
#undef PEA_PFCHUNK
}

}  // namespace pea
