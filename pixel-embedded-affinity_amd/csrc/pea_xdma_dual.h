// pea_xdma_dual.h -- the full-resolution PAIR of the reference's 2D training loops in one launch: embedding_loss(e, t, w, m) and
// ema_embedding_loss(e, ema, t, w, m) on the SAME target / weight / mask (scripts_cvppp/main.py:284,293; scripts_bbbc039v1/main.py
// likewise; scripts_cvppp/loss/loss_embedding_mse.py:18-47, 79-95).
//
// Why: as two launches (k_fwd_xdma, k_fwd_xdma<.., OTHER>) the pair reads e twice and t / w / m twice -- 194 + 40 (g) and
// 64 + 194 + 40 bytes per pixel at D = 16, K = 10: 532.  One kernel that stages BOTH operands' channel pairs side by side reads e,
// ema, t, w, m once and writes the self map and the two g maps: 64 + 64 + 90 + 40 + 80 = 338 bytes per pixel.  The own pixel of both
// dot products is e's (the staged centre of the first operand), so nothing is loaded per lane at all.
//
// Structure = k_fwd_xdma (pea_xdma.h) with four planes per chunk instead of two: a ring of NB buffers of (e ch 2c, e ch 2c + 1,
// ema ch 2c, ema ch 2c + 1) in the forward's one-sided 7.5 KB planes.  NB = 2: 60 KB, two workgroups per CU, the next chunk is
// requested when this one's buffer is free (one chunk of look-ahead, the other workgroup covers the rest); NB = 4 (the default): the
// SAME two buffers handed over in halves -- the e pair is gathered, its planes are requested again, then the ema pair: twice the
// barriers, but a half is on its way one and a half chunks ahead and the requests leave in a steadier stream (190 against 195 us);
// NB = 3: 90 KB, one workgroup per CU, two chunks of look-ahead (230 us).  40 packed accumulators (dot and neighbour |.|^2, self and
// cross, ten offsets).
// The sums run in the order of the single kernels -- even channels in .x, odd in .y, chunks ascending -- so every output is
// bit-identical to the two launches it replaces (tests/test_gpu_cross.py::test_dual_forward_equals_the_two_launches).
// 2D, D = 16, f32, axis-aligned stencil, K <= kXP; either border.  The cross loss' map is not written (the training loop drops it).
#pragma once
#include "pea_xdma.h"

namespace pea {

struct DualFwdArgs {
  const float* e2;       // the second operand (detached EMA embedding), [B, 16, Y, X]
  float* gout2;          // g of the cross loss, [B, K, Y, X]
  float* inv_other_out;  // 1 / norm plane of the second operand, [B, Y, X] (k_bwd_xdma<.., DUAL> reads it)
  LossState* st2;        // loss state of the cross loss
  float gs2[kXK];        // 2 lambda_i / N_i of the cross loss, per channel (the self loss': XParams::gs)
};

template <int TH, int TW, int PSU, bool CROP, int NB>
__global__ __launch_bounds__(TH* TW, NB == 3 ? 2 : 4) void k_fwd_xdma_dual(const KParams P, const XParams C, const float* __restrict__ e,
                                                                           const float* __restrict__ target,
                                                                           const float* __restrict__ weight,
                                                                           const uint8_t* __restrict__ mask, float* __restrict__ affs,
                                                                           float* __restrict__ gout, LossState* __restrict__ st,
                                                                           float* __restrict__ inv_out, const DualFwdArgs DA) {
  constexpr int D_T = 16, NT = TH * TW, PS = PSU * 256, NP = D_T / 2, TP = NT, QP = TP / 4, NSL = QP / 64, KMAX = kXP;
  constexpr int ITEMS = (KMAX * QP + NT - 1) / NT, BUF = 4 * PS;
  constexpr int NBUF = NB == 3 ? 3 : 2;  // whole four-plane buffers (NB = 4: the ring of two, handed over in HALVES -- e pair, ema pair)
  static_assert(TW == 32 && QP % 64 == 0, "lane mapping");
  static_assert(NB == 2 || NB == 3 || NB == 4, "ring depth");
  static_assert(2 * (KMAX * TP * 4 + KMAX * NSL * 4) <= NBUF * BUF && KMAX <= kXK, "the parked dot products fit the ring");
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  float* sA = (float*)lds;          // [K][TP] self dot products, laid over the ring once it is dead
  float* sA2 = sA + KMAX * TP;      // [K][TP] cross
  float* s_part = sA2 + KMAX * TP;  // [K][NSL]
  float* s_part2 = s_part + KMAX * NSL;
  int tile, b, z, y0, x0;
  if (!xdma_tile<TH, TW>(C, P, tile, b, z, y0, x0)) return;
  const size_t S = (size_t)P.S;
  const unsigned YX = (unsigned)(P.Y * P.X);
  const rsrc_t xB = mkbuf(e + (size_t)b * D_T * S), yB = mkbuf(DA.e2 + (size_t)b * D_T * S);
  const rsrc_t aB = mkbuf(affs ? affs + (size_t)b * P.K * S : nullptr), gB = mkbuf(gout + (size_t)b * P.K * S);
  const rsrc_t g2B = mkbuf(DA.gout2 + (size_t)b * P.K * S);
  const rsrc_t tB = mkbuf(target + (size_t)b * P.tbs), wB = mkbuf(weight + (size_t)b * P.wbs);
  const rsrc_t mB = mkbuf(mask ? mask + (size_t)b * P.mbs : nullptr);
  const rsrc_t iB = mkbuf(inv_out + (size_t)b * S), i2B = mkbuf(DA.inv_other_out + (size_t)b * S);
  const unsigned ecs = (unsigned)P.S * 4u, ezo = (unsigned)z * YX * 4u;
  const bool has_a = affs != nullptr, has_m = mask != nullptr;
  const unsigned af = P.flags & kActMask;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
  const int py = y0 + ly, px = x0 + lx;
  const bool live = py < P.Y && px < P.X;
  const unsigned pe = live ? (unsigned)(py * P.X + px) * 4u : kOOB;

  // ---- the (up to) two quads this lane moves per plane (k_fwd_xdma's geometry)
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
  const bool act0 = __builtin_amdgcn_ballot_w64(act[0]) != 0, act1 = __builtin_amdgcn_ballot_w64(act[1]) != 0;  // wave-uniform
  const int npt = 4 * ((int)act0 + (int)act1);  // DMA instructions of this wave per chunk: 0, 4 or 8
#define PEA_DDMA1(rs_, byte_, v_, so_) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_ptr_t)(lds + (byte_)), 16, v_, so_, 0, 0);
// chunk ch (channels 2 ch, 2 ch + 1 of both operands) into ring buffer buf
#define PEA_DDMA(buf, ch)                                       \
  {                                                             \
    const unsigned so_ = ezo + (unsigned)(2 * (ch)) * ecs;      \
    const int pb_ = (buf) * BUF;                                \
    if (act[0]) { /* per LANE: a lane beyond the region writes nothing (its 16 bytes would land in the next plane) */ \
      PEA_DDMA1(xB, pb_ + wbase, vo[0], so_)                    \
      PEA_DDMA1(xB, pb_ + PS + wbase, vo[0], so_ + ecs)         \
      PEA_DDMA1(yB, pb_ + 2 * PS + wbase, vo[0], so_)           \
      PEA_DDMA1(yB, pb_ + 3 * PS + wbase, vo[0], so_ + ecs)     \
    }                                                           \
    if (act[1]) {                                               \
      PEA_DDMA1(xB, pb_ + w1, vo[1], so_)                       \
      PEA_DDMA1(xB, pb_ + PS + w1, vo[1], so_ + ecs)            \
      PEA_DDMA1(yB, pb_ + 2 * PS + w1, vo[1], so_)              \
      PEA_DDMA1(yB, pb_ + 3 * PS + w1, vo[1], so_ + ecs)        \
    }                                                           \
  }
// the oldest chunk in flight has landed (n younger ones may still fly), and every wave is done with the buffer behind the barrier
#define PEA_DWAIT(n)                                                                                  \
  {                                                                                                   \
    const int fly_ = (n) * npt;                                                                       \
    if (fly_ == 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");         \
    else if (fly_ == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");      \
    else if (fly_ == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");      \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                     \
  }
// half a chunk: the e pair (op_ = 0, planes 0 and 1 of the buffer) or the ema pair (op_ = 1, planes 2 and 3)
#define PEA_DDMAH(buf, ch, op_)                                                          \
  {                                                                                      \
    const unsigned so_ = ezo + (unsigned)(2 * (ch)) * ecs;                               \
    const int pb_ = (buf) * BUF + (op_) * 2 * PS;                                        \
    if (act[0]) {                                                                        \
      PEA_DDMA1((op_) ? yB : xB, pb_ + wbase, vo[0], so_)                                \
      PEA_DDMA1((op_) ? yB : xB, pb_ + PS + wbase, vo[0], so_ + ecs)                     \
    }                                                                                    \
    if (act[1]) {                                                                        \
      PEA_DDMA1((op_) ? yB : xB, pb_ + w1, vo[1], so_)                                   \
      PEA_DDMA1((op_) ? yB : xB, pb_ + PS + w1, vo[1], so_ + ecs)                        \
    }                                                                                    \
  }
// at most `loads_` (a wave-uniform multiple of 2 up to 16) younger DMA instructions may still fly; then the barrier
#define PEA_DWAITL(loads_)                                                                            \
  {                                                                                                   \
    const int fl_ = (loads_);                                                                         \
    if (fl_ >= 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");          \
    else if (fl_ >= 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");     \
    else if (fl_ >= 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");       \
    else if (fl_ >= 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");       \
    else if (fl_ >= 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");       \
    else if (fl_ >= 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");       \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                     \
  }
  const int nh = npt / 2;  // DMA instructions of this wave per HALF chunk
  if (NB == 4) {  // (half by half: a wait counts whole halves)
    PEA_DDMAH(0, 0, 0) PEA_DDMAH(0, 0, 1) PEA_DDMAH(1, 1, 0) PEA_DDMAH(1, 1, 1)
  } else {
#pragma unroll
    for (int c = 0; c < NBUF; ++c) PEA_DDMA(c, c)
  }

  // ---- LDS slot of every offset's neighbour
  int an[kXP];
  const int vown = ((C.hy0 + ly) * TW + lx) * 4;
  const int hrow = (C.QV * 4 + ly * C.SW) * 4;
#pragma unroll
  for (int k = 0; k < kXP; ++k) {
    const int d = C.fd[k], c = lx + d;
    const int a_x = (unsigned)c < (unsigned)TW ? vown + d * 4 : hrow + (c & C.fm[k]) * 4;
    an[k] = C.fax[k] ? a_x : vown + d * TW * 4;  // unused offsets: d = 0, the own slot
  }
  if (NB == 4) PEA_DWAITL(3 * nh)  // the first e pair has landed
  else PEA_DWAIT(NBUF - 1)

  f2 dot[kXP], ssq[kXP], dotc[kXP], ssqc[kXP], oss = {0.f, 0.f}, css = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < kXP; ++k) {
    dot[k] = (f2){0.f, 0.f}; ssq[k] = (f2){0.f, 0.f};
    dotc[k] = (f2){0.f, 0.f}; ssqc[k] = (f2){0.f, 0.f};
  }
// one chunk out of the ring buffer at byte bo_: own pixel (e) and centre (ema), then the ten neighbours of both operands
#define PEA_DPROC(bo_)                                                                                                   \
  {                                                                                                                      \
    f2 o, oc;                                                                                                            \
    o.x = *(const float*)(lds + (bo_) + vown);                                                                           \
    o.y = *(const float*)(lds + (bo_) + PS + vown);                                                                      \
    oc.x = *(const float*)(lds + (bo_) + 2 * PS + vown);                                                                 \
    oc.y = *(const float*)(lds + (bo_) + 3 * PS + vown);                                                                 \
    oss = __builtin_elementwise_fma(o, o, oss);                                                                          \
    css = __builtin_elementwise_fma(oc, oc, css);                                                                        \
    _Pragma("unroll") for (int k = 0; k < kXP; ++k) {                                                                    \
      f2 v, u;                                                                                                           \
      v.x = *(const float*)(lds + (bo_) + an[k]);                                                                        \
      v.y = *(const float*)(lds + (bo_) + PS + an[k]);                                                                   \
      u.x = *(const float*)(lds + (bo_) + 2 * PS + an[k]);                                                               \
      u.y = *(const float*)(lds + (bo_) + 3 * PS + an[k]);                                                               \
      dot[k] = __builtin_elementwise_fma(o, v, dot[k]);                                                                  \
      ssq[k] = __builtin_elementwise_fma(v, v, ssq[k]);                                                                  \
      dotc[k] = __builtin_elementwise_fma(o, u, dotc[k]);                                                                \
      ssqc[k] = __builtin_elementwise_fma(u, u, ssqc[k]);                                                                \
      if (NB == 2 || k % 3 == 2) asm volatile("" ::: "memory"); /* (a few reads in flight: the accumulators leave no room for more) */ \
    }                                                                                                                    \
    /* the chunk's sums exist before its barrier */                                                                     \
    _Pragma("unroll") for (int k = 0; k < kXP; ++k) asm volatile("" : "+v"(dot[k]), "+v"(ssq[k]), "+v"(dotc[k]), "+v"(ssqc[k])); \
    asm volatile("" : "+v"(oss), "+v"(css));                                                                             \
  }
// the same chunk in two parts (NB = 4): the e pair first -- own pixel kept in o_ --, then the ema pair
#define PEA_DPROC_E(bo_, o_)                                                                                             \
  {                                                                                                                      \
    o_.x = *(const float*)(lds + (bo_) + vown);                                                                          \
    o_.y = *(const float*)(lds + (bo_) + PS + vown);                                                                     \
    oss = __builtin_elementwise_fma(o_, o_, oss);                                                                        \
    _Pragma("unroll") for (int k = 0; k < kXP; ++k) {                                                                    \
      f2 v;                                                                                                              \
      v.x = *(const float*)(lds + (bo_) + an[k]);                                                                        \
      v.y = *(const float*)(lds + (bo_) + PS + an[k]);                                                                   \
      dot[k] = __builtin_elementwise_fma(o_, v, dot[k]);                                                                 \
      ssq[k] = __builtin_elementwise_fma(v, v, ssq[k]);                                                                  \
      if (k % 2 == 1) asm volatile("" ::: "memory");                                                                     \
    }                                                                                                                    \
    _Pragma("unroll") for (int k = 0; k < kXP; ++k) asm volatile("" : "+v"(dot[k]), "+v"(ssq[k]));                       \
    asm volatile("" : "+v"(oss), "+v"(o_));                                                                              \
  }
#define PEA_DPROC_E2(bo_, o_)                                                                                            \
  {                                                                                                                      \
    f2 oc;                                                                                                               \
    oc.x = *(const float*)(lds + (bo_) + 2 * PS + vown);                                                                 \
    oc.y = *(const float*)(lds + (bo_) + 3 * PS + vown);                                                                 \
    css = __builtin_elementwise_fma(oc, oc, css);                                                                        \
    _Pragma("unroll") for (int k = 0; k < kXP; ++k) {                                                                    \
      f2 u;                                                                                                              \
      u.x = *(const float*)(lds + (bo_) + 2 * PS + an[k]);                                                               \
      u.y = *(const float*)(lds + (bo_) + 3 * PS + an[k]);                                                               \
      dotc[k] = __builtin_elementwise_fma(o_, u, dotc[k]);                                                               \
      ssqc[k] = __builtin_elementwise_fma(u, u, ssqc[k]);                                                                \
      if (k % 2 == 1) asm volatile("" ::: "memory");                                                                     \
    }                                                                                                                    \
    _Pragma("unroll") for (int k = 0; k < kXP; ++k) asm volatile("" : "+v"(dotc[k]), "+v"(ssqc[k]));                     \
    asm volatile("" : "+v"(css));                                                                                        \
  }
  if constexpr (NB == 4) {
    // the ring of two buffers handed over in HALVES: a half is requested as soon as its planes are free, three halves (one and a half
    // chunks) ahead of the gather instead of one chunk -- twice the barriers, a steadier stream of requests
#pragma unroll 1
    for (int it = 0; it < NP / 2; ++it) {
      const bool more = it + 1 < NP / 2;
      f2 o0, o1;
      PEA_DPROC_E(0, o0)
      PEA_DWAITL(2 * nh)  // (buffer 0, ema) landed; (1, e) and (1, ema) may fly
      if (more) PEA_DDMAH(0, 2 * it + 2, 0)
      PEA_DPROC_E2(0, o0)
      PEA_DWAITL(more ? 2 * nh : nh)  // (1, e) landed
      if (more) PEA_DDMAH(0, 2 * it + 2, 1)
      PEA_DPROC_E(BUF, o1)
      PEA_DWAITL(more ? 2 * nh : 0)  // (1, ema) landed
      if (more) PEA_DDMAH(1, 2 * it + 3, 0)
      PEA_DPROC_E2(BUF, o1)
      if (more) {
        PEA_DWAITL(2 * nh)  // the next (0, e) landed
        PEA_DDMAH(1, 2 * it + 3, 1)
      }
    }
  } else if constexpr (NB == 2) {
    // a ROLLED loop over pairs of chunks (buffer 0, buffer 1): unrolled eight times the compiler renames the 84 accumulator registers
    // from chunk to chunk and spills the neighbour slots inside the loop -- behind scratch loads whose s_waitcnt vmcnt(0) also waits
    // for the DMA in flight
#pragma unroll 1
    for (int it = 0; it < NP / 2; ++it) {
      const bool more = it + 1 < NP / 2;
      PEA_DPROC(0)
      PEA_DWAIT(0)
      if (more) PEA_DDMA(0, 2 * it + 2)
      PEA_DPROC(BUF)
      if (more) {
        PEA_DWAIT(0)
        PEA_DDMA(1, 2 * it + 3)
      }
    }
  } else {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      PEA_DPROC((ps % NBUF) * BUF)
      if (ps + 1 < NP) {
        PEA_DWAIT(NBUF - 2 < NP - 2 - ps ? NBUF - 2 : NP - 2 - ps)
        if (ps + NBUF < NP) PEA_DDMA(ps % NBUF, ps + NBUF)
      }
    }
  }
#undef PEA_DPROC
#undef PEA_DPROC_E
#undef PEA_DPROC_E2
#undef PEA_DDMAH
#undef PEA_DWAITL
#undef PEA_DDMA
#undef PEA_DDMA1
#undef PEA_DWAIT

  // ---- normalise; both 1 / norm planes for the backward
  const float inv_eps = 1.0f / P.eps;
  const float osum = oss.x + oss.y, csum = css.x + css.y;
  const float inv_own = rnorm(osum, inv_eps), inv_c = rnorm(csum, inv_eps);
  bs32(iB, osum < P.eps * P.eps ? -inv_own : inv_own, pe, ezo);
  bs32(i2B, csum < P.eps * P.eps ? -inv_c : inv_c, pe, ezo);
  lds_barrier();  // every lane is done with the ring: the parked maps go over it
#pragma unroll
  for (int k = 0; k < kXP; ++k) {
    if (k < C.nf) {  // uniform
      float a = (dot[k].x + dot[k].y) * inv_own * rnorm(ssq[k].x + ssq[k].y, inv_eps);
      float c = (dotc[k].x + dotc[k].y) * inv_own * rnorm(ssqc[k].x + ssqc[k].y, inv_eps);
      if (CROP) {
        const int q = (C.fax[k] ? px : py) + C.fd[k];
        const bool in = (unsigned)q < (unsigned)(C.fax[k] ? P.X : P.Y);
        a = in ? a : 0.f;
        c = in ? c : 0.f;
      }
      sA[C.fi[k] * TP + (int)threadIdx.x] = a;
      sA2[C.fi[k] * TP + (int)threadIdx.x] = c;
    }
  }

  // ---- epilogue: item = (offset, quad of 4 x-adjacent tile pixels); target / weight / mask requested only now (the accumulators
  //      are dead), read ONCE for both losses
  bool ion[ITEMS];
  unsigned ivo[ITEMS];
  int iqd[ITEMS], igy[ITEMS], igx[ITEMS], isl[ITEMS];
  f4 t4[ITEMS], w4[ITEMS];
  unsigned m4[ITEMS];
  int tid_i = (int)threadIdx.x;
  asm volatile("" : "+v"(tid_i));  // opaque: not hoisted back over the channel loop
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int tt = it * NT + tid_i;
    const int sl = __builtin_amdgcn_readfirstlane(tt / QP);
    ion[it] = sl < P.K;
    isl[it] = min(sl, P.K - 1);
    const int qd = tt - (tt / QP) * QP;
    iqd[it] = qd;
    const int l4 = qd * 4;
    igy[it] = y0 + l4 / TW;
    igx[it] = x0 + l4 % TW;
    const bool lv = ion[it] && igy[it] < P.Y && igx[it] < P.X;  // X % 4 == 0: a quad is inside or outside as a whole
    ivo[it] = lv ? (unsigned)(igy[it] * P.X + igx[it]) * 4u : kOOB;
  }
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const unsigned so = ezo + (unsigned)isl[it] * ecs;
    t4[it] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(tB, ivo[it], so, kAuxNT));
    w4[it] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(wB, ivo[it], so, kAuxNT));
    m4[it] = has_m ? __builtin_amdgcn_raw_buffer_load_b32(mB, ivo[it] == kOOB ? kOOB : ivo[it] >> 2,
                                                         (ezo >> 2) + (unsigned)isl[it] * (unsigned)P.S, kAuxNT)
                   : 0x01010101u;
  }
  lds_barrier();
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    if (!ion[it]) continue;  // wave-uniform
    const int sl = isl[it];
    const f4 a4 = *(const f4*)(sA + sl * TP + iqd[it] * 4);
    const f4 c4 = *(const f4*)(sA2 + sl * TP + iqd[it] * 4);
    const unsigned so = ezo + (unsigned)sl * ecs;
    if (has_a) {
      f4 o = a4;
      if (af) { o.x = act_affs(o.x, af); o.y = act_affs(o.y, af); o.z = act_affs(o.z, af); o.w = act_affs(o.w, af); }
      bs128<true>(aB, o, ivo[it], so);
    }
    float acc = 0.f, acc2 = 0.f;
    f4 g4, h4;
    const float gs = C.gs[sl], gs2 = DA.gs2[sl];
    const int ax_ = C.oax[sl], od_ = C.od[sl];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float m = (float)((m4[it] >> (8 * j)) & 0xffu);
      const float r = a4[j] * m - t4[it][j] * m;
      const float r2 = c4[j] * m - t4[it][j] * m;
      float wr = w4[it][j] * r, wr2 = w4[it][j] * r2;
      if (CROP) {  // a cropped-away neighbour carries no loss term (its a is already 0)
        const int q = (ax_ == 1 ? igx[it] + j : igy[it]) + od_;
        const bool in = (unsigned)q < (unsigned)(ax_ == 1 ? P.X : P.Y);
        wr = in ? wr : 0.f;
        wr2 = in ? wr2 : 0.f;
      }
      g4[j] = gs * wr * m;
      h4[j] = gs2 * wr2 * m;
      acc = fmaf(wr, r, acc);
      acc2 = fmaf(wr2, r2, acc2);
    }
    bs128<false>(gB, g4, ivo[it], so);
    bs128<false>(g2B, h4, ivo[it], so);
    const float red = wave_sum63(acc), red2 = wave_sum63(acc2);
    if ((threadIdx.x & 63) == 63) {
      s_part[sl * NSL + (iqd[it] >> 6)] = red;
      s_part2[sl * NSL + (iqd[it] >> 6)] = red2;
    }
  }
  lds_barrier();
  if (wave < 2) {  // the two waves that touch the loss states: wave 0 the self loss', wave 1 the cross loss'
    if (lane < P.K) {
      const float* sp = wave == 0 ? s_part : s_part2;
      float v = 0.f;
#pragma unroll
      for (int s = 0; s < NSL; ++s) v += sp[lane * NSL + s];
      loss_accumulate(wave == 0 ? st : DA.st2, tile, lane, v);
    }
  }
}

}  // namespace pea
