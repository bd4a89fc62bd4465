// pea_xdma_hq.h -- the f16-storage cross BACKWARD (pea_xdma_h16.h k_bwd_xdma_h<PF, HW>) with producer and consumer waves, the chunk
// staged through registers four channels at a time, on 8 x 64 tiles (round 4; BASELINE.json configs[4]: 262-287 -> 242-262 us).
//
// What the round-3 kernel pays for, found by compiling phases out (profiles/r4_c5_hq_ablation.txt) and by a microbenchmark of the
// vector-memory path (profiles/microbench/vmem_issue.hip: a LOAD wave instruction costs the CU 16-18 cycles whatever its width --
// 2 to 16 bytes per lane -- and however few of its lanes are active; a 2-byte store 10, a dword store 14.5):
//   * its phases ADD UP (no stores -90 us, no chunk loads -64 us, no gather -60 us of 340): loads and stores share vmcnt and a store may
//     retire before an older load (pea_zmarch.h), so a wave that does both can only say "my chunk has landed" with a count that also
//     waits for the gradient stores of the chunk before -- the store latency sits in every chunk's critical path;
//   * 64-byte rows (16 x 32 tiles of f16): half lines in every request, 1.26x write amplification.
// Here:
//   * eight CONSUMER waves (a pixel per lane: gather, finish four channels, store them) and four PRODUCER waves (request the chunks,
//     transpose, write the working buffers), one barrier per chunk of FOUR channels between them.  A producer issues loads only: its
//     counts are exact and it is the only wave that ever waits for memory; a consumer issues its stores and never waits for them;
//   * a producer lane takes an OCT of eight x-adjacent region pixels of four channel planes into registers (4 x buffer_load_dwordx4:
//     the fewest instructions the chunk can be requested in), transposes them (16 v_perm_b32) and writes [pixel][4 halves] -- 8 bytes
//     per region pixel -- into one of two working buffers: no DMA write and no interleave read (the LDS array sees the chunk once
//     instead of three times: SQ_LDS_IDX_ACTIVE 58 M -> 30 M), one ds_read_b64 per pair for four channels (two LDS cycles, like the
//     ds_read_b32 it replaces: MI355X_MICROARCH.md section LDS);
//   * 8 x 64 tiles: a tile row is one whole 128-byte line of an f16 plane, loads and stores alike.
// A quad's two 16-byte pieces swap places in every second 128-byte block (hq_addr): reads stay conflict-free for any start pixel of
// a row (u -> u ^ (bit4(u) << 1) is a bijection on the 32 eight-byte units of a read) and writers a quad (32 bytes) apart spread over
// the banks; the producers' octs are 64 bytes apart and take two passes per ds_write_b128 -- off the consumers' path.  Arithmetic, order of operations and results are those of k_bwd_xdma_h<PF, HW>, bit for bit.
// Measured and NOT kept (same file, removed): the forward in the same two forms -- every wave staging quads (162-169 us against
// 157-161 for k_fwd_xdma_h), producer / consumer waves with octs (184 us): the forward has no stores in its chunk loop to decouple and
// three or four workgroups per CU already hide its loads; the backward with every wave loading AND storing (285-307 us); three and
// four register sets per producer (no change: the chunk loads are not latency-bound); eight producer waves (no change).
// Self loss, 2D, X % 8 == 0, axis-aligned stencils whose region fits 1920 pixels (reach <= 9 both ways), D in {32, 64}.
#pragma once
#include "pea_xdma_h16.h"

namespace pea {

typedef unsigned u2_t __attribute__((ext_vector_type(2)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));

// byte address of region pixel p in a working buffer (8 bytes per pixel)
__device__ __forceinline__ int hq_addr(int p) { return (p ^ ((p >> 3) & 2)) * 8; }

// the two 16-byte pieces of a quad: [pixel 0: c0 c1 c2 c3][pixel 1: ...], [pixel 2][pixel 3] from four planes' (p0 p1 | p2 p3) halves
__device__ __forceinline__ void hq_transpose(const u2_t (&r)[4], u4_t& lo, u4_t& hi) {
  lo.x = __builtin_amdgcn_perm(r[1].x, r[0].x, 0x05040100u);
  lo.y = __builtin_amdgcn_perm(r[3].x, r[2].x, 0x05040100u);
  lo.z = __builtin_amdgcn_perm(r[1].x, r[0].x, 0x07060302u);
  lo.w = __builtin_amdgcn_perm(r[3].x, r[2].x, 0x07060302u);
  hi.x = __builtin_amdgcn_perm(r[1].y, r[0].y, 0x05040100u);
  hi.y = __builtin_amdgcn_perm(r[3].y, r[2].y, 0x05040100u);
  hi.z = __builtin_amdgcn_perm(r[1].y, r[0].y, 0x07060302u);
  hi.w = __builtin_amdgcn_perm(r[3].y, r[2].y, 0x07060302u);
}

// chunk c is requested into register set c % NS (NS sets: NS - 1 chunks fly while one is written) and written to working buffer c & 1.
// A lane stages OCTS (8 x-adjacent region pixels, one buffer_load_dwordx4 per plane): a vector-memory wave instruction costs the CU
// 16-18 cycles whatever its width and however few of its lanes are active (profiles/microbench/vmem_issue.hip), so the chunk is
// requested in as few instructions as it can be, and a wave none of whose octs exist (`ldw` false) issues none.
// (PEA_ABL_HQ_*: diagnostic builds of profiles/build_variant_tu.sh only -- timing with a phase compiled out)
#ifdef PEA_ABL_HQ_NOLOAD
#define PEA_HQ_LOAD(c)                                                                                                     \
  {                                                                                                                        \
    _Pragma("unroll") for (int s_ = 0; s_ < NQ; ++s_) {                                                                    \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) raw[(c) % NS][s_][i_] = (u4_t){voq[s_] + (c), voq[s_] ^ i_, 0u, 1u}; \
    }                                                                                                                      \
  }
#else
#define PEA_HQ_LOAD(c)                                                                                                     \
  if (ldw) {                                                                                                               \
    _Pragma("unroll") for (int s_ = 0; s_ < NQ; ++s_) {                                                                    \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                     \
        raw[(c) % NS][s_][i_] =                                                                                            \
            __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(xB, voq[s_], hzo + (unsigned)(4 * (c) + i_) * hcs, 0)); \
    }                                                                                                                      \
  }
#endif
#ifdef PEA_ABL_HQ_NOWRITE
#define PEA_HQ_WRITE(c)                                                                                                    \
  {                                                                                                                        \
    _Pragma("unroll") for (int s_ = 0; s_ < NQ; ++s_) {                                                                    \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) asm volatile("" ::"v"(raw[(c) % NS][s_][i_]));                      \
    }                                                                                                                      \
  }
#else
// (the oct's two quads: 32 bytes each, their two 16-byte pieces swapped in every second 128-byte block -- hq_addr)
#define PEA_HQ_WRITE(c)                                                                                                    \
  {                                                                                                                        \
    _Pragma("unroll") for (int s_ = 0; s_ < NQ; ++s_) {                                                                    \
      if (act[s_]) {                                                                                                       \
        u4_t lo_, hi_;                                                                                                     \
        const u4_t(&r_)[4] = raw[(c) % NS][s_];                                                                            \
        const u2_t qa_[4] = {{r_[0].x, r_[0].y}, {r_[1].x, r_[1].y}, {r_[2].x, r_[2].y}, {r_[3].x, r_[3].y}};              \
        hq_transpose(qa_, lo_, hi_);                                                                                       \
        *(u4_t*)(W + ((c) & 1) * WB + wq[s_]) = lo_;                                                                       \
        *(u4_t*)(W + ((c) & 1) * WB + (wq[s_] ^ 16)) = hi_;                                                                \
        const u2_t qb_[4] = {{r_[0].z, r_[0].w}, {r_[1].z, r_[1].w}, {r_[2].z, r_[2].w}, {r_[3].z, r_[3].w}};              \
        hq_transpose(qb_, lo_, hi_);                                                                                       \
        *(u4_t*)(W + ((c) & 1) * WB + wq[s_] + 32) = lo_;                                                                  \
        *(u4_t*)(W + ((c) & 1) * WB + (wq[s_] ^ 16) + 32) = hi_;                                                           \
      }                                                                                                                    \
    }                                                                                                                      \
  }
#endif
// the chunk whose registers are about to be written has landed: only the `younger` chunks requested after it (4 * NQ loads each) may
// still fly.  Loads retire in order, so "at most that many outstanding" means the older chunk is complete whatever the stores did
// (they share the counter and may retire early or late: pea_zmarch.h zm_bwd_wait -- a late store only makes this wait longer,
// never unsound)
__device__ __forceinline__ void hq_wait(int n) {
#define PEA_HQW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    PEA_HQW(0) PEA_HQW(4) PEA_HQW(8) PEA_HQW(12) PEA_HQW(16) PEA_HQW(20) PEA_HQW(24) PEA_HQW(28) PEA_HQW(32) PEA_HQW(36) PEA_HQW(40)
    PEA_HQW(48) PEA_HQW(56)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef PEA_HQW
}
#define PEA_HQ_LANDED(younger) hq_wait(4 * NQ * (younger));

// ------------------------------------------------------------------------------------------------------------------
// backward, self loss (both roles), f16 e / de, the projection first (pea_xdma_pf.h: `affs` = the forward's raw cosine map)
// ------------------------------------------------------------------------------------------------------------------
template <int TW, bool CROP>
__device__ __forceinline__ unsigned x_quad_pixel(const KParams& P, const XParams& C, int y0, int x0, int q, bool& act) {
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
  act = q < C.QA;
  bool oky, okx;
  gy = wrap1<CROP>(gy, P.Y, oky);
  gx = wrap1<CROP>(gx, P.X, okx);
  return (act && oky && okx) ? (unsigned)(gy * P.X + gx) : kOOB;
}

// oct o of the region (8 x-adjacent pixels; X % 8 == 0 and 8-pixel strip geometry: inside or outside as a whole)
template <int TW, bool CROP>
__device__ __forceinline__ unsigned x_oct_pixel(const KParams& P, const XParams& C, int y0, int x0, int o, bool& act) {
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
  act = o < (C.QA >> 1);
  bool oky, okx;
  gy = wrap1<CROP>(gy, P.Y, oky);
  gx = wrap1<CROP>(gx, P.X, okx);
  return (act && oky && okx) ? (unsigned)(gy * P.X + gx) : kOOB;
}

// NS: register sets of a producer lane = chunks it has requested ahead of the one it writes
template <int D_T, int TH, int TW, int PSU, bool CROP, int XP, int WPE, int NPW, int NS = 2>
__global__ __launch_bounds__(TH* TW + 64 * NPW, WPE) void k_bwd_xdma_hqs(const KParams P, const XParams C, const __half* __restrict__ xt,
                                                                         const float* __restrict__ invp, const float* __restrict__ gin,
                                                                         const float* __restrict__ affs, const float* __restrict__ dloss,
                                                                         __half* __restrict__ dx) {
  constexpr int NT = TH * TW, PS = PSU * 256, WB = 2 * PS, NC = D_T / 4, NPL = 64 * NPW, NQ = (PS / 32 + NPL - 1) / NPL;  // octs per lane
  static_assert((TW == 32 || TW == 64) && D_T % 4 == 0 && PS % 512 == 0 && XP % 2 == 0 && NC >= 3, "lane mapping / channel quads");
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  char* const W = lds;  // two working buffers; the 1 / norm plane (f32, PS bytes) sits in the second one first
  int tile, b, z, y0, x0;
  if (!xdma_tile<TH, TW>(C, P, tile, b, z, y0, x0)) return;
  const size_t S = (size_t)P.S;
  const unsigned YX = (unsigned)(P.Y * P.X);
  const rsrc_t xB = mkbuf(xt + (size_t)b * D_T * S);
  const unsigned hcs = (unsigned)P.S * 2u, hzo = (unsigned)z * YX * 2u;  // e / de (f16): channel stride, plane offset
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;

  if (wave >= NT / 64) {
    // ---------------- producer: quads pl, pl + NPL, ... of the region, four planes per chunk ----------------
    const int pl = (int)threadIdx.x - NT;
    unsigned voq[NQ];
    bool act[NQ];
    int wq[NQ];
    bool any = false;
#pragma unroll
    for (int s = 0; s < NQ; ++s) {
      const int o = s * NPL + pl;
      const unsigned px = x_oct_pixel<TW, CROP>(P, C, y0, x0, o, act[s]);
      voq[s] = px == kOOB ? kOOB : px * 2u;
      wq[s] = o * 64 + ((o >> 1) & 1) * 16;
      any |= act[s];
    }
    const bool ldw = __builtin_amdgcn_ballot_w64(any) != 0;
    static_assert(NS >= 2 && NC > NS, "register sets");
    u4_t raw[NS][NQ][4];
#pragma unroll
    for (int c = 0; c < NS; ++c) PEA_HQ_LOAD(c)
    PEA_HQ_LANDED(NS - 1)
    PEA_HQ_WRITE(0)
    PEA_HQ_LOAD(NS)
    lds_barrier();  // #0: chunk 0 is in the first buffer
    lds_barrier();  // #1: the consumers have read the 1 / norm plane: the second buffer may be written
#pragma unroll
    for (int c = 0; c + 1 < NC; ++c) {
      PEA_HQ_LANDED(NC - 2 - c < NS - 1 ? NC - 2 - c : NS - 1)  // exact: this wave issues loads only
      PEA_HQ_WRITE(c + 1)
      if (c + 1 + NS < NC) PEA_HQ_LOAD(c + 1 + NS)
      lds_barrier();
    }
    return;
  }

  // ---------------- consumer ----------------
  const rsrc_t dB = mkbuf(dx + (size_t)b * D_T * S);
  const rsrc_t gB = mkbuf(gin + (size_t)b * P.K * S), iB = mkbuf(invp + (size_t)b * S);
  const rsrc_t aB = mkbuf(affs + (size_t)b * P.K * S);
  const unsigned fcs = (unsigned)P.S * 4u, fzo = (unsigned)z * YX * 4u;  // g, affs, 1 / norm (f32)
  const float dl = dloss ? dloss[0] : 1.f;
  const int ly = threadIdx.x / TW, lx = threadIdx.x % TW;
  const int py = y0 + ly, px = x0 + lx;
  const bool live = py < P.Y && px < P.X;
  const unsigned po = (unsigned)(py * P.X + px);
  const unsigned ph = live ? po * 2u : kOOB;
  {
    // the 1 / norm plane (f32) -> the second working buffer, by LDS-DMA (planar: 4 bytes per region pixel)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bool a;
      const unsigned pq = x_quad_pixel<TW, CROP>(P, C, y0, x0, (s * (NT / 64) + wave) * 64 + lane, a);
      if (a) __builtin_amdgcn_raw_ptr_buffer_load_lds(iB, (lds_ptr_t)(W + WB + (s * (NT / 64) + wave) * 1024), 16,
                                                      pq == kOOB ? kOOB : pq * 4u, fzo, 0, 0);
    }
  }
  float proj = 0.f;
  const unsigned pg = live ? po * 4u : 0xC0000000u;
  f2 cx2[XP / 2], cy2[XP / 2];  // pair k in half (k & 1) of element k / 2
  int ax[XP], ay[XP];
  const int pown = (C.hy0 + ly) * TW + lx;
  const int prow = C.QV * 4 + ly * C.SW;
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    const int go = C.xgo[k];
    const int t = px + go;
    const bool out = (unsigned)t >= (unsigned)P.X;
    const int fix = go > 0 ? -P.X : P.X;
    const unsigned o = CROP ? (out ? kOOB : pg + (unsigned)(go * 4)) : pg + (unsigned)((out ? go + fix : go) * 4);
    const float gk = bl32(gB, k < C.npx ? o : kOOB, fzo + (unsigned)C.xgi[k] * fcs);
    cx2[k / 2][k & 1] = gk;
    proj = fmaf(gk, bl32(aB, k < C.npx ? o : kOOB, fzo + (unsigned)C.xgi[k] * fcs), proj);
    const int d = C.xd[k], c = lx + d;
    ax[k] = (unsigned)c < (unsigned)TW ? pown + d : prow + (c & C.xm[k]);
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
    proj = fmaf(gk, bl32(aB, k < C.npy ? o : kOOB, fzo + (unsigned)C.ygi[k] * fcs), proj);
    ay[k] = pown + C.yd[k] * TW;
  }
  // #0: the 1 / norm plane, g and the raw map have landed (the only time a consumer waits for memory)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  const float invo = *(const float*)(W + WB + pown * 4);
  const float inv_own = fabsf(invo);
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    cx2[k / 2][k & 1] *= fabsf(*(const float*)(W + WB + ax[k] * 4));
    cy2[k / 2][k & 1] *= fabsf(*(const float*)(W + WB + ay[k] * 4));
    if (k & 1) asm volatile("" : "+v"(cx2[k / 2]), "+v"(cy2[k / 2]));
  }
#pragma unroll
  for (int k = 0; k < XP; ++k) { ax[k] = hq_addr(ax[k]); ay[k] = hq_addr(ay[k]); }
  const int aown = hq_addr(pown);
  lds_barrier();  // #1: the 1 / norm plane is dead
  if (invo < 0.f) proj = 0.f;  // clamp branch of F.normalize
  asm volatile("" : "+v"(proj));

#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const char* const Wc = W + (c & 1) * WB;
    const h4_t oh = *(const h4_t*)(Wc + aown);
    const f4 o = (f4){(float)oh.x, (float)oh.y, (float)oh.z, (float)oh.w} * inv_own;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
#ifndef PEA_ABL_HQ_NOGATHER
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      const h4_t v = *(const h4_t*)(Wc + ax[k]);
      const float cf = (k & 1) ? cx2[k / 2].y : cx2[k / 2].x;
      acc.x = __builtin_fmaf((float)v.x, cf, acc.x);
      acc.y = __builtin_fmaf((float)v.y, cf, acc.y);
      acc.z = __builtin_fmaf((float)v.z, cf, acc.z);
      acc.w = __builtin_fmaf((float)v.w, cf, acc.w);
      if (k % 4 == 3) asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      const h4_t v = *(const h4_t*)(Wc + ay[k]);
      const float cf = (k & 1) ? cy2[k / 2].y : cy2[k / 2].x;
      acc.x = __builtin_fmaf((float)v.x, cf, acc.x);
      acc.y = __builtin_fmaf((float)v.y, cf, acc.y);
      acc.z = __builtin_fmaf((float)v.z, cf, acc.z);
      acc.w = __builtin_fmaf((float)v.w, cf, acc.w);
      if (k % 4 == 3) asm volatile("" ::: "memory");
    }
#else
    acc = (f4){cx2[c % (XP / 2)].x, cy2[c % (XP / 2)].y, cx2[0].y, cy2[1].x};
#endif
    // the f32 results exist before they are rounded to f16 (left alone the compiler folds `* dl` and the conversion into one
    // v_fma_mixlo_f16: a single rounding, one f16 ulp away from k_bwd_xdma_h's in 5 of 100 000 values)
    f4 sx = {pf_finish(acc.x, o.x, proj, inv_own, dl), pf_finish(acc.y, o.y, proj, inv_own, dl),
             pf_finish(acc.z, o.z, proj, inv_own, dl), pf_finish(acc.w, o.w, proj, inv_own, dl)};
    asm volatile("" : "+v"(sx));
#ifdef PEA_ABL_HQ_NOSTORE
    if (c == NC - 1) bs_emb<__half, true>(dB, sx.x + sx.y + sx.z + sx.w, ph, hzo);
#else
    bs_emb<__half, true>(dB, sx.x, ph, hzo + (unsigned)(4 * c) * hcs);
    bs_emb<__half, true>(dB, sx.y, ph, hzo + (unsigned)(4 * c + 1) * hcs);
    bs_emb<__half, true>(dB, sx.z, ph, hzo + (unsigned)(4 * c + 2) * hcs);
    bs_emb<__half, true>(dB, sx.w, ph, hzo + (unsigned)(4 * c + 3) * hcs);
#endif
    if (c + 1 < NC) lds_barrier();
  }
}

#undef PEA_HQ_LOAD
#undef PEA_HQ_WRITE
#undef PEA_HQ_LANDED

}  // namespace pea
