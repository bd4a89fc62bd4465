// pea_zmarch.h -- 3D volumes with axis-aligned stencils that step along z (AC3/AC4 embedding_loss_norm5 / norm1,
// scripts_ac3ac4/loss/loss_embedding_mse.py:7-27, 143-194): a workgroup MARCHES along z through one 16 x 32 tile column and keeps
// what the z offsets need in registers, instead of gathering the z neighbours of every plane from global memory again.
//
// Why: in k_fwd_xdma<.., ZF> / k_bwd_xdma<.., ZP> (pea_xdma.h) a z neighbour is the SAME (y, x) in another plane, read per channel
// chunk with one-dword loads: 4 (forward) / 8 (backward) loads per voxel and channel on top of the in-plane cross, 12.9 GB through
// L2 per backward launch on a 24 x 1024^2 sub-volume, 8.9 GB fetched beyond L2 for 2.9 GB of compulsory reads; the four z offsets
// cost 0.75 + 1.13 ms of the 1.70 + 2.55 ms (profiles/r3_3d_split.txt).  A lane that walks its column plane by plane has seen
// every one of those values already:
//   forward : a_s(z) = <ehat(z), ehat(z - s)>, s = 1 .. 4.  The lane keeps the RAW own pixel of the last four planes (4 x 16
//             registers, slot = plane mod 4) and their 1 / norm; per channel chunk the four dot products are four packed FMAs.
//   backward: G(z) += g_s(z) ehat(z - s)  (role A)  +  g_s(z + s) ehat(z + s)  (role B).  Role A reads the window of the last four
//             NORMALISED planes; role B is turned round: when plane z' is current, g_s(z') ehat(z') is ADDED to the pending sum of
//             plane z' - s.  So four planes are pending (4 x 16 registers), both roles use the one coefficient g_s(z') of the
//             current plane (g of a z offset is read ONCE), and plane z' - 4 is complete -- and stored -- while plane z' is
//             gathered.  <ehat, G> of a pending plane collects g_s(z') a_s(z') with a_s the raw cosine the forward wrote
//             (pea_affinity_bwd_ex2's affs): the projection-first identity of pea_xdma_pf.h, for the z pairs only.
// The in-plane cross is staged exactly as in pea_xdma.h (LDS-DMA ring of three two-channel buffers).  128 + 64 window registers
// do not fit two workgroups per CU: ONE workgroup of 8 waves per CU with up to 256 VGPRs; the march is what has to hide the
// latency the second workgroup used to hide (the next plane's requests are issued while the current one is gathered).
//
// A column may be cut into segments of zseg planes (more workgroups than CUs on small volumes): a segment [zb, ze) warms its
// window up on planes zb - 4 .. zb - 1 (own pixel only, global loads) and, in the backward, walks planes ze .. ze + 3 as
// contributors (role B into its last planes) before it drains.  CROP_ZERO border only (the reference's 3D border), D = 16, f32,
// every z offset negative with |oz| <= 4.
#pragma once
#include "pea_xdma.h"

namespace pea {

constexpr int kZS = 4;  // longest step along z the window covers

// how many (neighbour, channel pair) LDS reads the compiler may batch in the gather loops (a fence every N pairs; 0: no fence).
// Diagnostic builds (profiles/build_variant.sh) override them to A/B on one box.
#ifndef PEA_ZMV_BFENCE
#define PEA_ZMV_BFENCE 2
#endif
#ifndef PEA_ZMV_FFENCE
#define PEA_ZMV_FFENCE 0
#endif
#ifndef PEA_ZMV_STORE_LATE
#define PEA_ZMV_STORE_LATE 0  // 1: the march backward's gradient stores behind the hand-off instead of ahead of it
#endif
#ifndef PEA_ZMV_STORE_AUX
#define PEA_ZMV_STORE_AUX kAuxNT  // cache policy of the march backward's gradient stores (0: plain, 1: sc0, 2: nt, 16: sc1)
#endif

struct ZMParams {
  int zch[kZS];    // channel of the offset (-s, 0, 0), s = 1 .. kZS; -1: not in the table
  int zseg, nseg;  // planes per segment, segments per tile column
};

#define PEA_ZM_DMA(rsrc, plane_byte, so)                                                                          \
  {                                                                                                               \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + (plane_byte) + wbase), 16, vo0, so, 0, 0);   \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + (plane_byte) + w1), 16, vo1, so, 0, 0);      \
  }
#define PEA_ZM_WAIT(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(n) : "memory")

// ------------------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------------------
// slot of the plane d steps behind the current one (d = 1 .. 4; d = 4 is the slot the current plane takes over)
template <int JO, int D_>
struct ZmSlot { static constexpr int v = ((JO - D_) % kZS + kZS) % kZS; };

// one staged plane z of the segment: in-plane pairs from the LDS cross, z pairs from the window; finishes plane z - 4.
//
// The ring never stops: a plane is NINE items -- I (its 1 / norm plane, one LDS plane) and the chunks C0 .. C7 -- for a ring of
// NB = 3 or 4 buffers; item i of the plane whose step is JO (mod 4) lives in buffer (9 JO + i) mod NB (a compile-time constant: the
// steps are unrolled four to a turn), and while an item is worked on, the NB - 1 behind it are in flight -- the last of a plane
// already the NEXT plane's.  The g / a values a plane's coefficients are made of (16 in-plane pairs, 4 + 4 z values per lane) are
// prefetched the same way: one-dword LDS-DMA into a private [value][lane] block per wave (no registers: the window leaves none),
// requested together with the next plane's I.  Buffers beyond the 64 KB a ds_read immediate reaches get their addresses formed
// per chunk.
// Every staged plane issues the SAME sequence of LOAD instructions per wave (planes beyond the volume, and the plane after the
// last, are requested at out-of-range offsets), so the hand-offs are literal s_waitcnt vmcnt(N), N = the loads that may still fly
// behind the item waited for (zm_bwd_wait; the two gradient stores per chunk are NOT counted, see there):
//     item i: W_i: item i + 1 has landed; barrier; request item i + NB (of the next plane from i + NB >= 9 on; item 0 = I,
//     2 instructions, + G, 24); [2 stores of plane z - 4, i >= 1]
// NB = 3: vmcnt(4) everywhere, vmcnt(26) behind the request of I + G.   NB = 4: 8 and 30.
// The first step requests its items 0 .. NB - 1 itself.
constexpr int kZmG = 24;  // prefetched values per lane: 8 + 8 in-plane pairs, 4 g and 4 a of the z offsets

constexpr int zm_bwd_sz(int m) { return ((m % 9) + 9) % 9 == 0 ? 2 + kZmG : 4; }   // LOAD instructions of the request of item m
// vmcnt of W_i: the LOADS issued after the request of item i + 1 (made after item i + 1 - NB) and before W_i.  Stores are NOT
// counted although they sit on the same counter: loads retire in order among themselves, but a store may retire before an older
// load (LLVM's waitcnt pass treats a gfx9 vmcnt with loads and stores pending as out of order for the same reason).  Counting the
// stores in -- "only these may still fly" -- let a wait pass with part of the awaited chunk in flight whenever stores had retired
// early: with a ring of three the gradient differed between identical launches at 17 k of 403 M values (runs of eight pixels;
// profiles/r4_zm_race.txt), with a ring of four it never showed -- and was just as unsound.  With loads only, outstanding stores
// make a wait a little longer, never shorter.
constexpr int zm_bwd_wait(int NB, int i) {
  int n = 0;
  for (int j = i + 2 - NB; j <= i - 1; ++j) n += zm_bwd_sz(j + NB);
  return n;
}
static_assert(zm_bwd_wait(4, 0) == 8 && zm_bwd_wait(4, 3) == 8 && zm_bwd_wait(4, 6) == 30 && zm_bwd_wait(4, 7) == 30 &&
              zm_bwd_wait(4, 8) == 8 && zm_bwd_wait(3, 0) == 4 && zm_bwd_wait(3, 1) == 4 && zm_bwd_wait(3, 2) == 4 &&
              zm_bwd_wait(3, 7) == 26 && zm_bwd_wait(3, 8) == 4, "the hand-off counts derived by hand");

template <int JO, int TH, int TW, int PSU, int NB>
__device__ __forceinline__ void zm_bwd_full(const KParams& P, const XParams& C, const ZMParams& M, char* lds, const rsrc_t xB,
                                            const rsrc_t iB, const rsrc_t gB, const rsrc_t aB, const rsrc_t dB, const int z,
                                            const bool first, const bool nxt, const bool fin, const float dl, const unsigned ecs,
                                            const unsigned YX4, const unsigned vo0, const unsigned vo1, const int wbase, const int w1,
                                            const int gwave, const int glane, const unsigned pe, const int px, const int py,
                                            const int vown, const int (&ax)[8], const int (&ay)[8], f2 (&Eh)[8][kZS],
                                            f2 (&Pd)[8][kZS], float (&prj)[kZS], float (&ivs)[kZS]) {
  constexpr int PS = PSU * 256, NP = 8, XP = 8, GB = 2 * NB * PS;  // the g / a blocks sit behind the ring
  constexpr int JN = (JO + 1) % 4;                                 // the next plane's step
#define PEA_ZMB_BUF(jo_, item_) ((9 * (jo_) + (item_)) % NB)
  const bool here = z < P.Z, there = nxt && z + 1 < P.Z;  // the plane / the next one exists (else: every request out of range)
  const unsigned ezo = (unsigned)(here ? z : 0) * YX4;
  const unsigned ezn = (unsigned)(there ? z + 1 : 0) * YX4;
// one chunk (two planes) into ring buffer `buf`
#define PEA_ZMB_CHUNK(buf, so_, v0_, v1_)                                                                                        \
  {                                                                                                                              \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(lds + (2 * (buf)) * PS + wbase), 16, v0_, so_, 0, 0);               \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(lds + (2 * (buf)) * PS + w1), 16, v1_, so_, 0, 0);                  \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(lds + (2 * (buf) + 1) * PS + wbase), 16, v0_, (so_) + ecs, 0, 0);   \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(lds + (2 * (buf) + 1) * PS + w1), 16, v1_, (so_) + ecs, 0, 0);      \
  }
// I (the 1 / norm plane, into plane 0 of buffer `buf`) and G (kZmG one-dword DMA instructions into the wave's block) of the plane
// at plane offset so_; en_: the plane exists
#define PEA_ZMB_IG(buf, so_, v0_, v1_, en_)                                                                                      \
  {                                                                                                                              \
    /* (per-lane offsets from OPAQUE copies made HERE, afresh for every four requests: hoisted to the top of the plane or out */ \
    /*  of the march, or merely computed all at once ahead of the 24 requests, they would be kept -- spilled -- beside the    */ \
    /*  window) */                                                                                                               \
    unsigned pe_o = pe;                                                                                                          \
    int px_o = px, py_o = py;                                                                                                    \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(iB, (lds_ptr_t)(lds + (2 * (buf)) * PS + wbase), 16, v0_, so_, 0, 0);               \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(iB, (lds_ptr_t)(lds + (2 * (buf)) * PS + w1), 16, v1_, so_, 0, 0);                  \
    _Pragma("unroll") for (int k = 0; k < XP; ++k) {                                                                             \
      if (k % 4 == 0) asm volatile("" : "+v"(pe_o), "+v"(px_o));                                                                 \
      const unsigned pg = pe_o != kOOB ? pe_o : 0xC0000000u; /* dead lanes: out of range also with a displacement added */      \
      const int go = C.xgo[k];                                                                                                   \
      const bool out = (unsigned)(px_o + go) >= (unsigned)P.X;                                                                   \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(gB, (lds_ptr_t)(lds + GB + gwave + k * 256), 4,                                   \
                                               ((en_) && k < C.npx && !out) ? pg + (unsigned)(go * 4) : kOOB,                    \
                                               (so_) + (unsigned)C.xgi[k] * ecs, 0, 0);                                          \
    }                                                                                                                            \
    _Pragma("unroll") for (int k = 0; k < XP; ++k) {                                                                             \
      if (k % 4 == 0) asm volatile("" : "+v"(pe_o), "+v"(py_o));                                                                 \
      const unsigned pg = pe_o != kOOB ? pe_o : 0xC0000000u;                                                                     \
      const int go = C.ygo[k];                                                                                                   \
      const bool out = (unsigned)(py_o + go) >= (unsigned)P.Y;                                                                   \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(gB, (lds_ptr_t)(lds + GB + gwave + (XP + k) * 256), 4,                            \
                                               ((en_) && k < C.npy && !out) ? pg + (unsigned)(go * P.X * 4) : kOOB,              \
                                               (so_) + (unsigned)C.ygi[k] * ecs, 0, 0);                                          \
    }                                                                                                                            \
    asm volatile("" : "+v"(pe_o));                                                                                               \
    _Pragma("unroll") for (int s = 0; s < kZS; ++s) {                                                                            \
      const int ch = M.zch[s];                                                                                                   \
      const unsigned vz = ((en_) && ch >= 0) ? pe_o : kOOB;                                                                      \
      const unsigned sz = (so_) + (unsigned)(ch >= 0 ? ch : 0) * ecs;                                                            \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(gB, (lds_ptr_t)(lds + GB + gwave + (2 * XP + s) * 256), 4, vz, sz, 0, 0);         \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(aB, (lds_ptr_t)(lds + GB + gwave + (2 * XP + kZS + s) * 256), 4, vz, sz, 0, 0);   \
    }                                                                                                                            \
  }
// request item `it_` (0: I + G, else chunk it_ - 1) of the plane with step jo_ at plane offset so_
#define PEA_ZMB_ITEM(jo_, it_, so_, v0_, v1_, en_)                                                                   \
  {                                                                                                                  \
    if ((it_) == 0) PEA_ZMB_IG(PEA_ZMB_BUF(jo_, 0), so_, v0_, v1_, en_)                                              \
    else PEA_ZMB_CHUNK(PEA_ZMB_BUF(jo_, it_), (so_) + (unsigned)(2 * ((it_) - 1)) * ecs, v0_, v1_)                   \
  }
  const unsigned vh0 = here ? vo0 : kOOB, vh1 = here ? vo1 : kOOB;
  if (first) {  // the pipeline's head: items 0 .. NB - 1
#pragma unroll
    for (int m = 0; m < NB - 1; ++m) PEA_ZMB_ITEM(JO, m, ezo, vh0, vh1, here)
    PEA_ZM_WAIT(zm_bwd_wait(NB, 8));
    PEA_ZMB_ITEM(JO, NB - 1, ezo, vh0, vh1, here)
  }
  // ---- I-proc: coefficient of a pair = g * 1 / |e(q)|, two pairs to a register pair (pk_fma_c selects the half)
  constexpr int bI = PEA_ZMB_BUF(JO, 0);
  constexpr int boI = (bI % 2) * 2 * PS;
  int hiI = (bI / 2) * 4 * PS;
  if (bI >= 2) asm volatile("" : "+v"(hiI));
  const float invo = *(const float*)(lds + boI + (vown + hiI));
  const float inv_own = fabsf(invo);
  const char* gp = lds + GB + glane;
  f2 ccx[XP / 2], ccy[XP / 2];
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    const float vx = *(const float*)(gp + k * 256) * fabsf(*(const float*)(lds + boI + (ax[k] + hiI)));
    const float vy = *(const float*)(gp + (XP + k) * 256) * fabsf(*(const float*)(lds + boI + (ay[k] + hiI)));
    if (k & 1) { ccx[k / 2].y = vx; ccy[k / 2].y = vy; }
    else { ccx[k / 2].x = vx; ccy[k / 2].x = vy; }
  }
  float gz[kZS], az[kZS];
#pragma unroll
  for (int s = 0; s < kZS; ++s) {
    gz[s] = *(const float*)(gp + (2 * XP + s) * 256);
    az[s] = *(const float*)(gp + (2 * XP + kZS + s) * 256);
  }
#pragma unroll
  for (int k = 0; k < XP / 2; ++k) asm volatile("" : "+v"(ccx[k]), "+v"(ccy[k]));
  const f2 gz01 = {gz[0], gz[1]}, gz23 = {gz[2], gz[3]};
  PEA_ZM_WAIT(zm_bwd_wait(NB, 0));  // W_I: C0 has landed; the 1 / norm plane is dead: its buffer may be filled
  const unsigned vn0 = there ? vo0 : kOOB, vn1 = there ? vo1 : kOOB;
  PEA_ZMB_ITEM(JO, NB, ezo, vh0, vh1, here)
  // <ehat, G> of the pending planes: role B of plane z - d through offset d is g_d(z) a_d(z)
  float projF = prj[JO] + gz[kZS - 1] * az[kZS - 1];
  prj[ZmSlot<JO, 1>::v] = fmaf(gz[0], az[0], prj[ZmSlot<JO, 1>::v]);
  prj[ZmSlot<JO, 2>::v] = fmaf(gz[1], az[1], prj[ZmSlot<JO, 2>::v]);
  prj[ZmSlot<JO, 3>::v] = fmaf(gz[2], az[2], prj[ZmSlot<JO, 3>::v]);
  if (ivs[JO] < 0.f) projF = 0.f;  // clamp branch of F.normalize
  const float scF = fabsf(ivs[JO]) * dl;
  const unsigned pf = fin ? pe : kOOB;
  const unsigned fzo = (unsigned)(fin ? z - kZS : 0) * YX4;
  float pcur = 0.f;
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int buf = PEA_ZMB_BUF(JO, ps + 1);  // compile-time after unrolling
    const int bo = (buf % 2) * 2 * PS;
    int hi = (buf / 2) * 4 * PS;
    if (buf >= 2) asm volatile("" : "+v"(hi));
    f2 o;
    o.x = *(const float*)(lds + bo + (vown + hi));
    o.y = *(const float*)(lds + bo + PS + (vown + hi));
    o = o * inv_own;
    f2 acc = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      f2 v;
      v.x = *(const float*)(lds + bo + (ax[k] + hi));
      v.y = *(const float*)(lds + bo + PS + (ax[k] + hi));
      acc = (k & 1) ? pk_fma_c<true>(ccx[k / 2], v, acc) : pk_fma_c<false>(ccx[k / 2], v, acc);
      if (k % PEA_ZMV_BFENCE == PEA_ZMV_BFENCE - 1) asm volatile("" ::: "memory");  // bound the ds_read hoisting
    }
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      f2 v;
      v.x = *(const float*)(lds + bo + (ay[k] + hi));
      v.y = *(const float*)(lds + bo + PS + (ay[k] + hi));
      acc = (k & 1) ? pk_fma_c<true>(ccy[k / 2], v, acc) : pk_fma_c<false>(ccy[k / 2], v, acc);
      if (k % PEA_ZMV_BFENCE == PEA_ZMV_BFENCE - 1) asm volatile("" ::: "memory");
    }
    // z pairs, role A: the window (the slot of plane z - 4 is JO itself)
    acc = pk_fma_c<false>(gz01, Eh[ps][ZmSlot<JO, 1>::v], acc);
    acc = pk_fma_c<true>(gz01, Eh[ps][ZmSlot<JO, 2>::v], acc);
    acc = pk_fma_c<false>(gz23, Eh[ps][ZmSlot<JO, 3>::v], acc);
    acc = pk_fma_c<true>(gz23, Eh[ps][JO], acc);
    pcur = fmaf(o.x, acc.x, fmaf(o.y, acc.y, pcur));
    // role B: this plane's pixel into the pending planes
    Pd[ps][ZmSlot<JO, 1>::v] = pk_fma_c<false>(gz01, o, Pd[ps][ZmSlot<JO, 1>::v]);
    Pd[ps][ZmSlot<JO, 2>::v] = pk_fma_c<true>(gz01, o, Pd[ps][ZmSlot<JO, 2>::v]);
    Pd[ps][ZmSlot<JO, 3>::v] = pk_fma_c<false>(gz23, o, Pd[ps][ZmSlot<JO, 3>::v]);
    const f2 Gf = pk_fma_c<true>(gz23, o, Pd[ps][JO]);
    // plane z - 4 is complete: (G - ehat <ehat, G>) / n
    float vx = (Gf.x - Eh[ps][JO].x * projF) * scF, vy = (Gf.y - Eh[ps][JO].y * projF) * scF;
    Pd[ps][JO] = acc;
    Eh[ps][JO] = o;
    asm volatile("" : "+v"(Pd[ps][JO]), "+v"(Eh[ps][JO]), "+v"(pcur), "+v"(vx), "+v"(vy));  // the chunk's sums exist before its barrier
#if !PEA_ZMV_STORE_LATE
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vx), dB, pf, fzo + (unsigned)(2 * ps) * ecs, PEA_ZMV_STORE_AUX);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vy), dB, pf, fzo + (unsigned)(2 * ps + 1) * ecs, PEA_ZMV_STORE_AUX);
#endif
    // the next item has landed; everyone is done with this buffer; refill it with the item NB behind
    PEA_ZM_WAIT(zm_bwd_wait(NB, ps + 1));
    if (ps + 1 + NB < 9) PEA_ZMB_ITEM(JO, ps + 1 + NB, ezo, vh0, vh1, here)
    else PEA_ZMB_ITEM(JN, ps + 1 + NB - 9, ezn, vn0, vn1, there)
#if PEA_ZMV_STORE_LATE
    // (variant: the finished plane's two channels stored BEHIND the hand-off)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vx), dB, pf, fzo + (unsigned)(2 * ps) * ecs, PEA_ZMV_STORE_AUX);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vy), dB, pf, fzo + (unsigned)(2 * ps + 1) * ecs, PEA_ZMV_STORE_AUX);
#endif
  }
  prj[JO] = pcur;
  ivs[JO] = invo;
#undef PEA_ZMB_CHUNK
#undef PEA_ZMB_IG
#undef PEA_ZMB_ITEM
#undef PEA_ZMB_BUF
}

// xt: e [B, 16, S]; invp: its signed 1 / norm plane [B, S] (the forward's); gin: d loss / d affs [B, K, S]; affs: the RAW cosines
// [B, K, S] (only the z channels are read).  C: plan_xdma mode 0 with the tile walk set up by the host (zrun = nseg: xdma_tile's
// "z" is the segment).  Grid: tiles_per_xcd * 8 workgroups of TH * TW lanes, ONE per CU (launch bounds: 2 waves per SIMD).
template <int TH, int TW, int PSU, int NB>
__global__ __launch_bounds__(TH* TW, 2) void k_bwd_zm(const KParams P, const XParams C, const ZMParams M, const float* __restrict__ xt,
                                                       const float* __restrict__ invp, const float* __restrict__ gin,
                                                       const float* __restrict__ affs, const float* __restrict__ dloss,
                                                       float* __restrict__ dx) {
  constexpr int NT = TH * TW, D_T = 16;
  static_assert(TW == 32, "lane mapping");
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  int tile, b, seg, y0, x0;
  if (!march_tile<TH, TW>(C, P, tile, b, seg, y0, x0)) return;
  const int zb = seg * M.zseg, ze = min(zb + M.zseg, P.Z);
  const size_t S = (size_t)P.S;
  const unsigned YX4 = (unsigned)(P.Y * P.X) * 4u, ecs = (unsigned)P.S * 4u;
  const rsrc_t xB = mkbuf(xt + (size_t)b * D_T * S), dB = mkbuf(dx + (size_t)b * D_T * S);
  const rsrc_t gB = mkbuf(gin + (size_t)b * P.K * S), aB = mkbuf(affs + (size_t)b * P.K * S), iB = mkbuf(invp + (size_t)b * S);
  const float dl = dloss ? dloss[0] : 1.f;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
  const int py = y0 + ly, px = x0 + lx;
  const bool live = py < P.Y && px < P.X;
  const unsigned pe = live ? (unsigned)(py * P.X + px) * 4u : kOOB;
  // the two quads this lane moves per plane (pea_xdma.h; unconditional DMA: a wave without a second block repeats its first)
  unsigned vo[2];
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
    bool oky, okx;
    gy = wrap1<true>(gy, P.Y, oky);
    gx = wrap1<true>(gx, P.X, okx);
    vo[s] = (q < C.QA && oky && okx) ? (unsigned)(gy * P.X + gx) * 4u : kOOB;
  }
  const int wbase = wave * 1024;
  const bool two = __builtin_amdgcn_readfirstlane(((NT / 64) + wave) * 64 < C.QA);
  const unsigned vo0 = vo[0], vo1 = two ? vo[1] : vo[0];
  const int w1 = two ? wbase + (NT / 64) * 1024 : wbase;
  // LDS slot of every in-plane neighbour
  int ax[8], ay[8];
  const int vown = ((C.hy0 + ly) * TW + lx) * 4;
  const int hrow = (C.QV * 4 + ly * C.SW) * 4;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int d = C.xd[k], c = lx + d;
    ax[k] = (unsigned)c < (unsigned)TW ? vown + d * 4 : hrow + (c & C.xm[k]) * 4;
    ay[k] = vown + C.yd[k] * TW * 4;
  }
  const int gwave = wave * (kZmG * 256), glane = gwave + lane * 4;  // the wave's block of prefetched g / a values: [value][lane]
  f2 Eh[8][kZS], Pd[8][kZS];
  float prj[kZS], ivs[kZS];
#pragma unroll
  for (int j = 0; j < kZS; ++j) {
    prj[j] = 0.f;
    ivs[j] = 0.f;
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) { Eh[ps][j] = (f2){0.f, 0.f}; Pd[ps][j] = (f2){0.f, 0.f}; }
  }
  // Every step is a staged plane, four steps to a turn with NO branch between them (any other control flow between the steps made
  // the register allocator copy the window at the joins: 44 - 426 spilled registers).  The steps run from zs = max(zb - 4, 0) --
  // the segment's warm-up planes, staged like any other, their gradients not stored -- to at least ze + 3: planes ze .. ze + 3 are
  // the contributors (role B into the segment's last planes) and, where they lie beyond the volume (every request out of range:
  // zeros), the drain; the step count is rounded up to a multiple of four with more such empty planes.
  const int zs = max(zb - kZS, 0);
  const int zt = zs + (ze + kZS - zs + 3) / 4 * 4;
#define PEA_ZM_FULL(JO_, zz)                                                                                                     \
  zm_bwd_full<JO_, TH, TW, PSU, NB>(P, C, M, lds, xB, iB, gB, aB, dB, (zz), (zz) == zs, (zz) + 1 < zt, (zz) - kZS >= zb && (zz) - kZS < ze, \
                                dl, ecs, YX4, vo0, vo1, wbase, w1, gwave, glane, pe, px, py, vown, ax, ay, Eh, Pd, prj, ivs);
  for (int z = zs; z < zt; z += 4) {
    PEA_ZM_FULL(0, z)
    PEA_ZM_FULL(1, z + 1)
    PEA_ZM_FULL(2, z + 2)
    PEA_ZM_FULL(3, z + 3)
  }
#undef PEA_ZM_FULL
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last plane's look-ahead DMA (out-of-range: zeros) has drained
}

// ------------------------------------------------------------------------------------------------------------------
// forward (training: affs, g, loss, the 1 / norm plane; inference: affs)
// ------------------------------------------------------------------------------------------------------------------
// LDS: a ring of EIGHT two-channel buffers (16 planes of PS bytes = one plane's eight chunks: chunk c lives in buffer c in every
// plane) that never stops: while chunk c of plane z is gathered, the seven chunks behind it are in flight -- chunks c + 1 .. 7 of
// this plane and 0 .. c - 1 of the NEXT.  (One workgroup per CU has nobody to hide its latency behind: what it keeps in flight is
// what it gets.  A ring of four buffers -- 48 KB in flight -- staged the sub-volume's 6 GB at 4.9 TB/s = 1.15 ms before a single
// output byte; PMC: waves parked 49 % of their cycles, TA 55 % busy.)  The K dot products are parked in their own [K][tile pixel]
// array behind the ring.  The upper four buffers lie beyond the 64 KB a ds_read immediate reaches: their addresses are formed per
// chunk (one add per read; hoisted out of the march they would cost eleven registers beside the window).
// Every staged plane issues the SAME sequence of LOAD instructions per wave (absent operands and the plane after the last are
// requested at an out-of-range offset), so every hand-off is a literal s_waitcnt vmcnt(N), N = the loads that may still fly behind
// the chunk waited for (stores sit on the same counter but may retire before older loads: they are NOT counted, zm_bwd_wait):
//     T  : 3 * ITEMS loads (target, weight, mask quads of the plane's items)
//     c  : gather chunk c; wait W_c for chunk c + 1; barrier; 4 DMA instructions: chunk c of the NEXT plane into the same buffer
//     E  : 1 + 2 * ITEMS stores (the 1 / norm plane, affs and g quads)
// W_c, c <= 6, waits for DMA issued in the previous plane: behind it are the chunks c + 2 .. 7 of this plane, T (9) and the chunks
// 0 .. c - 1 of the next: six chunks (24) + 9 -> vmcnt(33); W_7 (chunk 0 of the next plane): six chunks -> vmcnt(24).
// The first plane of a segment issues its eight chunks itself.
// NXP: in-plane offsets the gather walks (8: the norm5 / norm1 tables; 10: every table the cross kernels take) -- an unused slot
// still costs its LDS reads
template <int TH, int TW, int PSU, bool TRAIN, int NXP>
__global__ __launch_bounds__(TH* TW, 2) void k_fwd_zm(const KParams P, const XParams C, const ZMParams M, const float* __restrict__ e,
                                                       const float* __restrict__ target, const float* __restrict__ weight,
                                                       const uint8_t* __restrict__ mask, float* __restrict__ affs,
                                                       float* __restrict__ gout, LossState* __restrict__ st,
                                                       float* __restrict__ inv_out) {
  constexpr int D_T = 16, NT = TH * TW, PS = PSU * 256, NP = D_T / 2, TP = NT, QP = TP / 4, NSL = QP / 64;
  constexpr int KMAX = kXP + 2, ITEMS = (KMAX * QP + NT - 1) / NT, RING = 16 * PS;
  constexpr int NLD = 3 * ITEMS;  // T
  static_assert(TW == 32 && QP % 64 == 0, "lane mapping");
  static_assert(KMAX <= kXK && 8 * PS == 65536, "the lower four buffers end where the ds_read immediate does");
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  float* sA = (float*)(lds + RING);                      // [K][TP] dot products of the plane
  float* s_part = (float*)(lds + RING + KMAX * TP * 4);  // [K][NSL]
  int tile, b, seg, y0, x0;
  if (!march_tile<TH, TW>(C, P, tile, b, seg, y0, x0)) return;
  const int zb = seg * M.zseg, ze = min(zb + M.zseg, P.Z);
  const size_t S = (size_t)P.S;
  const unsigned YX4 = (unsigned)(P.Y * P.X) * 4u, ecs = (unsigned)P.S * 4u;
  const rsrc_t xB = mkbuf(e + (size_t)b * D_T * S);
  const rsrc_t aB = mkbuf(affs ? affs + (size_t)b * P.K * S : nullptr), gB = mkbuf(gout ? gout + (size_t)b * P.K * S : nullptr);
  const rsrc_t tB = mkbuf(TRAIN ? target + (size_t)b * P.tbs : nullptr), wB = mkbuf(TRAIN ? weight + (size_t)b * P.wbs : nullptr);
  const rsrc_t mB = mkbuf(mask ? mask + (size_t)b * P.mbs : nullptr);
  const rsrc_t iB = mkbuf(inv_out ? inv_out + (size_t)b * S : nullptr);
  const bool has_a = affs != nullptr, has_g = gout != nullptr, has_m = mask != nullptr, has_i = inv_out != nullptr;
  const unsigned af = P.flags & kActMask;
  const float inv_eps = 1.0f / P.eps;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;

  // ---- the epilogue's items: (offset, quad of 4 x-adjacent tile pixels); the same for every plane
  bool ion[ITEMS];
  unsigned ivo[ITEMS];
  int iqd[ITEMS], igy[ITEMS], igx[ITEMS], isl[ITEMS];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int tt = it * NT + (int)threadIdx.x;
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
  const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
  const int py = y0 + ly, px = x0 + lx;
  const bool live = py < P.Y && px < P.X;
  const unsigned pe = live ? (unsigned)(py * P.X + px) * 4u : kOOB;
  unsigned vo[2];
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
    bool oky, okx;
    gy = wrap1<true>(gy, P.Y, oky);
    gx = wrap1<true>(gx, P.X, okx);
    vo[s] = (q < C.QA && oky && okx) ? (unsigned)(gy * P.X + gx) * 4u : kOOB;
  }
  const int wbase = wave * 1024;
  const bool two = __builtin_amdgcn_readfirstlane(((NT / 64) + wave) * 64 < C.QA);
  const unsigned vo0 = vo[0], vo1 = two ? vo[1] : vo[0];
  const int w1 = two ? wbase + (NT / 64) * 1024 : wbase;
  int an[NXP];
  const int vown = ((C.hy0 + ly) * TW + lx) * 4;
  const int hrow = (C.QV * 4 + ly * C.SW) * 4;
#pragma unroll
  for (int k = 0; k < NXP; ++k) {
    const int d = C.fd[k], c = lx + d;
    const int a_x = (unsigned)c < (unsigned)TW ? vown + d * 4 : hrow + (c & C.fm[k]) * 4;
    an[k] = C.fax[k] ? a_x : vown + d * TW * 4;  // unused offsets: d = 0, the own slot
  }
  // the window: raw own pixel of the last four planes (slot = step mod 4) and their 1 / norm
  f2 W[NP][kZS];
  float iw[kZS];
#pragma unroll
  for (int j = 0; j < kZS; ++j) {
    iw[j] = 0.f;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) W[ps][j] = (f2){0.f, 0.f};
  }
  // the loss: lane k < K of wave 0 sums the workgroup's partials of offset k over the planes it marches through (a fixed order:
  // bit-reproducible) and adds the sum to the integer accumulators ONCE at the end -- per plane, the 128-bit conversion and the
  // atomics made wave 0 late for the next plane's first barrier, and reading the partials cost a barrier of its own
  const bool lane_k = wave == 0 && (int)threadIdx.x < P.K;
  float lacc = 0.f;
// the DMA of one chunk (two planes) into ring buffer `buf`, from plane offset `so_`, with the given per-lane offsets
#define PEA_ZMF_CHUNK(buf, so_, v0_, v1_)                                                                                       \
  {                                                                                                                              \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(lds + (2 * (buf)) * PS + wbase), 16, v0_, so_, 0, 0);               \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(lds + (2 * (buf)) * PS + w1), 16, v1_, so_, 0, 0);                  \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(lds + (2 * (buf) + 1) * PS + wbase), 16, v0_, (so_) + ecs, 0, 0);   \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(lds + (2 * (buf) + 1) * PS + w1), 16, v1_, (so_) + ecs, 0, 0);      \
  }

#define PEA_ZMF_STEP(JO, zz)                                                                                                   \
  {                                                                                                                            \
    const int z = (zz);                                                                                                        \
    if (z < ze) {                                                                                                              \
      if (z < zb) {                                                                                                            \
        if (z >= 0) { /* warm-up: the own pixel of a plane below the segment */                                               \
          const unsigned ezo = (unsigned)z * YX4;                                                                              \
          f2 ss = {0.f, 0.f};                                                                                                  \
          _Pragma("unroll") for (int ps = 0; ps < NP; ++ps) {                                                                  \
            W[ps][JO].x = bl32(xB, pe, ezo + (unsigned)(2 * ps) * ecs);                                                        \
            W[ps][JO].y = bl32(xB, pe, ezo + (unsigned)(2 * ps + 1) * ecs);                                                    \
          }                                                                                                                    \
          _Pragma("unroll") for (int ps = 0; ps < NP; ++ps) ss = __builtin_elementwise_fma(W[ps][JO], W[ps][JO], ss);          \
          iw[JO] = rnorm(ss.x + ss.y, inv_eps);                                                                                \
        }                                                                                                                      \
      } else {                                                                                                                 \
        const unsigned ezo = (unsigned)z * YX4;                                                                                \
        const bool first = z == zb, nxt = z + 1 < ze;                                                                          \
        const unsigned ezn = (unsigned)(nxt ? z + 1 : z) * YX4;                                                                \
        const unsigned vn0 = nxt ? vo0 : kOOB, vn1 = nxt ? vo1 : kOOB;                                                         \
        if (first) { /* the pipeline's head: the plane's eight chunks */                                                       \
          _Pragma("unroll") for (int c = 0; c < NP; ++c) PEA_ZMF_CHUNK(c, ezo + (unsigned)(2 * c) * ecs, vo0, vo1)             \
        }                                                                                                                      \
        /* T: target / weight / mask of the plane's items: they land while the chunks are gathered */                        \
        f4 t4[ITEMS], w4[ITEMS];                                                                                               \
        unsigned m4[ITEMS];                                                                                                    \
        _Pragma("unroll") for (int it = 0; it < ITEMS; ++it) {                                                                 \
          const unsigned so = ezo + (unsigned)isl[it] * ecs;                                                                   \
          const unsigned vt = TRAIN ? ivo[it] : kOOB;                                                                          \
          t4[it] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(tB, vt, so, kAuxNT));                          \
          w4[it] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(wB, vt, so, kAuxNT));                          \
          const unsigned mm = __builtin_amdgcn_raw_buffer_load_b32(mB, (!has_m || vt == kOOB) ? kOOB : vt >> 2,                \
                                                                   (ezo >> 2) + (unsigned)isl[it] * (unsigned)P.S, kAuxNT);    \
          m4[it] = has_m ? mm : 0x01010101u;                                                                                   \
        }                                                                                                                      \
        if (first) PEA_ZM_WAIT(28 + NLD); /* chunk 0 has landed (chunks 1 - 7 and T may fly) */                                \
        f2 dot[NXP], ssq[NXP], dz[kZS], oss = {0.f, 0.f};                                                                      \
        _Pragma("unroll") for (int k = 0; k < NXP; ++k) { dot[k] = (f2){0.f, 0.f}; ssq[k] = (f2){0.f, 0.f}; }                  \
        _Pragma("unroll") for (int k = 0; k < kZS; ++k) dz[k] = (f2){0.f, 0.f};                                                \
        _Pragma("unroll") for (int ps = 0; ps < NP; ++ps) {                                                                    \
          const int bo = (ps % 4) * 2 * PS;                                                                                    \
          int hi = ps >= 4 ? 8 * PS : 0; /* the upper four buffers: beyond the immediate's reach */                           \
          if (ps >= 4) asm volatile("" : "+v"(hi));                                                                            \
          f2 o;                                                                                                                \
          o.x = *(const float*)(lds + bo + (vown + hi));                                                                       \
          o.y = *(const float*)(lds + bo + PS + (vown + hi));                                                                  \
          oss = __builtin_elementwise_fma(o, o, oss);                                                                          \
          _Pragma("unroll") for (int k = 0; k < NXP; ++k) {                                                                    \
            f2 v;                                                                                                              \
            v.x = *(const float*)(lds + bo + (an[k] + hi));                                                                    \
            v.y = *(const float*)(lds + bo + PS + (an[k] + hi));                                                               \
            dot[k] = __builtin_elementwise_fma(o, v, dot[k]);                                                                  \
            ssq[k] = __builtin_elementwise_fma(v, v, ssq[k]);                                                                  \
            if (PEA_ZMV_FFENCE && k % (PEA_ZMV_FFENCE ? PEA_ZMV_FFENCE : 1) == (PEA_ZMV_FFENCE ? PEA_ZMV_FFENCE : 1) - 1)      \
              asm volatile("" ::: "memory");                                                                                   \
          }                                                                                                                    \
          _Pragma("unroll") for (int j = 0; j < kZS; ++j) dz[j] = __builtin_elementwise_fma(o, W[ps][j], dz[j]);               \
          W[ps][JO] = o;                                                                                                       \
          _Pragma("unroll") for (int k = 0; k < NXP; ++k) asm volatile("" : "+v"(dot[k]), "+v"(ssq[k]));                       \
          _Pragma("unroll") for (int j = 0; j < kZS; ++j) asm volatile("" : "+v"(dz[j]));                                      \
          asm volatile("" : "+v"(oss), "+v"(W[ps][JO]));                                                                       \
          /* the next chunk has landed (W_c); everyone is done with this buffer; refill it with the next plane's chunk */   \
          if (ps < 7) PEA_ZM_WAIT(24 + NLD);                                                                                   \
          else PEA_ZM_WAIT(24);                                                                                                \
          PEA_ZMF_CHUNK(ps, ezn + (unsigned)(2 * ps) * ecs, vn0, vn1)                                                          \
          if (TRAIN && ps == 0 && !first && lane_k) { /* the PREVIOUS plane's loss partials: behind this barrier they are all */ \
            _Pragma("unroll") for (int s = 0; s < NSL; ++s) lacc += s_part[threadIdx.x * NSL + s]; /* written (no barrier of */ \
          }                                                                                        /* their own)            */ \
        }                                                                                                                      \
        const float osum = oss.x + oss.y;                                                                                      \
        const float inv_own = rnorm(osum, inv_eps);                                                                            \
        bs32(iB, osum < P.eps * P.eps ? -inv_own : inv_own, has_i ? pe : kOOB, ezo); /* E: the 1 / norm plane */                \
        _Pragma("unroll") for (int k = 0; k < NXP; ++k) {                                                                      \
          if (k < C.nf) {                                                                                                      \
            float a = (dot[k].x + dot[k].y) * inv_own * rnorm(ssq[k].x + ssq[k].y, inv_eps);                                   \
            const int q = (C.fax[k] ? px : py) + C.fd[k];                                                                      \
            a = (unsigned)q < (unsigned)(C.fax[k] ? P.X : P.Y) ? a : 0.f;                                                      \
            sA[C.fi[k] * TP + (int)threadIdx.x] = a;                                                                           \
          }                                                                                                                    \
        }                                                                                                                      \
        _Pragma("unroll") for (int s = 1; s <= kZS; ++s) {                                                                     \
          const int ch = M.zch[s - 1];                                                                                         \
          if (ch >= 0) {                                                                                                       \
            const int jj = ((JO - s) % kZS + kZS) % kZS;                                                                       \
            const float a = z - s >= 0 ? (dz[jj].x + dz[jj].y) * inv_own * iw[jj] : 0.f;                                       \
            sA[ch * TP + (int)threadIdx.x] = a;                                                                                \
          }                                                                                                                    \
        }                                                                                                                      \
        iw[JO] = inv_own;                                                                                                      \
        lds_barrier();                                                                                                         \
        _Pragma("unroll") for (int it = 0; it < ITEMS; ++it) { /* E: two stores per item, issued whether or not it is live */ \
          const int sl = isl[it];                                                                                              \
          const f4 a4 = *(const f4*)(sA + sl * TP + iqd[it] * 4);                                                              \
          const unsigned so = ezo + (unsigned)sl * ecs;                                                                        \
          f4 o4 = a4;                                                                                                          \
          if (af) { o4.x = act_affs(o4.x, af); o4.y = act_affs(o4.y, af); o4.z = act_affs(o4.z, af); o4.w = act_affs(o4.w, af); } \
          bs128<true>(aB, o4, (has_a && ion[it]) ? ivo[it] : kOOB, so);                                                        \
          float acc = 0.f;                                                                                                     \
          f4 g4;                                                                                                               \
          const float gs = C.gs[sl];                                                                                           \
          const int ax_ = C.oax[sl], od_ = C.od[sl];                                                                           \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                      \
            const float m = (float)((m4[it] >> (8 * j)) & 0xffu);                                                              \
            const float r = a4[j] * m - t4[it][j] * m;                                                                         \
            float wr = w4[it][j] * r;                                                                                          \
            const int q = (ax_ == 1 ? igx[it] + j : ax_ == 0 ? igy[it] : z) + od_;                                             \
            wr = (unsigned)q < (unsigned)(ax_ == 1 ? P.X : ax_ == 0 ? P.Y : P.Z) ? wr : 0.f;                                   \
            g4[j] = gs * wr * m;                                                                                               \
            acc = fmaf(wr, r, acc);                                                                                            \
          }                                                                                                                    \
          bs128<false>(gB, g4, (TRAIN && has_g && ion[it]) ? ivo[it] : kOOB, so);                                              \
          if (TRAIN && ion[it]) {                                                                                              \
            const float red = wave_sum63(acc);                                                                                 \
            if ((threadIdx.x & 63) == 63) s_part[sl * NSL + (iqd[it] >> 6)] = red;                                             \
          }                                                                                                                    \
        }                                                                                                                      \
      }                                                                                                                        \
    }                                                                                                                          \
  }
  for (int zq = zb - kZS; zq < ze; zq += kZS) {
    PEA_ZMF_STEP(0, zq)
    PEA_ZMF_STEP(1, zq + 1)
    PEA_ZMF_STEP(2, zq + 2)
    PEA_ZMF_STEP(3, zq + 3)
  }
#undef PEA_ZMF_STEP
#undef PEA_ZMF_CHUNK
  if (TRAIN) {
    lds_barrier();  // the last plane's partials
    if (lane_k) {
#pragma unroll
      for (int s = 0; s < NSL; ++s) lacc += s_part[threadIdx.x * NSL + s];
      loss_accumulate(st, tile, threadIdx.x, lacc);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last plane's look-ahead DMA (out-of-range: zeros) has drained
}

#undef PEA_ZM_DMA
#undef PEA_ZM_WAIT

// host: which channel holds the offset (-s, 0, 0); false: a z offset the march does not cover (positive, longer than kZS, twice)
inline bool plan_zmarch(const KParams& P, ZMParams* M) {
  if (P.border != PEA_BORDER_CROP_ZERO || P.D != 16 || P.Z < 2) return false;
  bool any = false;
  for (int s = 0; s < kZS; ++s) M->zch[s] = -1;
  for (int i = 0; i < P.K; ++i) {
    const int oz = P.off[i][0];
    if (oz == 0) continue;
    if (P.off[i][1] != 0 || P.off[i][2] != 0 || oz > 0 || oz < -kZS || oz <= -P.Z) return false;
    if (M->zch[-oz - 1] >= 0) return false;
    M->zch[-oz - 1] = i;
    any = true;
  }
  return any;
}

}  // namespace pea
