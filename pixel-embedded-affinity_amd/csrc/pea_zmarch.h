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

// one staged plane z of the segment: in-plane pairs from the LDS cross, z pairs from the window; finishes plane z - 4
template <int JO, int TH, int TW, int PSU>
__device__ __forceinline__ void zm_bwd_full(const KParams& P, const XParams& C, const ZMParams& M, char* lds, const rsrc_t xB,
                                            const rsrc_t iB, const rsrc_t gB, const rsrc_t aB, const rsrc_t dB, const int z,
                                            const bool fin, const float dl, const unsigned ecs, const unsigned YX4, const unsigned vo0,
                                            const unsigned vo1, const int wbase, const int w1, const unsigned pe, const int px,
                                            const int py, const int vown, const int (&ax)[8], const int (&ay)[8],
                                            f2 (&Eh)[8][kZS], f2 (&Pd)[8][kZS], float (&prj)[kZS], float (&ivs)[kZS]) {
  constexpr int PS = PSU * 256, NP = 8, XP = 8;
  const unsigned ezo = (unsigned)z * YX4;
  lds_barrier();  // every wave is done with the previous plane's last chunk: the ring is free
  PEA_ZM_DMA(iB, 4 * PS, ezo)
  PEA_ZM_DMA(xB, 0, ezo)
  PEA_ZM_DMA(xB, PS, ezo + ecs)
  // ---- g of every in-plane pair (role A at p, role B at p - o), of the z offsets (at p: both roles), and their raw cosines
  // (the offsets are the same for every plane; derived from OPAQUE copies so that the compiler does not hoist all sixteen of them
  //  out of the march and keep -- spill -- them: a few integer operations per load instead)
  unsigned pe_o = pe;
  int px_o = px, py_o = py;
  asm volatile("" : "+v"(pe_o), "+v"(px_o), "+v"(py_o));
  const unsigned pg = pe_o != kOOB ? pe_o : 0xC0000000u;  // dead lanes: stays out of range when a small displacement is added
  float cx[XP], cy[XP];
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    const int go = C.xgo[k];
    const bool out = (unsigned)(px_o + go) >= (unsigned)P.X;
    cx[k] = bl32(gB, (k < C.npx && !out) ? pg + (unsigned)(go * 4) : kOOB, ezo + (unsigned)C.xgi[k] * ecs);
  }
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    const int go = C.ygo[k];
    const bool out = (unsigned)(py_o + go) >= (unsigned)P.Y;
    cy[k] = bl32(gB, (k < C.npy && !out) ? pg + (unsigned)(go * P.X * 4) : kOOB, ezo + (unsigned)C.ygi[k] * ecs);
  }
  float gz[kZS], az[kZS];
#pragma unroll
  for (int s = 0; s < kZS; ++s) {
    const int ch = M.zch[s];
    const unsigned vz = ch >= 0 ? pe_o : kOOB;
    const unsigned so = ezo + (unsigned)(ch >= 0 ? ch : 0) * ecs;
    gz[s] = bl32(gB, vz, so);
    az[s] = bl32(aB, vz, so);
  }
  PEA_ZM_DMA(xB, 2 * PS, ezo + 2u * ecs)
  PEA_ZM_DMA(xB, 3 * PS, ezo + 3u * ecs)
  PEA_ZM_WAIT(4);  // inv, chunk 0, g, a have landed (chunk 1's four DMA instructions may still fly)
  const float invo = *(const float*)(lds + 4 * PS + vown);
  const float inv_own = fabsf(invo);
  // coefficient of a pair = g * 1 / |e(q)|, two pairs to a register pair (pk_fma_c selects the half: hipcc would otherwise keep
  // every coefficient twice, as (f2){c, c})
  f2 ccx[XP / 2], ccy[XP / 2];
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    const float vx = cx[k] * fabsf(*(const float*)(lds + 4 * PS + ax[k]));
    const float vy = cy[k] * fabsf(*(const float*)(lds + 4 * PS + ay[k]));
    if (k & 1) { ccx[k / 2].y = vx; ccy[k / 2].y = vy; }
    else { ccx[k / 2].x = vx; ccy[k / 2].x = vy; }
  }
#pragma unroll
  for (int k = 0; k < XP / 2; ++k) asm volatile("" : "+v"(ccx[k]), "+v"(ccy[k]));
  const f2 gz01 = {gz[0], gz[1]}, gz23 = {gz[2], gz[3]};
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the inv plane is dead: buffer 2 may be filled
  PEA_ZM_DMA(xB, 4 * PS, ezo + 4u * ecs)
  PEA_ZM_DMA(xB, 5 * PS, ezo + 5u * ecs)
  // <ehat, G> of the pending planes: role B of plane z - d through offset d is g_d(z) a_d(z)
  float projF = prj[JO] + gz[kZS - 1] * az[kZS - 1];
  prj[ZmSlot<JO, 1>::v] = fmaf(gz[0], az[0], prj[ZmSlot<JO, 1>::v]);
  prj[ZmSlot<JO, 2>::v] = fmaf(gz[1], az[1], prj[ZmSlot<JO, 2>::v]);
  prj[ZmSlot<JO, 3>::v] = fmaf(gz[2], az[2], prj[ZmSlot<JO, 3>::v]);
  if (ivs[JO] < 0.f) projF = 0.f;  // clamp branch of F.normalize
  const float scF = fabsf(ivs[JO]) * dl;
  const unsigned pf = fin ? pe_o : kOOB;
  const unsigned fzo = (unsigned)(fin ? z - kZS : 0) * YX4;
  float pcur = 0.f;
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int bo = (ps % 3) * 2 * PS;
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
      acc = (k & 1) ? pk_fma_c<true>(ccx[k / 2], v, acc) : pk_fma_c<false>(ccx[k / 2], v, acc);
      if (k % 2 == 1) asm volatile("" ::: "memory");  // bound the ds_read hoisting
    }
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      f2 v;
      v.x = *(const float*)(lds + bo + ay[k]);
      v.y = *(const float*)(lds + bo + PS + ay[k]);
      acc = (k & 1) ? pk_fma_c<true>(ccy[k / 2], v, acc) : pk_fma_c<false>(ccy[k / 2], v, acc);
      if (k % 2 == 1) asm volatile("" ::: "memory");
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
    const float vx = (Gf.x - Eh[ps][JO].x * projF) * scF, vy = (Gf.y - Eh[ps][JO].y * projF) * scF;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vx), dB, pf, fzo + (unsigned)(2 * ps) * ecs, kAuxNT);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vy), dB, pf, fzo + (unsigned)(2 * ps + 1) * ecs, kAuxNT);
    Pd[ps][JO] = acc;
    Eh[ps][JO] = o;
    asm volatile("" : "+v"(Pd[ps][JO]), "+v"(Eh[ps][JO]), "+v"(pcur));  // the chunk's sums exist before its barrier
    if (ps + 1 < NP) {
      // chunk ps + 1 has landed; what may still fly: the DMA of chunk ps + 2 (4) and this chunk's two stores
      if (ps + 2 < NP) PEA_ZM_WAIT(6);
      else PEA_ZM_WAIT(2);
      if (ps + 3 < NP) {
        PEA_ZM_DMA(xB, bo, ezo + (unsigned)(2 * ps + 6) * ecs)
        PEA_ZM_DMA(xB, bo + PS, ezo + (unsigned)(2 * ps + 7) * ecs)
      }
    }
  }
  prj[JO] = pcur;
  ivs[JO] = invo;
}

// a plane that is not staged: warm-up (z < zb: fills the window), contributor (ze <= z < Z: role B into the pending planes) or
// drain (z >= Z: nothing to add); finishes plane z - 4 where that is a plane of the segment.  The own pixel is taken in two halves
// of eight channels (a fence between them): sixteen loaded values on top of the window do not fit the register file.
template <int JO>
__device__ __forceinline__ void zm_bwd_light(const KParams& P, const ZMParams& M, const rsrc_t xB, const rsrc_t iB, const rsrc_t gB,
                                             const rsrc_t aB, const rsrc_t dB, const int z, const int zb, const int ze, const float dl,
                                             const unsigned ecs, const unsigned YX4, const unsigned pe, f2 (&Eh)[8][kZS],
                                             f2 (&Pd)[8][kZS], float (&prj)[kZS], float (&ivs)[kZS]) {
  constexpr int NP = 8, HP = NP / 2;
  const bool exists = z >= 0 && z < P.Z;  // uniform
  unsigned pe_o = pe;
  asm volatile("" : "+v"(pe_o));
  const unsigned pz = exists ? pe_o : kOOB;
  const unsigned ezo = (unsigned)(exists ? z : 0) * YX4;
  const float invo = bl32(iB, pz, ezo);
  const float inv_own = fabsf(invo);
  if (z < zb) {  // warm-up: the window only
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f2 o[HP];
#pragma unroll
      for (int q = 0; q < HP; ++q) {
        o[q].x = bl32(xB, pz, ezo + (unsigned)(2 * (h * HP + q)) * ecs);
        o[q].y = bl32(xB, pz, ezo + (unsigned)(2 * (h * HP + q) + 1) * ecs);
      }
#pragma unroll
      for (int q = 0; q < HP; ++q) {
        Eh[h * HP + q][JO] = o[q] * inv_own;
        Pd[h * HP + q][JO] = (f2){0.f, 0.f};
      }
      asm volatile("" ::: "memory");
    }
    prj[JO] = 0.f;
    ivs[JO] = invo;
    return;
  }
  float gz[kZS], az[kZS];
#pragma unroll
  for (int s = 0; s < kZS; ++s) {
    const int ch = M.zch[s];
    const unsigned vz = ch >= 0 ? pz : kOOB;
    const unsigned so = ezo + (unsigned)(ch >= 0 ? ch : 0) * ecs;
    gz[s] = bl32(gB, vz, so);
    az[s] = bl32(aB, vz, so);
  }
  float projF = prj[JO] + gz[kZS - 1] * az[kZS - 1];
  prj[ZmSlot<JO, 1>::v] = fmaf(gz[0], az[0], prj[ZmSlot<JO, 1>::v]);
  prj[ZmSlot<JO, 2>::v] = fmaf(gz[1], az[1], prj[ZmSlot<JO, 2>::v]);
  prj[ZmSlot<JO, 3>::v] = fmaf(gz[2], az[2], prj[ZmSlot<JO, 3>::v]);
  if (ivs[JO] < 0.f) projF = 0.f;
  const float scF = fabsf(ivs[JO]) * dl;
  const f2 gz01 = {gz[0], gz[1]}, gz23 = {gz[2], gz[3]};
  const int zf = z - kZS;
  const bool fin = zf >= zb && zf < ze;
  const unsigned pf = fin ? pe_o : kOOB;
  const unsigned fzo = (unsigned)(fin ? zf : 0) * YX4;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f2 o[HP];
#pragma unroll
    for (int q = 0; q < HP; ++q) {
      o[q].x = bl32(xB, pz, ezo + (unsigned)(2 * (h * HP + q)) * ecs);
      o[q].y = bl32(xB, pz, ezo + (unsigned)(2 * (h * HP + q) + 1) * ecs);
    }
#pragma unroll
    for (int q = 0; q < HP; ++q) {
      const int ps = h * HP + q;
      const f2 oh = o[q] * inv_own;
      Pd[ps][ZmSlot<JO, 1>::v] = pk_fma_c<false>(gz01, oh, Pd[ps][ZmSlot<JO, 1>::v]);
      Pd[ps][ZmSlot<JO, 2>::v] = pk_fma_c<true>(gz01, oh, Pd[ps][ZmSlot<JO, 2>::v]);
      Pd[ps][ZmSlot<JO, 3>::v] = pk_fma_c<false>(gz23, oh, Pd[ps][ZmSlot<JO, 3>::v]);
      const f2 Gf = pk_fma_c<true>(gz23, oh, Pd[ps][JO]);
      const float vx = (Gf.x - Eh[ps][JO].x * projF) * scF, vy = (Gf.y - Eh[ps][JO].y * projF) * scF;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vx), dB, pf, fzo + (unsigned)(2 * ps) * ecs, kAuxNT);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vy), dB, pf, fzo + (unsigned)(2 * ps + 1) * ecs, kAuxNT);
      Pd[ps][JO] = (f2){0.f, 0.f};
      Eh[ps][JO] = oh;
    }
    asm volatile("" ::: "memory");
  }
  prj[JO] = 0.f;
  ivs[JO] = invo;
}

// xt: e [B, 16, S]; invp: its signed 1 / norm plane [B, S] (the forward's); gin: d loss / d affs [B, K, S]; affs: the RAW cosines
// [B, K, S] (only the z channels are read).  C: plan_xdma mode 0 with the tile walk set up by the host (zrun = nseg: xdma_tile's
// "z" is the segment).  Grid: tiles_per_xcd * 8 workgroups of TH * TW lanes, ONE per CU (launch bounds: 2 waves per SIMD).
template <int TH, int TW, int PSU>
__global__ __launch_bounds__(TH* TW, 2) void k_bwd_zm(const KParams P, const XParams C, const ZMParams M, const float* __restrict__ xt,
                                                       const float* __restrict__ invp, const float* __restrict__ gin,
                                                       const float* __restrict__ affs, const float* __restrict__ dloss,
                                                       float* __restrict__ dx) {
  constexpr int NT = TH * TW, D_T = 16;
  static_assert(TW == 32, "lane mapping");
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  int tile, b, seg, y0, x0;
  if (!xdma_tile<TH, TW>(C, P, tile, b, seg, y0, x0)) return;
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
  f2 Eh[8][kZS], Pd[8][kZS];
  float prj[kZS], ivs[kZS];
#pragma unroll
  for (int j = 0; j < kZS; ++j) {
    prj[j] = 0.f;
    ivs[j] = 0.f;
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) { Eh[ps][j] = (f2){0.f, 0.f}; Pd[ps][j] = (f2){0.f, 0.f}; }
  }
#define PEA_ZM_STEP(JO_, zz)                                                                                                   \
  {                                                                                                                            \
    const int z_ = (zz);                                                                                                       \
    if (z_ < ze + kZS) {                                                                                                       \
      if (z_ >= zb && z_ < ze)                                                                                                 \
        zm_bwd_full<JO_, TH, TW, PSU>(P, C, M, lds, xB, iB, gB, aB, dB, z_, z_ - kZS >= zb, dl, ecs, YX4, vo0, vo1, wbase, w1, \
                                      pe, px, py, vown, ax, ay, Eh, Pd, prj, ivs);                                             \
      else if (z_ >= 0 || z_ >= zb)                                                                                            \
        zm_bwd_light<JO_>(P, M, xB, iB, gB, aB, dB, z_, zb, ze, dl, ecs, YX4, pe, Eh, Pd, prj, ivs);                           \
    }                                                                                                                          \
  }
  for (int zq = zb - kZS; zq < ze + kZS; zq += kZS) {
    PEA_ZM_STEP(0, zq)
    PEA_ZM_STEP(1, zq + 1)
    PEA_ZM_STEP(2, zq + 2)
    PEA_ZM_STEP(3, zq + 3)
  }
#undef PEA_ZM_STEP
}

// ------------------------------------------------------------------------------------------------------------------
// forward (training: affs, g, loss, the 1 / norm plane; inference: affs)
// ------------------------------------------------------------------------------------------------------------------
template <int TH, int TW, int PSU, bool TRAIN>
__global__ __launch_bounds__(TH* TW, 2) void k_fwd_zm(const KParams P, const XParams C, const ZMParams M, const float* __restrict__ e,
                                                       const float* __restrict__ target, const float* __restrict__ weight,
                                                       const uint8_t* __restrict__ mask, float* __restrict__ affs,
                                                       float* __restrict__ gout, LossState* __restrict__ st,
                                                       float* __restrict__ inv_out) {
  constexpr int D_T = 16, NT = TH * TW, PS = PSU * 256, NP = D_T / 2, TP = NT, QP = TP / 4, NSL = QP / 64;
  constexpr int KMAX = kXP + 2, ITEMS = (KMAX * QP + NT - 1) / NT;
  static_assert(TW == 32 && QP % 64 == 0, "lane mapping");
  static_assert(KMAX * TP * 4 + KMAX * NSL * 4 <= 6 * PS && KMAX <= kXK, "the parked dot products fit the ring");
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  float* sA = (float*)lds;                        // [K][TP] dot products, laid over the ring once a plane's chunks are done
  float* s_part = (float*)(lds + KMAX * TP * 4);  // [K][NSL]
  int tile, b, seg, y0, x0;
  if (!xdma_tile<TH, TW>(C, P, tile, b, seg, y0, x0)) return;
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
  int an[kXP];
  const int vown = ((C.hy0 + ly) * TW + lx) * 4;
  const int hrow = (C.QV * 4 + ly * C.SW) * 4;
#pragma unroll
  for (int k = 0; k < kXP; ++k) {
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
        lds_barrier(); /* the previous plane's epilogue is done with sA / s_part: the ring is free */                         \
        PEA_ZM_DMA(xB, 0, ezo)                                                                                                 \
        PEA_ZM_DMA(xB, PS, ezo + ecs)                                                                                          \
        PEA_ZM_DMA(xB, 2 * PS, ezo + 2u * ecs)                                                                                 \
        PEA_ZM_DMA(xB, 3 * PS, ezo + 3u * ecs)                                                                                 \
        PEA_ZM_WAIT(4);                                                                                                        \
        PEA_ZM_DMA(xB, 4 * PS, ezo + 4u * ecs)                                                                                 \
        PEA_ZM_DMA(xB, 5 * PS, ezo + 5u * ecs)                                                                                 \
        f2 dot[kXP], ssq[kXP], dz[kZS], oss = {0.f, 0.f};                                                                      \
        _Pragma("unroll") for (int k = 0; k < kXP; ++k) { dot[k] = (f2){0.f, 0.f}; ssq[k] = (f2){0.f, 0.f}; }                  \
        _Pragma("unroll") for (int k = 0; k < kZS; ++k) dz[k] = (f2){0.f, 0.f};                                                \
        _Pragma("unroll") for (int ps = 0; ps < NP; ++ps) {                                                                    \
          const int bo = (ps % 3) * 2 * PS;                                                                                    \
          f2 o;                                                                                                                \
          o.x = *(const float*)(lds + bo + vown);                                                                              \
          o.y = *(const float*)(lds + bo + PS + vown);                                                                         \
          oss = __builtin_elementwise_fma(o, o, oss);                                                                          \
          _Pragma("unroll") for (int k = 0; k < kXP; ++k) {                                                                    \
            f2 v;                                                                                                              \
            v.x = *(const float*)(lds + bo + an[k]);                                                                           \
            v.y = *(const float*)(lds + bo + PS + an[k]);                                                                      \
            dot[k] = __builtin_elementwise_fma(o, v, dot[k]);                                                                  \
            ssq[k] = __builtin_elementwise_fma(v, v, ssq[k]);                                                                  \
            if (k % 5 == 4) asm volatile("" ::: "memory");                                                                     \
          }                                                                                                                    \
          _Pragma("unroll") for (int j = 0; j < kZS; ++j) dz[j] = __builtin_elementwise_fma(o, W[ps][j], dz[j]);               \
          W[ps][JO] = o;                                                                                                       \
          _Pragma("unroll") for (int k = 0; k < kXP; ++k) asm volatile("" : "+v"(dot[k]), "+v"(ssq[k]));                       \
          _Pragma("unroll") for (int j = 0; j < kZS; ++j) asm volatile("" : "+v"(dz[j]));                                      \
          asm volatile("" : "+v"(oss), "+v"(W[ps][JO]));                                                                       \
          if (ps + 1 < NP) {                                                                                                   \
            if (ps + 2 < NP) PEA_ZM_WAIT(4);                                                                                   \
            else PEA_ZM_WAIT(0);                                                                                               \
            if (ps + 3 < NP) {                                                                                                 \
              PEA_ZM_DMA(xB, bo, ezo + (unsigned)(2 * ps + 6) * ecs)                                                           \
              PEA_ZM_DMA(xB, bo + PS, ezo + (unsigned)(2 * ps + 7) * ecs)                                                      \
            }                                                                                                                  \
          }                                                                                                                    \
        }                                                                                                                      \
        /* target / weight / mask of the plane's items, requested now: they land while the dot products are parked */        \
        f4 t4[ITEMS], w4[ITEMS];                                                                                               \
        unsigned m4[ITEMS];                                                                                                    \
        if (TRAIN) {                                                                                                           \
          _Pragma("unroll") for (int it = 0; it < ITEMS; ++it) {                                                               \
            const unsigned so = ezo + (unsigned)isl[it] * ecs;                                                                 \
            t4[it] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(tB, ivo[it], so, kAuxNT));                   \
            w4[it] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(wB, ivo[it], so, kAuxNT));                   \
            m4[it] = has_m ? __builtin_amdgcn_raw_buffer_load_b32(mB, ivo[it] == kOOB ? kOOB : ivo[it] >> 2,                   \
                                                                 (ezo >> 2) + (unsigned)isl[it] * (unsigned)P.S, kAuxNT)       \
                           : 0x01010101u;                                                                                      \
          }                                                                                                                    \
        }                                                                                                                      \
        const float osum = oss.x + oss.y;                                                                                      \
        const float inv_own = rnorm(osum, inv_eps);                                                                            \
        if (has_i) bs32(iB, osum < P.eps * P.eps ? -inv_own : inv_own, pe, ezo);                                               \
        lds_barrier(); /* every lane is done with the ring: sA goes over it */                                                 \
        _Pragma("unroll") for (int k = 0; k < kXP; ++k) {                                                                      \
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
            const float a = z - s >= 0 ? (dz[jj].x + dz[jj].y) * inv_own * iw[jj] : 0.f;                                      \
            sA[ch * TP + (int)threadIdx.x] = a;                                                                                \
          }                                                                                                                    \
        }                                                                                                                      \
        iw[JO] = inv_own;                                                                                                      \
        lds_barrier();                                                                                                         \
        _Pragma("unroll") for (int it = 0; it < ITEMS; ++it) {                                                                 \
          if (!ion[it]) continue;                                                                                              \
          const int sl = isl[it];                                                                                              \
          const f4 a4 = *(const f4*)(sA + sl * TP + iqd[it] * 4);                                                              \
          const unsigned so = ezo + (unsigned)sl * ecs;                                                                        \
          if (has_a) {                                                                                                         \
            f4 o4 = a4;                                                                                                        \
            if (af) { o4.x = act_affs(o4.x, af); o4.y = act_affs(o4.y, af); o4.z = act_affs(o4.z, af); o4.w = act_affs(o4.w, af); } \
            bs128<true>(aB, o4, ivo[it], so);                                                                                  \
          }                                                                                                                    \
          if (TRAIN) {                                                                                                         \
            float acc = 0.f;                                                                                                   \
            f4 g4;                                                                                                             \
            const float gs = C.gs[sl];                                                                                         \
            const int ax_ = C.oax[sl], od_ = C.od[sl];                                                                         \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                    \
              const float m = (float)((m4[it] >> (8 * j)) & 0xffu);                                                            \
              const float r = a4[j] * m - t4[it][j] * m;                                                                       \
              float wr = w4[it][j] * r;                                                                                        \
              const int q = (ax_ == 1 ? igx[it] + j : ax_ == 0 ? igy[it] : z) + od_;                                           \
              wr = (unsigned)q < (unsigned)(ax_ == 1 ? P.X : ax_ == 0 ? P.Y : P.Z) ? wr : 0.f;                                 \
              g4[j] = gs * wr * m;                                                                                             \
              acc = fmaf(wr, r, acc);                                                                                          \
            }                                                                                                                  \
            if (has_g) bs128<false>(gB, g4, ivo[it], so);                                                                      \
            const float red = wave_sum63(acc);                                                                                 \
            if ((threadIdx.x & 63) == 63) s_part[sl * NSL + (iqd[it] >> 6)] = red;                                             \
          }                                                                                                                    \
        }                                                                                                                      \
        if (TRAIN) {                                                                                                           \
          lds_barrier();                                                                                                       \
          if (wave == 0 && (int)threadIdx.x < P.K) {                                                                           \
            float v = 0.f;                                                                                                     \
            _Pragma("unroll") for (int s = 0; s < NSL; ++s) v += s_part[threadIdx.x * NSL + s];                                \
            loss_accumulate(st, tile + z, threadIdx.x, v);                                                                     \
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
