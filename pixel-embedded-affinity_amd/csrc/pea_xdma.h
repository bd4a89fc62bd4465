// pea_xdma.h -- backward for AXIS-ALIGNED in-plane stencils (every offset moves along y or along x only: the CVPPP /
// BBBC039V1 multi_offset(neighbor=4) tables, the in-plane AC3/AC4 tables), the whole stencil served from LDS, staged by
// LDS-DMA.  Included by pea_hip.hip only.
//
// Why a second backward next to k_bwd_tiled (pea_tiled.h): that kernel stages a (TH+2h) x (TW+2h) BOX of 64-byte
// pixels through registers; at h = 9 one 32x32 tile fills the CU's 160 KB (one workgroup per CU: staging, gather and
// stores of a tile run one after the other) and the +-27 offsets are gathered from global memory, 16 one-dword loads
// per (pixel, pair).  For an axis-aligned stencil the pixels a tile needs form a CROSS, not a box:
//     VF : the tile's TW columns over rows y0 - hy .. y0 + TH + hy      (the tile itself lives here)
//     H  : for each tile row, the columns right / left of the tile that the x offsets reach
// and  G(p) = sum_pairs g * ehat(q)  separates over channels once 1 / |e(q)| is known, so the channels go through LDS
// TWO AT A TIME as planar float planes (one float per region pixel and channel), moved by buffer_load_dwordx4 ... lds
// (no VGPR round trip, no ds_write, 4 pixels per lane and instruction, 1 KiB per wave instruction) into a ring of three
// buffers: the DMA of chunk p + 2 is in flight while chunk p is gathered.  1 / |e| comes from a plane the forward
// writes (4 bytes per pixel) or from k_inv_norm; it is staged like a channel and folded into the pair coefficients
// g * 1/|e(q)| once per tile.  A 16x32 tile with the full +-27 cross is 3264 region pixels = 78 KB for six planes: two
// workgroups per CU, every one of the 2K pairs an LDS read (ds_read2st64_b32: both channels of the chunk), no far
// gathers.  Vector-memory instructions per pixel: 34 x dwordx4 DMA + 20 g loads + 16 stores (k_bwd_tiled: 139 dword).
//
// LDS geometry (dword index inside a plane):  VF[r][c] = r * TW + c  (r = 0 is image row y0 - hy0);  H strips after it,
// HB + ly * SW + coord with SW = 32 or 64: the RIGHT strip at coord [0, SW/2), the LEFT strip at [SW/2, SW), so that a
// neighbour column c = lx + d outside the tile sits at coord (c & 31) resp. (c & (SW-1)): bank (c mod 32) -- the same
// bank it would have inside the tile, i.e. a wave whose lanes split between VF and a strip still reads 32 distinct banks.
#pragma once
#include "pea_tiled.h"

namespace pea {

constexpr int kXP = 10;  // (offset, role) pairs per axis held in registers (CVPPP: 5 shifts x 2 roles)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct XParams {
  int hy0, hy1;   // halo rows above / below the tile in VF
  int SW;         // strip row length: 32 (reach <= 16) or 64
  int QV, QA;     // quads (4 x-adjacent pixels) in VF; in VF + strips
  int QW;         // quads per wave = ceil(QA / waves), <= 128
  int tiles_y, tiles_x, tiles_per_plane, ntiles, tiles_per_xcd;
  int npx, npy;
  int xd[kXP], yd[kXP];    // pixel displacement of the neighbour along x / y  (0 for unused pairs)
  int xm[kXP];             // strip coordinate mask of the x pair: d < 0 ? SW - 1 : 31
  int xgi[kXP], ygi[kXP];  // g channel
  int xgo[kXP], ygo[kXP];  // role A: 0 (g at p); role B: -o (g at p - o)
};

// inv[b, z, y, x] = 1 / max(|e|, eps), NEGATED where |e| < eps (the clamp branch of F.normalize: d ehat / d e = I / eps)
template <typename T>
__global__ __launch_bounds__(256) void k_inv_norm(const KParams P, const T* __restrict__ e, float* __restrict__ inv) {
  const int tile = logical_tile(P);
  if (tile >= P.tiles) return;
  const int b = tile / P.chunks;
  const int p = (tile - b * P.chunks) * kBlock + threadIdx.x;
  if (p >= P.S) return;
  const T* eb = e + (size_t)b * P.D * P.S + p;
  float ss = 0.f;
  for (int c = 0; c < P.D; ++c) {
    const float v = ld(eb, (size_t)c * P.S);
    ss = fmaf(v, v, ss);
  }
  const float r = fminf(__builtin_amdgcn_rsqf(ss), 1.0f / P.eps);
  inv[(size_t)b * P.S + p] = ss < P.eps * P.eps ? -r : r;
}

// self-loss backward (both roles, nb == x); f32 storage; X % 4 == 0 and 16-byte aligned planes (host-checked)
// LDS: six planes of PS = PSU * 256 bytes: buffer b in {0,1,2}, channel j of the chunk at (2b + j) * PS; the 1 / norm
// plane starts out in plane 4 (buffer 2 is first filled after the coefficients are done).
template <int D_T, int TH, int TW, int PSU, bool CROP, int AUXS = 0>
__global__ __launch_bounds__(TH* TW, 4) void k_bwd_xdma(const KParams P, const XParams C, const float* __restrict__ xt,
                                                         const float* __restrict__ invp, const float* __restrict__ gin,
                                                         const float* __restrict__ dloss, float* __restrict__ dx) {
  constexpr int NT = TH * TW, PS = PSU * 256, NP = D_T / 2;
  static_assert(TW == 32 && D_T % 2 == 0, "lane mapping / channel pairs");
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  const int bid = blockIdx.x;
  const int tile = (bid % kXcd) * C.tiles_per_xcd + bid / kXcd;
  if (tile >= C.ntiles) return;
  const int plane = tile / C.tiles_per_plane;
  const int rem = tile - plane * C.tiles_per_plane;
  const int ty = rem / C.tiles_x;
  const int y0 = ty * TH, x0 = (rem - ty * C.tiles_x) * TW;
  const int b = plane / P.Z, z = plane - b * P.Z;
  const size_t S = (size_t)P.S;
  const unsigned YX = (unsigned)(P.Y * P.X);
  const rsrc_t xB = mkbuf(xt + (size_t)b * D_T * S), dB = mkbuf(dx + (size_t)b * D_T * S);
  const rsrc_t gB = mkbuf(gin + (size_t)b * P.K * S), iB = mkbuf(invp + (size_t)b * S);
  const unsigned ecs = (unsigned)P.S * 4u, ezo = (unsigned)z * YX * 4u;  // channel stride / plane offset (e, g, inv: all f32)
  const float dl = dloss ? dloss[0] : 1.f;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;

  const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
  const int py = y0 + ly, px = x0 + lx;
  const bool live = py < P.Y && px < P.X;
  const unsigned po4 = (unsigned)(py * P.X + px) * 4u;
  const unsigned pe = live ? po4 : kOOB;

  // ---- the two quads this lane moves per plane (wave w owns quads [w * QW, (w+1) * QW)): global byte offset
  unsigned vo[2];
  bool act[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int qw = s * 64 + lane;
    const int q = wave * C.QW + qw;
    int gy, gx;
    if (q < C.QV) {
      gy = y0 - C.hy0 + (q >> 3);
      gx = x0 + 4 * (q & 7);
    } else {
      const int k = q - C.QV;
      const int sh = C.SW == 64 ? 4 : 3;  // quads per strip row: 16 / 8
      const int cc = 4 * (k & ((1 << sh) - 1));
      gy = y0 + (k >> sh);
      gx = cc < (C.SW >> 1) ? x0 + TW + cc : x0 - C.SW + cc;  // right strip first, then the left one
    }
    act[s] = qw < C.QW && q < C.QA;
    bool oky, okx;
    gy = wrap1<CROP>(gy, P.Y, oky);
    gx = wrap1<CROP>(gx, P.X, okx);
    vo[s] = (oky && okx) ? (unsigned)(gy * P.X + gx) * 4u : kOOB;
  }
  const int wbase = wave * C.QW * 16;  // this wave's byte offset inside a plane
  // DMA instructions this wave issues per chunk (a slot without a live lane is skipped): what `vmcnt` has to count
  const int npc = 2 * ((__builtin_amdgcn_ballot_w64(act[0]) != 0) + (__builtin_amdgcn_ballot_w64(act[1]) != 0));
  // wait until only the youngest chunk's DMA may still be in flight, then the workgroup barrier
#define PEA_XWAIT1()                                                                             \
  {                                                                                              \
    if (npc == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");       \
    else if (npc == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");  \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                \
  }
#define PEA_XDMA(rsrc, plane_byte, so)                                                                              \
  {                                                                                                                 \
    if (act[0]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + (plane_byte) + wbase), 16, vo[0], so, 0, 0);        \
    if (act[1]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + (plane_byte) + wbase + 1024), 16, vo[1], so, 0, 0); \
  }
  PEA_XDMA(iB, 4 * PS, ezo)
  PEA_XDMA(xB, 0, ezo)
  PEA_XDMA(xB, PS, ezo + ecs)

  // ---- g of every pair (role A at p, role B at p - o) and the LDS slot of every neighbour
  // dead lanes: an offset that stays out of range when a small displacement is added
  const unsigned pg = live ? po4 : 0xC0000000u;
  float cx[kXP], cy[kXP];
  int ax[kXP], ay[kXP];
  const int vown = ((C.hy0 + ly) * TW + lx) * 4;
  const int hrow = (C.QV * 4 + ly * C.SW) * 4;
#pragma unroll
  for (int k = 0; k < kXP; ++k) {
    const int go = C.xgo[k];                       // uniform
    const int t = px + go;
    const bool out = (unsigned)t >= (unsigned)P.X;  // either side
    const int fix = go > 0 ? -P.X : P.X;            // uniform
    const unsigned o = CROP ? (out ? kOOB : pg + (unsigned)(go * 4)) : pg + (unsigned)((out ? go + fix : go) * 4);
    cx[k] = bl32(gB, k < C.npx ? o : kOOB, ezo + (unsigned)C.xgi[k] * ecs);
    const int d = C.xd[k], c = lx + d;
    ax[k] = (unsigned)c < (unsigned)TW ? vown + d * 4 : hrow + (c & C.xm[k]) * 4;
  }
#pragma unroll
  for (int k = 0; k < kXP; ++k) {
    const int go = C.ygo[k];
    const int t = py + go;
    const bool out = (unsigned)t >= (unsigned)P.Y;
    const int fix = go > 0 ? -P.Y : P.Y;
    const unsigned o = CROP ? (out ? kOOB : pg + (unsigned)(go * P.X * 4)) : pg + (unsigned)((out ? go + fix : go) * P.X * 4);
    cy[k] = bl32(gB, k < C.npy ? o : kOOB, ezo + (unsigned)C.ygi[k] * ecs);
    ay[k] = vown + C.yd[k] * TW * 4;
  }
  PEA_XDMA(xB, 2 * PS, ezo + 2u * ecs)
  PEA_XDMA(xB, 3 * PS, ezo + 3u * ecs)
  // inv, chunk 0 and g have landed (the 4 DMA instructions of chunk 1 may still fly); every wave's share of them too
  PEA_XWAIT1()

  // coefficient of a pair = g * 1 / |e(q)|
  const float invo = *(const float*)(lds + 4 * PS + vown);
  const float inv_own = fabsf(invo);
#pragma unroll
  for (int k = 0; k < kXP; ++k) {
    cx[k] *= fabsf(*(const float*)(lds + 4 * PS + ax[k]));
    cy[k] *= fabsf(*(const float*)(lds + 4 * PS + ay[k]));
    // computed HERE: volatile asm statements keep their order; the scheduler otherwise sinks the whole arithmetic
    // below the last barrier and keeps every LDS value of every chunk in registers (256 VGPRs + spills)
    asm volatile("" : "+v"(cx[k]), "+v"(cy[k]));
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the inv plane is dead: buffer 2 may be filled
  if (NP > 2) {
    PEA_XDMA(xB, 4 * PS, ezo + 4u * ecs)
    PEA_XDMA(xB, 5 * PS, ezo + 5u * ecs)
  }

  f2 G[NP], eh[NP];
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int bo = (ps % 3) * 2 * PS;
    f2 o;
    o.x = *(const float*)(lds + bo + vown);
    o.y = *(const float*)(lds + bo + PS + vown);
    eh[ps] = o * inv_own;
    asm volatile("" : "+v"(eh[ps]));
    f2 acc = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < kXP; ++k) {
      f2 v;
      v.x = *(const float*)(lds + bo + ax[k]);
      v.y = *(const float*)(lds + bo + PS + ax[k]);
      acc = __builtin_elementwise_fma((f2){cx[k], cx[k]}, v, acc);
      if (k % 5 == 4) asm volatile("" ::: "memory");  // bound the ds_read hoisting
    }
#pragma unroll
    for (int k = 0; k < kXP; ++k) {
      f2 v;
      v.x = *(const float*)(lds + bo + ay[k]);
      v.y = *(const float*)(lds + bo + PS + ay[k]);
      acc = __builtin_elementwise_fma((f2){cy[k], cy[k]}, v, acc);
      if (k % 5 == 4) asm volatile("" ::: "memory");
    }
    asm volatile("" : "+v"(acc));  // the chunk's sums exist before its barrier
    G[ps] = acc;
    if (ps + 1 < NP) {
      // chunk ps + 1 has landed (chunk ps + 2, issued after it, may still fly); everyone is done with buffer ps % 3
      if (ps + 2 < NP) PEA_XWAIT1()
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (ps + 3 < NP) {
        PEA_XDMA(xB, bo, ezo + (unsigned)(2 * ps + 6) * ecs)
        PEA_XDMA(xB, bo + PS, ezo + (unsigned)(2 * ps + 7) * ecs)
      }
    }
  }
#undef PEA_XDMA
#undef PEA_XWAIT1

  float proj = 0.f;
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) proj = fmaf(eh[ps].x, G[ps].x, fmaf(eh[ps].y, G[ps].y, proj));
  if (invo < 0.f) proj = 0.f;  // clamp branch of F.normalize
  const float sc = dl * inv_own;
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (G[ps].x - eh[ps].x * proj) * sc), dB, pe, ezo + (unsigned)(2 * ps) * ecs, AUXS);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (G[ps].y - eh[ps].y * proj) * sc), dB, pe, ezo + (unsigned)(2 * ps + 1) * ecs, AUXS);
  }
}

// host: the plan.  false = not an axis-aligned in-plane stencil that fits (the caller falls back to k_bwd_tiled)
inline bool plan_xdma(const KParams& P, int TH, int TW, int psu, XParams* out, size_t* lds_bytes) {
  if (P.border == PEA_BORDER_REPLICATE) return false;
  if ((long long)P.Y * P.X >= (1LL << 28)) return false;                           // plane byte offsets + displacement < 2^31
  if ((long long)(P.D > P.K ? P.D : P.K) * P.S * 4 > 0xFFFFFFFFLL) return false;   // 32-bit buffer soffset
  if (P.X % 4 || P.S % 4) return false;                                            // quads never straddle a row end
  XParams C = {};
  int hx = 0, hy = 0;
  for (int i = 0; i < P.K; ++i) {
    const int oz = P.off[i][0], oy = P.off[i][1], ox = P.off[i][2];
    if (oz != 0 || (oy != 0) == (ox != 0)) return false;  // exactly one in-plane component
    if (ox != 0) {
      if (C.npx + 2 > kXP) return false;
      hx = ox < 0 ? (hx > -ox ? hx : -ox) : (hx > ox ? hx : ox);
      C.xd[C.npx] = ox; C.xgi[C.npx] = i; C.xgo[C.npx] = 0; ++C.npx;     // role A: neighbour p + o, g at p
      C.xd[C.npx] = -ox; C.xgi[C.npx] = i; C.xgo[C.npx] = -ox; ++C.npx;  // role B: neighbour p - o, g at p - o
    } else {
      if (C.npy + 2 > kXP) return false;
      hy = oy < 0 ? (hy > -oy ? hy : -oy) : (hy > oy ? hy : oy);
      C.yd[C.npy] = oy; C.ygi[C.npy] = i; C.ygo[C.npy] = 0; ++C.npy;
      C.yd[C.npy] = -oy; C.ygi[C.npy] = i; C.ygo[C.npy] = -oy; ++C.npy;
    }
  }
  if (hx > TW) return false;  // a neighbour column is inside the tile or in the strip next to it
  C.hy0 = C.hy1 = hy;
  C.SW = hx <= 16 ? 32 : 64;
  for (int k = 0; k < kXP; ++k) C.xm[k] = C.xd[k] < 0 ? C.SW - 1 : 31;
  C.QV = (C.hy0 + TH + C.hy1) * TW / 4;
  C.QA = C.QV + TH * C.SW / 4;
  const int nw = TH * TW / 64;
  C.QW = (C.QA + nw - 1) / nw;
  if (C.QW > 128) return false;                      // two quads per lane and plane
  if (C.QW * nw * 16 > psu * 256) return false;      // the plane holds every wave's share
  // the kernels wrap with one conditional add
  if (P.Y < TH + C.hy1 || P.Y < C.hy0 || P.X < TW + C.SW / 2 || P.X < C.SW / 2) return false;
  C.tiles_y = (P.Y + TH - 1) / TH;
  C.tiles_x = (P.X + TW - 1) / TW;
  C.tiles_per_plane = C.tiles_y * C.tiles_x;
  const long long nt = (long long)C.tiles_per_plane * P.Z * P.B;
  if (nt > 0x7fffff00LL) return false;
  C.ntiles = (int)nt;
  C.tiles_per_xcd = (C.ntiles + kXcd - 1) / kXcd;
  *lds_bytes = (size_t)6 * psu * 256;
  *out = C;
  return true;
}

}  // namespace pea
