// pea_xdma.h -- backward for AXIS-ALIGNED in-plane stencils (every offset moves along y or along x only: the CVPPP /
// BBBC039V1 multi_offset(neighbor=4) tables, the in-plane AC3/AC4 tables), the whole stencil served from LDS, staged by
// LDS-DMA.  Included by pea_k_xdma.hip (and, for the staging geometry and the tile walk, by pea_box.h / pea_zmarch.h).
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
// HB + ly * SW + coord with SW = 32 or 64: the RIGHT strip at coord [0, split), the LEFT strip at [split, SW) (a one-sided
// stencil keeps one of them: split = 0 or SW), so that a neighbour column c = lx + d outside the tile sits at coord
// (c & 31) resp. (c & (SW-1)): bank (c mod 32) -- the same bank it would have inside the tile, i.e. a wave whose lanes
// split between VF and a strip still reads 32 distinct banks.
#pragma once
#include "pea_tiled.h"

namespace pea {

// In-kernel stamps (cdna_hip_programming.md section 7): only in the diagnostic build profiles/microbench/stamp_bwd.hip, which
// defines PEA_STAMPS; in the library no stamp executes.  Wave `w` of workgroup `b` < kStampWgs writes s_memtime into its own row
// of a buffer nothing else reads.
#ifdef PEA_STAMPS
constexpr int kStampWgs = 128, kStampN = 96;
__device__ unsigned long long* g_stamps;
#define PEA_STAMP(i)                                                                                              \
  {                                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    unsigned long long t_;                                                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    if (blockIdx.x < kStampWgs && (threadIdx.x & 63) == 0 && (i) < kStampN)                                       \
      g_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * kStampN + (i)] = t_;                               \
  }
#else
#define PEA_STAMP(i)
#endif
constexpr int kStampLast = 95;

constexpr int kXP = 10;  // (offset, role) pairs per axis held in registers (CVPPP: 5 shifts x 2 roles)
constexpr int kXZ = 8;   // (offset, role) pairs along z (AC3/AC4 norm5: shifts 1, 2, 3, 4)
constexpr int kXK = 16;  // channels (offsets) the forward's epilogue handles
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// acc += c * v for both channels of a pair, the coefficient c taken from the LOW (HI = false) or HIGH half of a register pair
// that holds TWO pairs' coefficients.  hipcc materialises (f2){c, c} in two registers per coefficient (20 pairs: 40 VGPRs of
// coefficients); v_pk_fma_f32's op_sel / op_sel_hi select the half per lane of the packed operation, so one register pair
// serves two pairs: half the coefficient registers in the gather loops (the f16 kernels and the box kernels need that).
template <bool HI>
__device__ __forceinline__ f2 pk_fma_c(f2 cpair, f2 v, f2 acc) {
  if (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(cpair), "v"(v));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(cpair), "v"(v));
  return acc;
}

struct XParams {
  int hy0, hy1;   // halo rows above / below the tile in VF
  int SW;         // strip row length in LDS: 32 or 64 pixels
  int split;      // strip coordinates < split hold columns x0 + TW + coord (right of the tile), the others x0 - SW + coord
  int QV, QA;     // quads (4 x-adjacent pixels) in VF; in VF + strips.  Blocks of 64 quads go round the waves.
  int tiles_y, tiles_x, tiles_per_plane, ntiles, tiles_per_xcd;
  int npx, npy;
  int xd[kXP], yd[kXP];    // pixel displacement of the neighbour along x / y  (0 for unused pairs)
  int xm[kXP];             // strip coordinate mask of the x pair: d < 0 ? SW - 1 : TW - 1
  int xgi[kXP], ygi[kXP];  // g channel
  int xgo[kXP], ygo[kXP];  // role A: 0 (g at p); role B: -o (g at p - o)
  // z offsets (3D volumes): not staged (a third LDS arm would cost 8 region pixels per pixel and does not fit beside the
  // ring); their neighbour is the SAME (y, x) in another plane, read per chunk straight from global memory -- plane and
  // validity are wave-uniform, so the displacement rides in the scalar offset and costs no VGPR
  int npz;                 // backward: (offset, role) pairs along z
  int zd[kXZ];             // plane displacement of the neighbour
  int zgi[kXZ], zgo[kXZ];  // g channel; plane displacement of the g sample (role A: 0, role B: -oz)
  int rev;                 // every XCD walks its tile range from the last tile to the first (PEA_BWD_REV: the backward starts where the
                           // forward ended)
  int zrun;                // > 1: tiles walk z fastest (= Z), so the planes a z offset reaches were staged just before
  int zgy, zgx;            // ... inside blocks of zgy x zgx tiles
  int sup_y, sup_x;        // > 0 (sup_y * sup_x == 8): the eight XCDs' blocks of one round lie side by side as ONE super-block of
                           // (sup_y zgy) x (sup_x zgx) tiles (march kernels: what the round fetches from HBM is the super-block's halo)
  // forward (role A only): in-plane offsets in their own order (nf <= kXP), z offsets (nfz <= kXZ / 2)
  int nf, nfz;
  int fd[kXP];             // displacement along the offset's axis
  int fax[kXP];            // 1: along x, 0: along y
  int fm[kXP];             // strip coordinate mask (x offsets)
  int fi[kXP];             // channel of the in-plane offset
  int fzd[kXZ / 2], fzi[kXZ / 2];  // plane displacement, channel of the z offset
  // per CHANNEL (the forward's epilogue walks channels): 2 * lambda_i / N_i, axis (0 y, 1 x, 2 z), displacement
  float gs[kXK];
  int oax[kXK], od[kXK];
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

// tile id -> (plane = b * Z + z, y0, x0).  XCD g walks tiles [g * tpx, (g+1) * tpx); volumes with z offsets walk z FASTEST
// (zrun = Z): the tiles running together on an XCD are a few (y, x) columns over all z, so the planes a z offset reaches
// are in that XCD's L2.  The returned tile id stays plane-major (the loss-partial slot of a tile does not depend on the walk).
template <int TH, int TW, typename CP = XParams>
__device__ __forceinline__ bool xdma_tile(const CP& C, const KParams& P, int& tile, int& b, int& z, int& y0, int& x0) {
  const int bid = blockIdx.x;
  int slot = bid / kXcd;
  if (C.rev) slot = C.tiles_per_xcd - 1 - slot;
  const int lin = (bid % kXcd) * C.tiles_per_xcd + slot;
  if (lin >= C.ntiles) return false;
  int plane, rem;
  if (C.zrun >= 1) {
    // walk: blocks of kGY x kGX tiles; inside a block z, then y, then x fastest -- the tiles in flight on an XCD (64) are a few
    // planes of one block: the z neighbours were staged 1-4 planes ago, the in-plane halos are shared inside the block.
    const int kGY = C.zgy, kGX = C.zgx;
    const int per_b = C.tiles_per_plane * C.zrun;
    b = lin / per_b;
    int r = lin - b * per_b;
    const int nbx = (C.tiles_x + kGX - 1) / kGX;
    // block row (all of its x blocks), then block, then (z, y, x) inside the block; edge blocks are smaller
    const int rows_full = kGY * C.tiles_x * C.zrun;            // tiles in a full block row
    const int by = r / rows_full;
    r -= by * rows_full;
    const int gy = min(kGY, C.tiles_y - by * kGY);             // rows of this block row
    const int blk_full = gy * kGX * C.zrun;                    // tiles in a full-width block of this block row
    const int bx = min(r / blk_full, nbx - 1);
    r -= bx * blk_full;
    const int gx = min(kGX, C.tiles_x - bx * kGX);
    z = r / (gy * gx);
    r -= z * gy * gx;
    const int ty_ = by * kGY + r / gx, tx_ = bx * kGX + r % gx;
    rem = ty_ * C.tiles_x + tx_;
    plane = b * C.zrun + z;
  } else {
    plane = lin / C.tiles_per_plane;
    rem = lin - plane * C.tiles_per_plane;
    b = plane / P.Z;
    z = plane - b * P.Z;
  }
  tile = plane * C.tiles_per_plane + rem;
  const int ty = rem / C.tiles_x;
  y0 = ty * TH;
  x0 = (rem - ty * C.tiles_x) * TW;
  return true;
}

// The march kernels' tile walk (zrun = segments per tile column).  sup_x > 0: the SUPER-BLOCK walk -- round r of the launch = super-block
// r, XCD g = its block (g / sup_x, g % sup_x) of zgy x zgx tile columns.  What one XCD's block shares with its neighbours' it finds in
// the Infinity Cache (they are fetched in the same round), so HBM sees the halo of (sup_y zgy) x (sup_x zgx) tiles, not of each block.
// Super-blocks are padded to the tile grid: a workgroup whose tile lies beyond it leaves.  (Its own function: inside xdma_tile the
// second path cost the 80-VGPR forward one spilled register.)
template <int TH, int TW, typename CP = XParams>
__device__ __forceinline__ bool march_tile(const CP& C, const KParams& P, int& tile, int& b, int& z, int& y0, int& x0) {
  if (C.sup_x <= 0) return xdma_tile<TH, TW, CP>(C, P, tile, b, z, y0, x0);
  const int bid = blockIdx.x, slot = bid / kXcd;
  const int G = C.zgy * C.zgx * C.zrun;
  const int sb = slot / G;
  int r = slot - sb * G;
  const int nsx = (C.tiles_x + C.zgx * C.sup_x - 1) / (C.zgx * C.sup_x), nsy = (C.tiles_y + C.zgy * C.sup_y - 1) / (C.zgy * C.sup_y);
  b = sb / (nsx * nsy);
  if (b >= P.B) return false;
  const int s2 = sb - b * nsx * nsy, sby = s2 / nsx, sbx = s2 - sby * nsx;
  const int xcd = bid % kXcd, xy = xcd / C.sup_x, xx = xcd - xy * C.sup_x;
  z = r / (C.zgy * C.zgx);
  r -= z * C.zgy * C.zgx;
  const int ty_ = (sby * C.sup_y + xy) * C.zgy + r / C.zgx, tx_ = (sbx * C.sup_x + xx) * C.zgx + r % C.zgx;
  if (ty_ >= C.tiles_y || tx_ >= C.tiles_x) return false;
  tile = (b * C.zrun + z) * C.tiles_per_plane + ty_ * C.tiles_x + tx_;
  y0 = ty_ * TH;
  x0 = tx_ * TW;
  return true;
}

// self-loss backward (both roles, nb == x); f32 storage; X % 4 == 0 and 16-byte aligned planes (host-checked)
// LDS: six planes of PS = PSU * 256 bytes: buffer b in {0,1,2}, channel j of the chunk at (2b + j) * PS; the 1 / norm
// plane starts out in plane 4 (buffer 2 is first filled after the coefficients are done).
// AUXS: cache policy of the gradient stores (non-temporal: they must not push the halo lines out of the L2, pea_tiled.h bs_emb)
// XP: (offset, role) pairs per axis held in registers (<= kXP; the D = 64 instantiation takes 8 to stay inside 128 VGPRs)
// ZP: pairs along z read from global memory (0: 2D; kXZ: the 3D instantiation)
// DUAL: the backward of the training loop's full-resolution PAIR in one launch (scripts_cvppp/main.py:284-293: embedding_loss and
//   ema_embedding_loss of the same `embedding`, the second with the detached EMA operand): after the self loss' chunk loop the
//   workgroup stages the SECOND operand's one-sided cross (plan_xdma mode 2) and adds the cross loss' role-A pairs into the same
//   registers; one projection, one store of de.  280 B/px instead of the 408 of the two launches (no second read of e, no
//   read-modify-write of de).  The two losses' grad_outputs scale the sums (G *= dl_self; the second phase's coefficients *= dl_cross).
struct DualArgs {
  XParams C2;               // plan_xdma(.., mode 2) of the cross loss' descriptor
  const float* ema;         // [B, D, S] second operand
  const float* inv_other;   // [B, S] its signed 1 / norm plane
  const float* g_cross;     // [B, K, S] d loss_cross / d affs
  const float* dloss_cross;
};
// second phase of k_bwd_xdma<.., DUAL>: role-A pairs of the second operand into G (D = 16: NP = 8 chunk pairs), 2D, no z.
// Round 5: the phase no longer starts from an empty pipeline.  While the first phase gathers its last two chunks the kernel has
// already requested this phase's 1 / norm plane (into plane 4, free behind chunk NP - 3), its g values (cx / cy: role A, the own
// pixel) and its first chunk (into buffer 0, free behind chunk NP - 2) -- dual_geom / the hand-offs of k_bwd_xdma; here the
// coefficients are formed at once, chunks 1 and 2 follow, and the ring runs as before.
struct DualGeom {
  unsigned vo[2];
  bool act[2];
  int npc;
};
template <int TH, int TW, bool CROP>
__device__ __forceinline__ DualGeom dual_geom(const KParams& P, const XParams& C, const int y0, const int x0) {
  constexpr int NT = TH * TW;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  DualGeom g;
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
    g.act[s] = q < C.QA;
    bool oky, okx;
    gy = wrap1<CROP>(gy, P.Y, oky);
    gx = wrap1<CROP>(gx, P.X, okx);
    g.vo[s] = (g.act[s] && oky && okx) ? (unsigned)(gy * P.X + gx) * 4u : kOOB;
  }
  g.npc = 2 * ((__builtin_amdgcn_ballot_w64(g.act[0]) != 0) + (__builtin_amdgcn_ballot_w64(g.act[1]) != 0));
  return g;
}
#define PEA_X2DMA(rsrc, plane_byte, so)                                                                                             \
  {                                                                                                                                 \
    if (DG.act[0]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + (plane_byte) + wbase2), 16, DG.vo[0], so, 0, 0); \
    if (DG.act[1]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + (plane_byte) + w12), 16, DG.vo[1], so, 0, 0);    \
  }
template <int TH, int TW, int PSU, bool CROP, int XP>
__device__ __forceinline__ void bwd_phase_role_a(const KParams& P, const XParams& C, char* lds, const rsrc_t xB, const DualGeom& DG,
                                                 float (&cx)[XP], float (&cy)[XP], const float dlx, const unsigned ezo,
                                                 const unsigned ecs, f2 (&G)[8]) {
  constexpr int NT = TH * TW, PS = PSU * 256, NP = 8;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
  const int wbase2 = wave * 1024, w12 = wbase2 + (NT / 64) * 1024;
  const int npc = DG.npc;
#define PEA_X2WAIT1()                                                                            \
  {                                                                                              \
    if (npc == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");       \
    else if (npc == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");  \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                \
  }
  int ax[XP], ay[XP];
  const int vown = ((C.hy0 + ly) * TW + lx) * 4;
  const int hrow = (C.QV * 4 + ly * C.SW) * 4;
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    const int d = C.xd[k], c = lx + d;
    ax[k] = (unsigned)c < (unsigned)TW ? vown + d * 4 : hrow + (c & C.xm[k]) * 4;
    ay[k] = vown + C.yd[k] * TW * 4;
  }
  // every wave is done with the first phase's last chunk; this phase's 1 / norm plane, g values and first chunk -- requested one and
  // two chunks ago -- have landed
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    cx[k] *= fabsf(*(const float*)(lds + 4 * PS + ax[k])) * dlx;
    cy[k] *= fabsf(*(const float*)(lds + 4 * PS + ay[k])) * dlx;
    asm volatile("" : "+v"(cx[k]), "+v"(cy[k]));
  }
  PEA_X2DMA(xB, 2 * PS, ezo + 2u * ecs)  // chunk 1 into buffer 1 (the first phase's last chunk is done)
  PEA_X2DMA(xB, 3 * PS, ezo + 3u * ecs)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the inv plane is dead: buffer 2 may be filled
  PEA_X2DMA(xB, 4 * PS, ezo + 4u * ecs)
  PEA_X2DMA(xB, 5 * PS, ezo + 5u * ecs)
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int bo = (ps % 3) * 2 * PS;
    f2 acc = G[ps];
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
    asm volatile("" : "+v"(acc));
    G[ps] = acc;
    if (ps + 1 < NP) {
      if (ps + 2 < NP) PEA_X2WAIT1()
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (ps + 3 < NP) {
        PEA_X2DMA(xB, bo, ezo + (unsigned)(2 * ps + 6) * ecs)
        PEA_X2DMA(xB, bo + PS, ezo + (unsigned)(2 * ps + 7) * ecs)
      }
    }
  }
#undef PEA_X2WAIT1
}

// OTHER: the cross loss with a detached second operand (ema_embedding_loss, scripts_cvppp/loss/loss_embedding_mse.py:79-95 with
//   convert_consistency_flip's detach): role A only.  xt / invp are the SECOND operand and its 1 / norm plane (the neighbours);
//   the own pixel and its 1 / norm come from O.own / O.own_inv (global loads, once per tile); O.accumulate: dx += instead of =
//   (the self loss' gradient of the same embedding is already there).  plan_xdma mode 2.
struct OtherArgs {
  const float* own;      // [B, D, S] the embedding that is differentiated
  const float* own_inv;  // [B, S] its signed 1 / norm plane
  int accumulate;
};
template <int D_T, int TH, int TW, int PSU, bool CROP, int XP = kXP, int AUXS = kAuxNT, int ZP = 0, bool OTHER = false, bool DUAL = false>
__global__ __launch_bounds__(TH* TW, 4) void k_bwd_xdma(const KParams P, const XParams C, const float* __restrict__ xt,
                                                         const float* __restrict__ invp, const float* __restrict__ gin,
                                                         const float* __restrict__ dloss, float* __restrict__ dx,
                                                         const OtherArgs O, const DualArgs Q) {
  static_assert(!OTHER || D_T <= 16, "role-A instantiation: D <= 16 (z pairs: role A only, ZP = kXZ / 2)");
  static_assert(!DUAL || (D_T == 16 && ZP == 0 && !OTHER), "pair instantiation: D = 16, in-plane");
  constexpr int NT = TH * TW, PS = PSU * 256, NP = D_T / 2;
  static_assert(TW == 32 && D_T % 2 == 0, "lane mapping / channel pairs");
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  int tile, b, z, y0, x0;
  if (!xdma_tile<TH, TW>(C, P, tile, b, z, y0, x0)) return;
  __builtin_amdgcn_sched_barrier(0);
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

  // ---- the (up to) two quads this lane moves per plane: blocks of 64 quads go round the waves (block s * NW + wave)
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
      const int sh = C.SW == 64 ? 4 : 3;  // quads per strip row: 16 / 8
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
  const int wbase = wave * 1024;  // this wave's first block inside a plane; its second one is NW KiB further
  // DUAL: the second phase's staging geometry, buffers and g registers (requested from the first phase's last hand-offs)
  DualGeom DG = {};
  float cx2[DUAL ? XP : 1], cy2[DUAL ? XP : 1];
  rsrc_t x2B = xB, i2B = xB, g2B = xB;
  if constexpr (DUAL) {
    DG = dual_geom<TH, TW, CROP>(P, Q.C2, y0, x0);
    x2B = mkbuf(Q.ema + (size_t)b * D_T * S);
    i2B = mkbuf(Q.inv_other + (size_t)b * S);
    g2B = mkbuf(Q.g_cross + (size_t)b * P.K * S);
  }
  // SDMA (the 3D instantiation): every wave issues BOTH slots for every plane, unconditionally -- a wave without a second block
  // repeats its first one (same bytes to the same place), the tail lanes of the last block write zeros into the plane's
  // padding.  The count of DMA instructions per chunk is then a compile-time 4, which is what lets the compiler wait for the
  // z gathers with vmcnt(4) instead of vmcnt(0): with a DMA inside an `if` it has to assume that none was issued.
  constexpr bool SDMA = ZP > 0;
  const bool two = __builtin_amdgcn_readfirstlane(((NT / 64) + wave) * 64 < C.QA);
  const unsigned vo1 = SDMA ? (two ? vo[1] : vo[0]) : vo[1];
  const int w1 = SDMA ? (two ? wbase + (NT / 64) * 1024 : wbase) : wbase + (NT / 64) * 1024;
  // DMA instructions this wave issues per chunk (a slot without a live lane is skipped): what `vmcnt` has to count
  const int npc = SDMA ? 4 : 2 * ((__builtin_amdgcn_ballot_w64(act[0]) != 0) + (__builtin_amdgcn_ballot_w64(act[1]) != 0));
  // wait until only the youngest chunk's DMA may still be in flight, then the workgroup barrier
#define PEA_XWAIT1()                                                                             \
  {                                                                                              \
    if (npc == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");       \
    else if (npc == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");  \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                \
  }
  // in the chunk loop of the 3D instantiation the z gathers of the NEXT chunk (2 * ZP loads, issued one iteration earlier) sit
  // between the DMA that has to have landed and the youngest DMA: they may stay in flight too
#define PEA_XWAITZ() asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * ZP + 4) : "memory");
#define PEA_XDMA(rsrc, plane_byte, so)                                                                              \
  {                                                                                                                 \
    if (SDMA) {                                                                                                     \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + (plane_byte) + wbase), 16, vo[0], so, 0, 0); \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + (plane_byte) + w1), 16, vo1, so, 0, 0);      \
    } else {                                                                                                        \
      if (act[0]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + (plane_byte) + wbase), 16, vo[0], so, 0, 0);        \
      if (act[1]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + (plane_byte) + w1), 16, vo1, so, 0, 0); \
    }                                                                                                               \
  }
  PEA_STAMP(0)
  PEA_XDMA(iB, 4 * PS, ezo)
  PEA_XDMA(xB, 0, ezo)
  PEA_XDMA(xB, PS, ezo + ecs)

  // ---- g of every pair (role A at p, role B at p - o) and the LDS slot of every neighbour
  // dead lanes: an offset that stays out of range when a small displacement is added
  const unsigned pg = live ? po4 : 0xC0000000u;
  float cx[XP], cy[XP];
  int ax[XP], ay[XP];
  const int vown = ((C.hy0 + ly) * TW + lx) * 4;
  const int hrow = (C.QV * 4 + ly * C.SW) * 4;
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    const int d = C.xd[k], c = lx + d;
    ax[k] = (unsigned)c < (unsigned)TW ? vown + d * 4 : hrow + (c & C.xm[k]) * 4;
    ay[k] = vown + C.yd[k] * TW * 4;
  }
  // one dword per lane and pair (the lane's own pixel)
#define PEA_XG_DWORDS()                                                                                                              \
  {                                                                                                                                  \
    _Pragma("unroll") for (int k = 0; k < XP; ++k) {                                                                                 \
      const int go = C.xgo[k];                       /* uniform */                                                                   \
      const int t = px + go;                                                                                                         \
      const bool out = (unsigned)t >= (unsigned)P.X; /* either side */                                                               \
      const int fix = go > 0 ? -P.X : P.X;           /* uniform */                                                                   \
      const unsigned o = CROP ? (out ? kOOB : pg + (unsigned)(go * 4)) : pg + (unsigned)((out ? go + fix : go) * 4);                 \
      cx[k] = bl32(gB, k < C.npx ? o : kOOB, ezo + (unsigned)C.xgi[k] * ecs);                                                        \
    }                                                                                                                                \
    _Pragma("unroll") for (int k = 0; k < XP; ++k) {                                                                                 \
      const int go = C.ygo[k];                                                                                                       \
      const int t = py + go;                                                                                                         \
      const bool out = (unsigned)t >= (unsigned)P.Y;                                                                                 \
      const int fix = go > 0 ? -P.Y : P.Y;                                                                                           \
      const unsigned o = CROP ? (out ? kOOB : pg + (unsigned)(go * P.X * 4)) : pg + (unsigned)((out ? go + fix : go) * P.X * 4);     \
      cy[k] = bl32(gB, k < C.npy ? o : kOOB, ezo + (unsigned)C.ygi[k] * ecs);                                                        \
    }                                                                                                                                \
  }
  PEA_XG_DWORDS()
#undef PEA_XG_DWORDS
  // z pairs: g and the neighbour's 1 / norm (another plane, same (y, x): scalar plane offsets); a pair whose plane does not
  // exist gets the coefficient 0 and reads plane z itself
  float cz[ZP > 0 ? ZP : 1];
  unsigned zso[ZP > 0 ? ZP : 1];  // byte offset of the neighbour's plane (scalar)
#pragma unroll
  for (int k = 0; k < ZP; ++k) {
    bool okq, okg;
    const int zq = wrap1<CROP>(z + C.zd[k], P.Z, okq), zg = wrap1<CROP>(z + C.zgo[k], P.Z, okg);
    const bool ok = okq && okg && k < C.npz;
    zso[k] = (unsigned)(ok ? zq : z) * YX * 4u;
    const float gk = bl32(gB, pe, (unsigned)(ok ? zg : z) * YX * 4u + (unsigned)C.zgi[k] * ecs);
    const float iq = bl32(iB, pe, zso[k]);
    cz[k] = ok ? gk * fabsf(iq) : 0.f;
  }
  // OTHER: the own pixel (all channels) and its 1 / norm, requested before the second chunk's DMA like the coefficients
  f2 eo[OTHER ? D_T / 2 : 1];
  float invo_g = 0.f;
  if (OTHER) {
    const rsrc_t oB = mkbuf(O.own + (size_t)b * D_T * S), oiB = mkbuf(O.own_inv + (size_t)b * S);
#pragma unroll
    for (int ps = 0; ps < D_T / 2; ++ps) {
      eo[ps].x = bl32(oB, pe, ezo + (unsigned)(2 * ps) * ecs);
      eo[ps].y = bl32(oB, pe, ezo + (unsigned)(2 * ps + 1) * ecs);
    }
    invo_g = bl32(oiB, pe, ezo);
  }
  PEA_XDMA(xB, 2 * PS, ezo + 2u * ecs)
  PEA_XDMA(xB, 3 * PS, ezo + 3u * ecs)
  // inv, chunk 0 and g have landed (the 4 DMA instructions of chunk 1 may still fly); every wave's share of them too
  PEA_STAMP(1)
  PEA_XWAIT1()
  PEA_STAMP(2)

  // coefficient of a pair = g * 1 / |e(q)|
  const float invo = OTHER ? invo_g : *(const float*)(lds + 4 * PS + vown);
  const float inv_own = fabsf(invo);
#pragma unroll
  for (int k = 0; k < XP; ++k) {
    cx[k] *= fabsf(*(const float*)(lds + 4 * PS + ax[k]));
    cy[k] *= fabsf(*(const float*)(lds + 4 * PS + ay[k]));
    // computed HERE: volatile asm statements keep their order; the scheduler otherwise sinks the whole arithmetic
    // below the last barrier and keeps every LDS value of every chunk in registers (256 VGPRs + spills)
    asm volatile("" : "+v"(cx[k]), "+v"(cy[k]));
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the inv plane is dead: buffer 2 may be filled
  // the z neighbours' two channels of a chunk are requested TWO chunks ahead (two register sets), before the DMA of the chunk
  // three ahead: a gather from another plane is an L2 miss more often than not here (section 5.7 item 9), and one chunk's LDS
  // pairs do not cover that latency.  vmcnt retires in order, so the waits count the younger DMA / gathers exactly.
  f2 zv[2][ZP > 0 ? ZP : 1];
#define PEA_XZLOAD(ch)                                                                  \
  {                                                                                     \
    _Pragma("unroll") for (int k = 0; k < ZP; ++k) {                                    \
      zv[(ch) & 1][k].x = bl32(xB, pe, zso[k] + (unsigned)(2 * (ch)) * ecs);            \
      zv[(ch) & 1][k].y = bl32(xB, pe, zso[k] + (unsigned)(2 * (ch) + 1) * ecs);        \
    }                                                                                   \
  }
  PEA_XZLOAD(0)
  if (NP > 1) PEA_XZLOAD(1)
  if (NP > 2) {
    PEA_XDMA(xB, 4 * PS, ezo + 4u * ecs)
    PEA_XDMA(xB, 5 * PS, ezo + 5u * ecs)
  }
  // D_T <= 16: the lane keeps its own normalised pixel (eh) for the projection at the end; wider embeddings have no
  // registers for it (G alone is D_T of them): <ehat, G> is accumulated chunk by chunk and the own pixel is read again
  // from global memory (an L2 hit: its tile was just staged) for the final (G - ehat <ehat, G>) / n
  constexpr bool KEEP = D_T <= 16;
  f2 G[NP], eh[KEEP ? NP : 1];
  float proj = 0.f;
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int bo = (ps % 3) * 2 * PS;
    f2 o;
    if (OTHER) o = eo[ps];
    else {
      o.x = *(const float*)(lds + bo + vown);
      o.y = *(const float*)(lds + bo + PS + vown);
    }
    o = o * inv_own;
    if (KEEP) {
      eh[ps] = o;
      asm volatile("" : "+v"(eh[ps]));
    }
    f2 acc = {0.f, 0.f};
#ifndef PEA_ABL_NOGATHER  // (diagnostic builds of profiles/microbench/stamp_bwd.hip only)
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      f2 v;
      v.x = *(const float*)(lds + bo + ax[k]);
      v.y = *(const float*)(lds + bo + PS + ax[k]);
      acc = __builtin_elementwise_fma((f2){cx[k], cx[k]}, v, acc);
      if (DUAL || k % 5 == 4) asm volatile("" ::: "memory");  // bound the ds_read hoisting (DUAL: the second phase's 20 g registers are live too)
    }
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      f2 v;
      v.x = *(const float*)(lds + bo + ay[k]);
      v.y = *(const float*)(lds + bo + PS + ay[k]);
      acc = __builtin_elementwise_fma((f2){cy[k], cy[k]}, v, acc);
      if (DUAL || k % 5 == 4) asm volatile("" ::: "memory");
    }
#else
    acc.x = cx[ps % XP] + cy[ps % XP];
#endif
#pragma unroll
    for (int k = 0; k < ZP; ++k) acc = __builtin_elementwise_fma((f2){cz[k], cz[k]}, zv[ps & 1][k], acc);
    if (!KEEP) {
      proj = fmaf(o.x, acc.x, fmaf(o.y, acc.y, proj));
      asm volatile("" : "+v"(proj));
    }
    asm volatile("" : "+v"(acc));  // the chunk's sums exist before its barrier
    G[ps] = acc;
    PEA_STAMP(3 + 3 * ps)
    if (ps + 1 < NP) {
      // chunk ps + 1 has landed (chunk ps + 2, issued after it, may still fly); everyone is done with buffer ps % 3
      if (ps + 2 < NP) {
        if (ZP > 0) PEA_XWAITZ()
        else PEA_XWAIT1()
      } else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      PEA_STAMP(4 + 3 * ps)
      if (ps + 2 < NP) PEA_XZLOAD(ps + 2)
#ifndef PEA_ABL_NODMA
      if (ps + 3 < NP) {
        PEA_XDMA(xB, bo, ezo + (unsigned)(2 * ps + 6) * ecs)
        PEA_XDMA(xB, bo + PS, ezo + (unsigned)(2 * ps + 7) * ecs)
      }
#endif
      if constexpr (DUAL) {  // the second phase's first requests ride in the ring slots the first phase no longer needs (see bwd_phase_role_a)
        const int wbase2 = wbase, w12 = wbase + (NT / 64) * 1024;
        if (ps == NP - 3) {  // buffer 2 is free: the second operand's 1 / norm plane into plane 4, and the cross loss' g values
          PEA_X2DMA(i2B, 4 * PS, ezo)
          const unsigned pg2 = live ? po4 : kOOB;  // role A: g at the own pixel
#pragma unroll
          for (int k = 0; k < XP; ++k) {
            cx2[k] = bl32(g2B, k < Q.C2.npx ? pg2 : kOOB, ezo + (unsigned)Q.C2.xgi[k] * ecs);
            cy2[k] = bl32(g2B, k < Q.C2.npy ? pg2 : kOOB, ezo + (unsigned)Q.C2.ygi[k] * ecs);
          }
        }
        if (ps == NP - 2) {  // buffer 0 is free: the second phase's chunk 0
          PEA_X2DMA(x2B, 0, ezo)
          PEA_X2DMA(x2B, PS, ezo + ecs)
        }
      }
      PEA_STAMP(5 + 3 * ps)
    }
  }
#undef PEA_XDMA
#undef PEA_XWAIT1
#undef PEA_XWAITZ
#undef PEA_XZLOAD

  if constexpr (DUAL) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) G[ps] = G[ps] * dl;
    bwd_phase_role_a<TH, TW, PSU, CROP, XP>(P, Q.C2, lds, x2B, DG, cx2, cy2, Q.dloss_cross ? Q.dloss_cross[0] : 1.f, ezo, ecs, G);
  }
  if (KEEP) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) proj = fmaf(eh[ps].x, G[ps].x, fmaf(eh[ps].y, G[ps].y, proj));
  }
  if (invo < 0.f) proj = 0.f;  // clamp branch of F.normalize
  const float sc = DUAL ? inv_own : dl * inv_own;
  const float pn = proj * inv_own;  // !KEEP: ehat * proj = e * (inv_own * proj)
  f2 old[OTHER ? NP : 1];
  const bool accum = OTHER && O.accumulate != 0;  // uniform
  if (OTHER && accum) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      old[ps].x = bl32(dB, pe, ezo + (unsigned)(2 * ps) * ecs);
      old[ps].y = bl32(dB, pe, ezo + (unsigned)(2 * ps + 1) * ecs);
    }
  }
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    // (G - ehat <ehat, G>) / n with ONE rounding of the difference in every channel: left to the compiler, one channel of sixteen came out
    // as multiply + subtract (packed with the scale's product) and the others fused
    float vx, vy;
    if (KEEP) { vx = __builtin_fmaf(-eh[ps].x, proj, G[ps].x) * sc; vy = __builtin_fmaf(-eh[ps].y, proj, G[ps].y) * sc; }
    else {
      vx = __builtin_fmaf(-bl32(xB, pe, ezo + (unsigned)(2 * ps) * ecs), pn, G[ps].x) * sc;
      vy = __builtin_fmaf(-bl32(xB, pe, ezo + (unsigned)(2 * ps + 1) * ecs), pn, G[ps].y) * sc;
    }
    if (OTHER && accum) {
      // the product is a rounded f32 BEFORE the sum (left alone the compiler contracts x * sc + old into one FMA): de then holds, bit
      // for bit, what the two launches and an add over [B,D,...] give (tests/test_gpu_zmarch.py::test_ac3ac4_section_backward_runs_the_march)
      asm volatile("" : "+v"(vx), "+v"(vy));
      vx += old[ps].x; vy += old[ps].y;
    }
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vx), dB, pe, ezo + (unsigned)(2 * ps) * ecs, AUXS);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vy), dB, pe, ezo + (unsigned)(2 * ps + 1) * ecs, AUXS);
  }
  PEA_STAMP(kStampLast)
}

// ------------------------------------------------------------------------------------------------------------------
// forward (self loss / inference), same staging: role A only, so the cross is one-sided (CVPPP: 27 rows up, one strip).
// The channels arrive RAW, two per chunk; every lane accumulates, per offset, the raw dot product with its own pixel
// and the neighbour's sum of squares (packed: the two channels of the chunk side by side), and normalises at the end:
// <ehat(p), ehat(q)> = <e(p), e(q)> / (|e(p)| |e(q)|).  The lane's own 1 / norm goes to the plane the backward stages.
// Epilogue as k_fwd_tiled_v: the K dot products are parked in LDS as [offset][tile pixel] (over the dead ring) and walked
// four x-adjacent pixels per lane, so target / weight / affs / g are dwordx4 and the four masks one dword.
// Needs K <= kXP, X % 4 == 0, 16-byte aligned planes.
// ------------------------------------------------------------------------------------------------------------------
// ZF: z offsets read from global memory (0: 2D; kXZ / 2: the 3D instantiation, which also issues its DMA unconditionally --
// SDMA, see k_bwd_xdma -- and requests target / weight / mask only after the channel loop: their registers go to the z sums)
// OTHER: the cross loss a_i(p) = <ehat(p), ehat_other(p + o_i)> (ema_embedding_loss): `e` is the SECOND operand (staged: the
//   neighbours), the own pixel comes from `own` (global loads, all channels up front); both 1 / norm planes are written
//   (inv_out: own, inv_other_out: the second operand's, from the staged centre) for the role-A backward
// WPE: waves per SIMD the register budget is cut for: 4 = two workgroups per CU (13 KB planes: the backward's geometry),
//   6 = THREE workgroups per CU -- the forward's one-sided cross needs 7.5 KB planes (6 x 7680 B = 45 KB of ring), so a third
//   workgroup fits the LDS if the kernel stays within 80 VGPRs: target / weight / mask are then requested after the channel loop
//   (LATE) instead of being held across it
// LAB: target / mask / weight come from a LABEL image (gen_affs_ours, scripts_cvppp/utils/affinity_ours.py
//   :17-39, and weight_binary_ratio as two scalars per (image, channel), pea_label_weights) instead of three tensors: the label
//   plane is staged like one more channel of the cross (a SEVENTH plane, alive until the loss is evaluated), so the neighbour's
//   label sits at the LDS slot of the neighbour's embedding.  The loss term and g are evaluated per PIXEL right after the channel
//   loop (own label, K neighbour labels: 11 LDS reads), a AND g are parked in LDS and leave as dwordx4 rows.  No t / w / m
//   traffic at all: 4D + 4 + 8K + 4 bytes per pixel.  2D only.
struct LabArgs {
  const int32_t* labels;  // [B, Z, Y, X]
  const float* wtab;      // [B, K, 2]: weight of target-1 pixels, of target-0 pixels
  unsigned lflags;        // PEA_TGT_*
};
template <int D_T, int TH, int TW, int PSU, bool CROP, bool TRAIN, int ZF = 0, bool OTHER = false, int WPE = 4, bool LAB = false>
__global__ __launch_bounds__(TH* TW, WPE) void k_fwd_xdma(const KParams P, const XParams C, const float* __restrict__ e,
                                                           const float* __restrict__ target, const float* __restrict__ weight,
                                                           const uint8_t* __restrict__ mask, float* __restrict__ affs,
                                                           float* __restrict__ gout, LossState* __restrict__ st,
                                                           float* __restrict__ inv_out,
                                                           const float* __restrict__ own, float* __restrict__ inv_other_out,
                                                           const LabArgs LA) {
  // OWNL: the cross loss at D > 16 -- the own pixel's D registers do not exist, so the OWN tile (16 x 32 pixels, 2 KB per channel, no
  // halo) is staged beside each chunk by the first two waves (one dwordx4 DMA instruction per channel and wave) and read from LDS
  constexpr bool OWNL = OTHER && D_T > 16;
  static_assert(!OWNL || ZF == 0, "cross-loss instantiation at D > 16: in-plane");
  static_assert(!LAB || (TRAIN && ZF == 0 && !OTHER), "labels-in instantiation: 2D self loss");
  constexpr int NT = TH * TW, PS = PSU * 256, NP = D_T / 2, TP = NT, QP = TP / 4, NSL = QP / 64;
  constexpr int KMAX = ZF > 0 ? kXP + 2 : kXP;      // channels the epilogue handles (norm5: 8 in-plane + 4 z offsets)
  constexpr int ITEMS = (KMAX * QP + NT - 1) / NT;
  constexpr bool SDMA = ZF > 0, LATE = ZF > 0 || OTHER || WPE > 4;  // OTHER: the own pixel's registers instead of the early t / w / m
  // the epilogue's item geometry after the channel loop: the 80-VGPR instantiations, and the 3D inference one (which spilled 13
  // registers with the items alive across the loop -- profiles/kernel_resources.py)
  constexpr bool ILATE = WPE > 4 || (ZF > 0 && !TRAIN);
  static_assert(TW == 32 && D_T % 2 == 0 && QP % 64 == 0, "lane mapping / channel pairs");
  constexpr int NW = NT / 64;
  static_assert(KMAX * TP * 4 + KMAX * NSL * 4 <= 6 * PS && KMAX <= kXK, "the parked dot products fit the ring");

  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  float* sA = (float*)lds;                          // [K][TP] dot products, laid over the ring once it is dead
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
  const unsigned ecs = (unsigned)P.S * 4u, ezo = (unsigned)z * YX * 4u;
  const bool has_a = affs != nullptr, has_g = gout != nullptr, has_m = mask != nullptr;
  const unsigned af = P.flags & kActMask;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;

  // ---- (0) the epilogue's operands: item = (offset, quad of 4 x-adjacent tile pixels); requested first
  bool ion[ITEMS];
  unsigned ivo[ITEMS];
  int iqd[ITEMS], igy[ITEMS], igx[ITEMS], isl[ITEMS];
  f4 t4[ITEMS], w4[ITEMS];
  unsigned m4[ITEMS];
  // (the 80-VGPR instantiations evaluate this after the channel loop: three registers less across it)
#define PEA_XITEMS()                                                                                                      \
  int tid_i = (int)threadIdx.x;                                                                                           \
  if (ILATE) asm volatile("" : "+v"(tid_i)); /* opaque: not hoisted back over the channel loop */                        \
  _Pragma("unroll") for (int it = 0; it < ITEMS; ++it) {                                                                  \
    const int tt = it * NT + tid_i;                                                                                       \
    const int sl = __builtin_amdgcn_readfirstlane(tt / QP);                                                               \
    ion[it] = sl < P.K;                                                                                                   \
    isl[it] = min(sl, P.K - 1);                                                                                           \
    const int qd = tt - (tt / QP) * QP;                                                                                   \
    iqd[it] = qd;                                                                                                         \
    const int l4 = qd * 4;                                                                                                \
    igy[it] = y0 + l4 / TW;                                                                                               \
    igx[it] = x0 + l4 % TW;                                                                                               \
    const bool lv = ion[it] && igy[it] < P.Y && igx[it] < P.X; /* X % 4 == 0: a quad is inside or outside as a whole */   \
    ivo[it] = lv ? (unsigned)(igy[it] * P.X + igx[it]) * 4u : kOOB;                                                       \
  }
  if (!ILATE) { PEA_XITEMS() }
#define PEA_XLOAD_TWM()                                                                                                   \
  if (TRAIN) {                                                                                                            \
    _Pragma("unroll") for (int it = 0; it < ITEMS; ++it) {                                                                \
      const unsigned so = ezo + (unsigned)isl[it] * ecs;                                                                  \
      t4[it] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(tB, ivo[it], so, kAuxNT));                    \
      w4[it] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(wB, ivo[it], so, kAuxNT));                    \
      m4[it] = has_m ? __builtin_amdgcn_raw_buffer_load_b32(mB, ivo[it] == kOOB ? kOOB : ivo[it] >> 2,                    \
                                                           (ezo >> 2) + (unsigned)isl[it] * (unsigned)P.S, kAuxNT)        \
                     : 0x01010101u;                                                                                       \
    }                                                                                                                     \
  }
  if (!LATE && !LAB) PEA_XLOAD_TWM()

  const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
  const int py = y0 + ly, px = x0 + lx;
  const bool live = py < P.Y && px < P.X;
  const unsigned pe = live ? (unsigned)(py * P.X + px) * 4u : kOOB;

  // ---- the (up to) two quads this lane moves per plane
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
  const int wbase = wave * 1024;
  const bool two = __builtin_amdgcn_readfirstlane(((NT / 64) + wave) * 64 < C.QA);
  const unsigned vo1 = SDMA ? (two ? vo[1] : vo[0]) : vo[1];
  const int w1 = SDMA ? (two ? wbase + (NT / 64) * 1024 : wbase) : wbase + (NT / 64) * 1024;
  const int npc = SDMA ? 4 : 2 * ((__builtin_amdgcn_ballot_w64(act[0]) != 0) + (__builtin_amdgcn_ballot_w64(act[1]) != 0));
  f2 eo[(OTHER && !OWNL) ? NP : 1];  // OTHER: the own pixel, requested before the first DMA (vmcnt retires in order)
  const rsrc_t ownB = mkbuf(OWNL ? own + (size_t)b * D_T * S : nullptr);
  constexpr int OWNB = 6 * PS;  // OWNL: three buffers x two channels x 2 KB behind the ring
  unsigned ownvo = kOOB;        // OWNL: the quad of the own tile this lane moves (waves 0 and 1)
  if (OWNL) {
    const int q = (int)threadIdx.x;  // quads 0 .. 127 of the tile: row q / 8, columns 4 (q % 8) ..
    const int qy = y0 + (q >> 3), qx = x0 + 4 * (q & 7);
    ownvo = (q < NT / 4 && qy < P.Y && qx < P.X) ? (unsigned)(qy * P.X + qx) * 4u : kOOB;
  }
  const bool ownw = OWNL && wave < NT / 256;  // uniform: this wave moves own tiles
#define PEA_XOWN(ch)                                                                                                          \
  if (ownw) {                                                                                                                 \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ownB, (lds_ptr_t)(lds + OWNB + (((ch) % 3) * 2) * 2048 + wave * 1024), 16, ownvo,   \
                                             ezo + (unsigned)(2 * (ch)) * ecs, 0, 0);                                          \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ownB, (lds_ptr_t)(lds + OWNB + (((ch) % 3) * 2 + 1) * 2048 + wave * 1024), 16, ownvo, \
                                             ezo + (unsigned)(2 * (ch) + 1) * ecs, 0, 0);                                      \
  }
  if (OTHER && !OWNL) {
    const rsrc_t oB = mkbuf(own + (size_t)b * D_T * S);
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      eo[ps].x = bl32(oB, pe, ezo + (unsigned)(2 * ps) * ecs);
      eo[ps].y = bl32(oB, pe, ezo + (unsigned)(2 * ps + 1) * ecs);
    }
  }
#define PEA_XDMA(plane_byte, so)                                                                                    \
  {                                                                                                                 \
    if (SDMA) {                                                                                                     \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(lds + (plane_byte) + wbase), 16, vo[0], so, 0, 0);   \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(lds + (plane_byte) + w1), 16, vo1, so, 0, 0);        \
    } else {                                                                                                        \
      if (act[0]) __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(lds + (plane_byte) + wbase), 16, vo[0], so, 0, 0);        \
      if (act[1]) __builtin_amdgcn_raw_ptr_buffer_load_lds(xB, (lds_ptr_t)(lds + (plane_byte) + w1), 16, vo1, so, 0, 0); \
    }                                                                                                               \
  }
  const int npt = npc + (ownw ? 2 : 0);  // DMA instructions per chunk of this wave, the own tile's included
#define PEA_XWAIT1()                                                                             \
  {                                                                                              \
    if (npt == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");       \
    else if (npt == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");  \
    else if (npt == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");  \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                \
  }
#define PEA_XWAITZ() asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * ZF + 4) : "memory");
  if (LAB) {  // the label plane: same geometry as a channel plane (int32, [Y][X]), its own (seventh) LDS plane
    const rsrc_t lB = mkbuf(LA.labels + (size_t)b * S);
    if (act[0]) __builtin_amdgcn_raw_ptr_buffer_load_lds(lB, (lds_ptr_t)(lds + 6 * PS + wbase), 16, vo[0], ezo, 0, 0);
    if (act[1]) __builtin_amdgcn_raw_ptr_buffer_load_lds(lB, (lds_ptr_t)(lds + 6 * PS + w1), 16, vo1, ezo, 0, 0);
  }
  PEA_XDMA(0, ezo)
  PEA_XDMA(PS, ezo + ecs)
  PEA_XOWN(0)
  if (NP > 1) {
    PEA_XDMA(2 * PS, ezo + 2u * ecs)
    PEA_XDMA(3 * PS, ezo + 3u * ecs)
    PEA_XOWN(1)
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
  if (NP > 1) PEA_XWAIT1()
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // z offsets: the neighbour is the same (y, x) in plane z + dz (wave-uniform plane and validity); requested two chunks ahead
  // into two register sets, used after the chunk's LDS offsets (k_bwd_xdma)
  unsigned zso[ZF > 0 ? ZF : 1];
  bool zok[ZF > 0 ? ZF : 1];
#pragma unroll
  for (int k = 0; k < ZF; ++k) {
    bool okq;
    const int zq = wrap1<CROP>(z + C.fzd[k], P.Z, okq);
    zok[k] = okq && k < C.nfz;
    zso[k] = (unsigned)(zok[k] ? zq : z) * YX * 4u;
  }
  f2 zv[2][ZF > 0 ? ZF : 1];
#define PEA_XZLOAD(ch)                                                                  \
  {                                                                                     \
    _Pragma("unroll") for (int k = 0; k < ZF; ++k) {                                    \
      zv[(ch) & 1][k].x = bl32(xB, pe, zso[k] + (unsigned)(2 * (ch)) * ecs);            \
      zv[(ch) & 1][k].y = bl32(xB, pe, zso[k] + (unsigned)(2 * (ch) + 1) * ecs);        \
    }                                                                                   \
  }
  PEA_XZLOAD(0)
  if (NP > 1) PEA_XZLOAD(1)
  if (NP > 2) {
    PEA_XDMA(4 * PS, ezo + 4u * ecs)
    PEA_XDMA(5 * PS, ezo + 5u * ecs)
    PEA_XOWN(2)
  }

  f2 dot[kXP], ssq[kXP], oss = {0.f, 0.f}, css = {0.f, 0.f}, dotz[ZF > 0 ? ZF : 1], ssqz[ZF > 0 ? ZF : 1];
#pragma unroll
  for (int k = 0; k < kXP; ++k) { dot[k] = (f2){0.f, 0.f}; ssq[k] = (f2){0.f, 0.f}; }
#pragma unroll
  for (int k = 0; k < ZF; ++k) { dotz[k] = (f2){0.f, 0.f}; ssqz[k] = (f2){0.f, 0.f}; }
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int bo = (ps % 3) * 2 * PS;
    f2 o;
    o.x = *(const float*)(lds + bo + vown);
    o.y = *(const float*)(lds + bo + PS + vown);
    if (OTHER) {  // the staged centre is the second operand's pixel: its norm goes to the backward; the own pixel is eo / the own tile
      css = __builtin_elementwise_fma(o, o, css);
      if (OWNL) {
        o.x = *(const float*)(lds + OWNB + ((ps % 3) * 2) * 2048 + (int)threadIdx.x * 4);
        o.y = *(const float*)(lds + OWNB + ((ps % 3) * 2 + 1) * 2048 + (int)threadIdx.x * 4);
      } else {
        o = eo[ps];
      }
    }
    oss = __builtin_elementwise_fma(o, o, oss);
#pragma unroll
    for (int k = 0; k < kXP; ++k) {
      f2 v;
      v.x = *(const float*)(lds + bo + an[k]);
      v.y = *(const float*)(lds + bo + PS + an[k]);
      dot[k] = __builtin_elementwise_fma(o, v, dot[k]);
      ssq[k] = __builtin_elementwise_fma(v, v, ssq[k]);
      if (k % (ZF > 0 && !TRAIN ? 2 : 5) == (ZF > 0 && !TRAIN ? 1 : 4)) asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int k = 0; k < ZF; ++k) {
      dotz[k] = __builtin_elementwise_fma(o, zv[ps & 1][k], dotz[k]);
      ssqz[k] = __builtin_elementwise_fma(zv[ps & 1][k], zv[ps & 1][k], ssqz[k]);
    }
#pragma unroll
    for (int k = 0; k < kXP; ++k) asm volatile("" : "+v"(dot[k]), "+v"(ssq[k]));  // the chunk's sums exist before its barrier
#pragma unroll
    for (int k = 0; k < ZF; ++k) asm volatile("" : "+v"(dotz[k]), "+v"(ssqz[k]));
    asm volatile("" : "+v"(oss));
    if (OTHER) asm volatile("" : "+v"(css));
    if (ps + 1 < NP) {
      if (ps + 2 < NP) {
        if (ZF > 0) PEA_XWAITZ()
        else PEA_XWAIT1()
      } else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (ps + 2 < NP) PEA_XZLOAD(ps + 2)
      if (ps + 3 < NP) {
        PEA_XDMA(bo, ezo + (unsigned)(2 * ps + 6) * ecs)
        PEA_XDMA(bo + PS, ezo + (unsigned)(2 * ps + 7) * ecs)
        PEA_XOWN(ps + 3)
      }
    }
  }
#undef PEA_XDMA
#undef PEA_XOWN
#undef PEA_XWAIT1
#undef PEA_XWAITZ
#undef PEA_XZLOAD
  if (LATE && !ILATE && !LAB) PEA_XLOAD_TWM()

  // ---- normalise; the lane's own 1 / norm for the backward
  const float osum = oss.x + oss.y;
  const float inv_eps = 1.0f / P.eps;
  const float inv_own = rnorm(osum, inv_eps);
  if (inv_out) bs32(iB, osum < P.eps * P.eps ? -inv_own : inv_own, pe, ezo);
  if (OTHER && inv_other_out) {
    const float csum = css.x + css.y;
    const float inv_c = rnorm(csum, inv_eps);
    bs32(mkbuf(inv_other_out + (size_t)b * S), csum < P.eps * P.eps ? -inv_c : inv_c, pe, ezo);
  }
  lds_barrier();  // every lane is done with the ring: sA goes over it
  {
#pragma unroll
  for (int k = 0; k < kXP; ++k) {
    if (k < C.nf) {  // uniform
      float a = (dot[k].x + dot[k].y) * inv_own * rnorm(ssq[k].x + ssq[k].y, inv_eps);
      if (CROP) {
        const int q = (C.fax[k] ? px : py) + C.fd[k];
        a = (unsigned)q < (unsigned)(C.fax[k] ? P.X : P.Y) ? a : 0.f;
      }
      sA[C.fi[k] * TP + (int)threadIdx.x] = a;
    }
  }
  }
#pragma unroll
  for (int k = 0; k < ZF; ++k) {
    if (k < C.nfz) {
      const float a = zok[k] ? (dotz[k].x + dotz[k].y) * inv_own * rnorm(ssqz[k].x + ssqz[k].y, inv_eps) : 0.f;
      sA[C.fzi[k] * TP + (int)threadIdx.x] = a;
    }
  }
  if (ILATE && !LAB) { PEA_XITEMS() }
  if (LATE && ILATE && !LAB) PEA_XLOAD_TWM()  // 80-VGPR budget: only now are the 40 accumulator registers free
#undef PEA_XLOAD_TWM
  lds_barrier();
  if (ILATE && LAB) { PEA_XITEMS() }
#undef PEA_XITEMS

  // ---- epilogue: 4 x-adjacent pixels of one offset per lane, dwordx4 everywhere
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    if (!ion[it]) continue;  // wave-uniform
    const int sl = isl[it];
    const f4 a4 = *(const f4*)(sA + sl * TP + iqd[it] * 4);
    const unsigned so = ezo + (unsigned)sl * ecs;
    if constexpr (LAB) {
      // target, mask, weight of the quad's four pairs from their labels (pea_fused_labels.h PEA_LAB_PAIR: the same rules, the same
      // order of operations) -- here in the store walk, per quad like the tensor path's epilogue, not per pixel after the channel
      // loop (that form spent 47 % more vector-ALU instructions than the tensor forward and parked g beside a).  The neighbour's
      // label is UNWRAPPED: a neighbour outside the image has none (inside = false).
      const bool pad = LA.lflags & PEA_TGT_PADDING, fg = LA.lflags & PEA_TGT_BOTH_FOREGROUND, msk = LA.lflags & PEA_TGT_MASK_INSIDE;
      const float* wt_b = LA.wtab + 2 * (size_t)b * P.K;
      const float w1 = wt_b[2 * sl], w0 = wt_b[2 * sl + 1], gs = C.gs[sl];
      const int ax_ = C.oax[sl], od_ = C.od[sl];
      const int l4 = iqd[it] * 4, qr = l4 / TW, qc = l4 % TW;
      const char* const lp = lds + 6 * PS;
      const int rown = ((C.hy0 + qr) * TW) * 4;
      typedef int i4 __attribute__((ext_vector_type(4)));
      const i4 lo = *(const i4*)(lp + rown + qc * 4);
      i4 ln;
      if (ax_ == 1) {  // uniform: along x, element by element (inside the tile row or in its strip)
        const int hr = (C.QV * 4 + qr * C.SW) * 4, fm = od_ < 0 ? C.SW - 1 : 31;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cc = qc + j + od_;
          ln[j] = *(const int*)(lp + ((unsigned)cc < (unsigned)TW ? rown + cc * 4 : hr + (cc & fm) * 4));
        }
      } else {
        ln = *(const i4*)(lp + rown + od_ * TW * 4 + qc * 4);
      }
      const bool lv = ivo[it] != kOOB;
      float acc = 0.f;
      f4 g4, o4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = (ax_ == 1 ? igx[it] + j : igy[it]) + od_;
        const bool inside = (unsigned)q < (unsigned)(ax_ == 1 ? P.X : P.Y);
        const bool eq = lo[j] == ln[j] && (!fg || (lo[j] > 0 && ln[j] > 0));
        const float t = (inside ? eq : pad) ? 1.f : 0.f;
        const float m = (msk && !inside) ? 0.f : 1.f;
        const float w = t != 0.f ? w1 : w0;
        const bool exists = lv && (!CROP || inside);
        const float a = exists ? a4[j] : 0.f;
        const float rr = a * m - t * m;
        const float wr = exists ? w * rr : 0.f;
        o4[j] = act_affs(a, af);
        g4[j] = gs * wr * m;
        acc = fmaf(wr, rr, acc);
      }
      if (has_a) bs128<true>(aB, o4, ivo[it], so);
      if (has_g) bs128<false>(gB, g4, ivo[it], so);
      const float red = wave_sum63(acc);
      if ((threadIdx.x & 63) == 63) s_part[sl * NSL + (iqd[it] >> 6)] = red;
      continue;
    }
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
        if (CROP) {  // a cropped-away neighbour carries no loss term (its a is already 0)
          const int q = (ax_ == 1 ? igx[it] + j : ax_ == 0 ? igy[it] : z) + od_;
          wr = (unsigned)q < (unsigned)(ax_ == 1 ? P.X : ax_ == 0 ? P.Y : P.Z) ? wr : 0.f;
        }
        g4[j] = gs * wr * m;
        acc = fmaf(wr, r, acc);
      }
      if (has_g) bs128<false>(gB, g4, ivo[it], so);
      const float red = wave_sum63(acc);
      if ((threadIdx.x & 63) == 63) s_part[sl * NSL + (iqd[it] >> 6)] = red;
    }
  }
  if (TRAIN) {
    lds_barrier();
    if (wave == 0) {  // the one wave that touches the loss state
      if ((int)threadIdx.x < P.K) {
        float v = 0.f;
#pragma unroll
        for (int s = 0; s < NSL; ++s) v += s_part[threadIdx.x * NSL + s];
        loss_accumulate(st, tile, threadIdx.x, v);
      }
    }
  }
}

// host: the plan.  false = not an axis-aligned in-plane stencil that fits (the caller falls back to k_bwd_tiled)
// fwd = true: role A only (one-sided halos, offsets in their own order for the forward kernel).
// Offsets along z (3D volumes) are allowed: they are not staged but gathered per chunk (npz / nfz > 0 selects the 3D
// instantiations, which issue their DMA unconditionally: the plane must then hold whole 64-quad blocks).
// mode 0: backward, both roles (self loss); 1: forward (role A, one-sided cross); 2: backward, role A only (the detached-EMA
// cross loss: the neighbours are the second operand's, the cross is the forward's one-sided one; in-plane stencils only)
inline bool plan_xdma(const KParams& P, int TH, int TW, int psu, XParams* out, size_t* lds_bytes, int mode = 0) {
  const bool fwd = mode == 1, role_a = mode == 2;
  if (P.border == PEA_BORDER_REPLICATE) return false;
  if ((long long)P.Y * P.X >= (1LL << 28)) return false;                           // plane byte offsets + displacement < 2^31
  if ((long long)(P.D > P.K ? P.D : P.K) * P.S * 4 >= (1LL << 31)) return false;   // soffset counts in the range check (plan_tiles)
  if (P.X % 4 || P.S % 4) return false;                                            // quads never straddle a row end
  XParams C = {};
  int hx = 0, hy = 0;
  if (P.K > kXK) return false;
  int up = 0, down = 0, left = 0, right = 0;
  for (int i = 0; i < P.K; ++i) {
    const int oz = P.off[i][0], oy = P.off[i][1], ox = P.off[i][2];
    if ((oz != 0) + (oy != 0) + (ox != 0) != 1) return false;  // exactly one component: axis-aligned
    C.gs[i] = P.gscale[i];
    C.oax[i] = ox != 0 ? 1 : (oy != 0 ? 0 : 2);
    C.od[i] = ox != 0 ? ox : (oy != 0 ? oy : oz);
    if (oz != 0) {
      if (oz <= -P.Z || oz >= P.Z) return false;
      if (role_a) {  // the detached second operand's backward: role A only (neighbour plane z + oz of the second operand, g at z)
        if (C.npz + 1 > kXZ / 2) return false;
        C.zd[C.npz] = oz; C.zgi[C.npz] = i; C.zgo[C.npz] = 0; ++C.npz;
      } else if (fwd) {
        if (C.nfz >= kXZ / 2) return false;
        C.fzd[C.nfz] = oz; C.fzi[C.nfz] = i; ++C.nfz;
      } else {
        if (C.npz + 2 > kXZ) return false;
        C.zd[C.npz] = oz; C.zgi[C.npz] = i; C.zgo[C.npz] = 0; ++C.npz;     // role A: neighbour plane z + oz, g at z
        C.zd[C.npz] = -oz; C.zgi[C.npz] = i; C.zgo[C.npz] = -oz; ++C.npz;  // role B: neighbour plane z - oz, g at z - oz
      }
      continue;
    }
    up = oy < -up ? -oy : up; down = oy > down ? oy : down; left = ox < -left ? -ox : left; right = ox > right ? ox : right;
    if (fwd) {
      if (C.nf >= kXP) return false;
      C.fd[C.nf] = ox != 0 ? ox : oy; C.fax[C.nf] = ox != 0; C.fi[C.nf] = i; ++C.nf;
      continue;
    }
    if (ox != 0) {
      if (C.npx + 2 > kXP) return false;
      hx = ox < 0 ? (hx > -ox ? hx : -ox) : (hx > ox ? hx : ox);
      C.xd[C.npx] = ox; C.xgi[C.npx] = i; C.xgo[C.npx] = 0; ++C.npx;     // role A: neighbour p + o, g at p
      if (!role_a) { C.xd[C.npx] = -ox; C.xgi[C.npx] = i; C.xgo[C.npx] = -ox; ++C.npx; }  // role B: neighbour p - o, g at p - o
    } else {
      if (C.npy + 2 > kXP) return false;
      hy = oy < 0 ? (hy > -oy ? hy : -oy) : (hy > oy ? hy : oy);
      C.yd[C.npy] = oy; C.ygi[C.npy] = i; C.ygo[C.npy] = 0; ++C.npy;
      if (!role_a) { C.yd[C.npy] = -oy; C.ygi[C.npy] = i; C.ygo[C.npy] = -oy; ++C.npy; }
    }
  }
  const bool has_z = C.npz > 0 || C.nfz > 0;
  if (has_z && P.Z > 1) C.zrun = P.Z;
  C.zgy = 4; C.zgx = 2;
  if (fwd || role_a) { hx = left > right ? left : right; C.hy0 = up; C.hy1 = down; }
  else { C.hy0 = C.hy1 = hy; left = right = hx; }
  if (hx > TW) return false;  // a neighbour column is inside the tile or in the strip next to it
  if (left > 0 && right > 0) { C.SW = hx <= 16 ? 32 : 64; C.split = C.SW / 2; }
  else { C.SW = 32; C.split = right > 0 ? 32 : 0; }  // one strip (or none: still one row of 32 per tile row)
  for (int k = 0; k < kXP; ++k) C.xm[k] = C.xd[k] < 0 ? C.SW - 1 : TW - 1;
  for (int k = 0; k < kXP; ++k) C.fm[k] = C.fd[k] < 0 ? C.SW - 1 : TW - 1;
  C.QV = (C.hy0 + TH + C.hy1) * TW / 4;
  C.QA = C.QV + TH * C.SW / 4;
  const int nw = TH * TW / 64;
  if (C.QA > 2 * nw * 64) return false;              // two quads per lane and plane
  const int plane_bytes = has_z ? (C.QA + 63) / 64 * 1024 : C.QA * 16;  // unconditional DMA writes whole blocks
  if (plane_bytes > psu * 256) return false;         // the plane holds the region
  // the kernels wrap with one conditional add
  if (P.Y < TH + C.hy1 || P.Y < C.hy0 || P.X < TW + C.SW || P.X < C.SW) return false;
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
