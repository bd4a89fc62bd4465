// pea_phased.h -- "phase machine" backward: every byte of embedding traffic goes through 16-byte staging.
// EXPERIMENTAL (opt-in with PEA_BWD_PHASED=1; parity-tested, tests/test_gpu_parity.py): correct, but at the CVPPP
// bench shape it takes 255 us against 162 us for k_bwd_tiled.  Kept because the measurements behind it shape the
// next design:
//   * a persistent workgroup (1024 lanes, one per CU) walks its tiles of 32x32 pixels; per tile it runs a short
//     list of PHASES planned on the host (pea_plan.h).  A phase stages one or two rectangular BLOCKS of normalised
//     embeddings into LDS -- the tile plus a vertical or horizontal halo for a group of near offsets, or the tile
//     shifted by a far offset -- and then serves every (offset, role) pair whose neighbour lives in those blocks;
//   * staging works on items = (4 x-adjacent pixels, 4 channels): four dwordx4 loads per item, the squared norm
//     reduced over the 4 lanes of a pixel quad with two DPP adds, four ds_write_b128 in the [slot][pixel] float4
//     layout (odd plane stride => conflict-free); rows of a block are 32 or 64 pixels, so quad q of a phase lands
//     at LDS pixel 4q and a lane's LDS address never changes;
//   * the items of the NEXT phase (or of the next tile's first phase) are requested before the current phase
//     computes and wait in registers; barriers are LDS-only (lds_barrier) so they do not drain those loads;
//   * the own pixel comes out of LDS with its 1/norm from a side buffer; the result leaves through an LDS
//     transpose as dwordx4 rows.
// What the s_memtime stamps (pea_debug_stamps, profiles/stamps.py) showed: the vector-memory pipe of a CU accepts
// L2-hit dwordx4 loads at ~27 B/clk (one-dword loads: ~14 B/clk) and a wave BLOCKS at issue while the queue is
// full: the 160 KB a phase requests take ~6000 cycles to issue, during which the issuing waves cannot compute;
// then write (1.2-2k cycles), two barriers and the latency tail leave the pipe idle.  With 135 KB of LDS per
// workgroup no second workgroup fits on the CU to fill those gaps, and one phase of register prefetch is all that
// 128 VGPRs allow.  The staged volume (7.6 pixels per tile pixel) puts an L2->L1 floor of ~80 us under the design.
#pragma once
#include "pea_tiled.h"

namespace pea {

constexpr int kPhTH = 32, kPhTW = 32, kPhNT = kPhTH * kPhTW;
constexpr int kPhPlq = 2049;   // LDS plane stride in pixels: 2048 staged pixels per phase, odd => conflict-free ds_write_b128
constexpr int kPhQuads = 512;  // pixel quads a phase can stage
constexpr int kMaxPair = 64, kMaxPhase = 16;
constexpr int kPChunk = 8;     // pairs whose g values are held in registers at a time
enum : int { BLK_UNALIGNED = 1, BLK_SRC_X = 2, BLK_OWN = 4 };

// A phase stages `nquads` pixel quads as rows of 2^wsh quads (32 or 64 pixels), one or two blocks stacked
// vertically, into LDS pixels [0, 4*nquads): quad q of the phase lands at LDS pixel 4q, so a lane's LDS address
// never changes.  Block 1 (if any) starts at row `rows0`.
struct MPhase {
  int wsh, nquads, rows0, nblk;
  int dz0, y00, x00, fl0;  // block 0: plane offset, origin relative to the tile origin, BLK_* flags
  int dz1, y01, x01, fl1;  // block 1: y01 has rows0 already subtracted
  int pair0, npair;
  int own_r, own_c;        // BLK_OWN block: phase row / pixel column of the tile's first pixel
};
struct MPair {  // one (offset, role): G(p) += g * nbhat(p + delta)
  int e_off;    // LDS byte offset (slot plane 0) of the neighbour of lane (0, 0)
  int g_so;     // byte offset of plane i inside one batch item of g
  int gyx;      // where g is sampled relative to p: (gdy << 16) | (gdx & 0xffff)   (0 for role A, -o for role B)
  int gz;       // gdz
};
struct MParams {
  int Z, Y, X, S, K;
  unsigned flags;
  float eps, inv_eps;
  int tiles_x, tiles_per_plane, ntiles, tiles_per_xcd;
  unsigned m_tpp, m_tx, m_z;  // floor(2^32 / d) + 1 for d = tiles_per_plane, tiles_x, Z
  int wg_per_xcd;             // persistent workgroups per XCD group (grid = 8 * wg_per_xcd)
  int nphase;
  int own_off;                // own pixel in the LAST phase: LDS byte offset of lane (0,0)
  MPhase ph[kMaxPhase];
  MPair pair[kMaxPair];
};

// ---- 4 x-adjacent pixels of one channel -------------------------------------------------------------
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ f4 ld_quad(rsrc_t r, unsigned vo, unsigned so) {
  if (sizeof(T) == 4) return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0));
  const h4 h = __builtin_bit_cast(h4, __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0));
  f4 o;
  o.x = (float)h.x; o.y = (float)h.y; o.z = (float)h.z; o.w = (float)h.w;
  return o;
}
template <typename T>
__device__ __forceinline__ void st_quad(rsrc_t r, f4 v, unsigned vo, unsigned so) {
  if (sizeof(T) == 4) {
    bs128<false>(r, v, vo, so);
  } else {
    h4 h;
    h.x = (_Float16)v.x; h.y = (_Float16)v.y; h.z = (_Float16)v.z; h.w = (_Float16)v.w;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, h), r, vo, so, 0);
    asm volatile("s_nop 1" ::: "memory");
  }
}

// sum over the 4 lanes of a quad (lanes 4q .. 4q+3), result in all four
__device__ __forceinline__ float quad_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
  return v;
}

// n / d with m = floor(2^32 / d) + 1 (exact while n * d < 2^32); m == 0 encodes d == 1
__device__ __forceinline__ int udiv_magic(int n, unsigned m) { return m ? (int)__umulhi((unsigned)n, m) : n; }

struct TileCtx {  // uniform per tile
  int b, z, y0, x0;
};
__device__ __forceinline__ TileCtx tile_ctx(const MParams& M, int tile) {
  TileCtx t;
  const int plane = udiv_magic(tile, M.m_tpp);
  const int rem = tile - plane * M.tiles_per_plane;
  const int ty = udiv_magic(rem, M.m_tx);
  t.y0 = ty * kPhTH;
  t.x0 = (rem - ty * M.tiles_x) * kPhTW;
  t.b = udiv_magic(plane, M.m_z);
  t.z = plane - t.b * M.Z;
  return t;
}

// Staged items of one phase held in registers between request and LDS write: item r of lane t is
// (quad (t >> 2) + 256 r, channels 4 (t & 3) .. 4 (t & 3) + 3).
constexpr int kPhNI = 2;
struct Items {
  f4 raw[kPhNI][4];  // [item][channel of the slot] x 4 pixels
};

template <typename T, int D_T, bool CROP>
__device__ __forceinline__ void load_items(const MParams& M, const TileCtx& tc, const MPhase& PH, const T* xt,
                                           const T* nbt, Items& it) {
  static_assert(D_T == 16, "one item = a quad x 4 channels, 4 lanes per quad");
  const unsigned cs = (unsigned)M.S * (unsigned)sizeof(T);
  const size_t bo = (size_t)tc.b * D_T * (size_t)M.S;
  const int q0 = (int)threadIdx.x >> 2;
  const unsigned chan = (unsigned)(4 * (threadIdx.x & 3)) * cs;
#pragma unroll
  for (int r = 0; r < kPhNI; ++r) {
    const int q = q0 + 256 * r;
    const int prow = q >> PH.wsh, col = (q & ((1 << PH.wsh) - 1)) * 4;
    // a wave holds 16 consecutive quads = 1 or 2 rows; rows0 is even, so the block is wave-uniform
    const bool b1 = PH.nblk > 1 && __builtin_amdgcn_readfirstlane(prow) >= PH.rows0;
    const int dz = b1 ? PH.dz1 : PH.dz0, yb = b1 ? PH.y01 : PH.y00, xb = b1 ? PH.x01 : PH.x00;
    const int flags = b1 ? PH.fl1 : PH.fl0;
    const bool valid = q < PH.nquads;
    bool okz, oky, okx;
    const int gz = wrap1<CROP>(tc.z + dz, M.Z, okz);
    const int gy = wrap1<CROP>(tc.y0 + yb + prow, M.Y, oky);
    const int gx = tc.x0 + xb + col;
    const int gxw = wrap1<CROP>(gx, M.X, okx);  // aligned block: X % 4 == 0 and gx % 4 == 0, the quad wraps / crops as a whole
    const unsigned plane = (unsigned)((CROP ? min(max(gz, 0), M.Z - 1) : gz) * M.Y + gy) * (unsigned)M.X;
    const rsrc_t src = mkbuf(((flags & BLK_SRC_X) ? xt : nbt) + bo);
    const bool okzy = valid && okz && oky;
    const bool un = flags & BLK_UNALIGNED;        // uniform
    const bool contig = gx >= 0 && gx + 3 < M.X;  // unaligned block: no wrap / crop inside the quad
    const bool whole = un ? contig : okx;
    const unsigned vo = (okzy && whole) ? (plane + (unsigned)(un ? gx : gxw)) * (unsigned)sizeof(T) + chan : kOOB;
#pragma unroll
    for (int c = 0; c < 4; ++c) it.raw[r][c] = ld_quad<T>(src, vo, (unsigned)c * cs);
    if (un) {
      // quads of an unaligned block that straddle the image edge: pixel by pixel (only waves that hold one)
      if (okzy && !contig) {
        unsigned vj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bool okj;
          const int gj = wrap1<CROP>(gx + j, M.X, okj);
          vj[j] = okj ? (plane + (unsigned)gj) * (unsigned)sizeof(T) + chan : kOOB;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int j = 0; j < 4; ++j) it.raw[r][c][j] = bl_emb<T>(src, vj[j], (unsigned)c * cs);
      }
    }
  }
}

// normalise the staged items and write them to LDS ([slot][pixel] float4); own phase: 1/norm of tile pixels to rnbuf
__device__ __forceinline__ void write_items(const MParams& M, const MPhase& PH, bool own_phase, Items& it,
                                            char* __restrict__ ldst, float* __restrict__ rnbuf) {
#pragma unroll
  for (int r = 0; r < kPhNI; ++r) {
    const int q = ((int)threadIdx.x >> 2) + 256 * r;
    float rn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float ss = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) ss = fmaf(it.raw[r][c][j], it.raw[r][c][j], ss);
      rn[j] = rnorm(quad_sum(ss), M.inv_eps);
    }
    if (q < PH.nquads) {
      char* dst = ldst + r * (256 * 64);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f4 t;
        t.x = it.raw[r][0][j] * rn[j]; t.y = it.raw[r][1][j] * rn[j];
        t.z = it.raw[r][2][j] * rn[j]; t.w = it.raw[r][3][j] * rn[j];
        *(f4*)(dst + j * 16) = t;
      }
      if (own_phase) {  // uniform
        const int s = threadIdx.x & 3;
        const int tr = (q >> PH.wsh) - PH.own_r, tcn = (q & ((1 << PH.wsh) - 1)) * 4 - PH.own_c;
        const bool b1 = PH.nblk > 1 && (q >> PH.wsh) >= PH.rows0;
        const bool is_own = ((b1 ? PH.fl1 : PH.fl0) & BLK_OWN) != 0;
        if (is_own && (unsigned)tr < (unsigned)kPhTH && (unsigned)tcn < (unsigned)kPhTW)
          rnbuf[tr * kPhTW + tcn + s] = s == 0 ? rn[0] : s == 1 ? rn[1] : s == 2 ? rn[2] : rn[3];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, phased: de(p) = dl * (G - xhat <xhat, G>) / n,  G(p) = sum over pairs of g * nbhat(p + delta)
// ------------------------------------------------------------------------------------------------
template <typename T, int D_T, bool CROP>
__global__ __launch_bounds__(kPhNT, 4) void k_bwd_phased(const MParams M, const T* __restrict__ xt,
                                                         const T* __restrict__ nbt, const float* __restrict__ gin,
                                                         const float* __restrict__ dloss, T* __restrict__ dx,
                                                         long long* __restrict__ dbg) {
  typedef Lds<D_T, kPhPlq> L;
  static_assert(D_T == 16, "phase plan geometry (items per pixel, epilogue) is laid out for D = 16");
  // diagnostic build only (PEA_STAMPS): s_memtime stamps of wave 0 for the workgroup's SECOND tile
  int nstamp = 0, tcount = 0;
#define PEA_STAMP()                                                                                   \
  if (dbg && tcount == 1 && threadIdx.x == 0 && nstamp < 64) dbg[blockIdx.x * 64 + nstamp++] = (long long)__builtin_amdgcn_s_memtime();
  extern __shared__ f4 lds4[];
  char* lds = (char*)lds4;
  float* rnbuf = (float*)(lds + L::kBytes);  // [1024] 1 / norm of the tile's own pixels
  const int xcd = blockIdx.x % kXcd, wj = blockIdx.x / kXcd;
  const int tile_lo = xcd * M.tiles_per_xcd, tile_hi = min(tile_lo + M.tiles_per_xcd, M.ntiles);
  int tile = tile_lo + wj;
  if (tile >= tile_hi) return;
  const float dl = dloss ? dloss[0] : 1.f;
  const unsigned ecs = (unsigned)M.S * (unsigned)sizeof(T);
  int ly, lx;
  lane_pixel<kPhTW>(ly, lx);
  char* ldst = lds + ((int)threadIdx.x >> 2) * 64 + (int)(threadIdx.x & 3) * L::kPlaneB;  // this lane's staging slot

  TileCtx tc = tile_ctx(M, tile);
  Items it;
  load_items<T, D_T, CROP>(M, tc, M.ph[0], xt, nbt, it);

  // g of pairs [k0, k0 + kPChunk) of a phase for this lane's pixel of tile `t` (role A: at p; role B: at p - o)
  auto load_g = [&](const TileCtx& t, const MPhase& PH, int k0, float* gn) {
    const rsrc_t gB = mkbuf(gin + (size_t)t.b * M.K * (size_t)M.S);
    const int py = t.y0 + ly, px = t.x0 + lx;
    const bool live = py < M.Y && px < M.X;
    const unsigned pb = live ? (unsigned)((t.z * M.Y + py) * M.X + px) * 4u : kOOB;
#pragma unroll
    for (int u = 0; u < kPChunk; ++u) {
      gn[u] = 0.f;
      if (k0 + u < PH.npair) {  // uniform
        const MPair pe = M.pair[PH.pair0 + k0 + u];
        if (pe.gyx == 0 && pe.gz == 0) {  // uniform: role A
          gn[u] = bl32(gB, pb, (unsigned)pe.g_so);
        } else {
          bool okz, oky, okx;
          const int zz = wrap1<CROP>(t.z + pe.gz, M.Z, okz);
          const int yy = wrap1<CROP>(py + (pe.gyx >> 16), M.Y, oky);
          const int xx = wrap1<CROP>(px + (int)(short)(pe.gyx & 0xffff), M.X, okx);
          const bool ok = live && okz && oky && okx;
          const unsigned zc = (unsigned)(CROP ? min(max(zz, 0), M.Z - 1) : zz);
          gn[u] = bl32(gB, ok ? ((zc * (unsigned)M.Y + (unsigned)yy) * (unsigned)M.X + (unsigned)xx) * 4u : kOOB,
                       (unsigned)pe.g_so);
        }
      }
    }
  };

  float gcur[kPChunk], gnext[kPChunk];
  load_g(tc, M.ph[0], 0, gcur);
  float G[D_T];
#pragma unroll
  for (int c = 0; c < D_T; ++c) G[c] = 0.f;

  int ph = 0;
  while (true) {
    const MPhase PH = M.ph[ph];
    const bool last_ph = ph + 1 == M.nphase;
    PEA_STAMP()
    write_items(M, PH, last_ph, it, ldst, rnbuf);
    PEA_STAMP()
    lds_barrier();
    PEA_STAMP()
    // Request the next step's items (next phase, or phase 0 of this workgroup's next tile) and serve this phase's
    // pairs from LDS.  Even waves request first, odd waves compute first, so that the load-issue burst of one half
    // of the workgroup overlaps the LDS / VALU work of the other half.
    const int ntile = tile + M.wg_per_xcd;
    const bool more = !last_ph || ntile < tile_hi;
    TileCtx tn = tc;
    if (last_ph && more) tn = tile_ctx(M, ntile);
    const int nph = last_ph ? 0 : ph + 1;
    const int parity = (int)(threadIdx.x >> 6) & 1;  // uniform per wave
    for (int step = 0; step < 2; ++step) {
      if ((step ^ parity) == 0) {
        if (more) {
          const MPhase PN = M.ph[nph];
          load_items<T, D_T, CROP>(M, tn, PN, xt, nbt, it);
          load_g(tn, PN, 0, gnext);
        }
        PEA_STAMP()
      } else {
        // ---- pairs of this phase: neighbour vector of pair u+1 requested from LDS before pair u is consumed
        const char* lbase = lds + ((ly << (PH.wsh + 2)) + lx) * 16;
        for (int k0 = 0; k0 < PH.npair; k0 += kPChunk) {
          if (k0 > 0) load_g(tc, PH, k0, gcur);
          int eo[kPChunk];
#pragma unroll
          for (int u = 0; u < kPChunk; ++u) eo[u] = M.pair[PH.pair0 + min(k0 + u, PH.npair - 1)].e_off;
          float va[D_T], vb[D_T];
          lds_pixel<D_T, kPhPlq>(lbase + eo[0], 0, va);
#pragma unroll
          for (int u = 0; u < kPChunk; u += 2) {
            if (k0 + u + 1 < PH.npair) lds_pixel<D_T, kPhPlq>(lbase + eo[u + 1], 0, vb);
            if (k0 + u < PH.npair) {
#pragma unroll
              for (int c = 0; c < D_T; ++c) G[c] = fmaf(gcur[u], va[c], G[c]);
            }
            asm volatile("" ::: "memory");
            if (u + 2 < kPChunk && k0 + u + 2 < PH.npair) lds_pixel<D_T, kPhPlq>(lbase + eo[u + 2], 0, va);
            if (k0 + u + 1 < PH.npair) {
#pragma unroll
              for (int c = 0; c < D_T; ++c) G[c] = fmaf(gcur[u + 1], vb[c], G[c]);
            }
            asm volatile("" ::: "memory");
          }
        }
        PEA_STAMP()
      }
    }
    PEA_STAMP()
    if (last_ph) {
      // own pixel (normalised) and 1 / norm: the own block is staged in the last phase
      float xh[D_T];
      lds_pixel<D_T, kPhPlq>(lds + ((ly << (PH.wsh + 2)) + lx) * 16 + M.own_off, 0, xh);
      const float rn = rnbuf[threadIdx.x];
      float proj = 0.f;
#pragma unroll
      for (int c = 0; c < D_T; ++c) proj = fmaf(xh[c], G[c], proj);
      if (rn >= M.inv_eps) proj = 0.f;  // clamp_min branch of F.normalize: d ehat / d e = I / eps
      const float sc = dl * rn;
      lds_barrier();  // every lane is done with the blocks: lay the result over them as [channel][tile pixel]
      float* sT = (float*)lds;
#pragma unroll
      for (int c = 0; c < D_T; ++c) {
        sT[c * kPhNT + (int)threadIdx.x] = (G[c] - xh[c] * proj) * sc;
        G[c] = 0.f;
      }
      lds_barrier();
      const rsrc_t dB = mkbuf(dx + (size_t)tc.b * D_T * (size_t)M.S);
      const int qd = threadIdx.x & 255;
      const int gy = tc.y0 + (qd >> 3), gx = tc.x0 + (qd & 7) * 4;
      const unsigned vo = (gy < M.Y && gx < M.X) ? ((unsigned)((tc.z * M.Y + gy) * M.X + gx)) * (unsigned)sizeof(T) : kOOB;
#pragma unroll
      for (int r = 0; r < D_T / 4; ++r) {
        const int c = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8) + 4 * r);  // 256 quads per channel: uniform per wave
        const f4 val = *(const f4*)(sT + c * kPhNT + qd * 4);
        st_quad<T>(dB, val, vo, (unsigned)c * ecs);
      }
    }
    PEA_STAMP()
    if (!more) break;
    lds_barrier();
    PEA_STAMP()
#pragma unroll
    for (int u = 0; u < kPChunk; ++u) gcur[u] = gnext[u];
    PEA_STAMP()
    if (last_ph) { tile = ntile; tc = tn; ++tcount; }
    ph = nph;
  }
}

}  // namespace pea
