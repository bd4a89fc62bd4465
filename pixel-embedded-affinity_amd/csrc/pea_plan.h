// pea_plan.h -- host-side tile planning of the LDS-tiled box kernels (pea_tiled.h, pea_chunked.h, pea_fused_labels.h).
#pragma once
#include <stdlib.h>

#include <algorithm>

#include "pea_host.h"
#include "pea_tiled.h"

namespace pea {

// ------------------------------------------------------------------------------------------------
// tile planning
// ------------------------------------------------------------------------------------------------
struct TileCfg { int TH, TW, PLQ; };  // workgroup = TH*TW lanes (one per pixel); PLQ = LDS plane stride in pixels
// compiled-in shapes; index chosen by PEA_FWD_CFG / PEA_BWD_CFG (defaults = the measured best, CVPPP stencil)
// compiled-in tile shapes (the measured best of the round-1 sweep at the CVPPP stencil; the losing shapes are gone)
constexpr TileCfg kFwdCfg[] = {{16, 32, 1041}};
constexpr TileCfg kBwdCfg[] = {{32, 32, 2505}};
// D = 32: 128 B of LDS per region pixel, so one shape: 16x32 tiles, 1041 region pixels (133 KB, one workgroup of 8 waves per CU)
constexpr TileCfg kCfg32 = {16, 32, 1041};
template <int D_T> constexpr TileCfg fwd_cfg(int ci) { return D_T == 32 ? kCfg32 : kFwdCfg[ci]; }
constexpr TileCfg kCfg32B = {16, 32, 1093};  // backward: two-sided halo of 5 (26 x 42 region pixels, 140 KB)
template <int D_T> constexpr TileCfg bwd_cfg(int ci) { return D_T == 32 ? kCfg32B : kBwdCfg[ci]; }
template <int D_T> constexpr TileCfg fwdv_cfg() { return D_T == 32 ? kCfg32 : TileCfg{16, 32, 1041}; }
constexpr int kLdsMax = 160 * 1024;  // gfx950: 160 KiB per CU, one workgroup may take all of it

// Choose the "near" offsets (served from LDS): the largest in-plane radius whose halo'd region still fits the
// LDS planes of this tile shape.  both_sides: the backward needs p - o as well as p + o.
inline bool plan_tiles(const KParams& P, TileCfg c, bool both_sides, TParams* Q, bool ksplit_ok = false) {
  // (PEA_BORDER_REPLICATE: the callers decide -- k_fwd_tiled / k_fwd_tiled_v / k_bwd_tiled take it at f32, D = 16: pea_k_tiled.hip)
  if ((long long)P.Y * P.X >= (1LL << 29)) return false;                     // plane byte offsets stay below 2^31 (kOOB)
  // A raw buffer access is in range iff voffset < num_records - soffset (gfx9 range check: the scalar offset COUNTS), and the
  // kernels select the channel / offset plane with soffset under num_records = 2^31: the [D or K, Z, Y, X] block of one batch
  // item must stay below 2 GiB, or planes past it read zeros and drop their stores without any error (found by the
  // full-size K = 26 test: 26 x 24 x 1024^2 x 4 B = 2.6 GB).  Larger blocks take the direct kernels (64-bit pointers).
  // k_fwd_tiled / k_bwd_tiled (ksplit_ok) reach the upper offset channels through a second resource based KParams::ksplit planes
  // further, so for them only each HALF of the K block has to stay below 2 GiB (the 26-neighbourhood of configs[3]: 2 x 1.3 GB).
  if ((long long)P.D * P.S * 4 >= (1LL << 31)) return false;
  if ((long long)P.K * P.S * 4 >= (1LL << 31)) {
    if (!ksplit_ok || P.ksplit >= P.K) return false;
    if ((long long)P.ksplit * P.S * 4 >= (1LL << 31) || (long long)(P.K - P.ksplit) * P.S * 4 >= (1LL << 31)) return false;
  }
  const int NT = c.TH * c.TW;
  int radii[PEA_MAX_K], nr = 0;
  for (int i = 0; i < P.K; ++i)
    if (P.off[i][0] == 0) radii[nr++] = std::max(abs(P.off[i][1]), abs(P.off[i][2]));
  std::sort(radii, radii + nr);
  for (int k = nr - 1; k >= 0; --k) {
    const int rc = radii[k];
    TParams q = {};
    unsigned near_mask = 0;
    for (int i = 0; i < P.K; ++i) {
      const int oy = P.off[i][1], ox = P.off[i][2];
      if (P.off[i][0] != 0 || std::max(abs(oy), abs(ox)) > rc) continue;
      near_mask |= 1u << i;
      q.hy0 = std::max(q.hy0, both_sides ? abs(oy) : -oy);
      q.hy1 = std::max(q.hy1, both_sides ? abs(oy) : oy);
      q.hx0 = std::max(q.hx0, both_sides ? abs(ox) : -ox);
      q.hx1 = std::max(q.hx1, both_sides ? abs(ox) : ox);
    }
    q.RH = c.TH + q.hy0 + q.hy1;
    q.RW = c.TW + q.hx0 + q.hx1;
    q.R = q.RH * q.RW;
    if (q.R > c.PLQ) continue;
    // the kernels wrap with one conditional add
    if (P.Y < c.TH + q.hy1 || P.Y < q.hy0 || P.X < c.TW + q.hx1 || P.X < q.hx0) continue;
    q.dr = NT / q.RW;
    q.dc = NT % q.RW;
    q.inv_rw = 1.0f / (float)q.RW;
    q.inv_sw = 1.0f / (float)std::max(1, q.hx0 + q.hx1);
    q.inv_eps = 1.0f / P.eps;
    q.tiles_y = (P.Y + c.TH - 1) / c.TH;
    q.tiles_x = (P.X + c.TW - 1) / c.TW;
    q.tiles_per_plane = q.tiles_y * q.tiles_x;
    const long long nt = (long long)q.tiles_per_plane * P.Z * P.B;
    if (nt > 0x7fffff00LL) return false;
    q.ntiles = (int)nt;
    q.tiles_per_xcd = (q.ntiles + kXcd - 1) / kXcd;
    for (int i = 0; i < P.K; ++i) {
      const int oy = P.off[i][1], ox = P.off[i][2];
      const int oyx = (int)(((unsigned)oy << 16) | ((unsigned)ox & 0xffffu));
      if (near_mask >> i & 1u) q.near[q.n_near++] = OffEnt{i, oy * q.RW + ox, oyx, P.gscale[i]};
      else q.far[q.n_far++] = OffEnt{i, P.off[i][0], oyx, P.gscale[i]};
    }
    q.zrun = 0;
    for (int k = 0; k < q.n_far; ++k)
      if (q.far[k].d != 0 && P.Z > 1) q.zrun = P.Z;
    *Q = q;
    return true;
  }
  return false;
}

struct TPlan { TParams Q; };
// memoised plan_tiles (per thread; keyed by KParams, the tile shape and the halo mode)
inline bool plan_tiles_cached(const KParams& P, TileCfg c, bool both_sides, TParams* Q, bool ksplit_ok = false) {
  static thread_local PlanCache<TPlan, 12> cache;
  TPlan t;
  const int mode = ((c.TH * 64 + c.TW) * 4096 + c.PLQ) * 4 + (both_sides ? 2 : 0) + (ksplit_ok ? 1 : 0);
  if (!cache.get(P, mode, &t, [&](TPlan* p) { return plan_tiles(P, c, both_sides, &p->Q, ksplit_ok); })) return false;
  *Q = t.Q;
  return true;
}

}  // namespace pea
