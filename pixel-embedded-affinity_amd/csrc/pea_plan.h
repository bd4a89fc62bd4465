// pea_plan.h -- host-side planner of the phase machine (pea_phased.h).  Included by pea_hip.hip only.
//
// Input: the stencil (offsets), which roles the launch needs and whether the neighbour tensor is the tensor
// differentiated (self loss).  Output: the blocks to stage per tile, grouped into phases that fit the LDS
// planes, and for each (offset, role) pair where its neighbour sits in those blocks.
//   role A: x is the first operand of <x(p), nb(p + o)>   -> neighbour displacement +o, g sampled at p
//   role B: x is the second operand of <nb(p - o), x(p)>  -> neighbour displacement -o, g sampled at p - o
// Blocks: a RECT (tile + halo) serves every pair whose displacement falls inside its halo; a SHIFT block is the
// tile displaced by one far offset.  The planner enumerates (square radius, vertical radius, horizontal radius)
// per plane and keeps the combination that stages the fewest pixels.
#pragma once
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "pea_phased.h"

namespace pea {

struct PlanPair { int i, role, dz, dy, dx; };
struct PlanBlk {
  int dz;
  int kind;                // 0 = V rect (32 px wide, halo rows), 1 = H rect (64 px wide: 16 px of halo left and right), 2 = shifted tile
  int hy0, hy1;            // V rect halo rows
  int sy, sx;              // shift blocks: displacement of the tile
  std::vector<int> pairs;  // indices into the pair list
  int rows() const { return kPhTH + hy0 + hy1; }
  int wsh() const { return kind == 1 ? 4 : 3; }
  int quads() const { return rows() << wsh(); }
  bool has_tile() const { return dz == 0 && (kind != 2 || (sy == 0 && sx == 0)); }
};
constexpr int kHRectHalo = 16;

inline bool plan_plane(const std::vector<PlanPair>& pr, int dz, std::vector<PlanBlk>* best) {
  std::vector<int> rv = {0};
  for (const PlanPair& p : pr)
    if (p.dz == dz && p.dx == 0) rv.push_back(abs(p.dy));
  long best_q = -1;
  for (int v : rv) for (int h = 0; h < 2; ++h) {
    std::vector<char> cov(pr.size(), 0);
    std::vector<PlanBlk> blks;
    if (v > 0) {
      PlanBlk b = {dz, 0, 0, 0, 0, 0, {}};
      for (size_t k = 0; k < pr.size(); ++k)
        if (pr[k].dz == dz && pr[k].dx == 0 && abs(pr[k].dy) <= v) {
          b.hy0 = std::max(b.hy0, -pr[k].dy); b.hy1 = std::max(b.hy1, pr[k].dy);
          b.pairs.push_back((int)k);
        }
      b.hy0 = (b.hy0 + 1) & ~1;  // even: a wave's 2 rows never straddle the tile rows / blocks
      b.hy1 = (b.hy1 + 1) & ~1;
      if (b.quads() > kPhQuads) continue;
      for (int k : b.pairs) cov[k] = 1;
      blks.push_back(b);
    }
    if (h > 0) {
      PlanBlk b = {dz, 1, 0, 0, 0, 0, {}};
      for (size_t k = 0; k < pr.size(); ++k)
        if (!cov[k] && pr[k].dz == dz && pr[k].dy == 0 && abs(pr[k].dx) <= kHRectHalo) b.pairs.push_back((int)k);
      if (b.pairs.empty()) continue;
      for (int k : b.pairs) cov[k] = 1;
      blks.push_back(b);
    }
    for (size_t k = 0; k < pr.size(); ++k) {
      if (cov[k] || pr[k].dz != dz) continue;
      bool found = false;
      for (PlanBlk& b : blks)
        if (b.kind == 2 && b.sy == pr[k].dy && b.sx == pr[k].dx) { b.pairs.push_back((int)k); found = true; break; }
      if (!found) blks.push_back(PlanBlk{dz, 2, 0, 0, pr[k].dy, pr[k].dx, {(int)k}});
      cov[k] = 1;
    }
    long q = 0;
    for (const PlanBlk& b : blks) q += b.quads();
    q = q * 64 + (long)blks.size();  // fewest staged pixels, then fewest blocks
    if (best_q < 0 || q < best_q) { best_q = q; *best = blks; }
  }
  return best_q >= 0;
}

inline unsigned magic_div(int d) { return d == 1 ? 0u : (unsigned)(0x100000000ULL / (unsigned long long)d) + 1u; }

// roles: bit 0 = A, bit 1 = B.  self: the neighbour tensor is x itself.  ncu: compute units of the device.
inline bool plan_phased(const KParams& P, int roles, bool self, size_t esize, int ncu, MParams* out) {
  if (P.D != 16 || P.X % 4 || P.border == PEA_BORDER_REPLICATE) return false;
  if ((long long)P.D * P.S * (long long)esize >= 0x7fffffffLL) return false;  // per-lane byte offsets carry the channel
  if ((long long)P.K * P.S * 4 > 0xFFFFFFFFLL) return false;                   // scalar plane offsets are 32-bit
  std::vector<PlanPair> pr;
  int my = 0, mx = 0;
  for (int i = 0; i < P.K; ++i) {
    const int oz = P.off[i][0], oy = P.off[i][1], ox = P.off[i][2];
    if (abs(oy) > 30000 || abs(ox) > 30000) return false;
    my = std::max(my, abs(oy)); mx = std::max(mx, abs(ox));
    if (roles & 1) pr.push_back(PlanPair{i, 0, oz, oy, ox});
    if (roles & 2) pr.push_back(PlanPair{i, 1, -oz, -oy, -ox});
  }
  if ((int)pr.size() > kMaxPair) return false;
  // the kernels wrap an index with one conditional add / subtract
  if (P.Y <= std::max(my, 2) + kPhTH || P.X <= std::max(mx, kHRectHalo) + kPhTW + 4) return false;
  std::vector<int> planes;
  for (const PlanPair& p : pr)
    if (std::find(planes.begin(), planes.end(), p.dz) == planes.end()) planes.push_back(p.dz);
  std::vector<PlanBlk> blks;
  for (int dz : planes) {
    std::vector<PlanBlk> b;
    if (!plan_plane(pr, dz, &b)) return false;
    blks.insert(blks.end(), b.begin(), b.end());
  }
  // the own block: x's tile, staged in the LAST phase
  int own = -1;
  if (self) {
    for (size_t k = 0; k < blks.size(); ++k)
      if (blks[k].has_tile() && (own < 0 || (blks[own].kind == 2 && blks[k].kind != 2))) own = (int)k;
  }
  bool own_extra = false;
  if (own < 0) {
    blks.push_back(PlanBlk{0, 2, 0, 0, 0, 0, {}});
    own = (int)blks.size() - 1;
    own_extra = !self;
  }
  // phases: a rect alone; shifted tiles two by two; the own block's phase last
  std::vector<std::vector<int>> phases;
  std::vector<int> open;  // a phase holding one shifted tile
  for (int k = 0; k < (int)blks.size(); ++k) {
    if (k == own) continue;
    if (blks[k].kind != 2) { phases.push_back({k}); continue; }
    if (!open.empty()) { open.push_back(k); phases.push_back(open); open.clear(); }
    else open.push_back(k);
  }
  if (blks[own].kind == 2 && !open.empty()) { open.push_back(own); phases.push_back(open); open.clear(); }
  else {
    if (!open.empty()) phases.push_back(open);
    phases.push_back({own});
  }
  if ((int)phases.size() > kMaxPhase) return false;

  MParams M = {};
  M.Z = P.Z; M.Y = P.Y; M.X = P.X; M.S = P.S; M.K = P.K;
  M.flags = P.flags; M.eps = P.eps; M.inv_eps = 1.0f / P.eps;
  const int tiles_y = (P.Y + kPhTH - 1) / kPhTH;
  M.tiles_x = (P.X + kPhTW - 1) / kPhTW;
  M.tiles_per_plane = tiles_y * M.tiles_x;
  const long long nt = (long long)M.tiles_per_plane * P.Z * P.B;
  if (nt * std::max(M.tiles_per_plane, P.Z) >= 0x7fffffffLL) return false;  // magic-number division stays exact
  M.ntiles = (int)nt;
  M.m_tpp = magic_div(M.tiles_per_plane); M.m_tx = magic_div(M.tiles_x); M.m_z = magic_div(P.Z);
  M.tiles_per_xcd = (M.ntiles + kXcd - 1) / kXcd;
  M.wg_per_xcd = std::max(1, std::min(M.tiles_per_xcd, ncu / kXcd));
  M.nphase = (int)phases.size();
  int np = 0;
  for (int ph = 0; ph < M.nphase; ++ph) {
    MPhase& H = M.ph[ph];
    H.nblk = (int)phases[ph].size();
    H.wsh = blks[phases[ph][0]].wsh();
    H.pair0 = np;
    const int pitch = 4 << H.wsh;
    int row0 = 0;
    for (int q = 0; q < H.nblk; ++q) {
      const PlanBlk& b = blks[phases[ph][q]];
      if (b.wsh() != H.wsh) return false;
      const int y0 = b.kind == 2 ? b.sy : -b.hy0;
      const int x0 = b.kind == 2 ? b.sx : (b.kind == 1 ? -kHRectHalo : 0);
      int fl = (x0 & 3) ? BLK_UNALIGNED : 0;
      const int tile_r = row0 + (b.kind == 0 ? b.hy0 : 0), tile_c = b.kind == 1 ? kHRectHalo : 0;  // where the tile sits (rects)
      if (phases[ph][q] == own) {
        if (ph != M.nphase - 1) return false;
        fl |= BLK_OWN | (own_extra ? BLK_SRC_X : 0);
        H.own_r = tile_r; H.own_c = tile_c;
        M.own_off = (tile_r * pitch + tile_c) * 16;
      }
      if (q == 0) { H.dz0 = b.dz; H.y00 = y0; H.x00 = x0; H.fl0 = fl; H.rows0 = b.rows(); }
      else { H.dz1 = b.dz; H.y01 = y0 - row0; H.x01 = x0; H.fl1 = fl; }
      if (q == 0 && H.nblk > 1 && (b.rows() & 1)) return false;
      for (int k : b.pairs) {
        const PlanPair& p = pr[k];
        if (np >= kMaxPair) return false;
        MPair& E = M.pair[np++];
        const int ey = b.kind == 2 ? row0 : tile_r + p.dy, ex = b.kind == 2 ? 0 : tile_c + p.dx;
        E.e_off = (ey * pitch + ex) * 16;
        E.g_so = (int)((unsigned)p.i * (unsigned)P.S * 4u);
        const int gdz = p.role ? p.dz : 0, gdy = p.role ? p.dy : 0, gdx = p.role ? p.dx : 0;
        E.gyx = (int)(((unsigned)gdy << 16) | ((unsigned)gdx & 0xffffu));
        E.gz = gdz;
      }
      row0 += b.rows();
    }
    H.nquads = row0 << H.wsh;
    if (H.nquads > kPhQuads) return false;
    H.npair = np - H.pair0;
  }
  *out = M;
  return true;
}

}  // namespace pea
