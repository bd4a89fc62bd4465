// pea_k_box.hip -- launchers of the unit-box kernels (pea_box.h): the 26-neighbourhood of BASELINE.json configs[3] and its subsets.
// One translation unit of libpea_hip.so (pea_host.h).
#include <algorithm>

#include "pea_host.h"
#include "pea_boxm.h"

namespace pea {

namespace {

struct BPlan { BParams C; };

// the XCDs' blocks of a round as one super-block (pea_xdma.h march_tile), (8 / sx) x sx: taken where the tile grid is a whole number
// of super-blocks (they are padded to the grid: XCDs would idle otherwise)
void set_super_blocks(const KParams& P, BParams* C, int sx) {
  if (C->zrun < 1 || !(sx == 1 || sx == 2 || sx == 4 || sx == 8)) return;
  const int sy = kXcd / sx;
  if (C->tiles_y % (C->zgy * sy) || C->tiles_x % (C->zgx * sx)) return;
  const long long per_xcd = (long long)P.B * (C->tiles_y / (C->zgy * sy)) * (C->tiles_x / (C->zgx * sx)) * C->zgy * C->zgx * C->zrun;
  if (per_xcd * kXcd > 0x7fffff00LL) return;
  C->sup_x = sx; C->sup_y = sy;
  C->tiles_per_xcd = (int)per_xcd;
}

bool plan(const KParams& P, BParams* out) {
  static thread_local PlanCache<BPlan, 8> cache;
  BPlan t;
  if (!cache.get(P, 0, &t, [&](BPlan* p) {
        if (!plan_box(P, &p->C)) return false;
        if (env().zblk_y > 0) p->C.zgy = env().zblk_y;
        if (env().zblk_x > 0) p->C.zgx = env().zblk_x;
        if (env().zblk_y < 0) p->C.zrun = 0;
        // Round 6 (profiles/r6_sup_n26*.txt, 24 x 1024^2, three rounds in one process): blocks of 2 x 8 tiles with the XCDs' blocks of a
        // round as one super-block, 4 high x 2 wide (march_tile) -- k_fwd_box 2.600 against 2.692 ms for every XCD on its own range of
        // 4 x 2 blocks; taken where the tile grid is a whole number of super-blocks (small volumes keep the 4 x 2 blocks)
        if (env().zm_sup != 0 && env().zblk_y == 0 && env().zblk_x == 0 && p->C.zrun > 1) {
          BParams T = p->C;
          T.zgy = 2; T.zgx = 8;
          set_super_blocks(P, &T, 2);
          if (T.sup_x) p->C = T;
        } else if (env().zm_sup > 0) {
          set_super_blocks(P, &p->C, env().zm_sup);
        }
        return true;
      }))
    return false;
  *out = t.C;
  return true;
}

}  // namespace

// axis-aligned in-plane stencils belong to the cross kernels (their caller asks them first); everything else inside the unit box
bool box_supported(const KParams& P, int dtype) {
  BParams C;
  return dtype == PEA_F32 && P.D == 16 && env().box && !env().force_direct && plan(P, &C);
}

bool box_fwd(const KParams& P, const FwdArgs& A, hipStream_t s) {
  if (A.dtype != PEA_F32 || P.D != 16 || A.eo != A.e || !env().box || env().force_direct) return false;
  if (misaligned(A.e, 16) || misaligned(A.t, 16) || misaligned(A.w, 16) || misaligned(A.affs, 16) || misaligned(A.gout, 16) ||
      misaligned(A.m, 4) || misaligned(A.inv_out, 4))
    return false;
  if (A.train && ((P.tbs | P.wbs | P.mbs) & 3)) return false;
  BParams C;
  if (!plan(P, &C)) return false;
  const dim3 grid((unsigned)(C.tiles_per_xcd * kXcd)), blk(kBoxTH * kBoxTW);
  const float* e = (const float*)A.e;
#define PEA_BF(CROP_, TRAIN_)                                                                                   \
  {                                                                                                             \
    constexpr auto kern = k_fwd_box<16, CROP_, TRAIN_>;                                                         \
    if (allow_lds<kern>(kBoxLds)) return false;                                                                 \
    hipLaunchKernelGGL(kern, grid, blk, kBoxLds, s, P, C, e, A.t, A.w, A.m, A.affs, A.gout, A.st, A.inv_out);   \
  }
  const bool crop = P.border != PEA_BORDER_CIRCULAR;
  if (A.train) {
    if (crop) PEA_BF(true, true) else PEA_BF(false, true)
  } else {
    if (crop) PEA_BF(true, false) else PEA_BF(false, false)
  }
#undef PEA_BF
  return true;
}

// the marching backward (pea_boxm.h): CROP_ZERO volumes with enough tile columns for the CUs (PEA_ZMARCH=2: any); false = not taken
static bool box_bwd_march(const KParams& P, const BParams& C0, const float* x, const float* inv, const float* g, const float* dl,
                          float* dx, hipStream_t s) {
  if (!env().zmarch || P.border != PEA_BORDER_CROP_ZERO || P.Z < 3) return false;
  if ((long long)P.D * P.S * 4 >= (1LL << 31)) return false;  // channel distance + plane offset in one 32-bit offset
  const long long cols = (long long)P.B * C0.tiles_per_plane;
  int nseg = 1;
  if (env().zseg > 0) nseg = (P.Z + env().zseg - 1) / env().zseg;
  else if (cols < 2 * device_cus()) nseg = (int)std::min<long long>((2 * device_cus() + cols - 1) / cols, std::max(1, P.Z / 6));
  if (env().zmarch < 2 && cols * nseg < device_cus()) return false;
  BMParams M;
  M.zseg = (P.Z + nseg - 1) / nseg;
  M.nseg = (P.Z + M.zseg - 1) / M.zseg;
  BParams C = C0;
  // xdma_tile's "z" is the segment; an XCD's 32 workgroups march through blocks of 2 x 8 tile columns (round 4 measured on the
  // 24 x 1024^2 sub-volume: 4 x 8 and 4 x 4 1.82 ms, 8 x 4 and 16 x 4 1.88, 16 x 2 -- the norm5 march's best -- 1.91, 32 x 1 2.25;
  // the region here has a halo of 1 - 4 pixels, not 27: what matters is that the blocks of the eight XCDs are spread out)
  C.zrun = M.nseg;
  C.zgy = env().zblk_y > 0 ? env().zblk_y : 2;
  C.zgx = env().zblk_x > 0 ? env().zblk_x : 8;
  const long long nt = cols * M.nseg;
  if (nt > 0x7fffff00LL) return false;
  C.ntiles = (int)nt;
  C.tiles_per_xcd = (C.ntiles + kXcd - 1) / kXcd;
  C.sup_x = C.sup_y = 0;  // (round 6: 2 x 8 blocks 1.935 ms, 4 x 8 1.96, super-blocks no better: profiles/r6_sup_n26b.txt)
  const dim3 grid((unsigned)(C.tiles_per_xcd * kXcd)), blk(kBoxTH * kBoxTW);
  constexpr auto kern = k_bwd_boxm;
  if (allow_lds<kern>(kBmLds)) return false;
  hipLaunchKernelGGL(kern, grid, blk, kBmLds, s, P, C, M, x, inv, g, dl, dx);
  return true;
}

bool box_bwd(const KParams& P, const float* x, const float* inv, const float* g, const float* dl, float* dx, hipStream_t s) {
  if (P.D != 16 || !inv || !env().box || env().force_direct) return false;
  if (misaligned(x, 16) || misaligned(inv, 16) || misaligned(g, 4) || misaligned(dx, 4)) return false;
  BParams C;
  if (!plan(P, &C)) return false;
  if (env().boxm && box_bwd_march(P, C, x, inv, g, dl, dx, s)) return true;
  const dim3 grid((unsigned)(C.tiles_per_xcd * kXcd)), blk(kBoxTH * kBoxTW);
  if (P.border != PEA_BORDER_CIRCULAR) {
    constexpr auto kern = k_bwd_box<16, true>;
    hipLaunchKernelGGL(kern, grid, blk, kBoxLdsBwd, s, P, C, x, inv, g, dl, dx);
  } else {
    constexpr auto kern = k_bwd_box<16, false>;
    hipLaunchKernelGGL(kern, grid, blk, kBoxLdsBwd, s, P, C, x, inv, g, dl, dx);
  }
  return true;
}

}  // namespace pea
