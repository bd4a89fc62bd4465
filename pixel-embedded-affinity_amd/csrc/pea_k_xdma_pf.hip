// pea_k_xdma_pf.hip -- launchers of the projection-first backward for f32 storage (pea_xdma_pf.h: D = 32 / 64, self loss and the cross loss
// with a detached second operand).  One translation unit of libpea_hip.so (pea_host.h); split from pea_k_xdma_h.hip for compile time
// (round 6: that file was the build's critical path, 55 s).
#include "pea_k_xdma_plan.h"
#include "pea_xdma_pf.h"

namespace pea {

namespace {

// ---- the backward with the projection first (pea_xdma_pf.h): 2D / in-plane stencils, self loss, f32, needs the raw affs map
template <int D_T>
bool bwd_self_pf(const KParams& P, const float* x, const float* inv, const float* g, const float* affs, const float* dl, float* dx,
                 hipStream_t s) {
  if (misaligned(x, 16) || misaligned(inv, 16) || misaligned(g, 4) || misaligned(affs, 4) || misaligned(dx, 4)) return false;
  XPlan X;
  const bool small = plan(P, kXdmaPSUS, 0, &X);
  if (!small && !plan(P, kXdmaPSU, 0, &X)) return false;
  if (X.C.npz > 0 || X.C.npx > kXP || X.C.npy > kXP) return false;
  const dim3 grid((unsigned)(X.C.tiles_per_xcd * kXcd)), blk(kXdmaTH * kXdmaTW);
#define PEA_PF(CROP_, PSU_, WPE_, RB_)                                                         \
  {                                                                                            \
    constexpr auto kern = k_bwd_xdma_pf<D_T, kXdmaTH, kXdmaTW, PSU_, CROP_, WPE_, RB_>;        \
    PEA_LAUNCH(kern, grid, blk, (size_t)2 * RB_ * PSU_ * 256, s, P, X.C, x, inv, g, affs, dl, dx) \
  }
  const bool crop = P.border != PEA_BORDER_CIRCULAR;
  if (small) { if (crop) PEA_PF(true, kXdmaPSUS, 6, 3) else PEA_PF(false, kXdmaPSUS, 6, 3) }
  else { if (crop) PEA_PF(true, kXdmaPSU, 4, 3) else PEA_PF(false, kXdmaPSU, 4, 3) }
#undef PEA_PF
  return true;
}

// ---- the same for the cross loss with a detached second operand at D = 32 / 64 (k_bwd_xdma_pfo): role A, one-sided cross
template <int D_T>
bool bwd_other_pf(const KParams& P, const float* e, const float* e_other, const float* inv2, const float* g, const float* affs,
                  const float* dl, float* de, hipStream_t s) {
  if (misaligned(e, 16) || misaligned(e_other, 16) || misaligned(inv2, 16) || ((size_t)P.B * P.S) % 4 || misaligned(g, 4) ||
      misaligned(affs, 4) || misaligned(de, 4))
    return false;
  XPlan X;
  if (!plan(P, kXdmaPSUF, 2, &X) || X.C.npz > 0 || X.C.npx > kXP || X.C.npy > kXP) return false;
  const dim3 grid((unsigned)(X.C.tiles_per_xcd * kXcd)), blk(kXdmaTH * kXdmaTW);
  constexpr int RB = 3;
  const size_t lds = (size_t)2 * RB * kXdmaPSUF * 256 + (size_t)2 * RB * 2048;  // the ring + the own tiles
  const float* inv_other = inv2 + (size_t)P.B * P.S;
  if (P.border != PEA_BORDER_CIRCULAR) {
    constexpr auto kern = k_bwd_xdma_pfo<D_T, kXdmaTH, kXdmaTW, kXdmaPSUF, true, 4, RB>;
    PEA_LAUNCH(kern, grid, blk, lds, s, P, X.C, e_other, inv_other, e, inv2, g, affs, dl, de)
  } else {
    constexpr auto kern = k_bwd_xdma_pfo<D_T, kXdmaTH, kXdmaTW, kXdmaPSUF, false, 4, RB>;
    PEA_LAUNCH(kern, grid, blk, lds, s, P, X.C, e_other, inv_other, e, inv2, g, affs, dl, de)
  }
  return true;
}

}  // namespace

bool xdma_pf_bwd_other(const KParams& P, const float* e, const float* e_other, const float* inv2, const float* g, const float* affs,
                       const float* dl, float* de, hipStream_t s) {
  if (P.D == 32) return bwd_other_pf<32>(P, e, e_other, inv2, g, affs, dl, de, s);
  if (P.D == 64) return bwd_other_pf<64>(P, e, e_other, inv2, g, affs, dl, de, s);
  return false;
}

bool xdma_pf_bwd_self(const KParams& P, const float* x, const float* inv, const float* g, const float* affs, const float* dl, float* dx,
                      hipStream_t s) {
  if (P.D == 32) return bwd_self_pf<32>(P, x, inv, g, affs, dl, dx, s);
  if (P.D == 64) return bwd_self_pf<64>(P, x, inv, g, affs, dl, dx, s);
  return false;
}

}  // namespace pea
