// pea_k_xdma_h.hip -- launchers of the LDS-DMA cross kernels for f16 storage (pea_xdma_h16.h).  One translation unit of libpea_hip.so
// (pea_host.h); split from pea_k_xdma.hip for compile time (the f32 projection-first backward: pea_k_xdma_pf.hip).
#include "pea_k_xdma_plan.h"
#include "pea_xdma_h16.h"
#include "pea_xdma_pf.h"

namespace pea {

namespace {

// ---- f16 storage (pea_xdma_h16.h): 2D self loss / inference, X % 8 == 0 ---------------------------------------------------
template <int D_T, bool TRAIN>
bool fwd_self_h(const KParams& P, const FwdArgs& A, hipStream_t s) {
  const __half* e = (const __half*)A.e;
  if (P.X % 8 || P.Z != 1 || misaligned(e, 16) || misaligned(A.t, 16) || misaligned(A.w, 16) || misaligned(A.affs, 16) ||
      misaligned(A.gout, 16) || misaligned(A.m, 4) || misaligned(A.inv_out, 4))
    return false;
  if (TRAIN && ((P.tbs | P.wbs | P.mbs) & 3)) return false;
  XPlan X;
  if (!plan(P, kXdmaPSUF, 1, &X) || X.C.nfz > 0 || P.K > kXP) return false;
  const size_t lds = (size_t)5 * kXdmaPSUF * 256;  // two f32 working planes + six half-size ring planes
  const dim3 grid((unsigned)(X.C.tiles_per_xcd * kXcd)), blk(kXdmaTH * kXdmaTW);
  if (env().h16_hw) {  // half-precision working buffer, v_dot2 gather: 48 VGPRs and 30 KB -- four workgroups per CU (five: 173 against 168 us)
    const size_t ldsh = (size_t)4 * kXdmaPSUF * 256;
    // (D = 64 with at most eight offsets -- BASELINE configs[4] -- walks eight slots instead of ten)
#define PEA_HF(CROP_, NXP_)                                                                                          \
  {                                                                                                                  \
    constexpr auto kern = k_fwd_xdma_h<D_T, kXdmaTH, kXdmaTW, kXdmaPSUF, CROP_, TRAIN, 8, true, NXP_>;               \
    PEA_LAUNCH(kern, grid, blk, ldsh, s, P, X.C, e, A.t, A.w, A.m, A.affs, A.gout, A.st, A.inv_out, (const __half*)nullptr, (float*)nullptr) \
  }
    const bool crop = P.border != PEA_BORDER_CIRCULAR;
    if (D_T == 64 && X.C.nf <= 8) { if (crop) PEA_HF(true, (D_T == 64 ? 8 : kXP)) else PEA_HF(false, (D_T == 64 ? 8 : kXP)) }
    else { if (crop) PEA_HF(true, kXP) else PEA_HF(false, kXP) }
#undef PEA_HF
    return true;
  }
  if (P.border != PEA_BORDER_CIRCULAR) {
    constexpr auto kern = k_fwd_xdma_h<D_T, kXdmaTH, kXdmaTW, kXdmaPSUF, true, TRAIN, 6>;
    PEA_LAUNCH(kern, grid, blk, lds, s, P, X.C, e, A.t, A.w, A.m, A.affs, A.gout, A.st, A.inv_out, (const __half*)nullptr, (float*)nullptr)
  } else {
    constexpr auto kern = k_fwd_xdma_h<D_T, kXdmaTH, kXdmaTW, kXdmaPSUF, false, TRAIN, 6>;
    PEA_LAUNCH(kern, grid, blk, lds, s, P, X.C, e, A.t, A.w, A.m, A.affs, A.gout, A.st, A.inv_out, (const __half*)nullptr, (float*)nullptr)
  }
  return true;
}

template <int D_T>
bool bwd_self_h(const KParams& P, const __half* x, const float* inv, const float* g, const float* affs, const float* dl, __half* dx,
                hipStream_t s) {
  if (P.X % 8 || P.Z != 1 || misaligned(x, 16) || misaligned(inv, 16) || misaligned(g, 4) || misaligned(dx, 2) || misaligned(affs, 4))
    return false;
  XPlan X;
  const dim3 blk(kXdmaTH * kXdmaTW);
  const bool crop = P.border != PEA_BORDER_CIRCULAR;
  if constexpr (D_T > 16) {  // the projection first (pea_xdma_pf.h); at D = 16 it loses and is not instantiated
   if (affs && env().bwd_pf && !(P.flags & kActMask)) {
    const bool small = plan(P, kXdmaPSUHS, 0, &X);
    if ((small || plan(P, kXdmaPSUH, 0, &X)) && X.C.npz == 0 && X.C.npx <= kXP && X.C.npy <= kXP) {
      const dim3 grid((unsigned)(X.C.tiles_per_xcd * kXcd));
      // (D = 64 with at most eight pairs per axis -- BASELINE configs[4]: offsets[:8] = four shifts per axis, two roles each -- walks
      //  eight pairs per axis instead of ten: an unused pair costs its LDS read and its two FMAs all the same)
      constexpr int XPS = D_T == 64 ? 8 : kXP;
      const bool few = D_T == 64 && X.C.npx <= 8 && X.C.npy <= 8;
#define PEA_HPF(CROP_, PSU_, WPE_)                                                                   \
  {                                                                                                  \
    if (env().h16_hw && few) {                                                                       \
      constexpr auto kern = k_bwd_xdma_h<D_T, kXdmaTH, kXdmaTW, PSU_, CROP_, XPS, true, WPE_, true>;  \
      PEA_LAUNCH(kern, grid, blk, (size_t)5 * PSU_ * 256, s, P, X.C, x, inv, g, affs, dl, dx, (const __half*)nullptr, (const float*)nullptr) \
    } else if (env().h16_hw) {                                                                       \
      constexpr auto kern = k_bwd_xdma_h<D_T, kXdmaTH, kXdmaTW, PSU_, CROP_, kXP, true, WPE_, true>;  \
      PEA_LAUNCH(kern, grid, blk, (size_t)5 * PSU_ * 256, s, P, X.C, x, inv, g, affs, dl, dx, (const __half*)nullptr, (const float*)nullptr) \
    } else {                                                                                         \
      constexpr auto kern = k_bwd_xdma_h<D_T, kXdmaTH, kXdmaTW, PSU_, CROP_, kXP, true, WPE_>;        \
      PEA_LAUNCH(kern, grid, blk, (size_t)5 * PSU_ * 256, s, P, X.C, x, inv, g, affs, dl, dx, (const __half*)nullptr, (const float*)nullptr) \
    }                                                                                                \
  }
      // (87 VGPRs: the conversion's temporaries keep it above the 80 a third workgroup would need; small planes all the same --
      //  less LDS per workgroup never hurts the other kernels sharing the CU in a multi-stream section)
      if (small) { if (crop) PEA_HPF(true, kXdmaPSUHS, 4) else PEA_HPF(false, kXdmaPSUHS, 4) }
      else { if (crop) PEA_HPF(true, kXdmaPSUH, 4) else PEA_HPF(false, kXdmaPSUH, 4) }
#undef PEA_HPF
      return true;
    }
   }
  }
  if (!plan(P, kXdmaPSUH, 0, &X) || X.C.npz > 0) return false;
  constexpr int XP = D_T > 32 ? 8 : kXP;
  if (X.C.npx > XP || X.C.npy > XP) return false;
  const size_t lds = (size_t)5 * kXdmaPSUH * 256;
  const dim3 grid((unsigned)(X.C.tiles_per_xcd * kXcd));
  if (crop) {
    constexpr auto kern = k_bwd_xdma_h<D_T, kXdmaTH, kXdmaTW, kXdmaPSUH, true, XP>;
    PEA_LAUNCH(kern, grid, blk, lds, s, P, X.C, x, inv, g, (const float*)nullptr, dl, dx, (const __half*)nullptr, (const float*)nullptr)
  } else {
    constexpr auto kern = k_bwd_xdma_h<D_T, kXdmaTH, kXdmaTW, kXdmaPSUH, false, XP>;
    PEA_LAUNCH(kern, grid, blk, lds, s, P, X.C, x, inv, g, (const float*)nullptr, dl, dx, (const __half*)nullptr, (const float*)nullptr)
  }
  return true;
}

// ---- f16 storage, the cross loss with a detached second operand (k_fwd_xdma_h<.., OTHER>, k_bwd_xdma_h<.., PF, HW, OTHER>): 2D, X % 8 == 0
template <int D_T>
bool fwd_other_h(const KParams& P, const FwdArgs& A, hipStream_t s) {
  const __half *e = (const __half*)A.e, *eo = (const __half*)A.eo;
  if (P.X % 8 || P.Z != 1 || misaligned(e, 16) || misaligned(eo, 16) || misaligned(A.t, 16) || misaligned(A.w, 16) || misaligned(A.affs, 16) ||
      misaligned(A.gout, 16) || misaligned(A.m, 4) || misaligned(A.inv_out, 4) || ((P.tbs | P.wbs | P.mbs) & 3))
    return false;
  XPlan X;
  if (!plan(P, kXdmaPSUF, 1, &X) || X.C.nfz > 0 || P.K > kXP) return false;
  const size_t lds = (size_t)4 * kXdmaPSUF * 256 + 6 * 1024;  // working plane + ring + the own tiles
  const dim3 grid((unsigned)(X.C.tiles_per_xcd * kXcd)), blk(kXdmaTH * kXdmaTW);
  float* inv_other = A.inv_out + (size_t)P.B * P.S;
#define PEA_HFO(CROP_, NXP_)                                                                                                  \
  {                                                                                                                          \
    constexpr auto kern = k_fwd_xdma_h<D_T, kXdmaTH, kXdmaTW, kXdmaPSUF, CROP_, true, 6, true, NXP_, true>;                   \
    PEA_LAUNCH(kern, grid, blk, lds, s, P, X.C, eo, A.t, A.w, A.m, A.affs, A.gout, A.st, A.inv_out, e, inv_other)             \
  }
  const bool crop = P.border != PEA_BORDER_CIRCULAR;
  if (D_T == 64 && X.C.nf <= 8) { if (crop) PEA_HFO(true, (D_T == 64 ? 8 : kXP)) else PEA_HFO(false, (D_T == 64 ? 8 : kXP)) }
  else { if (crop) PEA_HFO(true, kXP) else PEA_HFO(false, kXP) }
#undef PEA_HFO
  return true;
}

template <int D_T>
bool bwd_other_h(const KParams& P, const __half* e, const __half* eo, const float* inv2, const float* g, const float* affs, const float* dl,
                 __half* de, hipStream_t s) {
  if (P.X % 8 || P.Z != 1 || misaligned(e, 16) || misaligned(eo, 16) || misaligned(inv2, 16) || ((size_t)P.B * P.S) % 4 ||
      misaligned(g, 4) || misaligned(affs, 4) || misaligned(de, 2))
    return false;
  XPlan X;
  if (!plan(P, kXdmaPSUF, 2, &X) || X.C.npz > 0 || X.C.npx > kXP || X.C.npy > kXP) return false;
  const dim3 grid((unsigned)(X.C.tiles_per_xcd * kXcd)), blk(kXdmaTH * kXdmaTW);
  const size_t lds = (size_t)5 * kXdmaPSUF * 256 + 6 * 1024;
  const float* inv_other = inv2 + (size_t)P.B * P.S;
  constexpr int XPS = D_T == 64 ? 8 : kXP;
  const bool few = D_T == 64 && X.C.npx <= 8 && X.C.npy <= 8;
#define PEA_HBO(CROP_, XP_)                                                                                                   \
  {                                                                                                                          \
    constexpr auto kern = k_bwd_xdma_h<D_T, kXdmaTH, kXdmaTW, kXdmaPSUF, CROP_, XP_, true, 4, true, true>;                    \
    PEA_LAUNCH(kern, grid, blk, lds, s, P, X.C, eo, inv_other, g, affs, dl, de, e, inv2)                                      \
  }
  const bool crop = P.border != PEA_BORDER_CIRCULAR;
  if (few) { if (crop) PEA_HBO(true, XPS) else PEA_HBO(false, XPS) }
  else { if (crop) PEA_HBO(true, kXP) else PEA_HBO(false, kXP) }
#undef PEA_HBO
  return true;
}

}  // namespace


bool xdma_h_fwd_other(const KParams& P, const FwdArgs& A, hipStream_t s) {
  if (!env().h16_hw) return false;
  if (P.D == 16) return fwd_other_h<16>(P, A, s);
  if (P.D == 32) return fwd_other_h<32>(P, A, s);
  if (P.D == 64) return fwd_other_h<64>(P, A, s);
  return false;
}

bool xdma_h_bwd_other(const KParams& P, const void* e, const void* e_other, const float* inv2, const float* g, const float* affs,
                      const float* dl, void* de, hipStream_t s) {
  if (!env().h16_hw || !env().bwd_pf || !affs || (P.flags & kActMask)) return false;
  if (P.D == 16) return bwd_other_h<16>(P, (const __half*)e, (const __half*)e_other, inv2, g, affs, dl, (__half*)de, s);
  if (P.D == 32) return bwd_other_h<32>(P, (const __half*)e, (const __half*)e_other, inv2, g, affs, dl, (__half*)de, s);
  if (P.D == 64) return bwd_other_h<64>(P, (const __half*)e, (const __half*)e_other, inv2, g, affs, dl, (__half*)de, s);
  return false;
}

// entry points used by pea_k_xdma.hip's dispatchers
bool xdma_h_fwd_self(const KParams& P, const FwdArgs& A, hipStream_t s) {
  if (P.D == 16) return A.train ? fwd_self_h<16, true>(P, A, s) : fwd_self_h<16, false>(P, A, s);
  if (P.D == 32) return A.train ? fwd_self_h<32, true>(P, A, s) : fwd_self_h<32, false>(P, A, s);
  if (P.D == 64) return A.train ? fwd_self_h<64, true>(P, A, s) : fwd_self_h<64, false>(P, A, s);
  return false;
}


bool xdma_bwd_self_h(const KParams& P, const void* x, const float* inv, const float* g, const float* affs, const float* dl, void* dx,
                     hipStream_t s) {
  if (!inv || !env().bwd_xdma || env().force_direct) return false;
  if (env().h16_hw == 2 && env().bwd_pf && xdma_hq_bwd_self(P, x, inv, g, affs, dl, dx, s)) return true;  // pea_k_xdma_hq.hip
  if (P.D == 16) return bwd_self_h<16>(P, (const __half*)x, inv, g, affs, dl, (__half*)dx, s);
  if (P.D == 32) return bwd_self_h<32>(P, (const __half*)x, inv, g, affs, dl, (__half*)dx, s);
  if (P.D == 64) return bwd_self_h<64>(P, (const __half*)x, inv, g, affs, dl, (__half*)dx, s);
  return false;
}

}  // namespace pea
