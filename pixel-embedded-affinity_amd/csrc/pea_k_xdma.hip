// pea_k_xdma.hip -- launchers of the LDS-DMA cross kernels (pea_xdma.h): plan, pick the instantiation, launch.
// One translation unit of libpea_hip.so (pea_host.h).
#include "pea_k_xdma_plan.h"
#include "pea_xdma_dual.h"

namespace pea {

namespace {

template <int D_T, bool TRAIN>
bool fwd_self(const KParams& P, const FwdArgs& A, hipStream_t s) {
  const float *e = (const float*)A.e, *t = A.t, *w = A.w;
  if (misaligned(e, 16) || misaligned(t, 16) || misaligned(w, 16) || misaligned(A.affs, 16) || misaligned(A.gout, 16) ||
      misaligned(A.m, 4) || misaligned(A.inv_out, 4))
    return false;
  if (TRAIN && ((P.tbs | P.wbs | P.mbs) & 3)) return false;
  XPlan X;
  bool z3 = false, wg3 = false;
  if (env().fwd_wg3 && plan(P, kXdmaPSUF, 1, &X) && X.C.nfz == 0) {
    wg3 = true;
  } else if (!plan(P, kXdmaPSU, 1, &X) || X.C.nfz > 0) {
    if (D_T != 16 || !plan(P, kXdmaPSU3F, 1, &X) || X.C.nfz == 0) return false;
    z3 = true;
  }
  if (!z3 && P.K > kXP) return false;
  if (z3 && P.K > kXP + 2) return false;
  const XParams& C = X.C;
  const dim3 grid((unsigned)(C.tiles_per_xcd * kXcd)), blk(kXdmaTH * kXdmaTW);
#define PEA_XF(CROP_, PSU_, ZF_, WPE_)                                                                                      \
  {                                                                                                                         \
    constexpr auto kern = k_fwd_xdma<D_T, kXdmaTH, kXdmaTW, PSU_, CROP_, TRAIN, ZF_, false, WPE_>;                          \
    PEA_LAUNCH(kern, grid, blk, X.lds, s, P, C, e, t, w, A.m, A.affs, A.gout, A.st, A.inv_out, (const float*)nullptr,       \
               (float*)nullptr, LabArgs{})                                                                                  \
  }
  const bool crop = P.border != PEA_BORDER_CIRCULAR;
  if constexpr (D_T == 16) {
    if (z3) {
      if (crop) PEA_XF(true, kXdmaPSU3F, kXZ / 2, 4) else PEA_XF(false, kXdmaPSU3F, kXZ / 2, 4)
      return true;
    }
  }
  if (wg3) {
    if (crop) PEA_XF(true, kXdmaPSUF, 0, 6) else PEA_XF(false, kXdmaPSUF, 0, 6)
  } else {
    if (crop) PEA_XF(true, kXdmaPSU, 0, 4) else PEA_XF(false, kXdmaPSU, 0, 4)
  }
#undef PEA_XF
  return true;
}

template <int D_T>
bool bwd_self(const KParams& P, const float* x, const float* inv, const float* g, const float* dl, float* dx, hipStream_t s) {
  if (misaligned(x, 16) || misaligned(inv, 16) || misaligned(g, 4) || misaligned(dx, 4)) return false;
  XPlan X;
  bool z3 = false;
  if (!plan(P, kXdmaPSU, 0, &X)) {
    if (D_T != 16 || !plan(P, kXdmaPSU3, 0, &X) || X.C.npz == 0) return false;
    z3 = true;
  } else if (X.C.npz > 0) {
    if (D_T != 16 || !plan(P, kXdmaPSU3, 0, &X)) return false;
    z3 = true;
  }
  const XParams& C = X.C;
  constexpr int XP = D_T > 32 ? 8 : kXP;  // pairs per axis the instantiation keeps in registers
  if (C.npx > (z3 ? 8 : XP) || C.npy > (z3 ? 8 : XP)) return false;
  const dim3 grid((unsigned)(C.tiles_per_xcd * kXcd)), blk(kXdmaTH * kXdmaTW);
#define PEA_XB(CROP_, XP_, PSU_, ZP_)                                                             \
  {                                                                                               \
    constexpr auto kern = k_bwd_xdma<D_T, kXdmaTH, kXdmaTW, PSU_, CROP_, XP_, kAuxNT, ZP_>;       \
    PEA_LAUNCH(kern, grid, blk, X.lds, s, P, C, x, inv, g, dl, dx, OtherArgs{}, DualArgs{})       \
  }
  const bool crop = P.border != PEA_BORDER_CIRCULAR;
  if constexpr (D_T == 16) {
    if (z3) {
      if (crop) PEA_XB(true, 8, kXdmaPSU3, kXZ) else PEA_XB(false, 8, kXdmaPSU3, kXZ)
      return true;
    }
  }
  if (crop) PEA_XB(true, XP, kXdmaPSU, 0) else PEA_XB(false, XP, kXdmaPSU, 0)
#undef PEA_XB
  return true;
}

}  // namespace

// the LDS-DMA forward of the self loss (f32 storage, D in {16, 32, 64}, axis-aligned stencil, K <= kXP; 3D norm1 / norm5 at D = 16).
// (Inference keeps k_fwd_tiled: 60 us against 68 us at B=8 x 544^2 -- without the epilogue streams the one-sided box of the
//  tiled kernel moves fewer bytes than six ring planes do.)
bool xdma_fwd_self(const KParams& P, const FwdArgs& A, hipStream_t s) {
  if (!env().fwd_xdma || env().force_direct || A.eo != A.e) return false;
  if (A.dtype == PEA_F16) {  // f16 storage: pea_xdma_h16.h
    return xdma_h_fwd_self(P, A, s);  // pea_k_xdma_h.hip
  }
  if (!A.train) {
    if (P.D == 16) return fwd_self<16, false>(P, A, s);
    if (P.D == 32) return fwd_self<32, false>(P, A, s);
    if (P.D == 64) return fwd_self<64, false>(P, A, s);
    return false;
  }
  if (P.D == 16) return fwd_self<16, true>(P, A, s);
  if (P.D == 32) return fwd_self<32, true>(P, A, s);
  if (P.D == 64) return fwd_self<64, true>(P, A, s);
  return false;
}

// the full-resolution pair of the 2D training loops as one forward launch (pea_xdma_dual.h): 2D, D = 16, f32, axis-aligned stencil
bool xdma_fwd_dual_supported(const KParams& P, int dtype) {
  if (!env().fwd_dual || !env().fwd_xdma || env().force_direct || dtype != PEA_F32 || P.D != 16 || P.Z != 1 || P.K > kXP) return false;
  XPlan X;
  return plan(P, kXdmaPSUF, 1, &X) && X.C.nfz == 0;
}

bool xdma_fwd_dual(const KParams& P, const KParams& P2, const FwdArgs& A, const FwdArgs& A2, hipStream_t s) {
  if (!xdma_fwd_dual_supported(P, A.dtype) || !A.gout || !A2.gout || !A.inv_out || !A2.inv_out) return false;
  const float *e = (const float*)A.e, *e2 = (const float*)A2.eo;
  if (misaligned(e, 16) || misaligned(e2, 16) || misaligned(A.t, 16) || misaligned(A.w, 16) || misaligned(A.affs, 16) ||
      misaligned(A.gout, 16) || misaligned(A2.gout, 16) || misaligned(A.m, 4) || misaligned(A.inv_out, 4) || misaligned(A2.inv_out, 4))
    return false;
  if ((P.tbs | P.wbs | P.mbs) & 3) return false;
  XPlan X;
  if (!plan(P, kXdmaPSUF, 1, &X) || X.C.nfz > 0) return false;
  DualFwdArgs DA = {};
  DA.e2 = e2; DA.gout2 = A2.gout; DA.inv_other_out = A2.inv_out; DA.st2 = A2.st;
  for (int i = 0; i < kXK; ++i) DA.gs2[i] = i < P2.K ? P2.gscale[i] : 0.f;
  const dim3 grid((unsigned)(X.C.tiles_per_xcd * kXcd)), blk(kXdmaTH * kXdmaTW);
  const bool crop = P.border != PEA_BORDER_CIRCULAR;
#define PEA_XFD(CROP_, NB_)                                                                            \
  {                                                                                                    \
    constexpr auto kern = k_fwd_xdma_dual<kXdmaTH, kXdmaTW, kXdmaPSUF, CROP_, NB_>;                    \
    const size_t lds = (size_t)(NB_ == 3 ? 3 : 2) * 4 * kXdmaPSUF * 256;                               \
    PEA_LAUNCH(kern, grid, blk, lds, s, P, X.C, e, A.t, A.w, A.m, A.affs, A.gout, A.st, A.inv_out, DA) \
  }
  // (NB = 4: the ring of two four-plane buffers handed over in halves; the whole-buffer hand-off -- NB = 2, +2.6 % -- and the ring of
  //  three at one workgroup per CU -- NB = 3, +21 % -- were switches until round 6 and are no longer compiled: EXPERIMENTS.md)
  if (crop) PEA_XFD(true, 4) else PEA_XFD(false, 4)
#undef PEA_XFD
  return true;
}

// the cross loss with a second operand: D = 16, f32, axis-aligned stencil; 2D images (either border) and 3D volumes whose stencil
// steps along z (ema_embedding_loss_norm5 / norm1, scripts_ac3ac4/loss/loss_embedding_mse.py:30-51, 237-289: CROP_ZERO; the z
// neighbours are the second operand's, gathered per chunk like the self loss' tile-per-plane kernels do).  e_other staged, own pixel
// from e, both 1 / norm planes written (inv_out[0 .. B*S) own, inv_out[B*S .. 2*B*S) second operand)
// D = 32 / 64 (2D): the own pixel comes from own TILES staged beside each chunk (k_fwd_xdma OWNL); one-sided cross in 7.5 KB planes
template <int D_T>
static bool fwd_other_wide(const KParams& P, const FwdArgs& A, hipStream_t s) {
  const float *e = (const float*)A.e, *e_other = (const float*)A.eo;
  if (misaligned(e, 16)) return false;
  XPlan X;
  if (!plan(P, kXdmaPSUF, 1, &X) || X.C.nfz > 0 || P.K > kXP || P.Z != 1) return false;
  const size_t lds = X.lds + (size_t)6 * 2048;  // + the own tiles: three buffers x two channels x 2 KB
  const dim3 grid((unsigned)(X.C.tiles_per_xcd * kXcd)), blk(kXdmaTH * kXdmaTW);
  if (P.border != PEA_BORDER_CIRCULAR) {
    constexpr auto kern = k_fwd_xdma<D_T, kXdmaTH, kXdmaTW, kXdmaPSUF, true, true, 0, true, 4>;
    PEA_LAUNCH(kern, grid, blk, lds, s, P, X.C, e_other, A.t, A.w, A.m, A.affs, A.gout, A.st, A.inv_out, e,
               A.inv_out + (size_t)P.B * P.S, LabArgs{})
  } else {
    constexpr auto kern = k_fwd_xdma<D_T, kXdmaTH, kXdmaTW, kXdmaPSUF, false, true, 0, true, 4>;
    PEA_LAUNCH(kern, grid, blk, lds, s, P, X.C, e_other, A.t, A.w, A.m, A.affs, A.gout, A.st, A.inv_out, e,
               A.inv_out + (size_t)P.B * P.S, LabArgs{})
  }
  return true;
}

bool xdma_fwd_other(const KParams& P, const FwdArgs& A, hipStream_t s) {
  if (!env().fwd_xdma || env().force_direct || !A.train || !A.inv_out) return false;
  if (A.dtype == PEA_F16) return xdma_h_fwd_other(P, A, s);  // pea_k_xdma_h.hip
  if (A.dtype != PEA_F32) return false;
  if (P.D != 16 && P.D != 32 && P.D != 64) return false;
  const float *e = (const float*)A.e, *e_other = (const float*)A.eo;
  if (misaligned(e, 4) || misaligned(e_other, 16) || misaligned(A.t, 16) || misaligned(A.w, 16) || misaligned(A.affs, 16) ||
      misaligned(A.gout, 16) || misaligned(A.m, 4) || misaligned(A.inv_out, 4))
    return false;
  if ((P.tbs | P.wbs | P.mbs) & 3) return false;
  if (P.D == 32) return fwd_other_wide<32>(P, A, s);
  if (P.D == 64) return fwd_other_wide<64>(P, A, s);
  XPlan X;
  // (three workgroups per CU do not fit here: the own pixel's 16 registers on top of the accumulators spill at 80 VGPRs)
  bool z3 = false;
  if (!plan(P, kXdmaPSU, 1, &X) || X.C.nfz > 0) {
    if (!plan(P, kXdmaPSU3F, 1, &X) || X.C.nfz == 0) return false;
    z3 = true;
  }
  if (P.K > (z3 ? kXP + 2 : kXP)) return false;
  const dim3 grid((unsigned)(X.C.tiles_per_xcd * kXcd)), blk(kXdmaTH * kXdmaTW);
  const bool crop = P.border != PEA_BORDER_CIRCULAR;
#define PEA_XFO(CROP_, PSU_, ZF_)                                                                                           \
  {                                                                                                                         \
    constexpr auto kern = k_fwd_xdma<16, kXdmaTH, kXdmaTW, PSU_, CROP_, true, ZF_, true>;                                   \
    PEA_LAUNCH(kern, grid, blk, X.lds, s, P, X.C, e_other, A.t, A.w, A.m, A.affs, A.gout, A.st, A.inv_out, e,               \
               A.inv_out + (size_t)P.B * P.S, LabArgs{})                                                                    \
  }
  if (z3) {
    if (crop) PEA_XFO(true, kXdmaPSU3F, kXZ / 2) else PEA_XFO(false, kXdmaPSU3F, kXZ / 2)
  } else {
    if (crop) PEA_XFO(true, kXdmaPSU, 0) else PEA_XFO(false, kXdmaPSU, 0)
  }
#undef PEA_XFO
  return true;
}

// the labels-in forward (k_fwd_xdma<.., LAB>): self loss, 2D, f32, axis-aligned stencil, K <= kXP; g_out and inv_out are required
// (the backward that follows is xdma_bwd_self).  true = launched.
template <int D_T>
static bool fwd_labels(const KParams& P, const FwdArgs& A, const LabArgs& LA, hipStream_t s) {
  const float* e = (const float*)A.e;
  if (misaligned(e, 16) || misaligned(A.affs, 16) || misaligned(A.gout, 16) || misaligned(A.inv_out, 4) || misaligned(LA.labels, 16))
    return false;
  XPlan X;
  if (!plan(P, kXdmaPSUF, 1, &X) || X.C.nfz > 0 || P.K > kXP || P.Z != 1) return false;
  const size_t lds = X.lds + (size_t)kXdmaPSUF * 256;  // + the label plane
  const dim3 grid((unsigned)(X.C.tiles_per_xcd * kXcd)), blk(kXdmaTH * kXdmaTW);
  if (P.border != PEA_BORDER_CIRCULAR) {
    constexpr auto kern = k_fwd_xdma<D_T, kXdmaTH, kXdmaTW, kXdmaPSUF, true, true, 0, false, 6, true>;
    PEA_LAUNCH(kern, grid, blk, lds, s, P, X.C, e, (const float*)nullptr, (const float*)nullptr, (const uint8_t*)nullptr, A.affs, A.gout,
               A.st, A.inv_out, (const float*)nullptr, (float*)nullptr, LA)
  } else {
    constexpr auto kern = k_fwd_xdma<D_T, kXdmaTH, kXdmaTW, kXdmaPSUF, false, true, 0, false, 6, true>;
    PEA_LAUNCH(kern, grid, blk, lds, s, P, X.C, e, (const float*)nullptr, (const float*)nullptr, (const uint8_t*)nullptr, A.affs, A.gout,
               A.st, A.inv_out, (const float*)nullptr, (float*)nullptr, LA)
  }
  return true;
}

// would xdma_fwd_labels + xdma_bwd_self take this descriptor (16-byte aligned tensors assumed)?
bool xdma_labels_supported(const KParams& P, int dtype) {
  if (!env().fwd_xdma || !env().bwd_xdma || env().force_direct || dtype != PEA_F32 || (P.D != 16 && P.D != 32) || P.Z != 1) return false;
  XPlan X;
  if (!plan(P, kXdmaPSUF, 1, &X) || X.C.nfz > 0 || P.K > kXP) return false;
  return plan(P, kXdmaPSU, 0, &X) && X.C.npz == 0 && X.C.npx <= kXP && X.C.npy <= kXP;
}

bool xdma_fwd_labels(const KParams& P, const FwdArgs& A, const int32_t* labels, const float* wtab, unsigned lflags, hipStream_t s) {
  if (!env().fwd_xdma || !env().bwd_xdma || env().force_direct || A.dtype != PEA_F32 || !A.gout || !A.inv_out) return false;
  const LabArgs LA = {labels, wtab, lflags};
  if (P.D == 16) return fwd_labels<16>(P, A, LA, s);
  if (P.D == 32) return fwd_labels<32>(P, A, LA, s);
  return false;
}

// the cross backward (self loss, f32 storage, axis-aligned stencil): needs the 1 / norm plane; with the raw affinity map too
// (affs: what the forward wrote with no activation flag) the projection-first kernel runs
bool xdma_bwd_self(const KParams& P, const float* x, const float* inv, const float* g, const float* affs, const float* dl, float* dx,
                   hipStream_t s) {
  if (!inv || !env().bwd_xdma || env().force_direct) return false;
  // (D = 16 keeps k_bwd_xdma: with G in 16 registers and the own pixel kept there is no second read to save, and the stores
  //  inside the chunk loop cost more than they give: 137 against 94-107 us at the bench shape)
  if (affs && env().bwd_pf && !(P.flags & kActMask) && P.D > 16) {
    bool done = false;
    done = xdma_pf_bwd_self(P, x, inv, g, affs, dl, dx, s);  // pea_k_xdma_pf.hip
    if (done) return true;
  }
  if (P.D == 16) return bwd_self<16>(P, x, inv, g, dl, dx, s);
  if (P.D == 32) return bwd_self<32>(P, x, inv, g, dl, dx, s);
  if (P.D == 64) return bwd_self<64>(P, x, inv, g, dl, dx, s);
  return false;
}

// backward, role A only (the second operand is detached): de (+)= dloss * d loss / d e.  2D (either border) and 3D volumes with z steps
bool xdma_bwd_other(const KParams& P, const float* e, const float* e_other, const float* inv2, const float* g, const float* affs,
                    const float* dl, float* de, bool accumulate, hipStream_t s) {
  if (!inv2 || !env().bwd_xdma || env().force_direct) return false;
  if (P.D == 32 || P.D == 64) {  // projection first: needs the forward's raw map; no accumulate form
    if (!affs || accumulate || (P.flags & kActMask) || !env().bwd_pf || P.Z != 1) return false;
    return xdma_pf_bwd_other(P, e, e_other, inv2, g, affs, dl, de, s);
  }
  if (P.D != 16) return false;
  if (misaligned(e_other, 16) || misaligned(inv2, 16) || ((size_t)P.B * P.S) % 4) return false;
  XPlan X;
  bool z3 = false;
  if (!plan(P, kXdmaPSU, 2, &X) || X.C.npz > 0) {
    if (!plan(P, kXdmaPSU3, 2, &X) || X.C.npz == 0) return false;
    z3 = true;
  }
  if (X.C.npx > (z3 ? 8 : kXP) || X.C.npy > (z3 ? 8 : kXP)) return false;
  const dim3 grid((unsigned)(X.C.tiles_per_xcd * kXcd)), blk(kXdmaTH * kXdmaTW);
  OtherArgs O;
  O.own = e; O.own_inv = inv2; O.accumulate = accumulate ? 1 : 0;
  const bool crop = P.border != PEA_BORDER_CIRCULAR;
#define PEA_XBO(CROP_, XP_, PSU_, ZP_)                                                                                      \
  {                                                                                                                         \
    constexpr auto kern = k_bwd_xdma<16, kXdmaTH, kXdmaTW, PSU_, CROP_, XP_, kAuxNT, ZP_, true>;                            \
    PEA_LAUNCH(kern, grid, blk, X.lds, s, P, X.C, e_other, inv2 + (size_t)P.B * P.S, g, dl, de, O, DualArgs{})              \
  }
  if (z3) {
    if (crop) PEA_XBO(true, 8, kXdmaPSU3, kXZ / 2) else PEA_XBO(false, 8, kXdmaPSU3, kXZ / 2)
  } else {
    if (crop) PEA_XBO(true, kXP, kXdmaPSU, 0) else PEA_XBO(false, kXP, kXdmaPSU, 0)
  }
#undef PEA_XBO
  return true;
}

// the pair's backward in one launch (k_bwd_xdma<.., DUAL>): 2D, D = 16, f32, circular border
bool xdma_bwd_dual(const KParams& P, const float* e, const float* ema, const float* inv, const float* inv_other, const float* g,
                   const float* g_cross, const float* dl, const float* dl_cross, float* de, hipStream_t s) {
  if (!inv || !inv_other || !env().bwd_xdma || env().force_direct || P.D != 16 || P.border != PEA_BORDER_CIRCULAR) return false;
  if (misaligned(e, 16) || misaligned(ema, 16) || misaligned(inv, 16) || misaligned(inv_other, 16)) return false;
  XPlan X, X2;
  if (!plan(P, kXdmaPSU, 0, &X) || X.C.npz > 0 || X.C.npx > kXP || X.C.npy > kXP) return false;
  if (!plan(P, kXdmaPSU, 2, &X2)) return false;
  DualArgs Q;
  Q.C2 = X2.C;
  Q.ema = ema; Q.inv_other = inv_other; Q.g_cross = g_cross; Q.dloss_cross = dl_cross;
  const dim3 grid((unsigned)(X.C.tiles_per_xcd * kXcd)), blk(kXdmaTH * kXdmaTW);
  constexpr auto kern = k_bwd_xdma<16, kXdmaTH, kXdmaTW, kXdmaPSU, false, kXP, kAuxNT, 0, false, true>;
  PEA_LAUNCH(kern, grid, blk, X.lds, s, P, X.C, e, inv, g, dl, de, OtherArgs{}, Q)
  return true;
}

void launch_inv_norm(const KParams& P, int dtype, const void* e, float* inv, hipStream_t s) {
  const dim3 grid((unsigned)(P.tiles_per_xcd * kXcd)), blk(kBlock);
  if (dtype == PEA_F16) hipLaunchKernelGGL(k_inv_norm<__half>, grid, blk, 0, s, P, (const __half*)e, inv);
  else hipLaunchKernelGGL(k_inv_norm<float>, grid, blk, 0, s, P, (const float*)e, inv);
}

// pea_cross_supported: would the cross kernels take this descriptor?  mode 0: forward, 1: backward (self loss),
// 2: the cross loss with a detached second operand (forward + role-A backward)
int xdma_cross_supported(const KParams& P, int dtype, int mode) {
  if ((P.D != 16 && P.D != 32 && P.D != 64) || env().force_direct) return 0;
  if (!(mode ? env().bwd_xdma : env().fwd_xdma)) return 0;
  XPlan X;
  if (dtype == PEA_F16) {  // pea_xdma_h16.h: 2D self loss
    if (P.X % 8 || P.Z != 1) return 0;
    if (mode == 2 || mode == 4) {  // the cross loss with a detached second operand: forward + projection-first role-A backward
      if (!env().h16_hw || !env().bwd_pf || !env().fwd_xdma || (P.flags & kActMask) || P.K > kXP) return 0;
      if (!plan(P, kXdmaPSUF, 1, &X) || X.C.nfz > 0) return 0;
      return (plan(P, kXdmaPSUF, 2, &X) && X.C.npz == 0 && X.C.npx <= kXP && X.C.npy <= kXP) ? 1 : 0;
    }
    if (mode == 0) return (plan(P, kXdmaPSUF, 1, &X) && X.C.nfz == 0 && P.K <= kXP) ? 1 : 0;
    if (!plan(P, kXdmaPSUH, 0, &X) || X.C.npz > 0) return 0;
    const int xp = P.D > 32 ? 8 : kXP;
    return (X.C.npx <= xp && X.C.npy <= xp) ? 1 : 0;
  }
  if (mode == 2 || mode == 4) {  // 4: does the backward of the cross loss with a detached second operand READ the raw map?
    if (!env().fwd_xdma) return 0;
    if (P.D == 32 || P.D == 64) {
      if (!env().bwd_pf || (P.flags & kActMask) || P.Z != 1 || P.K > kXP) return 0;
      if (!plan(P, kXdmaPSUF, 1, &X) || X.C.nfz > 0) return 0;
      return (plan(P, kXdmaPSUF, 2, &X) && X.C.npz == 0 && X.C.npx <= kXP && X.C.npy <= kXP) ? 1 : 0;
    }
    if (P.D != 16 || mode == 4) return 0;
    if (plan(P, kXdmaPSU, 1, &X) && X.C.nfz == 0)
      return (P.K <= kXP && plan(P, kXdmaPSU, 2, &X) && X.C.npx <= kXP && X.C.npy <= kXP) ? 1 : 0;
    // 3D volumes whose stencil steps along z (ema_embedding_loss_norm5 / norm1): the tile-per-plane instantiations
    if (!plan(P, kXdmaPSU3F, 1, &X) || X.C.nfz == 0 || P.K > kXP + 2) return 0;
    return (plan(P, kXdmaPSU3, 2, &X) && X.C.npz > 0 && X.C.npx <= 8 && X.C.npy <= 8) ? 1 : 0;
  }
  const int pm = mode == 0 ? 1 : 0;
  if (!plan(P, kXdmaPSU3, pm, &X)) return 0;
  const bool z3 = X.C.npz > 0 || X.C.nfz > 0;
  if (z3) return (P.D == 16 && X.C.npx <= 8 && X.C.npy <= 8 && P.K <= kXP + 2 && (mode || plan(P, kXdmaPSU3F, 1, &X))) ? 1 : 0;
  if (!plan(P, kXdmaPSU, pm, &X)) return 0;
  if (!mode && P.K > kXP) return 0;
  return (mode && P.D > 32 && (X.C.npx > 8 || X.C.npy > 8)) ? 0 : 1;
}

}  // namespace pea
