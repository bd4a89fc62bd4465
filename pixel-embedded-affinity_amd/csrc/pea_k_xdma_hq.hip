// pea_k_xdma_hq.hip -- launcher of the producer / consumer f16-storage cross backward (pea_xdma_hq.h).  One translation unit of
// libpea_hip.so (pea_host.h); tried by pea_k_xdma_h.hip's dispatcher before the LDS-DMA form (PEA_H16_HW=2, the default).
#include "pea_k_xdma_plan.h"
#include "pea_xdma_hq.h"

namespace pea {

namespace {

constexpr int kHqTH = 8, kHqTW = 64;  // a row of the tile is one 128-byte line of an f16 plane
constexpr int kHqPSU = 30;            // working buffers of 2 x 7680 bytes: crosses whose region fits 1920 pixels (reach <= 9 both ways)
constexpr int kHqNPW = 4;             // producer waves

template <int D_T>
bool bwd_hq(const KParams& P, const __half* x, const float* inv, const float* g, const float* affs, const float* dl, __half* dx,
            hipStream_t s) {
  if (P.X % 8 || P.Z != 1 || misaligned(x, 16) || misaligned(inv, 16) || misaligned(g, 4) || misaligned(dx, 2) || misaligned(affs, 4))
    return false;
  if (!affs || (P.flags & kActMask)) return false;  // the projection comes from the forward's raw map
  XPlan X;
  if (!plan(P, kHqPSU, 0, &X, kHqTH, kHqTW) || X.C.npz > 0 || X.C.npx > kXP || X.C.npy > kXP) return false;
  if (X.C.QA > kHqPSU * 16 || X.C.QA > 2 * 64 * kHqNPW) return false;  // one oct per producer lane
  const dim3 grid((unsigned)(X.C.tiles_per_xcd * kXcd)), blk(kHqTH * kHqTW + 64 * kHqNPW);
  const bool crop = P.border != PEA_BORDER_CIRCULAR;
  // (D = 64 with at most eight pairs per axis -- BASELINE configs[4] -- walks eight pairs per axis instead of ten)
  constexpr int XPS = D_T == 64 ? 8 : kXP;
  const bool few = D_T == 64 && X.C.npx <= 8 && X.C.npy <= 8;
  // 59-67 VGPRs: two workgroups of 12 waves per CU
#define PEA_HQS(CROP_, XP_)                                                                        \
  {                                                                                                \
    constexpr auto kern = k_bwd_xdma_hqs<D_T, kHqTH, kHqTW, kHqPSU, CROP_, XP_, 6, kHqNPW, 2>;      \
    PEA_LAUNCH(kern, grid, blk, (size_t)4 * kHqPSU * 256, s, P, X.C, x, inv, g, affs, dl, dx)      \
  }
  if (few) { if (crop) PEA_HQS(true, XPS) else PEA_HQS(false, XPS) }
  else { if (crop) PEA_HQS(true, kXP) else PEA_HQS(false, kXP) }
#undef PEA_HQS
  return true;
}

}  // namespace

bool xdma_hq_bwd_self(const KParams& P, const void* x, const float* inv, const float* g, const float* affs, const float* dl, void* dx,
                      hipStream_t s) {
  if (P.D == 32) return bwd_hq<32>(P, (const __half*)x, inv, g, affs, dl, (__half*)dx, s);
  if (P.D == 64) return bwd_hq<64>(P, (const __half*)x, inv, g, affs, dl, (__half*)dx, s);
  return false;
}

}  // namespace pea
