// pea_k_zmarch.hip -- launchers of the z-march kernels (pea_zmarch.h): 3D volumes whose axis-aligned stencil steps along z
// (embedding_loss_norm5 / norm1, scripts_ac3ac4/loss/loss_embedding_mse.py:7-27, 143-194).  One translation unit of libpea_hip.so.
#include "pea_host.h"
#include "pea_zmarch.h"

namespace pea {

namespace {

constexpr int kTH = 16, kTW = 32;
constexpr int kPSUB = 52;  // backward: the two-sided +-27 cross in whole 64-quad blocks (13 KB planes, 78 KB of ring)
constexpr int kPSUF = 32;  // forward: the one-sided cross in 8 KB planes (48 KB of ring)

struct ZPlan { XParams C; ZMParams M; size_t lds; int nb; };  // nb: ring depth of the backward the LDS was sized for

// mode 0: backward (both roles), 1: forward.  The tile walk of xdma_tile is reused with the SEGMENT in the place of z: blocks of
// tile columns (one XCD's 32 workgroups march through one block: its halos are shared out of that XCD's L2), the segments of a
// block one after the other.
bool plan(const KParams& P, int mode, ZPlan* out) {
  static thread_local PlanCache<ZPlan, 8> cache;
  // (the segmentation depends on the CURRENT device's CU count: part of the key -- a thread that serves devices of different sizes must
  //  not be handed the other device's plan; round-4 advice)
  const Env& E = env();
  return cache.get(P, mode * 4096 + E.zmarch * 64 + E.zseg + (device_cus() << 13) + (E.zm_sup << 24), out, [&](ZPlan* p) {
    p->nb = 4;  // (a ring of three -- k_bwd_zm<.., 3>, a switch until round 6 -- is 5 % slower once the waits count loads only)
    if (!env().zmarch || !plan_zmarch(P, &p->M)) return false;
    if (!plan_xdma(P, kTH, kTW, mode ? kPSUF : kPSUB, &p->C, &p->lds, mode)) return false;
    XParams& C = p->C;
    if (mode == 0 && (C.npx > 8 || C.npy > 8)) return false;
    // forward: a ring of eight buffers (16 planes) + the parked dot products [kXP + 2][tile] + the loss partials
    if (mode == 1) p->lds = (size_t)16 * kPSUF * 256 + (size_t)(kXP + 2) * kTH * kTW * 4 + 256;
    // backward: the ring of four two-plane buffers + the waves' blocks of prefetched g / a values
    else p->lds = (size_t)2 * p->nb * kPSUB * 256 + (size_t)(kTH * kTW / 64) * kZmG * 256;
    const long long cols = (long long)P.B * C.tiles_per_plane;
    // one workgroup per CU: whole columns when there are enough of them for two rounds, else segments of >= 8 planes
    int nseg = 1;
    if (env().zseg > 0) nseg = (P.Z + env().zseg - 1) / env().zseg;
    else if (cols < 2 * device_cus()) nseg = (int)std::min<long long>((2 * device_cus() + cols - 1) / cols, std::max(1, P.Z / 8));
    // a volume that small keeps the tile-per-plane kernels (two workgroups per CU, no warm-up planes); PEA_ZMARCH=2 forces the march
    if (env().zmarch < 2 && cols * nseg < device_cus()) return false;
    p->M.nseg = nseg;
    p->M.zseg = (P.Z + nseg - 1) / nseg;
    p->M.nseg = (P.Z + p->M.zseg - 1) / p->M.zseg;
    C.zrun = p->M.nseg;
    // The walk.  Round 6 (profiles/r6_sup.txt, r6_sup_wide*.txt: 60 combinations, same process): the eight XCDs' blocks of a round as ONE
    // super-block (march_tile) -- blocks of 8 x 2 tile columns (128 rows x 64 pixels), the XCDs 4 high x 2 wide: 512 rows x 128 pixels per
    // super-block, two super-blocks per XCD in flight -- 1.422 + 1.743 ms against 1.546 + 1.805 for round 4's walk (every XCD on its own
    // contiguous range of 16 x 2 blocks).  Every arrangement whose super-block is 512 rows tall and 32 .. 128 pixels wide lands within
    // 1 % of it (4 x 2 / 4 x 1 / 4 x 4 stacked eight high, 16 x 1 two high); 256 pixels wide +1.7 %; 16 x 2 side by side +5 %.
    // Super-blocks are padded to the tile grid, so they are taken only where the grid is a whole number of them (else XCDs would idle).
    const int sx = E.zm_sup < 0 ? 2 : E.zm_sup;
    const bool sup_ok = sx == 1 || sx == 2 || sx == 4 || sx == 8;
    C.zgy = E.zblk_y > 0 ? E.zblk_y : 16;
    C.zgx = E.zblk_x > 0 ? E.zblk_x : 2;
    const long long nt = cols * p->M.nseg;
    if (nt > 0x7fffff00LL) return false;
    C.ntiles = (int)nt;
    C.tiles_per_xcd = (C.ntiles + kXcd - 1) / kXcd;
    if (sup_ok) {
      const int gy = E.zblk_y > 0 ? E.zblk_y : 8, gx = E.zblk_x > 0 ? E.zblk_x : 2, sy = kXcd / sx;
      const bool whole = C.tiles_y % (gy * sy) == 0 && C.tiles_x % (gx * sx) == 0;
      if (whole || E.zm_sup > 0) {  // (an explicit PEA_ZM_SUP: also on ragged grids -- the tests' padded super-blocks)
        C.zgy = gy; C.zgx = gx; C.sup_x = sx; C.sup_y = sy;
        const long long nsx = (C.tiles_x + gx * sx - 1) / (gx * sx), nsy = (C.tiles_y + gy * sy - 1) / (gy * sy);
        const long long per_xcd = (long long)P.B * nsx * nsy * gy * gx * p->M.nseg;
        if (per_xcd * kXcd > 0x7fffff00LL) return false;
        C.tiles_per_xcd = (int)per_xcd;
      }
    }
    return true;
  });
}

#define PEA_LAUNCH(kern, grid, blk, lds, s, ...)              \
  {                                                           \
    if (allow_lds<kern>(lds)) return false;                   \
    hipLaunchKernelGGL(kern, grid, blk, lds, s, __VA_ARGS__); \
  }

}  // namespace

// forward / inference of the self loss on a 3D volume (f32, D = 16, CROP_ZERO, every z offset in {-1 .. -4}); true = launched
bool zmarch_fwd(const KParams& P, const FwdArgs& A, hipStream_t s) {
  if (env().force_direct || !env().fwd_xdma || A.eo != A.e || A.dtype != PEA_F32) return false;
  const float* e = (const float*)A.e;
  if (misaligned(e, 16) || misaligned(A.t, 16) || misaligned(A.w, 16) || misaligned(A.affs, 16) || misaligned(A.gout, 16) ||
      misaligned(A.m, 4) || misaligned(A.inv_out, 4))
    return false;
  if (A.train && ((P.tbs | P.wbs | P.mbs) & 3)) return false;
  ZPlan Z;
  if (!plan(P, 1, &Z)) return false;
  const dim3 grid((unsigned)(Z.C.tiles_per_xcd * kXcd)), blk(kTH * kTW);
#define PEA_ZF(TRAIN_, NXP_)                                                                                                      \
  {                                                                                                                               \
    constexpr auto kern = k_fwd_zm<kTH, kTW, kPSUF, TRAIN_, NXP_>;                                                                \
    PEA_LAUNCH(kern, grid, blk, Z.lds, s, P, Z.C, Z.M, e, A.train ? A.t : nullptr, A.train ? A.w : nullptr,                       \
               A.train ? A.m : nullptr, A.affs, A.train ? A.gout : nullptr, A.train ? A.st : nullptr, A.train ? A.inv_out : nullptr) \
  }
  const bool few = Z.C.nf <= 8;  // the reference's 3D tables have at most eight in-plane offsets
  if (A.train) { if (few) PEA_ZF(true, 8) else PEA_ZF(true, kXP) }
  else { if (few) PEA_ZF(false, 8) else PEA_ZF(false, kXP) }
#undef PEA_ZF
  return true;
}

// would zmarch_bwd take this descriptor (given the 1 / norm plane and the raw affinity map)?
bool zmarch_bwd_supported(const KParams& P, int dtype) {
  if (env().force_direct || !env().bwd_xdma || dtype != PEA_F32 || (P.flags & kActMask)) return false;
  ZPlan Z;
  return plan(P, 0, &Z);
}

// backward of the self loss: needs the forward's 1 / norm plane AND its raw affinity map (the z channels are read)
bool zmarch_bwd(const KParams& P, const float* x, const float* inv, const float* g, const float* affs, const float* dl, float* dx,
                hipStream_t s) {
  if (!inv || !affs || !zmarch_bwd_supported(P, PEA_F32)) return false;
  if (misaligned(x, 16) || misaligned(inv, 16) || misaligned(g, 4) || misaligned(affs, 4) || misaligned(dx, 4)) return false;
  ZPlan Z;
  if (!plan(P, 0, &Z)) return false;
  const dim3 grid((unsigned)(Z.C.tiles_per_xcd * kXcd)), blk(kTH * kTW);
  constexpr auto kern = k_bwd_zm<kTH, kTW, kPSUB, 4>;  // (the ring depth the plan sized its LDS for)
  PEA_LAUNCH(kern, grid, blk, Z.lds, s, P, Z.C, Z.M, x, inv, g, affs, dl, dx)
  return true;
}

}  // namespace pea
