// pea_hip.hip — MI355X (gfx950 / CDNA4) kernels + C ABI of the embedding -> affinity hot path.
//
// Replaces the Python op loops of the reference (weih527/Pixel-Embedded-Affinity):
//   scripts_cvppp/loss/loss_embedding_mse.py:7-95   (2D: normalize -> K x torch.roll/mul/sum -> WeightedMSE)
//   scripts_ac3ac4/loss/loss_embedding_mse.py:7-289 (3D: cropped slices, norm1 / norm5 / ema / inf)
//   loss/loss.py:106-124 WeightedMSE                (fused)
// and the autograd backward of those, by one forward launch (+ a tiny deterministic loss
// reduction) and one backward launch.  See include/pea.h for the contract and DESIGN.md for the
// data layout, the per-kernel roofline and the algorithmic byte counts.
//
// This file is the host side: descriptor validation, tile planning and kernel dispatch.
//   pea_tiled.h   LDS-tiled kernels (fast path; D = 16 this round)
//   pea_direct.h  global-memory kernels (general fallback) + the deterministic loss finalize
// Common to both: HBM-bound op (2-7 flop/B) => no MFMA; the L2 norm of a neighbour is accumulated while
// its channels stream in, so ehat is never materialised in HBM; the 1-D grid is remapped so each of the 8
// XCDs walks a contiguous span of tiles (stencil re-reads hit that XCD's own L2); loss partials per
// workgroup in a caller-provided workspace -> fixed-order f64 reduction; backward in gather form.  No
// float atomics anywhere: results are bit-reproducible run to run.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "pea_direct.h"
#include "pea_tiled.h"
#include "pea_targets.h"
#include "pea_fused_labels.h"
#include "pea_head.h"
#include "pea_chunked.h"
#include "pea_xdma.h"

using namespace pea;

namespace {

// ------------------------------------------------------------------------------------------------
// descriptor -> kernel parameters
// ------------------------------------------------------------------------------------------------
int validate(const PeaDesc* d) {
  if (!d) return PEA_E_NULL;
  if (d->abi != PEA_ABI_VERSION) return PEA_E_DESC;
  if (d->ndim != 2 && d->ndim != 3) return PEA_E_DESC;
  if (d->B < 1 || d->D < 1 || d->K < 1 || d->K > PEA_MAX_K) return PEA_E_DESC;
  for (int a = 0; a < 3; ++a)
    if (d->dims[a] < 1) return PEA_E_DESC;
  if (d->ndim == 2 && d->dims[0] != 1) return PEA_E_DESC;
  if (d->border != PEA_BORDER_CIRCULAR && d->border != PEA_BORDER_CROP_ZERO && d->border != PEA_BORDER_REPLICATE) return PEA_E_DESC;
  if (d->dtype != PEA_F32 && d->dtype != PEA_F16) return PEA_E_DESC;
  if (d->norm < PEA_NORM_BX || d->norm > PEA_NORM_FULL) return PEA_E_DESC;
  if (!(d->eps > 0.f)) return PEA_E_DESC;
  if (d->target_bstride < 0 || d->weight_bstride < 0 || d->mask_bstride < 0) return PEA_E_DESC;
  const long long S = (long long)d->dims[0] * d->dims[1] * d->dims[2];
  if (S > 0x7fffffffLL - kBlock) return PEA_E_UNSUPPORTED;
  if ((S + kBlock - 1) / kBlock * (long long)d->B > 0x7fffff00LL) return PEA_E_UNSUPPORTED;
  for (int i = 0; i < d->K; ++i)
    for (int a = 0; a < 3; ++a) {
      const int o = d->offsets[i][a];
      // |o| < dim: torch.roll would wrap further, but no reference stencil does; cropped slices need it
      if (o <= -d->dims[a] || o >= d->dims[a]) return PEA_E_DESC;
    }
  return PEA_OK;
}

KParams make_params(const PeaDesc* d) {
  KParams P;
  P.B = d->B; P.D = d->D; P.Z = d->dims[0]; P.Y = d->dims[1]; P.X = d->dims[2]; P.K = d->K;
  P.S = P.Z * P.Y * P.X;
  P.border = d->border; P.flags = d->flags; P.eps = d->eps;
  P.ksplit = ((long long)d->K * P.S * 4 >= (1LL << 31)) ? (d->K + 1) / 2 : d->K;
  P.chunks = (P.S + kBlock - 1) / kBlock;
  P.tiles = P.B * P.chunks;
  P.tiles_per_xcd = (P.tiles + kXcd - 1) / kXcd;
  const long long dense = (long long)P.K * P.S;
  P.tbs = d->target_bstride ? d->target_bstride : dense;
  P.wbs = d->weight_bstride ? d->weight_bstride : dense;
  P.mbs = d->mask_bstride ? d->mask_bstride : dense;
  for (int i = 0; i < PEA_MAX_K; ++i) {
    const bool on = i < d->K;
    double n = 1.0;
    if (on) {
      if (d->norm == PEA_NORM_BX) n = (double)d->B * d->dims[2];
      else if (d->norm == PEA_NORM_FULL) n = (double)d->B * P.S;
      else {
        n = d->B;
        for (int a = 0; a < 3; ++a) n *= (double)(d->dims[a] - abs(d->offsets[i][a]));
      }
    }
    for (int a = 0; a < 3; ++a) P.off[i][a] = on ? d->offsets[i][a] : 0;
    P.lam[i] = on ? d->lambda[i] : 0.f;
    P.inv_n[i] = on ? (float)(1.0 / n) : 0.f;
    P.gscale[i] = on ? (float)(2.0 * (double)d->lambda[i] / n) : 0.f;
  }
  return P;
}

inline int hip_rc() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? PEA_OK : (int)e;
}

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

bool misaligned(const void* p, size_t a) { return ((uintptr_t)p & (a - 1)) != 0; }

// CUs of the CURRENT device (asked every time: the reference runs replicas under nn.DataParallel threads, one device each, so a
// process-wide cache of the first device's answer would be wrong for the others, and a static would not be thread-safe)
int device_cus() {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
    return v;
  (void)hipGetLastError();
  return 256;
}

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
bool plan_tiles(const KParams& P, TileCfg c, bool both_sides, TParams* Q, bool ksplit_ok = false) {
  if (P.border == PEA_BORDER_REPLICATE) return false;  // direct kernels only (row a-15: an unused variant of the reference)
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

// Raise the kernel's dynamic-LDS limit above the 64 KB default.  Asked for on every launch that needs it: the attribute is
// kept per device, so a cached "already raised" flag of the first device would leave the launch on a second device of the
// same process failing; the call is a host-side table update (no state of ours, thread-safe).  A failure surfaces through
// hip_rc() after the launch.
template <auto KERNEL>
void allow_lds(size_t bytes) {
  if (bytes > 64 * 1024) (void)hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// workspace = [K][fwd_partials] f32 loss partials
// [K][nparts] float partials, then (8-byte aligned) the [K][kFinSlices] double slice sums of the two-level loss reduction
constexpr size_t kSliceBytes = (size_t)PEA_MAX_K * kFinSlices * sizeof(double);
size_t ws_bytes(size_t nparts_max, int K) { return ((nparts_max * K * sizeof(float) + 7) & ~(size_t)7) + kSliceBytes; }

size_t fwd_partials(const KParams& P) {
  // worst case over the paths pea_affinity_fwd may take
  size_t n = (size_t)P.tiles;
  for (const TileCfg& c : kFwdCfg) {
    TParams q;
    if (plan_tiles(P, c, false, &q)) n = std::max(n, (size_t)q.ntiles);
  }
  n = std::max(n, (size_t)((P.Y + 15) / 16) * ((P.X + 31) / 32) * P.Z * P.B);      // the LDS-DMA forward's 16x32 tiles
  return n;
}

void launch_loss_finalize(const KParams& P, float* partials, int nparts, float* loss_out, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// 1 / norm plane (pea_xdma.h): written by the tiled D = 16 forward while it stages; by this kernel otherwise
// ------------------------------------------------------------------------------------------------
template <typename T>
void launch_inv_norm(const KParams& P, const T* e, float* inv, hipStream_t s) {
  hipLaunchKernelGGL(k_inv_norm<T>, dim3((unsigned)(P.tiles_per_xcd * kXcd)), dim3(kBlock), 0, s, P, e, inv);
}

// the cross backward (self loss, f32 storage, axis-aligned in-plane stencil): needs the 1 / norm plane
constexpr int kXdmaTH = 16, kXdmaTW = 32, kXdmaPSU = 51;
constexpr int kXdmaPSU3 = 52;  // 3D instantiations: whole 64-quad blocks (13 KB planes), 6 x 13312 B = 78 KB, still two workgroups per CU
template <int D_T>
bool try_bwd_xdma(const KParams& P, const float* x, const float* inv, const float* g, const float* dl, float* dx, hipStream_t s) {
  if (!inv || env_int("PEA_BWD_XDMA", 1) == 0) return false;
  if (misaligned(x, 16) || misaligned(inv, 16) || misaligned(g, 4) || misaligned(dx, 4)) return false;
  XParams C;
  size_t lds;
  bool z3 = false;
  if (!plan_xdma(P, kXdmaTH, kXdmaTW, kXdmaPSU, &C, &lds)) {
    if (D_T != 16 || !plan_xdma(P, kXdmaTH, kXdmaTW, kXdmaPSU3, &C, &lds) || C.npz == 0) return false;
    z3 = true;
  } else if (C.npz > 0) {
    if (D_T != 16 || !plan_xdma(P, kXdmaTH, kXdmaTW, kXdmaPSU3, &C, &lds)) return false;
    z3 = true;
  }
  constexpr int XP = D_T > 32 ? 8 : kXP;  // pairs per axis the instantiation keeps in registers
  if (C.npx > (z3 ? 8 : XP) || C.npy > (z3 ? 8 : XP)) return false;
  const dim3 grid((unsigned)(C.tiles_per_xcd * kXcd)), blk(kXdmaTH * kXdmaTW);
#define PEA_XB(CROP_, XP_, PSU_, ZP_)                                                                  \
  {                                                                                                    \
    constexpr auto kern = k_bwd_xdma<D_T, kXdmaTH, kXdmaTW, PSU_, CROP_, XP_, kAuxNT, ZP_>;            \
    allow_lds<kern>(lds);                                                                              \
    hipLaunchKernelGGL(kern, grid, blk, lds, s, P, C, x, inv, g, dl, dx, HeadArgs{}, OtherArgs{}, DualArgs{}); \
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

// the cross backward with the embedding head's backward in its epilogue (k_bwd_xdma<.., HC>): 2D, D = 16, circular border
// (every CVPPP call), C = 32 input channels (outconv_emb of ResidualUNet2D_deep).  `ntiles_out`: rows of H.partials written.
constexpr int kHeadFuseC = 32;
bool plan_bwd_head(const KParams& P, XParams* C, size_t* lds) {
  if (P.D != 16 || P.border != PEA_BORDER_CIRCULAR) return false;
  if (!plan_xdma(P, kXdmaTH, kXdmaTW, kXdmaPSU, C, lds) || C->npz > 0) return false;
  return C->npx <= kXP && C->npy <= kXP;
}
bool try_bwd_xdma_head(const KParams& P, const float* x, const float* inv, const float* g, const float* dl, float* de,
                       const HeadArgs& H, hipStream_t s, int* ntiles_out) {
  XParams C;
  size_t lds;
  if (!plan_bwd_head(P, &C, &lds)) return false;
  const dim3 grid((unsigned)(C.tiles_per_xcd * kXcd)), blk(kXdmaTH * kXdmaTW);
  constexpr auto kern = k_bwd_xdma<16, kXdmaTH, kXdmaTW, kXdmaPSU, false, kXP, kAuxNT, 0, kHeadFuseC>;
  allow_lds<kern>(lds);
  hipLaunchKernelGGL(kern, grid, blk, lds, s, P, C, x, inv, g, dl, de, H, OtherArgs{}, DualArgs{});
  *ntiles_out = C.ntiles;
  return true;
}

// the cross loss with a second operand on the cross kernels: 2D, D = 16, f32, circular border, axis-aligned stencil.
// forward: e_other staged, own pixel from e, both 1 / norm planes written (inv2[0 .. B*S) own, inv2[B*S .. 2*B*S) second operand)
bool try_fwd_xdma_other(const KParams& P, const float* e, const float* e_other, const float* t, const float* w, const uint8_t* m,
                        float* affs, float* gout, float* partials, float* inv2, hipStream_t s, int* nparts) {
  if (env_int("PEA_FWD_XDMA", 1) == 0 || P.D != 16 || P.border != PEA_BORDER_CIRCULAR) return false;
  if (misaligned(e, 4) || misaligned(e_other, 16) || misaligned(t, 16) || misaligned(w, 16) || misaligned(affs, 16) ||
      misaligned(gout, 16) || misaligned(m, 4) || misaligned(inv2, 4))
    return false;
  if ((P.tbs | P.wbs | P.mbs) & 3) return false;
  XParams C;
  size_t lds;
  if (!plan_xdma(P, kXdmaTH, kXdmaTW, kXdmaPSU, &C, &lds, 1) || C.nfz > 0 || P.K > kXP) return false;
  const dim3 grid((unsigned)(C.tiles_per_xcd * kXcd)), blk(kXdmaTH * kXdmaTW);
  constexpr auto kern = k_fwd_xdma<16, kXdmaTH, kXdmaTW, kXdmaPSU, false, true, 0, true>;
  allow_lds<kern>(lds);
  hipLaunchKernelGGL(kern, grid, blk, lds, s, P, C, e_other, t, w, m, affs, gout, partials, inv2, e,
                     inv2 ? inv2 + (size_t)P.B * P.S : nullptr);
  *nparts = C.ntiles;
  return true;
}
// backward, role A only (the second operand is detached): de (+)= dloss * d loss / d e
bool try_bwd_xdma_other(const KParams& P, const float* e, const float* e_other, const float* inv2, const float* g, const float* dl,
                        float* de, bool accumulate, hipStream_t s) {
  if (!inv2 || env_int("PEA_BWD_XDMA", 1) == 0 || P.D != 16 || P.border != PEA_BORDER_CIRCULAR) return false;
  if (misaligned(e_other, 16) || misaligned(inv2, 16) || ((size_t)P.B * P.S) % 4) return false;
  XParams C;
  size_t lds;
  if (!plan_xdma(P, kXdmaTH, kXdmaTW, kXdmaPSU, &C, &lds, 2) || C.npx > kXP || C.npy > kXP) return false;
  const dim3 grid((unsigned)(C.tiles_per_xcd * kXcd)), blk(kXdmaTH * kXdmaTW);
  constexpr auto kern = k_bwd_xdma<16, kXdmaTH, kXdmaTW, kXdmaPSU, false, kXP, kAuxNT, 0, 0, true>;
  allow_lds<kern>(lds);
  OtherArgs O;
  O.own = e; O.own_inv = inv2; O.accumulate = accumulate ? 1 : 0;
  hipLaunchKernelGGL(kern, grid, blk, lds, s, P, C, e_other, inv2 + (size_t)P.B * P.S, g, dl, de, HeadArgs{}, O, DualArgs{});
  return true;
}

// the pair's backward in one launch on the cross kernels (k_bwd_xdma<.., DUAL>): 2D, D = 16, f32, circular border
bool try_bwd_xdma_dual(const KParams& P, const float* e, const float* ema, const float* inv, const float* inv_other, const float* g,
                       const float* g_cross, const float* dl, const float* dl_cross, float* de, hipStream_t s) {
  if (!inv || !inv_other || env_int("PEA_BWD_XDMA", 1) == 0 || P.D != 16 || P.border != PEA_BORDER_CIRCULAR) return false;
  if (misaligned(e, 16) || misaligned(ema, 16) || misaligned(inv, 16) || misaligned(inv_other, 16)) return false;
  XParams C;
  DualArgs Q;
  size_t lds, lds2;
  if (!plan_xdma(P, kXdmaTH, kXdmaTW, kXdmaPSU, &C, &lds, 0) || C.npz > 0 || C.npx > kXP || C.npy > kXP) return false;
  if (!plan_xdma(P, kXdmaTH, kXdmaTW, kXdmaPSU, &Q.C2, &lds2, 2)) return false;
  Q.ema = ema; Q.inv_other = inv_other; Q.g_cross = g_cross; Q.dloss_cross = dl_cross;
  const dim3 grid((unsigned)(C.tiles_per_xcd * kXcd)), blk(kXdmaTH * kXdmaTW);
  constexpr auto kern = k_bwd_xdma<16, kXdmaTH, kXdmaTW, kXdmaPSU, false, kXP, kAuxNT, 0, 0, false, true>;
  allow_lds<kern>(lds);
  hipLaunchKernelGGL(kern, grid, blk, lds, s, P, C, e, inv, g, dl, de, HeadArgs{}, OtherArgs{}, Q);
  return true;
}

// the LDS-DMA forward (self loss / inference, D = 16, f32, axis-aligned in-plane stencil, K <= kXP)
template <int D_T, bool TRAIN>
bool try_fwd_xdma(const KParams& P, const float* e, const float* t, const float* w, const uint8_t* m, float* affs, float* gout,
                  float* partials, float* inv_out, hipStream_t s, int* nparts) {
  if (env_int("PEA_FWD_XDMA", 1) == 0) return false;
  if (misaligned(e, 16) || misaligned(t, 16) || misaligned(w, 16) || misaligned(affs, 16) || misaligned(gout, 16) ||
      misaligned(m, 4) || misaligned(inv_out, 4))
    return false;
  if (TRAIN && ((P.tbs | P.wbs | P.mbs) & 3)) return false;
  XParams C;
  size_t lds;
  bool z3 = false;
  if (!plan_xdma(P, kXdmaTH, kXdmaTW, kXdmaPSU, &C, &lds, true) || C.nfz > 0) {
    if (D_T != 16 || !plan_xdma(P, kXdmaTH, kXdmaTW, kXdmaPSU3, &C, &lds, true) || C.nfz == 0) return false;
    z3 = true;
  }
  if (!z3 && P.K > kXP) return false;
  if (z3 && P.K > kXP + 2) return false;
  const dim3 grid((unsigned)(C.tiles_per_xcd * kXcd)), blk(kXdmaTH * kXdmaTW);
#define PEA_XF(CROP_, PSU_, ZF_)                                                                       \
  {                                                                                                    \
    constexpr auto kern = k_fwd_xdma<D_T, kXdmaTH, kXdmaTW, PSU_, CROP_, TRAIN, ZF_>;                  \
    allow_lds<kern>(lds);                                                                              \
    hipLaunchKernelGGL(kern, grid, blk, lds, s, P, C, e, t, w, m, affs, gout, partials, inv_out, (const float*)nullptr, (float*)nullptr); \
  }
  const bool crop = P.border != PEA_BORDER_CIRCULAR;
  bool done = false;
  if constexpr (D_T == 16) {
    if (z3) {
      if (crop) PEA_XF(true, kXdmaPSU3, kXZ / 2) else PEA_XF(false, kXdmaPSU3, kXZ / 2)
      done = true;
    }
  }
  if (!done) {
    if (crop) PEA_XF(true, kXdmaPSU, 0) else PEA_XF(false, kXdmaPSU, 0)
  }
#undef PEA_XF
  *nparts = C.ntiles;
  return true;
}

// ------------------------------------------------------------------------------------------------
// forward dispatch
// ------------------------------------------------------------------------------------------------
template <typename T, int D_T, bool TRAIN, bool SELF, int CI>
void launch_fwd_cfg(const KParams& P, const TParams& Q, const T* e, const T* eo, const float* t, const float* w,
                    const uint8_t* m, float* affs, float* gout, float* partials, float* inv_out, hipStream_t s) {
  constexpr TileCfg c = fwd_cfg<D_T>(CI);
  constexpr int NT = c.TH * c.TW;
  const size_t lds = Lds<D_T, c.PLQ>::kBytes + (TRAIN ? (size_t)(NT / 64) * P.K * sizeof(float) : 0);
  const dim3 grid((unsigned)(Q.tiles_per_xcd * kXcd)), blk(NT);
  if (P.border == PEA_BORDER_CIRCULAR) {
    constexpr auto kern = k_fwd_tiled<T, D_T, c.TH, c.TW, c.PLQ, false, TRAIN, SELF>;
    allow_lds<kern>(lds);
    hipLaunchKernelGGL(kern, grid, blk, lds, s, P, Q, e, eo, t, w, m, affs, gout, partials, inv_out);
  } else {
    constexpr auto kern = k_fwd_tiled<T, D_T, c.TH, c.TW, c.PLQ, true, TRAIN, SELF>;
    allow_lds<kern>(lds);
    hipLaunchKernelGGL(kern, grid, blk, lds, s, P, Q, e, eo, t, w, m, affs, gout, partials, inv_out);
  }
}

// forward with the LDS-transposed, dwordx4 epilogue (k_fwd_tiled_v): 16x32 tiles, the dot products laid over the dead region,
// two workgroups of 8 waves per CU.  The training forward wherever the LDS-DMA kernel (k_fwd_xdma) does not apply.
template <typename T, int D_T, bool TRAIN, bool SELF>
void launch_fwd_v(const KParams& P, const TParams& Q, size_t lds, const T* e, const T* eo, const float* t, const float* w,
                  const uint8_t* m, float* affs, float* gout, float* partials, float* inv_out, hipStream_t s) {
  constexpr TileCfg c = fwdv_cfg<D_T>();
  const dim3 grid((unsigned)(Q.tiles_per_xcd * kXcd)), blk(c.TH * c.TW);
  if (P.border == PEA_BORDER_CIRCULAR) {
    constexpr auto kern = k_fwd_tiled_v<T, D_T, c.TH, c.TW, c.PLQ, true, false, TRAIN, SELF>;
    allow_lds<kern>(lds);
    hipLaunchKernelGGL(kern, grid, blk, lds, s, P, Q, e, eo, t, w, m, affs, gout, partials, inv_out);
  } else {
    constexpr auto kern = k_fwd_tiled_v<T, D_T, c.TH, c.TW, c.PLQ, true, true, TRAIN, SELF>;
    allow_lds<kern>(lds);
    hipLaunchKernelGGL(kern, grid, blk, lds, s, P, Q, e, eo, t, w, m, affs, gout, partials, inv_out);
  }
}

template <typename T, int D_T, bool TRAIN>
bool try_fwd_v(const KParams& P, const T* e, const T* eo, const float* t, const float* w, const uint8_t* m, float* affs,
               float* gout, float* partials, float* inv_out, hipStream_t s, int* nparts) {
  if (P.K > kKV || P.X % 4) return false;
  if (misaligned(t, 16) || misaligned(w, 16) || misaligned(affs, 16) || misaligned(gout, 16) || misaligned(m, 4)) return false;
  if ((P.tbs | P.wbs | P.mbs | (long long)P.S) & 3) return false;
  const TileCfg c = fwdv_cfg<D_T>();
  const size_t tp = (size_t)c.TH * c.TW;
  const size_t region = Lds<D_T, 1>::kBytes * (size_t)c.PLQ, dots = (size_t)P.K * tp * 4, parts = (size_t)P.K * (tp / 256) * 4;
  if (dots > region) return false;
  const size_t lds = region + parts;
  if (lds > (size_t)kLdsMax) return false;
  TParams Q;
  if (!plan_tiles(P, c, false, &Q) || Q.n_near > kKV || Q.n_far > kFV) return false;
  if (eo == e) launch_fwd_v<T, D_T, TRAIN, true>(P, Q, lds, e, eo, t, w, m, affs, gout, partials, inv_out, s);
  else launch_fwd_v<T, D_T, TRAIN, false>(P, Q, lds, e, eo, t, w, m, affs, gout, partials, inv_out, s);
  *nparts = Q.ntiles;
  return true;
}

// returns true if a tiled kernel was launched (nparts = number of partial rows written)
template <typename T, int D_T, bool TRAIN>
bool try_fwd_tiled(const KParams& P, const T* e, const T* eo, const float* t, const float* w, const uint8_t* m, float* affs,
                   float* gout, float* partials, float* inv_out, hipStream_t s, int* nparts) {
  TParams Q;
  if (!plan_tiles(P, fwd_cfg<D_T>(0), false, &Q, true)) return false;
  if (eo == e) launch_fwd_cfg<T, D_T, TRAIN, true, 0>(P, Q, e, eo, t, w, m, affs, gout, partials, inv_out, s);
  else launch_fwd_cfg<T, D_T, TRAIN, false, 0>(P, Q, e, eo, t, w, m, affs, gout, partials, inv_out, s);
  *nparts = Q.ntiles;
  return true;
}

// D = 64: channels through LDS in two chunks of 32 (k_fwd_tiled_chunked, the D = 32 region geometry);
// D = 32: two chunks of 16 in the D = 16 geometry, i.e. two workgroups per CU instead of one (inference 176 -> 139 us,
// training forward 255 -> 237 us at B=8 x 32 x 544^2; PEA_FWD_CHUNKED32=0 restores the one-region kernels)
template <typename T, int D_T, int DC, bool TRAIN, bool SELF>
void launch_fwd_chunked(const KParams& P, const TParams& Q, size_t lds, const T* e, const T* eo, const float* t, const float* w,
                        const uint8_t* m, float* affs, float* gout, float* partials, hipStream_t s) {
  constexpr TileCfg c = kCfg32;  // 16 x 32 tile, 1041 region pixels: 128 B (DC = 32) or 64 B (DC = 16) of LDS each
  const dim3 grid((unsigned)(Q.tiles_per_xcd * kXcd)), blk(c.TH * c.TW);
  if (P.border == PEA_BORDER_CIRCULAR) {
    constexpr auto kern = k_fwd_tiled_chunked<T, D_T, DC, c.TH, c.TW, c.PLQ, false, TRAIN, SELF>;
    allow_lds<kern>(lds);
    hipLaunchKernelGGL(kern, grid, blk, lds, s, P, Q, e, eo, t, w, m, affs, gout, partials);
  } else {
    constexpr auto kern = k_fwd_tiled_chunked<T, D_T, DC, c.TH, c.TW, c.PLQ, true, TRAIN, SELF>;
    allow_lds<kern>(lds);
    hipLaunchKernelGGL(kern, grid, blk, lds, s, P, Q, e, eo, t, w, m, affs, gout, partials);
  }
}

template <typename T, int D_T, int DC, bool TRAIN>
bool try_fwd_chunked(const KParams& P, const T* e, const T* eo, const float* t, const float* w, const uint8_t* m, float* affs,
                     float* gout, float* partials, hipStream_t s, int* nparts) {
  if (P.D != D_T) return false;
  constexpr TileCfg c = kCfg32;
  TParams Q;
  if (!plan_tiles(P, c, false, &Q) || Q.n_near > kChN || Q.n_far > kChF) return false;
  const size_t lds = Lds<DC, c.PLQ>::kBytes + (size_t)c.PLQ * 4 + (size_t)(c.TH * c.TW / 64) * P.K * 4;
  if (lds > (size_t)kLdsMax) return false;
  if (eo == e) {
    launch_fwd_chunked<T, D_T, DC, TRAIN, true>(P, Q, lds, e, eo, t, w, m, affs, gout, partials, s);
  } else {
    // a second operand under the 128-VGPR budget of DC = 16 spills: not instantiated (the caller keeps the one-region kernels)
    if constexpr (DC > 16) launch_fwd_chunked<T, D_T, DC, TRAIN, false>(P, Q, lds, e, eo, t, w, m, affs, gout, partials, s);
    else return false;
  }
  *nparts = Q.ntiles;
  return true;
}

template <typename T, bool TRAIN>
int launch_fwd(const KParams& P, const void* e, const void* eo, const float* t, const float* w, const uint8_t* m,
               float* affs, float* gout, float* partials, float* inv_out, hipStream_t s, int* nparts) {
  const T* ep = (const T*)e;
  const T* op = eo ? (const T*)eo : ep;
  if (inv_out && (eo != nullptr && eo != e)) inv_out = nullptr;  // self loss only (caller runs k_inv_norm otherwise)
  if (env_int("PEA_FORCE_DIRECT", 0) == 0) {
    bool done = false;
    if constexpr (sizeof(T) == 4 && TRAIN) {
      // (inference keeps k_fwd_tiled: 60 us against 68 us at B=8 x 544^2 -- without the epilogue streams the one-sided
      //  box of the tiled kernel moves fewer bytes than six ring planes do)
      if (op == ep) {
        if (P.D == 16) done = try_fwd_xdma<16, TRAIN>(P, (const float*)ep, t, w, m, affs, gout, partials, inv_out, s, nparts);
        else if (P.D == 32) done = try_fwd_xdma<32, TRAIN>(P, (const float*)ep, t, w, m, affs, gout, partials, inv_out, s, nparts);
        else if (P.D == 64) done = try_fwd_xdma<64, TRAIN>(P, (const float*)ep, t, w, m, affs, gout, partials, inv_out, s, nparts);
        if (done) return hip_rc();
      }
    }
    if (!done && P.D == 16 && TRAIN) done = try_fwd_v<T, 16, TRAIN>(P, ep, op, t, w, m, affs, gout, partials, inv_out, s, nparts);
    if (!done && P.D == 16) done = try_fwd_tiled<T, 16, TRAIN>(P, ep, op, t, w, m, affs, gout, partials, inv_out, s, nparts);
    if (done && P.D == 16) return hip_rc();
    if (inv_out) launch_inv_norm<T>(P, ep, inv_out, s);  // the kernels below do not write the plane themselves
    // 64 B of LDS per region pixel: two workgroups per CU.  Self loss / inference only: with a second operand the
    // 128-VGPR budget of that occupancy spills (and see pea_chunked.h on spill stores), so EMA calls keep the one-region kernels
    if (P.D == 32)
      done = try_fwd_chunked<T, 32, 16, TRAIN>(P, ep, op, t, w, m, affs, gout, partials, s, nparts);
    if (!done && P.D == 32 && TRAIN) done = try_fwd_v<T, 32, TRAIN>(P, ep, op, t, w, m, affs, gout, partials, nullptr, s, nparts);
    if (!done && P.D == 32) done = try_fwd_tiled<T, 32, TRAIN>(P, ep, op, t, w, m, affs, gout, partials, nullptr, s, nparts);
    if (P.D == 64) done = try_fwd_chunked<T, 64, 32, TRAIN>(P, ep, op, t, w, m, affs, gout, partials, s, nparts);
    if (done) return hip_rc();
  } else if (inv_out) {
    launch_inv_norm<T>(P, ep, inv_out, s);
  }
  const size_t lds = TRAIN ? (size_t)P.K * kBlock * sizeof(float) : 0;
  const dim3 g((unsigned)(P.tiles_per_xcd * kXcd)), blk(kBlock);
  *nparts = P.tiles;
  switch (P.D) {
    case 16: hipLaunchKernelGGL((k_fwd_direct<T, 16, TRAIN>), g, blk, lds, s, P, ep, op, t, w, m, affs, gout, partials); break;
    case 32: hipLaunchKernelGGL((k_fwd_direct<T, 32, TRAIN>), g, blk, lds, s, P, ep, op, t, w, m, affs, gout, partials); break;
    case 64: hipLaunchKernelGGL((k_fwd_direct<T, 64, TRAIN>), g, blk, lds, s, P, ep, op, t, w, m, affs, gout, partials); break;
    default: hipLaunchKernelGGL((k_fwd_direct<T, 0, TRAIN>), g, blk, lds, s, P, ep, op, t, w, m, affs, gout, partials); break;
  }
  return hip_rc();
}

// ------------------------------------------------------------------------------------------------
// backward dispatch
// ------------------------------------------------------------------------------------------------
template <typename T, int D_T, bool RA, bool RB, int CI>
void launch_bwd_cfg(const KParams& P, const TParams& Q, const T* x, const T* nb, const float* g, const float* dl, T* dx,
                    hipStream_t s) {
  constexpr TileCfg c = bwd_cfg<D_T>(CI);
  const size_t lds = Lds<D_T, c.PLQ>::kBytes;
  const dim3 grid((unsigned)(Q.tiles_per_xcd * kXcd)), blk(c.TH * c.TW);
  if (P.border == PEA_BORDER_CIRCULAR) {
    constexpr auto kern = k_bwd_tiled<T, D_T, c.TH, c.TW, c.PLQ, false, RA, RB>;
    allow_lds<kern>(lds);
    hipLaunchKernelGGL(kern, grid, blk, lds, s, P, Q, x, nb, g, dl, dx);
  } else {
    constexpr auto kern = k_bwd_tiled<T, D_T, c.TH, c.TW, c.PLQ, true, RA, RB>;
    allow_lds<kern>(lds);
    hipLaunchKernelGGL(kern, grid, blk, lds, s, P, Q, x, nb, g, dl, dx);
  }
}

template <typename T, int D_T, bool RA, bool RB>
bool try_bwd_tiled(const KParams& P, const T* x, const T* nb, const float* g, const float* dl, T* dx, hipStream_t s) {
  TParams Q;
  // role A alone (a detached second operand's cross loss) reaches only p + o: a one-sided halo; role B needs p - o
  if (!plan_tiles(P, bwd_cfg<D_T>(0), !(RA && !RB), &Q, true)) return false;
  launch_bwd_cfg<T, D_T, RA, RB, 0>(P, Q, x, nb, g, dl, dx, s);
  return true;
}

template <typename T, int D_T>
int launch_bwd_roles(const KParams& P, int roles, const T* x, const T* nbA, const T* nbB, const float* g, const float* dl,
                     T* dx, hipStream_t s) {
  if ((D_T == 16 || D_T == 32) && env_int("PEA_FORCE_DIRECT", 0) == 0) {
    constexpr int DT = D_T == 32 ? 32 : 16;
    bool done = false;
    if (roles == 3) done = try_bwd_tiled<T, DT, true, true>(P, x, nbA, g, dl, dx, s);
    else if (roles == 1) done = try_bwd_tiled<T, DT, true, false>(P, x, nbA, g, dl, dx, s);
    else done = try_bwd_tiled<T, DT, false, true>(P, x, nbB, g, dl, dx, s);
    if (done) return hip_rc();
  }
  const dim3 grid((unsigned)(P.tiles_per_xcd * kXcd)), blk(kBlock);
  if (roles == 3) hipLaunchKernelGGL((k_bwd_direct<T, D_T, true, true>), grid, blk, 0, s, P, x, nbA, nbB, g, dl, dx);
  else if (roles == 1) hipLaunchKernelGGL((k_bwd_direct<T, D_T, true, false>), grid, blk, 0, s, P, x, nbA, nbB, g, dl, dx);
  else hipLaunchKernelGGL((k_bwd_direct<T, D_T, false, true>), grid, blk, 0, s, P, x, nbA, nbB, g, dl, dx);
  return hip_rc();
}

template <typename T>
int launch_bwd(const KParams& P, int roles, const void* x, const void* nbA, const void* nbB, const float* g, const float* dl,
               void* dx, hipStream_t s) {
  switch (P.D) {
    case 16: return launch_bwd_roles<T, 16>(P, roles, (const T*)x, (const T*)nbA, (const T*)nbB, g, dl, (T*)dx, s);
    case 32: return launch_bwd_roles<T, 32>(P, roles, (const T*)x, (const T*)nbA, (const T*)nbB, g, dl, (T*)dx, s);
    case 64: return launch_bwd_roles<T, 64>(P, roles, (const T*)x, (const T*)nbA, (const T*)nbB, g, dl, (T*)dx, s);
    case 4: return launch_bwd_roles<T, 4>(P, roles, (const T*)x, (const T*)nbA, (const T*)nbB, g, dl, (T*)dx, s);
    case 8: return launch_bwd_roles<T, 8>(P, roles, (const T*)x, (const T*)nbA, (const T*)nbB, g, dl, (T*)dx, s);
    default: break;
  }
  // any other width: the runtime-D kernel (pea_direct.h); the REPLICATE border keeps the specialised kernels
  if (P.border == PEA_BORDER_REPLICATE) return PEA_E_UNSUPPORTED;
  const size_t lds = (size_t)2 * P.K * kBlock * sizeof(float);
  const dim3 grid((unsigned)(P.tiles_per_xcd * kXcd)), blk(kBlock);
#define PEA_ANYD(RA_, RB_)                                                                                           \
  {                                                                                                                  \
    constexpr auto kern = k_bwd_direct_anyd<T, RA_, RB_>;                                                            \
    allow_lds<kern>(lds);                                                                                            \
    hipLaunchKernelGGL(kern, grid, blk, lds, s, P, (const T*)x, (const T*)nbA, (const T*)nbB, g, dl, (T*)dx);        \
  }
  if (roles == 3) PEA_ANYD(true, true) else if (roles == 1) PEA_ANYD(true, false) else PEA_ANYD(false, true)
#undef PEA_ANYD
  return hip_rc();
}


// ------------------------------------------------------------------------------------------------
// training step from labels (pea_fused_labels.h): same tile plan as the tiled backward
// ------------------------------------------------------------------------------------------------
template <typename T, int D_T, bool RB>
bool try_fused_labels(const KParams& P, const T* x, const T* nb, const int32_t* labels, const float* wtab, unsigned lflags,
                      float* affs, float* partials, const float* dl, T* dx, hipStream_t s, int* nparts) {
  constexpr TileCfg c = bwd_cfg<D_T>(0);
  TParams Q;
  if (!plan_tiles(P, c, RB, &Q)) return false;
  const size_t lds = Lds<D_T, c.PLQ>::kBytes + (size_t)(c.TH * c.TW / 64) * P.K * sizeof(float);
  if (lds > (size_t)kLdsMax) return false;
  const dim3 grid((unsigned)(Q.tiles_per_xcd * kXcd)), blk(c.TH * c.TW);
  if (P.border == PEA_BORDER_CIRCULAR) {
    constexpr auto kern = k_fused_labels<T, D_T, c.TH, c.TW, c.PLQ, false, RB>;
    allow_lds<kern>(lds);
    hipLaunchKernelGGL(kern, grid, blk, lds, s, P, Q, x, nb, labels, wtab, lflags, affs, partials, dl, dx);
  } else {
    constexpr auto kern = k_fused_labels<T, D_T, c.TH, c.TW, c.PLQ, true, RB>;
    allow_lds<kern>(lds);
    hipLaunchKernelGGL(kern, grid, blk, lds, s, P, Q, x, nb, labels, wtab, lflags, affs, partials, dl, dx);
  }
  *nparts = Q.ntiles;
  return true;
}

// self + detached-EMA cross loss from labels in one launch (k_fused_labels_dual): both plans must split the stencil
// into the same near / far entries
template <typename T, int D_T>
bool try_fused_labels_dual(const KParams& P, const KParams& P2, const T* x, const T* ema, const int32_t* labels, const float* wtab,
                           unsigned lflags, float* affs, float* partials, float* partials2, const float* dl, const float* dl2, T* dx,
                           hipStream_t s, int* nparts) {
  constexpr TileCfg c = bwd_cfg<D_T>(0);
  TParams Q, Q2;
  if (!plan_tiles(P, c, true, &Q) || !plan_tiles(P2, c, false, &Q2)) return false;
  if (Q.n_near > kDualNear || Q2.n_near != Q.n_near || Q2.n_far != Q.n_far || Q2.ntiles != Q.ntiles) return false;
  CrossPar C2 = {};
  for (int k = 0; k < Q.n_near; ++k) {
    if (Q2.near[k].i != Q.near[k].i) return false;
    C2.d2[k] = Q2.near[k].d;
  }
  for (int k = 0; k < Q.n_far; ++k)
    if (Q2.far[k].i != Q.far[k].i) return false;
  for (int i = 0; i < PEA_MAX_K; ++i) C2.gscale[i] = P2.gscale[i];
  const size_t lds = Lds<D_T, c.PLQ>::kBytes + 2 * (size_t)(c.TH * c.TW / 64) * P.K * sizeof(float);
  if (lds > (size_t)kLdsMax) return false;
  const dim3 grid((unsigned)(Q.tiles_per_xcd * kXcd)), blk(c.TH * c.TW);
  if (P.border == PEA_BORDER_CIRCULAR) {
    constexpr auto kern = k_fused_labels_dual<T, D_T, c.TH, c.TW, c.PLQ, false>;
    allow_lds<kern>(lds);
    hipLaunchKernelGGL(kern, grid, blk, lds, s, P, Q, Q2, C2, x, ema, labels, wtab, lflags, affs, partials, partials2, dl, dl2, dx);
  } else {
    constexpr auto kern = k_fused_labels_dual<T, D_T, c.TH, c.TW, c.PLQ, true>;
    allow_lds<kern>(lds);
    hipLaunchKernelGGL(kern, grid, blk, lds, s, P, Q, Q2, C2, x, ema, labels, wtab, lflags, affs, partials, partials2, dl, dl2, dx);
  }
  *nparts = Q.ntiles;
  return true;
}

// self backward + detached-EMA cross backward in one launch (k_bwd_tiled_dual)
template <typename T, int D_T>
bool try_bwd_dual(const KParams& P, const T* x, const T* ema, const float* g, const float* g2, const float* dl, const float* dl2, T* dx,
                  hipStream_t s) {
  constexpr TileCfg c = bwd_cfg<D_T>(0);
  TParams Q, Q2;
  if (!plan_tiles(P, c, true, &Q) || !plan_tiles(P, c, false, &Q2)) return false;
  if (Q.n_near > 8 || Q2.n_near != Q.n_near || Q2.n_far != Q.n_far) return false;
  BwdCross C2 = {};
  for (int k = 0; k < Q.n_near; ++k) {
    if (Q2.near[k].i != Q.near[k].i) return false;
    C2.d2[k] = Q2.near[k].d;
  }
  for (int k = 0; k < Q.n_far; ++k)
    if (Q2.far[k].i != Q.far[k].i) return false;
  const size_t lds = Lds<D_T, c.PLQ>::kBytes;
  const dim3 grid((unsigned)(Q.tiles_per_xcd * kXcd)), blk(c.TH * c.TW);
  if (P.border == PEA_BORDER_CIRCULAR) {
    constexpr auto kern = k_bwd_tiled_dual<T, D_T, c.TH, c.TW, c.PLQ, false>;
    allow_lds<kern>(lds);
    hipLaunchKernelGGL(kern, grid, blk, lds, s, P, Q, Q2, C2, x, ema, g, g2, dl, dl2, dx);
  } else {
    constexpr auto kern = k_bwd_tiled_dual<T, D_T, c.TH, c.TW, c.PLQ, true>;
    allow_lds<kern>(lds);
    hipLaunchKernelGGL(kern, grid, blk, lds, s, P, Q, Q2, C2, x, ema, g, g2, dl, dl2, dx);
  }
  return true;
}

template <typename T>
__global__ __launch_bounds__(256) void k_scale_inplace(T* __restrict__ buf, size_t n4, size_t n, const float* __restrict__ scale) {
  const float sc = scale[0];
  if (sc == 1.0f) return;  // the common loss.backward() case: nothing to do, nothing touched
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (sizeof(T) == 4) {
    if (i < n4) {
      f4 v = ((f4*)buf)[i];
      v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
      ((f4*)buf)[i] = v;
    }
    if (i < n - 4 * n4) ((float*)buf)[4 * n4 + i] *= sc;
  } else {
    for (size_t k = i * 4; k < min(n, i * 4 + 4); ++k) st(buf, k, ld(buf, k) * sc);
  }
}

// Caller epilogue of the 3D path (scripts_ac3ac4/main.py:233-237,296-300; inference.py:160-164): for c in {0,1,2} the
// first `shift` slices of affs[:, c] along axis c (z, y, x) are overwritten with slices shift .. 2*shift-1, then relu.
// shift == 0: plain F.relu in place, 4 floats per lane (the general kernel below spends its time on index divisions)
__global__ __launch_bounds__(256) void k_relu_inplace(float* __restrict__ a, size_t n4, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    f4 v = ((f4*)a)[i];
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    ((f4*)a)[i] = v;
  }
  if (i < n - 4 * n4) a[4 * n4 + i] = fmaxf(a[4 * n4 + i], 0.f);
}

__global__ __launch_bounds__(256) void k_fill_border_relu(float* __restrict__ affs, int B, int K, int Z, int Y, int X, int shift,
                                                          int relu) {
  const size_t S = (size_t)Z * Y * X, n = (size_t)B * K * S;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)((i / S) % (size_t)K);
  const size_t p = i % S;
  const int z = (int)(p / ((size_t)Y * X)), y = (int)((p / X) % Y), x = (int)(p % X);
  size_t src = i;
  if (shift > 0 && c < 3) {
    const int a = c == 0 ? z : c == 1 ? y : x;
    const size_t stride = c == 0 ? (size_t)Y * X : c == 1 ? (size_t)X : 1;
    if (a < shift) src = i + (size_t)shift * stride;  // pred[..., :shift] = pred[..., shift:2*shift]
  }
  float v = affs[src];
  if (relu) v = fmaxf(v, 0.f);
  if (src != i || relu) affs[i] = v;
}

// 3D inference stitcher (scripts_ac3ac4/data/provider_valid.py:320-349): out[:, window] += vol * w ; wmap[window] += w,
// then out /= wmap.  Product and sum are rounded separately (no FMA) so the result is bit-identical to numpy's.
__global__ __launch_bounds__(256) void k_stitch_add(float* __restrict__ out, float* __restrict__ wmap, const float* __restrict__ vol,
                                                    const float* __restrict__ wv, int C, int Z, int Y, int X, int oz, int oy, int ox,
                                                    int z0, int y0, int x0) {
  const size_t n = (size_t)oz * oy * ox;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int z = (int)(i / ((size_t)oy * ox)), y = (int)((i / ox) % oy), x = (int)(i % ox);
  const size_t o = ((size_t)(z0 + z) * Y + (y0 + y)) * X + (x0 + x), S = (size_t)Z * Y * X;
  const float w = wv[i];
  for (int c = 0; c < C; ++c) {
    float prod = vol[c * n + i] * w;
    asm volatile("" : "+v"(prod));  // keep the product a rounded f32: hipcc would contract a * b + c into one FMA
    out[c * S + o] = out[c * S + o] + prod;
  }
  wmap[o] = wmap[o] + w;
}

__global__ __launch_bounds__(256) void k_stitch_finalize(float* __restrict__ out, const float* __restrict__ wmap, int C, size_t S) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= S) return;
  const float w = wmap[i];
  for (int c = 0; c < C; ++c) out[c * S + i] = __fdiv_rn(out[c * S + i], w);
}

struct ScaleMulti { void* buf[8]; unsigned long long n[8]; };
// up to 8 buffers in one launch (blockIdx.y = buffer): the gradients of one loss section share their grad_output
template <typename T>
__global__ __launch_bounds__(256) void k_scale_multi(const ScaleMulti M, const float* __restrict__ scale) {
  const float sc = scale[0];
  if (sc == 1.0f) return;  // the loss.backward() case: a few hundred workgroups that read one float
  T* buf = (T*)M.buf[blockIdx.y];
  const size_t n = (size_t)M.n[blockIdx.y];
  const size_t stride = (size_t)gridDim.x * 256, t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (sizeof(T) == 4 && (((uintptr_t)buf) & 15) == 0) {
    const size_t n4 = n / 4;
    for (size_t i = t; i < n4; i += stride) {
      f4 v = ((f4*)buf)[i];
      v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
      ((f4*)buf)[i] = v;
    }
    for (size_t i = 4 * n4 + t; i < n; i += stride) st(buf, i, ld(buf, i) * sc);
  } else {
    for (size_t i = t; i < n; i += stride) st(buf, i, ld(buf, i) * sc);
  }
}

// One workgroup: a second reduction launch costs more than it saves (measured: 16-workgroup slice sums + a final kernel
// took 8 + 5 us against 10 us for this one; any launch is >= 4.5 us here and one CU pulls ~20 GB/s)
void launch_loss_finalize(const KParams& P, float* partials, int nparts, float* loss_out, hipStream_t s) {
  if ((size_t)nparts * P.K <= 65536) {  // one workgroup: 8-10 us at the CVPPP batch (4624 tiles x 10), less than two launches
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(1024), 0, s, P, (const float*)partials, nparts, loss_out);
    return;
  }
  double* slices = (double*)((char*)partials + ws_bytes(fwd_partials(P), P.K) - kSliceBytes);
  hipLaunchKernelGGL(k_loss_slices, dim3(kFinSlices, P.K), dim3(256), 0, s, (const float*)partials, nparts, slices);
  hipLaunchKernelGGL(k_loss_finalize2, dim3(1), dim3(1024), 0, s, P, (const double*)slices, loss_out);
}

}  // namespace

extern "C" {

int pea_version(void) { return PEA_ABI_VERSION; }

const char* pea_strerror(int code) {
  switch (code) {
    case PEA_OK: return "ok";
    case PEA_E_NULL: return "required pointer is NULL";
    case PEA_E_DESC: return "descriptor field out of range";
    case PEA_E_UNSUPPORTED: return "unsupported combination";
    case PEA_E_WORKSPACE: return "workspace missing or too small";
    case PEA_E_ALIGN: return "pointer not aligned to its element size";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown pea error";
  }
}

int pea_desc_validate(const PeaDesc* desc) { return validate(desc); }

size_t pea_workspace_bytes(const PeaDesc* desc) {
  if (validate(desc)) return 0;
  const KParams P = make_params(desc);
  return ws_bytes(fwd_partials(P), P.K);
}

int pea_affinity_infer(const PeaDesc* desc, const void* e, const void* e_other, float* affs, void* stream) {
  const int rc = validate(desc);
  if (rc) return rc;
  if (!e || !affs) return PEA_E_NULL;
  const size_t es = desc->dtype == PEA_F16 ? 2 : 4;
  if (misaligned(e, es) || misaligned(e_other, es) || misaligned(affs, 4)) return PEA_E_ALIGN;
  const KParams P = make_params(desc);
  hipStream_t s = (hipStream_t)stream;
  int nparts = 0;
  return desc->dtype == PEA_F16
             ? launch_fwd<__half, false>(P, e, e_other, nullptr, nullptr, nullptr, affs, nullptr, nullptr, nullptr, s, &nparts)
             : launch_fwd<float, false>(P, e, e_other, nullptr, nullptr, nullptr, affs, nullptr, nullptr, nullptr, s, &nparts);
}

int pea_affinity_fwd_ex(const PeaDesc* desc, const void* e, const void* e_other, const float* target,
                        const float* weight, const uint8_t* mask, float* affs, float* g_out, float* inv_norm_out,
                        float* loss_out, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = validate(desc);
  if (rc) return rc;
  if (!e || !target || !weight || !loss_out) return PEA_E_NULL;
  const size_t es = desc->dtype == PEA_F16 ? 2 : 4;
  if (misaligned(e, es) || misaligned(e_other, es) || misaligned(affs, 4) || misaligned(g_out, 4) ||
      misaligned(target, 4) || misaligned(weight, 4) || misaligned(loss_out, 4) || misaligned(workspace, 8) ||
      misaligned(inv_norm_out, 4))
    return PEA_E_ALIGN;
  const KParams P = make_params(desc);
  if (!workspace || workspace_bytes < ws_bytes(fwd_partials(P), P.K)) return PEA_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  float* partials = (float*)workspace;
  int nparts = 0;
  const bool self = !e_other || e_other == e;
  if (!self && inv_norm_out && desc->dtype == PEA_F32 && env_int("PEA_FORCE_DIRECT", 0) == 0 &&
      try_fwd_xdma_other(P, (const float*)e, (const float*)e_other, target, weight, mask, affs, g_out, partials, inv_norm_out, s,
                         &nparts)) {
    launch_loss_finalize(P, partials, nparts, loss_out, s);
    return hip_rc();
  }
  rc = desc->dtype == PEA_F16
           ? launch_fwd<__half, true>(P, e, e_other, target, weight, mask, affs, g_out, partials, self ? inv_norm_out : nullptr, s, &nparts)
           : launch_fwd<float, true>(P, e, e_other, target, weight, mask, affs, g_out, partials, self ? inv_norm_out : nullptr, s, &nparts);
  if (rc) return rc;
  if (inv_norm_out && !self) {  // second operand, not on the cross kernels: the two planes take their own launches
    float* inv_o = inv_norm_out + (size_t)P.B * P.S;
    if (desc->dtype == PEA_F16) {
      launch_inv_norm<__half>(P, (const __half*)e, inv_norm_out, s);
      launch_inv_norm<__half>(P, (const __half*)e_other, inv_o, s);
    } else {
      launch_inv_norm<float>(P, (const float*)e, inv_norm_out, s);
      launch_inv_norm<float>(P, (const float*)e_other, inv_o, s);
    }
  }
  launch_loss_finalize(P, partials, nparts, loss_out, s);
  return hip_rc();
}

int pea_affinity_fwd(const PeaDesc* desc, const void* e, const void* e_other, const float* target,
                     const float* weight, const uint8_t* mask, float* affs, float* g_out, float* loss_out,
                     void* workspace, size_t workspace_bytes, void* stream) {
  return pea_affinity_fwd_ex(desc, e, e_other, target, weight, mask, affs, g_out, nullptr, loss_out, workspace, workspace_bytes, stream);
}

int pea_cross_supported(const PeaDesc* desc, int backward) {
  if (validate(desc)) return 0;
  const KParams P = make_params(desc);
  if (desc->dtype != PEA_F32 || (P.D != 16 && P.D != 32 && P.D != 64) || env_int("PEA_FORCE_DIRECT", 0) != 0) return 0;
  if (env_int(backward ? "PEA_BWD_XDMA" : "PEA_FWD_XDMA", 1) == 0) return 0;
  XParams C;
  size_t lds;
  if (backward == 2) {  // the cross loss with a detached second operand: forward and role-A backward
    if (P.D != 16 || P.border != PEA_BORDER_CIRCULAR || P.K > kXP || env_int("PEA_FWD_XDMA", 1) == 0) return 0;
    if (!plan_xdma(P, kXdmaTH, kXdmaTW, kXdmaPSU, &C, &lds, 1) || C.nfz > 0) return 0;
    return plan_xdma(P, kXdmaTH, kXdmaTW, kXdmaPSU, &C, &lds, 2) ? 1 : 0;
  }
  if (!plan_xdma(P, kXdmaTH, kXdmaTW, kXdmaPSU3, &C, &lds, backward == 0)) return 0;
  const bool z3 = C.npz > 0 || C.nfz > 0;
  if (z3) return (P.D == 16 && C.npx <= 8 && C.npy <= 8 && P.K <= kXP + 2) ? 1 : 0;
  if (!plan_xdma(P, kXdmaTH, kXdmaTW, kXdmaPSU, &C, &lds, backward == 0)) return 0;
  if (!backward && P.K > kXP) return 0;
  return (backward && P.D > 32 && (C.npx > 8 || C.npy > 8)) ? 0 : 1;
}

int pea_inv_norm(const PeaDesc* desc, const void* e, float* inv_norm_out, void* stream) {
  const int rc = validate(desc);
  if (rc) return rc;
  if (!e || !inv_norm_out) return PEA_E_NULL;
  if (misaligned(e, desc->dtype == PEA_F16 ? 2 : 4) || misaligned(inv_norm_out, 4)) return PEA_E_ALIGN;
  const KParams P = make_params(desc);
  if (desc->dtype == PEA_F16) launch_inv_norm<__half>(P, (const __half*)e, inv_norm_out, (hipStream_t)stream);
  else launch_inv_norm<float>(P, (const float*)e, inv_norm_out, (hipStream_t)stream);
  return hip_rc();
}

int pea_affinity_bwd_ex(const PeaDesc* desc, const void* e, const void* e_other, const float* g, const float* inv_norm,
                        const float* dloss, void* de, void* de_other, void* stream) {
  int rc = validate(desc);
  if (rc) return rc;
  if (!e || !g || (!de && !de_other)) return PEA_E_NULL;
  if (de_other && !e_other) return PEA_E_NULL;
  const size_t es = desc->dtype == PEA_F16 ? 2 : 4;
  if (misaligned(e, es) || misaligned(e_other, es) || misaligned(de, es) || misaligned(de_other, es) ||
      misaligned(g, 4) || misaligned(dloss, 4) || misaligned(inv_norm, 4))
    return PEA_E_ALIGN;
  const KParams P = make_params(desc);
  hipStream_t s = (hipStream_t)stream;
  const bool h = desc->dtype == PEA_F16;
  if (!e_other) {
    // self loss: the LDS-DMA cross kernel when the 1 / norm plane came along and the stencil is axis-aligned
    if (!h && env_int("PEA_FORCE_DIRECT", 0) == 0) {
      bool done = false;
      if (P.D == 16) done = try_bwd_xdma<16>(P, (const float*)e, inv_norm, g, dloss, (float*)de, s);
      else if (P.D == 32) done = try_bwd_xdma<32>(P, (const float*)e, inv_norm, g, dloss, (float*)de, s);
      else if (P.D == 64) done = try_bwd_xdma<64>(P, (const float*)e, inv_norm, g, dloss, (float*)de, s);
      if (done) return hip_rc();
    }
    return h ? launch_bwd<__half>(P, 3, e, e, e, g, dloss, de, s) : launch_bwd<float>(P, 3, e, e, e, g, dloss, de, s);
  }
  const bool accumulate = (desc->flags & PEA_FLAG_ACCUMULATE_DE) != 0;
  if (de && !de_other && !h && env_int("PEA_FORCE_DIRECT", 0) == 0 &&
      try_bwd_xdma_other(P, (const float*)e, (const float*)e_other, inv_norm, g, dloss, (float*)de, accumulate, s))
    return hip_rc();  // detached second operand: the role-A cross kernel (inv_norm = the two planes pea_affinity_fwd_ex wrote)
  if (accumulate) return PEA_E_UNSUPPORTED;
  if (de) {
    rc = h ? launch_bwd<__half>(P, 1, e, e_other, nullptr, g, dloss, de, s)
           : launch_bwd<float>(P, 1, e, e_other, nullptr, g, dloss, de, s);
    if (rc) return rc;
  }
  if (!de_other) return PEA_OK;
  return h ? launch_bwd<__half>(P, 2, e_other, nullptr, e, g, dloss, de_other, s)
           : launch_bwd<float>(P, 2, e_other, nullptr, e, g, dloss, de_other, s);
}

int pea_affinity_bwd(const PeaDesc* desc, const void* e, const void* e_other, const float* g, const float* dloss,
                     void* de, void* de_other, void* stream) {
  return pea_affinity_bwd_ex(desc, e, e_other, g, nullptr, dloss, de, de_other, stream);
}

int pea_scale_inplace(void* buf, int dtype, size_t n, const float* scale, void* stream) {
  if (!buf || !scale) return PEA_E_NULL;
  if (dtype != PEA_F32 && dtype != PEA_F16) return PEA_E_DESC;
  if (misaligned(buf, dtype == PEA_F32 ? 16 : 2) || misaligned(scale, 4)) return PEA_E_ALIGN;
  if (n == 0) return PEA_OK;
  hipStream_t s = (hipStream_t)stream;
  const size_t n4 = dtype == PEA_F32 ? n / 4 : 0;
  const size_t items = dtype == PEA_F32 ? std::max(n4, n - 4 * n4) : (n + 3) / 4;
  const size_t blocks = (items + 255) / 256;
  if (blocks > 0x7fffffffULL) return PEA_E_UNSUPPORTED;
  if (dtype == PEA_F32) hipLaunchKernelGGL(k_scale_inplace<float>, dim3((unsigned)blocks), dim3(256), 0, s, (float*)buf, n4, n, scale);
  else hipLaunchKernelGGL(k_scale_inplace<__half>, dim3((unsigned)blocks), dim3(256), 0, s, (__half*)buf, n4, n, scale);
  return hip_rc();
}

int pea_fill_border_relu(float* affs, int B, int K, int Z, int Y, int X, int shift, int relu, void* stream) {
  if (!affs) return PEA_E_NULL;
  if (B < 1 || K < 1 || Z < 1 || Y < 1 || X < 1 || shift < 0) return PEA_E_DESC;
  if (shift > 0 && K >= 3 && (2 * shift > Z || 2 * shift > Y || 2 * shift > X)) return PEA_E_DESC;
  if (misaligned(affs, 4)) return PEA_E_ALIGN;
  const size_t n = (size_t)B * K * Z * Y * X, blocks = (n + 255) / 256;
  if (blocks > 0x7fffffffULL) return PEA_E_UNSUPPORTED;
  if (shift == 0 && !misaligned(affs, 16)) {
    if (relu) {
      const size_t n4 = n / 4, b4 = (std::max(n4, n - 4 * n4) + 255) / 256;
      hipLaunchKernelGGL(k_relu_inplace, dim3((unsigned)b4), dim3(256), 0, (hipStream_t)stream, affs, n4, n);
    }
    return hip_rc();
  }
  // source slices [shift, 2*shift) are never themselves rewritten (relu is idempotent), so in place is race-free
  hipLaunchKernelGGL(k_fill_border_relu, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, affs, B, K, Z, Y, X, shift, relu);
  return hip_rc();
}

size_t pea_targets_workspace_bytes(const PeaDesc* desc) {
  if (validate(desc)) return 0;
  // pea_gen_targets: one count per (image, channel); pea_label_weights: one partial per (image, channel, workgroup)
  const size_t wgs = (size_t)((desc->dims[2] + 63) / 64) * ((desc->dims[1] + 4 * kCntRows - 1) / (4 * kCntRows)) * desc->dims[0];
  return (size_t)desc->B * desc->K * sizeof(unsigned) * std::max<size_t>(1, wgs);
}

int pea_gen_targets(const PeaDesc* desc, const int32_t* labels, unsigned flags, float* target, uint8_t* mask, float* weight,
                    void* workspace, size_t workspace_bytes, void* stream) {
  const int rc = validate(desc);
  if (rc) return rc;
  if (!labels || !target) return PEA_E_NULL;
  if (misaligned(labels, 4) || misaligned(target, 4) || misaligned(weight, 4) || misaligned(workspace, 4)) return PEA_E_ALIGN;
  if (flags & ~(PEA_TGT_PADDING | PEA_TGT_BOTH_FOREGROUND | PEA_TGT_MASK_INSIDE)) return PEA_E_DESC;
  const size_t need = (size_t)desc->B * desc->K * sizeof(unsigned);
  if (!workspace || workspace_bytes < need) return PEA_E_WORKSPACE;
  GParams G;
  G.B = desc->B; G.Z = desc->dims[0]; G.Y = desc->dims[1]; G.X = desc->dims[2]; G.K = desc->K;
  G.S = G.Z * G.Y * G.X;
  G.flags = flags;
  for (int i = 0; i < PEA_MAX_K; ++i)
    for (int a = 0; a < 3; ++a) G.off[i][a] = i < desc->K ? desc->offsets[i][a] : 0;
  if (desc->B > 65535 || (long long)desc->B * desc->K > 65535) return PEA_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(workspace, 0, need, s) != hipSuccess) return hip_rc();
  const unsigned chunks = (unsigned)((G.S + 255) / 256);
  const unsigned gx = (chunks + kTgtNit - 1) / kTgtNit;  // kTgtNit * 256 pixels per workgroup
  hipLaunchKernelGGL(k_gen_targets, dim3(gx, (unsigned)G.B), dim3(256), 0, s, G, labels, target, mask, (unsigned*)workspace);
  if (weight)
    hipLaunchKernelGGL(k_gen_weights, dim3(chunks, (unsigned)(G.B * G.K)), dim3(256), 0, s, G, target, (const unsigned*)workspace, weight);
  return hip_rc();
}

int pea_stitch_add(float* out_affs, float* weight_map, const float* affs_vol, const float* weight_vol, int C, int Z, int Y,
                   int X, int oz, int oy, int ox, int z0, int y0, int x0, void* stream) {
  if (!out_affs || !weight_map || !affs_vol || !weight_vol) return PEA_E_NULL;
  if (C < 1 || oz < 1 || oy < 1 || ox < 1 || z0 < 0 || y0 < 0 || x0 < 0 || z0 + oz > Z || y0 + oy > Y || x0 + ox > X) return PEA_E_DESC;
  if (misaligned(out_affs, 4) || misaligned(weight_map, 4) || misaligned(affs_vol, 4) || misaligned(weight_vol, 4)) return PEA_E_ALIGN;
  const size_t n = (size_t)oz * oy * ox, blocks = (n + 255) / 256;
  if (blocks > 0x7fffffffULL) return PEA_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_stitch_add, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out_affs, weight_map, affs_vol,
                     weight_vol, C, Z, Y, X, oz, oy, ox, z0, y0, x0);
  return hip_rc();
}

int pea_stitch_finalize(float* out_affs, const float* weight_map, int C, size_t voxels, void* stream) {
  if (!out_affs || !weight_map) return PEA_E_NULL;
  if (C < 1) return PEA_E_DESC;
  if (misaligned(out_affs, 4) || misaligned(weight_map, 4)) return PEA_E_ALIGN;
  const size_t blocks = (voxels + 255) / 256;
  if (blocks > 0x7fffffffULL) return PEA_E_UNSUPPORTED;
  if (voxels) hipLaunchKernelGGL(k_stitch_finalize, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out_affs, weight_map, C, voxels);
  return hip_rc();
}

int pea_label_weights(const PeaDesc* desc, const int32_t* labels, unsigned flags, float* wtab, void* workspace,
                      size_t workspace_bytes, void* stream) {
  const int rc = validate(desc);
  if (rc) return rc;
  if (!labels || !wtab) return PEA_E_NULL;
  if (misaligned(labels, 4) || misaligned(wtab, 4) || misaligned(workspace, 4)) return PEA_E_ALIGN;
  if (flags & ~(PEA_TGT_PADDING | PEA_TGT_BOTH_FOREGROUND | PEA_TGT_MASK_INSIDE)) return PEA_E_DESC;
  const size_t need = pea_targets_workspace_bytes(desc);
  if (!workspace || workspace_bytes < need) return PEA_E_WORKSPACE;
  GParams G;
  G.B = desc->B; G.Z = desc->dims[0]; G.Y = desc->dims[1]; G.X = desc->dims[2]; G.K = desc->K;
  G.S = G.Z * G.Y * G.X;
  G.flags = flags;
  for (int i = 0; i < PEA_MAX_K; ++i)
    for (int a = 0; a < 3; ++a) G.off[i][a] = i < desc->K ? desc->offsets[i][a] : 0;
  hipStream_t s = (hipStream_t)stream;
  if ((long long)G.B * G.Z > 65535 || (G.Y + 4 * kCntRows - 1) / (4 * kCntRows) > 65535) return PEA_E_UNSUPPORTED;
  const dim3 cgrid((unsigned)((G.X + 63) / 64), (unsigned)((G.Y + 4 * kCntRows - 1) / (4 * kCntRows)), (unsigned)(G.B * G.Z));
  hipLaunchKernelGGL(k_label_counts, cgrid, dim3(256), 0, s, G, labels, (unsigned*)workspace);
  const int per_img = (int)(cgrid.x * cgrid.y) * G.Z;
  const int n = G.B * G.K;
  hipLaunchKernelGGL(k_weight_table, dim3((unsigned)n), dim3(64), 0, s, G.S, per_img, (const unsigned*)workspace, wtab);
  return hip_rc();
}

int pea_affinity_fwd_bwd_labels(const PeaDesc* desc, const void* e, const void* e_other, const int32_t* labels,
                                const float* wtab, unsigned flags, float* affs, float* loss_out, const float* dloss, void* de,
                                void* workspace, size_t workspace_bytes, void* stream) {
  int rc = validate(desc);
  if (rc) return rc;
  if (!e || !labels || !wtab || !loss_out || !de) return PEA_E_NULL;
  const size_t es = desc->dtype == PEA_F16 ? 2 : 4;
  if (misaligned(e, es) || misaligned(e_other, es) || misaligned(de, es) || misaligned(affs, 4) || misaligned(labels, 4) ||
      misaligned(wtab, 4) || misaligned(loss_out, 4) || misaligned(dloss, 4) || misaligned(workspace, 8))
    return PEA_E_ALIGN;
  if (flags & ~(PEA_TGT_PADDING | PEA_TGT_BOTH_FOREGROUND | PEA_TGT_MASK_INSIDE | PEA_TGT_ACCUMULATE)) return PEA_E_DESC;
  const KParams P = make_params(desc);
  if (!workspace || workspace_bytes < ws_bytes(fwd_partials(P), P.K)) return PEA_E_WORKSPACE;
  if ((P.D != 16 && P.D != 32) || env_int("PEA_FORCE_DIRECT", 0) != 0) return PEA_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  float* partials = (float*)workspace;
  int nparts = 0;
  bool done;
#define PEA_LAB_DISPATCH(TT, DD)                                                                                         \
  {                                                                                                                      \
    const TT *x = (const TT*)e, *nb = (const TT*)e_other;                                                                \
    done = nb ? try_fused_labels<TT, DD, false>(P, x, nb, labels, wtab, flags, affs, partials, dloss, (TT*)de, s, &nparts) \
              : try_fused_labels<TT, DD, true>(P, x, x, labels, wtab, flags, affs, partials, dloss, (TT*)de, s, &nparts);  \
  }
  if (desc->dtype == PEA_F16) {
    if (P.D == 16) PEA_LAB_DISPATCH(__half, 16) else PEA_LAB_DISPATCH(__half, 32)
  } else {
    if (P.D == 16) PEA_LAB_DISPATCH(float, 16) else PEA_LAB_DISPATCH(float, 32)
  }
#undef PEA_LAB_DISPATCH
  if (!done) return PEA_E_UNSUPPORTED;
  rc = hip_rc();
  if (rc) return rc;
  launch_loss_finalize(P, partials, nparts, loss_out, s);
  return hip_rc();
}

int pea_affinity_fwd_bwd_labels_dual(const PeaDesc* desc, const PeaDesc* desc_cross, const void* e, const void* ema,
                                     const int32_t* labels, const float* wtab, unsigned flags, float* affs, float* loss_out,
                                     float* loss_cross_out, const float* dloss, const float* dloss_cross, void* de,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  int rc = validate(desc);
  if (rc) return rc;
  rc = validate(desc_cross);
  if (rc) return rc;
  if (!e || !ema || !labels || !wtab || !loss_out || !loss_cross_out || !de) return PEA_E_NULL;
  if (desc->B != desc_cross->B || desc->D != desc_cross->D || desc->K != desc_cross->K || desc->dtype != desc_cross->dtype ||
      desc->border != desc_cross->border || memcmp(desc->dims, desc_cross->dims, sizeof(desc->dims)) != 0 ||
      memcmp(desc->offsets, desc_cross->offsets, sizeof(desc->offsets)) != 0 || desc->flags != desc_cross->flags ||
      desc->eps != desc_cross->eps)
    return PEA_E_DESC;  // the two losses may differ in lambda and in the normaliser only
  const size_t es = desc->dtype == PEA_F16 ? 2 : 4;
  if (misaligned(e, es) || misaligned(ema, es) || misaligned(de, es) || misaligned(affs, 4) || misaligned(labels, 4) ||
      misaligned(wtab, 4) || misaligned(loss_out, 4) || misaligned(loss_cross_out, 4) || misaligned(dloss, 4) ||
      misaligned(dloss_cross, 4) || misaligned(workspace, 8))
    return PEA_E_ALIGN;
  if (flags & ~(PEA_TGT_PADDING | PEA_TGT_BOTH_FOREGROUND | PEA_TGT_MASK_INSIDE)) return PEA_E_DESC;
  const KParams P = make_params(desc), P2 = make_params(desc_cross);
  const size_t wsb = ws_bytes(fwd_partials(P), P.K);  // each loss has its own [partials | slice sums] block
  if (!workspace || workspace_bytes < 2 * wsb) return PEA_E_WORKSPACE;
  if ((P.D != 16 && P.D != 32) || env_int("PEA_FORCE_DIRECT", 0) != 0 || env_int("PEA_LABELS_DUAL", 1) == 0) return PEA_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  float *partials = (float*)workspace, *partials2 = (float*)((char*)workspace + wsb);
  int nparts = 0;
  bool done;
#define PEA_LD(T_, D_) try_fused_labels_dual<T_, D_>(P, P2, (const T_*)e, (const T_*)ema, labels, wtab, flags, affs, partials, partials2, \
                                                     dloss, dloss_cross, (T_*)de, s, &nparts)
  if (P.D == 16) done = desc->dtype == PEA_F16 ? PEA_LD(__half, 16) : PEA_LD(float, 16);
  else done = desc->dtype == PEA_F16 ? PEA_LD(__half, 32) : PEA_LD(float, 32);
#undef PEA_LD
  if (!done) return PEA_E_UNSUPPORTED;
  rc = hip_rc();
  if (rc) return rc;
  launch_loss_finalize(P, partials, nparts, loss_out, s);
  launch_loss_finalize(P2, partials2, nparts, loss_cross_out, s);
  return hip_rc();
}

int pea_scale_inplace_multi(void* const* bufs, const size_t* counts, int nbuf, int dtype, const float* scale, void* stream) {
  if (!bufs || !counts || !scale) return PEA_E_NULL;
  if (nbuf < 1 || nbuf > 8 || (dtype != PEA_F32 && dtype != PEA_F16)) return PEA_E_DESC;
  ScaleMulti M = {};
  size_t nmax = 0;
  for (int i = 0; i < nbuf; ++i) {
    if (!bufs[i]) return PEA_E_NULL;
    if (misaligned(bufs[i], dtype == PEA_F32 ? 4 : 2)) return PEA_E_ALIGN;
    M.buf[i] = bufs[i];
    M.n[i] = counts[i];
    nmax = std::max(nmax, counts[i]);
  }
  if (misaligned(scale, 4)) return PEA_E_ALIGN;
  if (nmax == 0) return PEA_OK;
  const unsigned gx = (unsigned)std::min<size_t>((nmax / 4 + 255) / 256 + 1, 512);  // grid-stride: a fixed, small grid
  if (dtype == PEA_F32) hipLaunchKernelGGL(k_scale_multi<float>, dim3(gx, (unsigned)nbuf), dim3(256), 0, (hipStream_t)stream, M, scale);
  else hipLaunchKernelGGL(k_scale_multi<__half>, dim3(gx, (unsigned)nbuf), dim3(256), 0, (hipStream_t)stream, M, scale);
  return hip_rc();
}

int pea_affinity_bwd_dual(const PeaDesc* desc, const void* e, const void* ema, const float* g, const float* g_cross,
                          const float* dloss, const float* dloss_cross, void* de, void* stream) {
  return pea_affinity_bwd_dual_ex(desc, e, ema, g, g_cross, nullptr, nullptr, dloss, dloss_cross, de, stream);
}

int pea_affinity_bwd_dual_ex(const PeaDesc* desc, const void* e, const void* ema, const float* g, const float* g_cross,
                             const float* inv_norm, const float* inv_norm_other, const float* dloss, const float* dloss_cross,
                             void* de, void* stream) {
  const int rc = validate(desc);
  if (rc) return rc;
  if (!e || !ema || !g || !g_cross || !de) return PEA_E_NULL;
  const size_t es = desc->dtype == PEA_F16 ? 2 : 4;
  if (misaligned(e, es) || misaligned(ema, es) || misaligned(de, es) || misaligned(g, 4) || misaligned(g_cross, 4) ||
      misaligned(dloss, 4) || misaligned(dloss_cross, 4) || misaligned(inv_norm, 4) || misaligned(inv_norm_other, 4))
    return PEA_E_ALIGN;
  const KParams P = make_params(desc);
  if ((P.D != 16 && P.D != 32) || env_int("PEA_FORCE_DIRECT", 0) != 0 || env_int("PEA_BWD_DUAL", 1) == 0) return PEA_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (desc->dtype == PEA_F32 && try_bwd_xdma_dual(P, (const float*)e, (const float*)ema, inv_norm, inv_norm_other, g, g_cross, dloss,
                                                  dloss_cross, (float*)de, s))
    return hip_rc();
  bool done;
#define PEA_BD(T_, D_) try_bwd_dual<T_, D_>(P, (const T_*)e, (const T_*)ema, g, g_cross, dloss, dloss_cross, (T_*)de, s)
  if (P.D == 16) done = desc->dtype == PEA_F16 ? PEA_BD(__half, 16) : PEA_BD(float, 16);
  else done = desc->dtype == PEA_F16 ? PEA_BD(__half, 32) : PEA_BD(float, 32);
#undef PEA_BD
  if (!done) return PEA_E_UNSUPPORTED;
  return hip_rc();
}

// ---- the embedding head (1x1 convolution, pea_head.h) ---------------------------------------------------------------
// (C, D) pairs of the reference's heads: ResUNet 32 / 64 / 128 / 256 -> 16 (CVPPP) or 32 (BBBC039V1),
// superhuman 3D U-Net 28 / 36 / 48 / 64 / 80 -> 16
#define PEA_HEAD_CASES(X) \
  X(28, 16) X(32, 16) X(36, 16) X(48, 16) X(64, 16) X(80, 16) X(128, 16) X(256, 16) X(32, 32) X(64, 32) X(128, 32) X(256, 32)

static bool head_supported(int C, int D) {
#define PEA_HEAD_Q(c, d) if (C == c && D == d) return true;
  PEA_HEAD_CASES(PEA_HEAD_Q)
#undef PEA_HEAD_Q
  return false;
}

size_t pea_head_workspace_bytes(int C, int D) {
  if (C < 1 || D < 1) return 0;
  return (size_t)kHeadMaxWg * ((size_t)D * C + D) * sizeof(float);
}

int pea_head_fwd(int B, int C, int D, size_t S, const float* x, const float* W, const float* bias, float* e, void* stream) {
  if (B < 1 || C < 1 || D < 1 || S < 1) return PEA_E_DESC;
  if (!x || !W || !e) return PEA_E_NULL;
  if (misaligned(x, 4) || misaligned(W, 4) || misaligned(bias, 4) || misaligned(e, 4)) return PEA_E_ALIGN;
  if (!head_supported(C, D)) return PEA_E_UNSUPPORTED;
  const size_t chunks = (S + kHeadBlock - 1) / kHeadBlock;
  if (chunks * (size_t)B > 0x7fffffffULL) return PEA_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)(chunks * B)), blk(kHeadBlock);
#define PEA_HEAD_F(c, d) \
  if (C == c && D == d) hipLaunchKernelGGL((k_head_fwd<c, d>), grid, blk, 0, s, x, W, bias, e, (long long)S, (int)chunks);
  PEA_HEAD_CASES(PEA_HEAD_F)
#undef PEA_HEAD_F
  return hip_rc();
}

int pea_head_bwd(int B, int C, int D, size_t S, const float* x, const float* W, const float* de, float* dx, float* dW, float* db,
                 void* workspace, size_t workspace_bytes, void* stream) {
  if (B < 1 || C < 1 || D < 1 || S < 1) return PEA_E_DESC;
  if (!x || !W || !de || !dW) return PEA_E_NULL;
  if (misaligned(x, 4) || misaligned(W, 4) || misaligned(de, 4) || misaligned(dx, 4) || misaligned(dW, 4) || misaligned(db, 4) ||
      misaligned(workspace, 4))
    return PEA_E_ALIGN;
  if (!head_supported(C, D)) return PEA_E_UNSUPPORTED;
  if (!workspace || workspace_bytes < pea_head_workspace_bytes(C, D)) return PEA_E_WORKSPACE;
  const size_t chunks = (S + kHeadBlock - 1) / kHeadBlock;
  if (chunks * (size_t)B > 0x7fffffffULL) return PEA_E_UNSUPPORTED;
  const int nchunks = (int)(chunks * B);
  // dW: a multiple of the CU count that the instantiation keeps resident, no partial round
  const int nwg = std::min(nchunks, std::min(kHeadMaxWg, ((C <= 48 && D == 16) ? 4 : 2) * device_cus()));
  hipStream_t s = (hipStream_t)stream;
  float* partials = (float*)workspace;
  if (dx) {
    const dim3 grid((unsigned)nchunks), blk(kHeadBlock);
#define PEA_HEAD_X(c, d) \
  if (C == c && D == d) hipLaunchKernelGGL((k_head_dx<c, d>), grid, blk, 0, s, W, de, dx, (long long)S, (int)chunks);
    PEA_HEAD_CASES(PEA_HEAD_X)
#undef PEA_HEAD_X
  }
#define PEA_HEAD_B(c, d)                                                                                              \
  if (C == c && D == d)                                                                                               \
    hipLaunchKernelGGL((k_head_dw<c, d>), dim3((unsigned)nwg), dim3(kHeadBlock), 0, s, x, de, partials, (long long)S, \
                       (int)chunks, nchunks);
  PEA_HEAD_CASES(PEA_HEAD_B)
#undef PEA_HEAD_B
  const int rc = hip_rc();
  if (rc) return rc;
  const int n = D * C + D;
  hipLaunchKernelGGL(k_head_finalize, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, partials, nwg, D * C, n, dW, db);
  return hip_rc();
}

// f1: the self-loss backward with the embedding head's backward in its epilogue
size_t pea_bwd_head_workspace_bytes(const PeaDesc* desc, int C) {
  if (validate(desc) || C != kHeadFuseC) return 0;
  const KParams P = make_params(desc);
  XParams X;
  size_t lds;
  if (desc->dtype != PEA_F32 || !plan_bwd_head(P, &X, &lds)) return 0;
  return (size_t)X.ntiles * ((size_t)P.D * C + P.D) * sizeof(float);
}

int pea_affinity_bwd_head(const PeaDesc* desc, const void* e, const float* g, const float* inv_norm, const float* dloss,
                          const float* de_add, const float* x, const float* W, int C, float* dx, float* dW, float* db, void* de,
                          void* workspace, size_t workspace_bytes, void* stream) {
  const int v = validate(desc);
  if (v) return v;
  if (!e || !g || !inv_norm || !x || !W || !dW) return PEA_E_NULL;
  if (misaligned(e, 16) || misaligned(inv_norm, 16) || misaligned(g, 4) || misaligned(dloss, 4) || misaligned(de_add, 4) ||
      misaligned(x, 4) || misaligned(W, 4) || misaligned(dx, 4) || misaligned(dW, 4) || misaligned(db, 4) || misaligned(de, 4) ||
      misaligned(workspace, 4))
    return PEA_E_ALIGN;
  if (desc->dtype != PEA_F32 || C != kHeadFuseC || env_int("PEA_FORCE_DIRECT", 0) != 0 || env_int("PEA_BWD_XDMA", 1) == 0)
    return PEA_E_UNSUPPORTED;
  const size_t need = pea_bwd_head_workspace_bytes(desc, C);
  if (need == 0) return PEA_E_UNSUPPORTED;
  if (!workspace || workspace_bytes < need) return PEA_E_WORKSPACE;
  const KParams P = make_params(desc);
  if ((size_t)C * P.S * 4 >= (1ull << 31)) return PEA_E_UNSUPPORTED;  // the head tensors are addressed like e (2 GiB rule)
  hipStream_t s = (hipStream_t)stream;
  HeadArgs H;
  H.x = x; H.W = W; H.de_add = de_add; H.dx = dx; H.partials = (float*)workspace;
  int ntiles = 0;
  if (!try_bwd_xdma_head(P, (const float*)e, inv_norm, g, dloss, (float*)de, H, s, &ntiles)) return PEA_E_UNSUPPORTED;
  const int rc = hip_rc();
  if (rc) return rc;
  const int n = P.D * C + P.D;
  hipLaunchKernelGGL(k_head_finalize, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, (const float*)workspace, ntiles, P.D * C, n, dW, db);
  return hip_rc();
}

}  // extern "C"
