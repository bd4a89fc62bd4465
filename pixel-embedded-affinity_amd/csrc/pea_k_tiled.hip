// pea_k_tiled.hip -- launchers of the LDS-tiled box kernels (pea_tiled.h) and of the channel-chunked forward (pea_chunked.h):
// what runs where the cross kernels do not apply (diagonal / mixed-sign stencils, f16 storage, a second operand with both
// gradients, inference).  One translation unit of libpea_hip.so (pea_host.h).
#include <type_traits>

#include "pea_plan.h"
#include "pea_chunked.h"

namespace pea {

namespace {

#define PEA_LAUNCH(kern, grid, blk, lds, s, ...)              \
  {                                                           \
    if (allow_lds<kern>(lds)) return false;                   \
    hipLaunchKernelGGL(kern, grid, blk, lds, s, __VA_ARGS__); \
  }

// REPLICATE tables are generic (embedding_loss_norm6: diagonals, z steps, reach 27): where the forward's one-sided region serves fewer
// than half of the offsets from LDS the global-memory forward is faster (shift_func(17) at 24 x 1024^2: 2 near of 17, 4.06 against
// 3.80 ms); the backward's larger two-sided region (8 near of 17) wins (4.73 against 6.20 ms) -- profiles/r5_c4r6_*.json
inline bool rep_mostly_far(const KParams& P, const TParams& Q) { return P.border == PEA_BORDER_REPLICATE && 2 * Q.n_near < P.K; }

struct FwdT {  // typed view of FwdArgs
  const void *e, *eo;
  const float *t, *w;
  const uint8_t* m;
  float *affs, *gout;
  LossState* st;
  float* inv_out;
};

// forward with the LDS-transposed, dwordx4 epilogue (k_fwd_tiled_v): 16x32 tiles, the dot products laid over the dead region,
// two workgroups of 8 waves per CU.  The training forward wherever the LDS-DMA kernel (k_fwd_xdma) does not apply.
template <typename T, int D_T, bool TRAIN>
bool try_fwd_v(const KParams& P, const FwdT& A, float* inv_out, hipStream_t s) {
  if (P.K > kKV || P.X % 4) return false;
  if (P.border == PEA_BORDER_REPLICATE && !(std::is_same<T, float>::value && D_T == 16)) return false;
  if (misaligned(A.t, 16) || misaligned(A.w, 16) || misaligned(A.affs, 16) || misaligned(A.gout, 16) || misaligned(A.m, 4)) return false;
  if ((P.tbs | P.wbs | P.mbs | (long long)P.S) & 3) return false;
  constexpr TileCfg c = fwdv_cfg<D_T>();
  const size_t tp = (size_t)c.TH * c.TW;
  const size_t region = Lds<D_T, 1>::kBytes * (size_t)c.PLQ, dots = (size_t)P.K * tp * 4, parts = (size_t)P.K * (tp / 256) * 4;
  if (dots > region) return false;
  const size_t lds = region + parts;
  if (lds > (size_t)kLdsMax) return false;
  TParams Q;
  if (!plan_tiles_cached(P, c, false, &Q) || Q.n_near > kKV || Q.n_far > kFV) return false;
  if (rep_mostly_far(P, Q)) return false;
  const T *e = (const T*)A.e, *eo = (const T*)A.eo;
  const dim3 grid((unsigned)(Q.tiles_per_xcd * kXcd)), blk(c.TH * c.TW);
#define PEA_FV(CROP_, SELF_)                                                                                  \
  {                                                                                                           \
    constexpr auto kern = k_fwd_tiled_v<T, D_T, c.TH, c.TW, c.PLQ, true, CROP_, TRAIN, SELF_>;                \
    PEA_LAUNCH(kern, grid, blk, lds, s, P, Q, e, eo, A.t, A.w, A.m, A.affs, A.gout, A.st, inv_out)            \
  }
  const bool crop = P.border != PEA_BORDER_CIRCULAR;
  if (eo == e) { if (crop) PEA_FV(true, true) else PEA_FV(false, true) }
  else { if (crop) PEA_FV(true, false) else PEA_FV(false, false) }
#undef PEA_FV
  return true;
}

template <typename T, int D_T, bool TRAIN>
bool try_fwd_tiled(const KParams& P, const FwdT& A, float* inv_out, hipStream_t s) {
  constexpr TileCfg c = fwd_cfg<D_T>(0);
  constexpr int NT = c.TH * c.TW;
  if (P.border == PEA_BORDER_REPLICATE && !(std::is_same<T, float>::value && D_T == 16)) return false;
  TParams Q;
  if (!plan_tiles_cached(P, c, false, &Q, true) || rep_mostly_far(P, Q)) return false;
  const size_t lds = Lds<D_T, c.PLQ>::kBytes + (TRAIN ? (size_t)(NT / 64) * P.K * sizeof(float) : 0);
  const T *e = (const T*)A.e, *eo = (const T*)A.eo;
  const dim3 grid((unsigned)(Q.tiles_per_xcd * kXcd)), blk(NT);
#define PEA_FT(CROP_, SELF_)                                                                                  \
  {                                                                                                           \
    constexpr auto kern = k_fwd_tiled<T, D_T, c.TH, c.TW, c.PLQ, CROP_, TRAIN, SELF_>;                        \
    PEA_LAUNCH(kern, grid, blk, lds, s, P, Q, e, eo, A.t, A.w, A.m, A.affs, A.gout, A.st, inv_out)            \
  }
  const bool crop = P.border != PEA_BORDER_CIRCULAR;
  if (eo == e) { if (crop) PEA_FT(true, true) else PEA_FT(false, true) }
  else { if (crop) PEA_FT(true, false) else PEA_FT(false, false) }
#undef PEA_FT
  return true;
}

// D = 64: channels through LDS in two chunks of 32 (k_fwd_tiled_chunked, the D = 32 region geometry);
// D = 32: two chunks of 16 in the D = 16 geometry, i.e. two workgroups per CU instead of one (inference 176 -> 139 us,
// training forward 255 -> 237 us at B=8 x 32 x 544^2)
template <typename T, int D_T, int DC, bool TRAIN>
bool try_fwd_chunked(const KParams& P, const FwdT& A, hipStream_t s) {
  if (P.D != D_T || P.border == PEA_BORDER_REPLICATE) return false;
  constexpr TileCfg c = kCfg32;  // 16 x 32 tile, 1041 region pixels: 128 B (DC = 32) or 64 B (DC = 16) of LDS each
  TParams Q;
  if (!plan_tiles_cached(P, c, false, &Q) || Q.n_near > kChN || Q.n_far > kChF) return false;
  const size_t lds = Lds<DC, c.PLQ>::kBytes + (size_t)c.PLQ * 4 + (size_t)(c.TH * c.TW / 64) * P.K * 4;
  if (lds > (size_t)kLdsMax) return false;
  const T *e = (const T*)A.e, *eo = (const T*)A.eo;
  const dim3 grid((unsigned)(Q.tiles_per_xcd * kXcd)), blk(c.TH * c.TW);
#define PEA_FC(CROP_, SELF_)                                                                                  \
  {                                                                                                           \
    constexpr auto kern = k_fwd_tiled_chunked<T, D_T, DC, c.TH, c.TW, c.PLQ, CROP_, TRAIN, SELF_>;            \
    PEA_LAUNCH(kern, grid, blk, lds, s, P, Q, e, eo, A.t, A.w, A.m, A.affs, A.gout, A.st)                     \
  }
  const bool crop = P.border != PEA_BORDER_CIRCULAR;
  if (eo == e) {
    if (crop) PEA_FC(true, true) else PEA_FC(false, true)
  } else {
    // a second operand under the 128-VGPR budget of DC = 16 spills: not instantiated (the caller keeps the one-region kernels)
    if constexpr (DC > 16) { if (crop) PEA_FC(true, false) else PEA_FC(false, false) }
    else return false;
  }
#undef PEA_FC
  return true;
}

template <typename T, bool TRAIN>
bool fwd_any(const KParams& P, const FwdT& A, hipStream_t s, bool* wrote_inv) {
  *wrote_inv = false;
  if (P.D == 16) {
    bool done = false;
    if (TRAIN) done = try_fwd_v<T, 16, TRAIN>(P, A, A.inv_out, s);
    if (!done) done = try_fwd_tiled<T, 16, TRAIN>(P, A, A.inv_out, s);
    *wrote_inv = done && A.inv_out != nullptr;
    return done;
  }
  // 64 B of LDS per region pixel: two workgroups per CU.  Self loss / inference only: with a second operand the
  // 128-VGPR budget of that occupancy spills (and see pea_chunked.h on spill stores), so EMA calls keep the one-region kernels
  if (P.D == 32) {
    bool done = try_fwd_chunked<T, 32, 16, TRAIN>(P, A, s);
    if (!done && TRAIN) done = try_fwd_v<T, 32, TRAIN>(P, A, nullptr, s);
    if (!done) done = try_fwd_tiled<T, 32, TRAIN>(P, A, nullptr, s);
    return done;
  }
  if (P.D == 64) return try_fwd_chunked<T, 64, 32, TRAIN>(P, A, s);
  return false;
}

template <typename T, int D_T, bool RA, bool RB>
bool try_bwd_tiled(const KParams& P, const T* x, const T* nb, const float* g, const float* dl, T* dx, hipStream_t s) {
  constexpr TileCfg c = bwd_cfg<D_T>(0);
  if (P.border == PEA_BORDER_REPLICATE && !(std::is_same<T, float>::value && D_T == 16)) return false;
  TParams Q;
  // role A alone (a detached second operand's cross loss) reaches only p + o: a one-sided halo; role B needs p - o
  if (!plan_tiles_cached(P, c, !(RA && !RB), &Q, true)) return false;
  const size_t lds = Lds<D_T, c.PLQ>::kBytes;
  const dim3 grid((unsigned)(Q.tiles_per_xcd * kXcd)), blk(c.TH * c.TW);
  if (P.border == PEA_BORDER_CIRCULAR) {
    constexpr auto kern = k_bwd_tiled<T, D_T, c.TH, c.TW, c.PLQ, false, RA, RB>;
    PEA_LAUNCH(kern, grid, blk, lds, s, P, Q, x, nb, g, dl, dx)
  } else {
    constexpr auto kern = k_bwd_tiled<T, D_T, c.TH, c.TW, c.PLQ, true, RA, RB>;
    PEA_LAUNCH(kern, grid, blk, lds, s, P, Q, x, nb, g, dl, dx)
  }
  return true;
}

template <typename T, int D_T>
bool bwd_roles(const KParams& P, int roles, const void* x, const void* nbA, const void* nbB, const float* g, const float* dl, void* dx,
               hipStream_t s) {
  if (roles == 3) return try_bwd_tiled<T, D_T, true, true>(P, (const T*)x, (const T*)nbA, g, dl, (T*)dx, s);
  if (roles == 1) return try_bwd_tiled<T, D_T, true, false>(P, (const T*)x, (const T*)nbA, g, dl, (T*)dx, s);
  return try_bwd_tiled<T, D_T, false, true>(P, (const T*)x, (const T*)nbB, g, dl, (T*)dx, s);
}

}  // namespace

bool tiled_fwd(const KParams& P, const FwdArgs& A, hipStream_t s, bool* wrote_inv) {
  *wrote_inv = false;
  if (env().force_direct) return false;
  const FwdT T_ = {A.e, A.eo, A.t, A.w, A.m, A.affs, A.gout, A.st, A.inv_out};
  if (A.dtype == PEA_F16) return A.train ? fwd_any<__half, true>(P, T_, s, wrote_inv) : fwd_any<__half, false>(P, T_, s, wrote_inv);
  return A.train ? fwd_any<float, true>(P, T_, s, wrote_inv) : fwd_any<float, false>(P, T_, s, wrote_inv);
}

bool tiled_bwd(const KParams& P, int dtype, int roles, const void* x, const void* nbA, const void* nbB, const float* g,
               const float* dl, void* dx, hipStream_t s) {
  if (env().force_direct || (P.D != 16 && P.D != 32)) return false;
  if (dtype == PEA_F16)
    return P.D == 16 ? bwd_roles<__half, 16>(P, roles, x, nbA, nbB, g, dl, dx, s) : bwd_roles<__half, 32>(P, roles, x, nbA, nbB, g, dl, dx, s);
  return P.D == 16 ? bwd_roles<float, 16>(P, roles, x, nbA, nbB, g, dl, dx, s) : bwd_roles<float, 32>(P, roles, x, nbA, nbB, g, dl, dx, s);
}

}  // namespace pea
