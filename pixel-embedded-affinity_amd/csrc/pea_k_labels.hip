// pea_k_labels.hip -- launchers of the labels-in training step (pea_fused_labels.h) and of the label-weight tables.
// One translation unit of libpea_hip.so (pea_host.h).
#include "pea_plan.h"
#include "pea_fused_labels.h"

namespace pea {

namespace {

#define PEA_LAUNCH(kern, grid, blk, lds, s, ...)              \
  {                                                           \
    if (allow_lds<kern>(lds)) return false;                   \
    hipLaunchKernelGGL(kern, grid, blk, lds, s, __VA_ARGS__); \
  }

// same tile plan as the tiled backward
template <typename T, int D_T, bool RB>
bool try_fused_labels(const KParams& P, const T* x, const T* nb, const int32_t* labels, const float* wtab, unsigned lflags,
                      float* affs, LossState* st, const float* dl, T* dx, hipStream_t s) {
  constexpr TileCfg c = bwd_cfg<D_T>(0);
  if (P.border == PEA_BORDER_REPLICATE) return false;  // (the labels-in kernels test, they do not clamp)
  TParams Q;
  if (!plan_tiles_cached(P, c, RB, &Q)) return false;
  const size_t lds = Lds<D_T, c.PLQ>::kBytes + (size_t)(c.TH * c.TW / 64) * P.K * sizeof(float);
  if (lds > (size_t)kLdsMax) return false;
  const dim3 grid((unsigned)(Q.tiles_per_xcd * kXcd)), blk(c.TH * c.TW);
  if (P.border == PEA_BORDER_CIRCULAR) {
    constexpr auto kern = k_fused_labels<T, D_T, c.TH, c.TW, c.PLQ, false, RB>;
    PEA_LAUNCH(kern, grid, blk, lds, s, P, Q, x, nb, labels, wtab, lflags, affs, st, dl, dx)
  } else {
    constexpr auto kern = k_fused_labels<T, D_T, c.TH, c.TW, c.PLQ, true, RB>;
    PEA_LAUNCH(kern, grid, blk, lds, s, P, Q, x, nb, labels, wtab, lflags, affs, st, dl, dx)
  }
  return true;
}

// self + detached-EMA cross loss from labels in one launch (k_fused_labels_dual): both plans must split the stencil
// into the same near / far entries
template <typename T, int D_T>
bool try_fused_labels_dual(const KParams& P, const KParams& P2, const T* x, const T* ema, const int32_t* labels, const float* wtab,
                           unsigned lflags, float* affs, LossState* st, LossState* st2, const float* dl, const float* dl2, T* dx,
                           hipStream_t s) {
  constexpr TileCfg c = bwd_cfg<D_T>(0);
  TParams Q, Q2;
  if (P.border == PEA_BORDER_REPLICATE) return false;
  if (!plan_tiles_cached(P, c, true, &Q) || !plan_tiles_cached(P2, c, false, &Q2)) return false;
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
    PEA_LAUNCH(kern, grid, blk, lds, s, P, Q, Q2, C2, x, ema, labels, wtab, lflags, affs, st, st2, dl, dl2, dx)
  } else {
    constexpr auto kern = k_fused_labels_dual<T, D_T, c.TH, c.TW, c.PLQ, true>;
    PEA_LAUNCH(kern, grid, blk, lds, s, P, Q, Q2, C2, x, ema, labels, wtab, lflags, affs, st, st2, dl, dl2, dx)
  }
  return true;
}

GParams gparams(const PeaDesc* desc, unsigned flags) {
  GParams G;
  G.B = desc->B; G.Z = desc->dims[0]; G.Y = desc->dims[1]; G.X = desc->dims[2]; G.K = desc->K;
  G.S = G.Z * G.Y * G.X;
  G.flags = flags;
  for (int i = 0; i < PEA_MAX_K; ++i)
    for (int a = 0; a < 3; ++a) G.off[i][a] = i < desc->K ? desc->offsets[i][a] : 0;
  return G;
}

}  // namespace

bool labels_step(const KParams& P, int dtype, const void* e, const void* e_other, const int32_t* labels, const float* wtab,
                 unsigned lflags, float* affs, LossState* st, const float* dl, void* de, hipStream_t s) {
  if ((P.D != 16 && P.D != 32) || env().force_direct) return false;
#define PEA_LAB(TT, DD)                                                                                                       \
  {                                                                                                                           \
    const TT *x = (const TT*)e, *nb = (const TT*)e_other;                                                                     \
    return nb ? try_fused_labels<TT, DD, false>(P, x, nb, labels, wtab, lflags, affs, st, dl, (TT*)de, s)                     \
              : try_fused_labels<TT, DD, true>(P, x, x, labels, wtab, lflags, affs, st, dl, (TT*)de, s);                      \
  }
  if (dtype == PEA_F16) {
    if (P.D == 16) PEA_LAB(__half, 16) else PEA_LAB(__half, 32)
  } else {
    if (P.D == 16) PEA_LAB(float, 16) else PEA_LAB(float, 32)
  }
#undef PEA_LAB
}

bool labels_step_dual(const KParams& P, const KParams& P2, int dtype, const void* e, const void* ema, const int32_t* labels,
                      const float* wtab, unsigned lflags, float* affs, LossState* st, LossState* st2, const float* dl,
                      const float* dl2, void* de, hipStream_t s) {
  if ((P.D != 16 && P.D != 32) || env().force_direct) return false;
#define PEA_LD(T_, D_) try_fused_labels_dual<T_, D_>(P, P2, (const T_*)e, (const T_*)ema, labels, wtab, lflags, affs, st, st2, dl, dl2, (T_*)de, s)
  if (P.D == 16) return dtype == PEA_F16 ? PEA_LD(__half, 16) : PEA_LD(float, 16);
  return dtype == PEA_F16 ? PEA_LD(__half, 32) : PEA_LD(float, 32);
#undef PEA_LD
}

// pea_gen_targets: one count per (image, channel); pea_label_weights: one partial per (image, channel, workgroup)
size_t label_counts_bytes(const PeaDesc* desc) {
  const size_t wgs = (size_t)((desc->dims[2] + 63) / 64) * ((desc->dims[1] + 4 * kCntRows - 1) / (4 * kCntRows)) * desc->dims[0];
  return (size_t)desc->B * desc->K * sizeof(unsigned) * std::max<size_t>(1, wgs);
}

int label_weights(const PeaDesc* desc, const int32_t* labels, unsigned flags, float* wtab, void* ws, hipStream_t s) {
  const GParams G = gparams(desc, flags);
  if ((long long)G.B * G.Z > 65535 || (G.Y + 4 * kCntRows - 1) / (4 * kCntRows) > 65535) return PEA_E_UNSUPPORTED;
  const dim3 cgrid((unsigned)((G.X + 63) / 64), (unsigned)((G.Y + 4 * kCntRows - 1) / (4 * kCntRows)), (unsigned)(G.B * G.Z));
  // in-plane tables within the halo, quads inside rows: the LDS-staged counts (the same integers)
  bool lds_ok = G.X % 4 == 0 && !misaligned(labels, 16) && !env().force_direct;
  for (int i = 0; i < G.K && lds_ok; ++i) lds_ok = G.off[i][0] == 0 && abs(G.off[i][1]) <= kCntHalo && abs(G.off[i][2]) <= kCntHalo;
  if (lds_ok) hipLaunchKernelGGL(k_label_counts_lds, cgrid, dim3(256), 0, s, G, labels, (unsigned*)ws);
  else hipLaunchKernelGGL(k_label_counts, cgrid, dim3(256), 0, s, G, labels, (unsigned*)ws);
  const int per_img = (int)(cgrid.x * cgrid.y) * G.Z;
  const int n = G.B * G.K;
  hipLaunchKernelGGL(k_weight_table, dim3((unsigned)n), dim3(64), 0, s, G.S, per_img, (const unsigned*)ws, wtab);
  return hip_rc();
}

int gen_targets(const PeaDesc* desc, const int32_t* labels, unsigned flags, float* target, uint8_t* mask, float* weight,
                void* workspace, size_t need, hipStream_t s) {
  const GParams G = gparams(desc, flags);
  if (desc->B > 65535 || (long long)desc->B * desc->K > 65535) return PEA_E_UNSUPPORTED;
  if (hipMemsetAsync(workspace, 0, need, s) != hipSuccess) return hip_rc();
  const unsigned chunks = (unsigned)((G.S + 255) / 256);
  const unsigned gx = (chunks + kTgtNit - 1) / kTgtNit;  // kTgtNit * 256 pixels per workgroup
  hipLaunchKernelGGL(k_gen_targets, dim3(gx, (unsigned)G.B), dim3(256), 0, s, G, labels, target, mask, (unsigned*)workspace);
  if (weight)
    hipLaunchKernelGGL(k_gen_weights, dim3(chunks, (unsigned)(G.B * G.K)), dim3(256), 0, s, G, target, (const unsigned*)workspace, weight);
  return hip_rc();
}

}  // namespace pea
