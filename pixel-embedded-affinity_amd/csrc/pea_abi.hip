// pea_abi.hip -- the C ABI of the embedding -> affinity hot path (include/pea.h): validation, descriptor -> kernel
// parameters, dispatch over the kernel families.  One translation unit of libpea_hip.so (pea_host.h); no kernels here.
//
// Replaces the Python op loops of the reference (weih527/Pixel-Embedded-Affinity):
//   scripts_cvppp/loss/loss_embedding_mse.py:7-95   (2D: normalize -> K x torch.roll/mul/sum -> WeightedMSE)
//   scripts_ac3ac4/loss/loss_embedding_mse.py:7-289 (3D: cropped slices, norm1 / norm5 / ema / inf)
//   loss/loss.py:106-124 WeightedMSE                (fused)
// and the autograd backward of those, by one forward launch and one backward launch.  Order of the families for a call:
// the LDS-DMA cross kernels (pea_k_xdma.hip) where the stencil is axis-aligned and the storage f32; else the LDS-tiled box
// kernels (pea_k_tiled.hip); else the global-memory kernels (pea_k_direct.hip), which take everything.  HBM-bound op
// (2-7 flop/B) => no MFMA; no float atomics anywhere (the loss is summed in integers, pea_loss.h): bit-reproducible.
#include <stdlib.h>
#include <string.h>

#include "pea_host.h"

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
  memset(&P, 0, sizeof(P));  // padding bytes too: KParams is the key of the plan memo (pea_host.h PlanCache)
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

constexpr size_t kStateBytes = sizeof(LossState);

// forward dispatch.  Returns PEA_OK or an error; the loss (training) is complete in stream order when it returns.
int run_fwd(const KParams& P, FwdArgs A, hipStream_t s) {
  const bool self = A.eo == A.e;
  bool launched = false;
  if (self) launched = zmarch_fwd(P, A, s);  // 3D volumes whose stencil steps along z (norm5 / norm1): pea_zmarch.h
  if (!launched) {
    if (A.train) launched = self ? xdma_fwd_self(P, A, s) : xdma_fwd_other(P, A, s);
    else if (self) launched = xdma_fwd_self(P, A, s);
  }
  if (!launched && self) launched = box_fwd(P, A, s);
  if (!launched) {
    // 1 / norm planes: the tiled D = 16 self forward writes its plane while it stages; everything else gets k_inv_norm
    float* inv = A.inv_out;
    A.inv_out = self ? inv : nullptr;
    bool wrote_inv = false;
    launched = tiled_fwd(P, A, s, &wrote_inv);
    if (!launched) direct_fwd(P, A, s);
    if (inv) {
      if (!wrote_inv) launch_inv_norm(P, A.dtype, A.e, inv, s);
      if (!self) launch_inv_norm(P, A.dtype, A.eo, inv + (size_t)P.B * P.S, s);
    }
    A.inv_out = inv;
  }
  int rc = hip_rc();
  if (!rc && A.train) {
    launch_loss_finish(P, A.st, A.loss_out, s);
    rc = hip_rc();
  }
  if (rc && A.train) {
    // the state block must be ZERO between calls (pea_workspace_init's contract) and only the finish puts it back: after an error
    // partials may have been queued without a finish -- prepare the block again, or every later loss on it would be offset
    launch_loss_state_init(A.st, 1, s);
    (void)hipGetLastError();
  }
  return rc;
}

int run_bwd(const KParams& P, int dtype, int roles, const void* x, const void* nbA, const void* nbB, const float* g, const float* dl,
            void* dx, hipStream_t s) {
  if (tiled_bwd(P, dtype, roles, x, nbA, nbB, g, dl, dx, s)) return hip_rc();
  return direct_bwd(P, dtype, roles, x, nbA, nbB, g, dl, dx, s);
}

}  // namespace

extern "C" {

int pea_version(void) { return PEA_ABI_VERSION; }

void pea_reload_env(void) { env_reload(); }

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

size_t pea_workspace_bytes(const PeaDesc* desc) { return validate(desc) ? 0 : kStateBytes; }

int pea_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
  if (!workspace) return PEA_E_NULL;
  if (misaligned(workspace, 8)) return PEA_E_ALIGN;
  const size_t n = workspace_bytes / kStateBytes;
  if (n < 1 || n > 0x7fffffff) return PEA_E_WORKSPACE;
  launch_loss_state_init((LossState*)workspace, (int)n, (hipStream_t)stream);
  return hip_rc();
}

int pea_affinity_infer(const PeaDesc* desc, const void* e, const void* e_other, float* affs, void* stream) {
  const int rc = validate(desc);
  if (rc) return rc;
  if (!e || !affs) return PEA_E_NULL;
  const size_t es = desc->dtype == PEA_F16 ? 2 : 4;
  if (misaligned(e, es) || misaligned(e_other, es) || misaligned(affs, 4)) return PEA_E_ALIGN;
  const KParams P = make_params(desc);
  FwdArgs A = {};
  A.e = e; A.eo = e_other ? e_other : e; A.affs = affs; A.dtype = desc->dtype; A.train = false;
  return run_fwd(P, A, (hipStream_t)stream);
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
  if (!workspace || workspace_bytes < kStateBytes) return PEA_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  FwdArgs A = {};
  A.e = e; A.eo = e_other ? e_other : e;
  A.t = target; A.w = weight; A.m = mask; A.affs = affs; A.gout = g_out;
  A.st = (LossState*)workspace; A.loss_out = loss_out; A.inv_out = inv_norm_out;
  A.dtype = desc->dtype; A.train = true;
  rc = run_fwd(P, A, s);
  if (rc) return rc;
  if (e_other && e_other == e && inv_norm_out) {
    // a second operand that aliases the first (ema_embedding_loss(e, e.detach(), ..)): the forward ran as a self loss and wrote
    // ONE plane; the contract with a second operand is two planes (the role-A backward reads the second): same values
    const size_t n = (size_t)P.B * P.S * sizeof(float);
    if (hipMemcpyAsync(inv_norm_out + (size_t)P.B * P.S, inv_norm_out, n, hipMemcpyDeviceToDevice, s) != hipSuccess) return hip_rc();
  }
  return PEA_OK;
}

int pea_affinity_fwd_dual_ex(const PeaDesc* desc, const PeaDesc* desc_cross, const void* e, const void* ema, const float* target,
                             const float* weight, const uint8_t* mask, float* affs, float* g_out, float* g_cross_out,
                             float* inv_norm_out, float* inv_norm_other_out, float* loss_out, float* loss_cross_out, void* workspace,
                             void* workspace_cross, size_t workspace_bytes, void* stream) {
  int rc = validate(desc);
  if (rc) return rc;
  rc = validate(desc_cross);
  if (rc) return rc;
  if (!e || !ema || !target || !weight || !g_out || !g_cross_out || !inv_norm_out || !inv_norm_other_out || !loss_out || !loss_cross_out)
    return PEA_E_NULL;
  // one geometry, one stencil, one normaliser: the two descriptors may differ in lambda (affs0_weight of the cross loss) and in the
  // activation of a map (the cross loss writes none)
  const PeaDesc &a = *desc, &b = *desc_cross;
  bool same = a.ndim == b.ndim && a.B == b.B && a.D == b.D && a.K == b.K && a.border == b.border && a.dtype == b.dtype &&
              a.norm == b.norm && a.eps == b.eps && ((a.flags ^ b.flags) & ~kActMask) == 0 && a.target_bstride == b.target_bstride &&
              a.weight_bstride == b.weight_bstride && a.mask_bstride == b.mask_bstride;
  for (int i = 0; same && i < 3; ++i) same = a.dims[i] == b.dims[i];
  for (int i = 0; same && i < a.K; ++i) same = a.offsets[i][0] == b.offsets[i][0] && a.offsets[i][1] == b.offsets[i][1] && a.offsets[i][2] == b.offsets[i][2];
  if (!same) return PEA_E_DESC;
  if (misaligned(e, 4) || misaligned(ema, 4) || misaligned(affs, 4) || misaligned(g_out, 4) || misaligned(g_cross_out, 4) ||
      misaligned(target, 4) || misaligned(weight, 4) || misaligned(loss_out, 4) || misaligned(loss_cross_out, 4) ||
      misaligned(workspace, 8) || misaligned(workspace_cross, 8) || misaligned(inv_norm_out, 4) || misaligned(inv_norm_other_out, 4))
    return PEA_E_ALIGN;
  if (!workspace || !workspace_cross || workspace == workspace_cross || workspace_bytes < kStateBytes) return PEA_E_WORKSPACE;
  if (e == ema) return PEA_E_UNSUPPORTED;  // (an aliased second operand is a self loss twice: the two calls handle it)
  // ONE kernel writes both losses' outputs: aliased outputs would interleave (the two calls this replaces give the last writer's)
  if (g_out == g_cross_out || inv_norm_out == inv_norm_other_out || loss_out == loss_cross_out) return PEA_E_DESC;
  const KParams P = make_params(desc), P2 = make_params(desc_cross);
  hipStream_t s = (hipStream_t)stream;
  FwdArgs A = {}, A2 = {};
  A.e = e; A.eo = e; A.t = target; A.w = weight; A.m = mask; A.affs = affs; A.gout = g_out;
  A.st = (LossState*)workspace; A.loss_out = loss_out; A.inv_out = inv_norm_out; A.dtype = desc->dtype; A.train = true;
  A2 = A;
  A2.eo = ema; A2.affs = nullptr; A2.gout = g_cross_out; A2.st = (LossState*)workspace_cross; A2.loss_out = loss_cross_out;
  A2.inv_out = inv_norm_other_out;
  if (!xdma_fwd_dual(P, P2, A, A2, s)) {
    const int pe = hip_rc();
    return pe ? pe : PEA_E_UNSUPPORTED;  // nothing was launched: pea_affinity_fwd_ex twice
  }
  rc = hip_rc();
  if (!rc) {
    launch_loss_finish(P, A.st, A.loss_out, s);
    launch_loss_finish(P2, A2.st, A2.loss_out, s);
    rc = hip_rc();
  }
  if (rc) {  // (run_fwd: the state blocks must be zero between calls)
    launch_loss_state_init(A.st, 1, s);
    launch_loss_state_init(A2.st, 1, s);
    (void)hipGetLastError();
  }
  return rc;
}

int pea_affinity_fwd(const PeaDesc* desc, const void* e, const void* e_other, const float* target,
                     const float* weight, const uint8_t* mask, float* affs, float* g_out, float* loss_out,
                     void* workspace, size_t workspace_bytes, void* stream) {
  return pea_affinity_fwd_ex(desc, e, e_other, target, weight, mask, affs, g_out, nullptr, loss_out, workspace, workspace_bytes, stream);
}

int pea_cross_supported(const PeaDesc* desc, int backward) {
  if (validate(desc)) return 0;
  const KParams P = make_params(desc);
  if (backward == 5) return xdma_fwd_dual_supported(P, desc->dtype) ? 1 : 0;  // does pea_affinity_fwd_dual_ex fuse the pair?
  if (backward == 4) return xdma_cross_supported(P, desc->dtype, 4);  // ... with a detached second operand?
  if (backward == 3) {  // does pea_affinity_bwd_ex2 READ the raw affinity map for this descriptor (self loss)?
    if (zmarch_bwd_supported(P, desc->dtype)) return 1;
    return (env().bwd_pf && !(P.flags & kActMask) && P.D > 16 && xdma_cross_supported(P, desc->dtype, 1)) ? 1 : 0;
  }
  if (backward == 1 && zmarch_bwd_supported(P, desc->dtype)) return 1;
  if (xdma_cross_supported(P, desc->dtype, backward)) return 1;
  return (backward == 0 || backward == 1) && box_supported(P, desc->dtype) ? 1 : 0;
}

int pea_inv_norm(const PeaDesc* desc, const void* e, float* inv_norm_out, void* stream) {
  const int rc = validate(desc);
  if (rc) return rc;
  if (!e || !inv_norm_out) return PEA_E_NULL;
  if (misaligned(e, desc->dtype == PEA_F16 ? 2 : 4) || misaligned(inv_norm_out, 4)) return PEA_E_ALIGN;
  const KParams P = make_params(desc);
  launch_inv_norm(P, desc->dtype, e, inv_norm_out, (hipStream_t)stream);
  return hip_rc();
}

int pea_affinity_bwd_ex(const PeaDesc* desc, const void* e, const void* e_other, const float* g, const float* inv_norm,
                        const float* dloss, void* de, void* de_other, void* stream) {
  return pea_affinity_bwd_ex2(desc, e, e_other, g, inv_norm, nullptr, dloss, de, de_other, stream);
}

int pea_affinity_bwd_ex2(const PeaDesc* desc, const void* e, const void* e_other, const float* g, const float* inv_norm,
                         const float* affs, const float* dloss, void* de, void* de_other, void* stream) {
  int rc = validate(desc);
  if (rc) return rc;
  if (!e || !g || (!de && !de_other)) return PEA_E_NULL;
  if (de_other && !e_other) return PEA_E_NULL;
  const size_t es = desc->dtype == PEA_F16 ? 2 : 4;
  if (misaligned(e, es) || misaligned(e_other, es) || misaligned(de, es) || misaligned(de_other, es) ||
      misaligned(g, 4) || misaligned(dloss, 4) || misaligned(inv_norm, 4) || misaligned(affs, 4))
    return PEA_E_ALIGN;
  const KParams P = make_params(desc);
  hipStream_t s = (hipStream_t)stream;
  const int dt = desc->dtype;
  const bool accumulate = (desc->flags & PEA_FLAG_ACCUMULATE_DE) != 0;
  if (!e_other) {
    if (accumulate) return PEA_E_UNSUPPORTED;  // de += is implemented for the detached second operand's role-A backward only
    // self loss: the LDS-DMA cross kernel when the 1 / norm plane came along and the stencil is axis-aligned
    // a 3D stencil inside the unit box with a step along z (norm1): the box backward serves the z neighbours from LDS where the
    // cross backward gathers them from global memory -- 1.34 against 1.91 ms on a 24 x 1024^2 sub-volume (profiles/exp_3d_norm1.py;
    // the forward stays on the cross kernels: 1.25 against 1.54 ms, the box forward walks all 26 displacements whatever K is)
    bool zstep = false;
    for (int i = 0; i < P.K; ++i) zstep |= P.off[i][0] != 0;
    if (dt == PEA_F32 && zstep && zmarch_bwd(P, (const float*)e, inv_norm, g, affs, dloss, (float*)de, s)) return hip_rc();
    if (dt == PEA_F32 && zstep && box_bwd(P, (const float*)e, inv_norm, g, dloss, (float*)de, s)) return hip_rc();
    if (dt == PEA_F32 && xdma_bwd_self(P, (const float*)e, inv_norm, g, affs, dloss, (float*)de, s)) return hip_rc();
    if (dt == PEA_F16 && xdma_bwd_self_h(P, e, inv_norm, g, affs, dloss, de, s)) return hip_rc();
    if (dt == PEA_F32 && box_bwd(P, (const float*)e, inv_norm, g, dloss, (float*)de, s)) return hip_rc();
    return run_bwd(P, dt, 3, e, e, e, g, dloss, de, s);
  }
  // (a second operand that aliases the first is still a second operand: its roles are separate)
  if (de && !de_other && dt == PEA_F32 &&
      xdma_bwd_other(P, (const float*)e, (const float*)e_other, inv_norm, g, affs, dloss, (float*)de, accumulate, s))
    return hip_rc();  // detached second operand: the role-A cross kernel (inv_norm = the two planes pea_affinity_fwd_ex wrote)
  if (de && !de_other && dt == PEA_F16 && inv_norm && !accumulate && env().bwd_xdma && !env().force_direct &&
      xdma_h_bwd_other(P, e, e_other, inv_norm, g, affs, dloss, de, s))
    return hip_rc();
  if (accumulate) return PEA_E_UNSUPPORTED;
  if (de) {
    rc = run_bwd(P, dt, 1, e, e_other, nullptr, g, dloss, de, s);
    if (rc) return rc;
  }
  if (!de_other) return PEA_OK;
  return run_bwd(P, dt, 2, e_other, nullptr, e, g, dloss, de_other, s);
}

int pea_affinity_bwd(const PeaDesc* desc, const void* e, const void* e_other, const float* g, const float* dloss,
                     void* de, void* de_other, void* stream) {
  return pea_affinity_bwd_ex(desc, e, e_other, g, nullptr, dloss, de, de_other, stream);
}

size_t pea_targets_workspace_bytes(const PeaDesc* desc) {
  if (validate(desc)) return 0;
  return label_counts_bytes(desc);
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
  return gen_targets(desc, labels, flags, target, mask, weight, workspace, need, (hipStream_t)stream);
}

int pea_label_weights(const PeaDesc* desc, const int32_t* labels, unsigned flags, float* wtab, void* workspace,
                      size_t workspace_bytes, void* stream) {
  const int rc = validate(desc);
  if (rc) return rc;
  if (!labels || !wtab) return PEA_E_NULL;
  if (misaligned(labels, 4) || misaligned(wtab, 4) || misaligned(workspace, 4)) return PEA_E_ALIGN;
  if (flags & ~(PEA_TGT_PADDING | PEA_TGT_BOTH_FOREGROUND | PEA_TGT_MASK_INSIDE)) return PEA_E_DESC;
  if (!workspace || workspace_bytes < label_counts_bytes(desc)) return PEA_E_WORKSPACE;
  return label_weights(desc, labels, flags, wtab, workspace, (hipStream_t)stream);
}

size_t pea_labels_scratch_bytes(const PeaDesc* desc) {
  if (validate(desc)) return 0;
  const KParams P = make_params(desc);
  if (!xdma_labels_supported(P, desc->dtype)) return 0;
  return ((size_t)(P.K + 1) * P.B * P.S * sizeof(float) + 15) & ~(size_t)15;
}

int pea_affinity_fwd_bwd_labels_ex(const PeaDesc* desc, const void* e, const void* e_other, const int32_t* labels,
                                   const float* wtab, unsigned flags, float* affs, float* loss_out, const float* dloss, void* de,
                                   void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes, void* stream) {
  int rc = validate(desc);
  if (rc) return rc;
  if (!e || !labels || !wtab || !loss_out || !de) return PEA_E_NULL;
  const size_t es = desc->dtype == PEA_F16 ? 2 : 4;
  if (misaligned(e, es) || misaligned(e_other, es) || misaligned(de, es) || misaligned(affs, 4) || misaligned(labels, 4) ||
      misaligned(wtab, 4) || misaligned(loss_out, 4) || misaligned(dloss, 4) || misaligned(workspace, 8) || misaligned(scratch, 16))
    return PEA_E_ALIGN;
  if (flags & ~(PEA_TGT_PADDING | PEA_TGT_BOTH_FOREGROUND | PEA_TGT_MASK_INSIDE | PEA_TGT_ACCUMULATE)) return PEA_E_DESC;
  const KParams P = make_params(desc);
  if (!workspace || workspace_bytes < kStateBytes) return PEA_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  LossState* st = (LossState*)workspace;
  // Two launches on the cross kernels where they apply and the caller lent the scratch for g and the 1 / norm plane: the
  // labels-in forward (k_fwd_xdma<.., LAB>: 360 MB instead of the tensor forward's 596) and the cross backward -- 70 + 94 us
  // against 202 us for the one-launch box kernel, whose single workgroup per CU and far gathers the cross structure avoids.
  // (The one-launch form on the cross structure would stage every channel pair twice -- the dot products need all channels
  // before the first coefficient exists -- i.e. twice the backward's L2 -> LDS traffic, which is what bounds it.)
  if (scratch && !e_other && !(flags & PEA_TGT_ACCUMULATE) && scratch_bytes >= pea_labels_scratch_bytes(desc) &&
      xdma_labels_supported(P, desc->dtype)) {
    float* g = (float*)scratch;
    float* inv = g + (size_t)P.K * P.B * P.S;
    FwdArgs A = {};
    A.e = e; A.eo = e; A.affs = affs; A.gout = g; A.st = st; A.loss_out = loss_out; A.inv_out = inv;
    A.dtype = desc->dtype; A.train = true;
    if (xdma_fwd_labels(P, A, labels, wtab, flags, s)) {
      rc = hip_rc();
      if (rc) return rc;
      launch_loss_finish(P, st, loss_out, s);
      // (the map this forward wrote is raw unless the caller asked for an activation: the projection-first backward takes it)
      if (!xdma_bwd_self(P, (const float*)e, inv, g, (P.flags & kActMask) ? nullptr : affs, dloss, (float*)de, s))
        return run_bwd(P, desc->dtype, 3, e, e, e, g, dloss, de, s);
      return hip_rc();
    }
    rc = hip_rc();
    if (rc) return rc;
  }
  if (!labels_step(P, desc->dtype, e, e_other, labels, wtab, flags, affs, st, dloss, de, s)) {
    const int pe = hip_rc();
    return pe ? pe : PEA_E_UNSUPPORTED;
  }
  rc = hip_rc();
  if (rc) return rc;
  launch_loss_finish(P, st, loss_out, s);
  return hip_rc();
}

int pea_affinity_fwd_bwd_labels(const PeaDesc* desc, const void* e, const void* e_other, const int32_t* labels,
                                const float* wtab, unsigned flags, float* affs, float* loss_out, const float* dloss, void* de,
                                void* workspace, size_t workspace_bytes, void* stream) {
  return pea_affinity_fwd_bwd_labels_ex(desc, e, e_other, labels, wtab, flags, affs, loss_out, dloss, de, workspace, workspace_bytes,
                                        nullptr, 0, stream);
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
  if (!workspace || workspace_bytes < 2 * kStateBytes) return PEA_E_WORKSPACE;  // each loss has its own state
  hipStream_t s = (hipStream_t)stream;
  LossState *st = (LossState*)workspace, *st2 = st + 1;
  if (!labels_step_dual(P, P2, desc->dtype, e, ema, labels, wtab, flags, affs, st, st2, dloss, dloss_cross, de, s)) {
    const int pe = hip_rc();
    return pe ? pe : PEA_E_UNSUPPORTED;
  }
  rc = hip_rc();
  if (rc) return rc;
  launch_loss_finish(P, st, loss_out, s);
  launch_loss_finish(P2, st2, loss_cross_out, s);
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
  // one launch only where the cross kernel's second phase applies (the tiled two-phase kernel it superseded is gone:
  // 241 us against 214 us); everything else: two pea_affinity_bwd_ex calls and an add, as the header says
  if (desc->dtype == PEA_F32 && xdma_bwd_dual(P, (const float*)e, (const float*)ema, inv_norm, inv_norm_other, g, g_cross, dloss,
                                              dloss_cross, (float*)de, (hipStream_t)stream))
    return hip_rc();
  const int pe = hip_rc();
  return pe ? pe : PEA_E_UNSUPPORTED;
}

}  // extern "C"
