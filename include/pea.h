/*
 * pea.h — C ABI of the MI355X-native embedding -> affinity hot path.
 *
 * One shared library (libpea_hip.so, built by hipcc for gfx950) exports exactly the
 * entry points declared here.  Every data pointer is a DEVICE pointer owned by the
 * caller; the library allocates nothing, keeps no global state, never synchronises the
 * host, and launches on the HIP stream it is handed (`void *stream` is a hipStream_t;
 * NULL = the default stream).  No torch types appear in any signature.
 *
 * What each entry point replaces in the reference (weih527/Pixel-Embedded-Affinity):
 *
 *   pea_affinity_infer  <- scripts_cvppp/loss/loss_embedding_mse.py:58-66   embedding2affs
 *                          scripts_ac3ac4/loss/loss_embedding_mse.py:54-67  inf_embedding_loss_norm1
 *                          scripts_ac3ac4/loss/loss_embedding_mse.py:212-234 inf_embedding_loss_norm5
 *   pea_affinity_fwd    <- scripts_cvppp/loss/loss_embedding_mse.py:18-47   embedding_loss
 *                          scripts_cvppp/loss/loss_embedding_mse.py:79-95   ema_embedding_loss
 *                          scripts_ac3ac4/loss/loss_embedding_mse.py:7-27   embedding_loss_norm1
 *                          scripts_ac3ac4/loss/loss_embedding_mse.py:169-194 embedding_loss_norm5
 *                          scripts_ac3ac4/loss/loss_embedding_mse.py:30-51,263-289 ema_..._norm1/5
 *                          with loss/loss.py:106-124 WeightedMSE (same in all three script trees) fused in
 *   pea_affinity_bwd    <- the torch.autograd backward of the functions above (also the vjp for foreign criteria)
 *                          (the reference has no explicit backward; loss.backward() at
 *                          scripts_cvppp/main.py:311, scripts_ac3ac4/main.py:232)
 *
 * Tensor layouts (all C-contiguous, exactly the reference's):
 *   e, e_other, de, de_other : [B, D, Z, Y, X]   f32 (PEA_F32) or f16 (PEA_F16);  2D => Z = 1
 *   target, weight, affs     : [B, K, Z, Y, X]   f32
 *   mask                     : [B, K, Z, Y, X]   u8   (NULL => all ones; the 3D path has none)
 *
 * Semantics (SURVEY.md section 8a closed forms; n(p) = max(||e(p)||_2, eps), ehat = e / n):
 *   a_i(p)  = < ehat(p), ehat_other(p + o_i) >          (ehat_other = ehat when e_other == NULL)
 *   border CIRCULAR : p + o_i taken modulo (Y, X)   (torch.roll, 2D reference path)
 *   border CROP_ZERO: a_i(p) = 0 and no loss where p + o_i leaves the volume (3D reference path)
 *   border REPLICATE: p + o_i clamped into the volume (embedding_loss_norm6; every pair exists)
 *   r_i(p)  = a_i(p) * m_i(p) - t_i(p) * m_i(p)
 *   L_i     = sum_{b,p} w_i(p) * r_i(p)^2 / N_i ,   loss = sum_i lambda_i * L_i
 *   N_i     = B * X                      (PEA_NORM_BX: the 2D WeightedMSE quirk, pred is [B,H,W])
 *           = B * prod(dims - |o_i|)     (PEA_NORM_CROPPED: 3D, pred is the cropped [B,1,Z',Y',X'])
 *           = B * Z * Y * X              (PEA_NORM_FULL)
 */
#ifndef PEA_H_
#define PEA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PEA_ABI_VERSION 2 /* version of PeaDesc and of the semantics of the existing calls; new entry points do not bump it.
                            2: the workspace of the training forward is a small STATE block that pea_workspace_init prepares once (was: a
                               per-workgroup table, no preparation); pea_affinity_fwd_bwd (round 1) and pea_affinity_bwd_head are gone */
#define PEA_MAX_K 32 /* 26-neighbourhood (BASELINE config 4) fits */

/* border modes */
#define PEA_BORDER_CIRCULAR 0
#define PEA_BORDER_CROP_ZERO 1
#define PEA_BORDER_REPLICATE 2 /* neighbour index clamped into the volume (nn.ReplicationPad3d + slice: shift_tensor,
                                  scripts_ac3ac4/loss/loss_embedding_mse.py:294-344; embedding_loss_norm6 :346-354) */
/* storage dtype of e / de (arithmetic is always f32) */
#define PEA_F32 0
#define PEA_F16 1
/* loss normaliser */
#define PEA_NORM_BX 0
#define PEA_NORM_CROPPED 1
#define PEA_NORM_FULL 2
/* flags */
/* activation of the affs OUTPUT (the loss and its gradient always use the raw cosine a); applied in this order: */
#define PEA_FLAG_HALF_SHIFT 4u /* a -> (a + 1) / 2   (scripts_cvppp/loss/loss_embedding.py:10,35; for unit vectors also the L2 affinity
                                  1 - |ehat_p - ehat_q|^2 / 4 of scripts_ac3ac4/loss/embedding2affs_3d_l2.py:10-11) */
#define PEA_FLAG_RELU_AFFS 1u  /* a -> max(a, 0)     (F.relu(pred), scripts_cvppp/main.py:312, inference.py:193) */
#define PEA_FLAG_CLAMP01 8u    /* a -> clamp(a, 0, 1) (torch.clamp(affs_temp, 0.0, 1.0), loss_embedding.py:11,36) */
#define PEA_FLAG_ACCUMULATE_DE 16u /* pea_affinity_bwd_ex with a detached second operand: de += (the self loss' gradient of the same
                                    embedding is already in `de`: loss_embedding + loss_embedding_cross, main.py:306-310, one buffer);
                                    PEA_E_UNSUPPORTED where the role-A cross kernel does not apply (the caller adds two buffers),
                                    and for a self loss (e_other == NULL); ignored by the forward / inference calls */
#define PEA_FLAG_ONE_MINUS 2u  /* a -> 1 - a         (what elf's mutex_watershed is handed: scripts_cvppp/utils/seg_mutex.py:4-5) */

/* error codes: 0 = ok, negative = PEA_E_*, positive = a hipError_t from the runtime */
#define PEA_OK 0
#define PEA_E_NULL (-1)        /* a required pointer is NULL */
#define PEA_E_DESC (-2)        /* descriptor field out of range */
#define PEA_E_UNSUPPORTED (-3) /* valid but not implemented combination */
#define PEA_E_WORKSPACE (-4)   /* workspace too small / missing */
#define PEA_E_ALIGN (-5)       /* pointer not aligned to its element size */

typedef struct PeaDesc {
  int32_t abi;     /* must equal PEA_ABI_VERSION */
  int32_t ndim;    /* 2 or 3 (informational; 2 requires dims[0] == 1) */
  int32_t B;       /* batch */
  int32_t D;       /* embedding channels */
  int32_t dims[3]; /* Z, Y, X */
  int32_t K;       /* number of offsets, 1..PEA_MAX_K */
  int32_t border;  /* PEA_BORDER_* */
  int32_t dtype;   /* PEA_F32 / PEA_F16 */
  int32_t norm;    /* PEA_NORM_* */
  uint32_t flags;  /* PEA_FLAG_* */
  float eps;       /* clamp of the L2 norm: 1e-12 (F.normalize) or 1e-6 (nn.CosineSimilarity) */
  int32_t offsets[PEA_MAX_K][3]; /* o_i = (dz, dy, dx); the neighbour of p is p + o_i */
  float lambda[PEA_MAX_K];       /* per-offset loss weight (affs0_weight on the first channels) */
  /* batch strides, in elements, of target / weight / mask; 0 = dense (K*Z*Y*X).  Lets the caller pass the
   * channel slices down1[:, 0:k], [k:2k], [2k:3k] of one packed tensor (scripts_cvppp/main.py:284-287)
   * without a copy; inside one batch item the [K,Z,Y,X] block must be dense. */
  int64_t target_bstride, weight_bstride, mask_bstride;
} PeaDesc;

/* ABI version of the loaded library (== PEA_ABI_VERSION it was built with). */
int pea_version(void);

/* Static string for an error code returned by any pea_* call. */
const char *pea_strerror(int code);

/* Host-only check of a descriptor: PEA_OK or PEA_E_DESC / PEA_E_UNSUPPORTED. No GPU needed. */
int pea_desc_validate(const PeaDesc *desc);

/* The workspace of the training forward (pea_affinity_fwd / _fwd_ex / _fwd_bwd_labels): a STATE block of pea_workspace_bytes(desc)
 * bytes (the same ~16 KB for every descriptor; 8-byte aligned) into which the workgroups add their loss partials as 128-bit
 * fixed-point integers with integer atomics (csrc/pea_loss.h): the sum does not depend on the order of arrival, is exact, and
 * bit-reproducible.  Contract:
 *   - call pea_workspace_init(workspace, bytes, stream) ONCE after allocating it (it may hold several states back to back:
 *     pea_affinity_fwd_bwd_labels_dual takes two);
 *   - every call leaves the block ready for the next one (it is zero between calls), so one block can serve every later call
 *     ON THE SAME STREAM, of any descriptor; calls that may run concurrently (different streams) need a block each;
 *   - a block that was never initialised yields NaN losses (never a silently wrong number). */
size_t pea_workspace_bytes(const PeaDesc *desc);
int pea_workspace_init(void *workspace, size_t workspace_bytes, void *stream);

/* Re-read the PEA_* environment switches (they are read once, at the first call) -- fifteen A/B and debugging switches, each a default-ON
 * selector with its fallback kernels or a walk parameter (csrc/pea_host.h says what each does): PEA_FORCE_DIRECT, PEA_FWD_XDMA,
 * PEA_BWD_XDMA, PEA_FWD_WG3, PEA_BWD_PF, PEA_BOX, PEA_BOXM, PEA_H16_HW, PEA_ZMARCH, PEA_ZSEG, PEA_ZM_SUP, PEA_ZBLK_Y / _X, PEA_BWD_REV,
 * PEA_FWD_DUAL; tests that change one call this.  Memoised launch plans are dropped with it.  The switch set is replaced as a whole (no
 * half-written set is ever seen), but one entry point may read it more than once: a launch that runs CONCURRENTLY with a reload may
 * mix the two sets.  Call it between launches (tests and A/B runs do). */
void pea_reload_env(void);

/* Inference: affs[B,K,Z,Y,X] only.  e_other may be NULL. */
int pea_affinity_infer(const PeaDesc *desc, const void *e, const void *e_other, float *affs,
                       void *stream);

/* Training forward: affs (nullable), loss_out[1 + K] = { loss, L_0 .. L_{K-1} } (device, f32; L_i is the
 * un-weighted per-offset loss, i.e. the reference's all_loss list) and, when g_out != NULL,
 * g_out[B,K,Z,Y,X] = d loss / d affs = lambda_i * 2 w m (a m - t m) / N_i  (0 where the neighbour is cropped
 * away) -- the only thing the backward needs besides the embeddings.
 * Deterministic: per-workgroup partials added into `workspace` as integers (see pea_workspace_bytes). */
int pea_affinity_fwd(const PeaDesc *desc, const void *e, const void *e_other, const float *target,
                     const float *weight, const uint8_t *mask, float *affs, float *g_out, float *loss_out,
                     void *workspace, size_t workspace_bytes, void *stream);

/* Backward / vector-Jacobian product of the affinity map:
 *     de = dloss * sum_i g_i * d a_i / d e        (same dtype/layout as e)
 * g [B,K,Z,Y,X] f32 is either what pea_affinity_fwd wrote (fused WeightedMSE path) or any upstream gradient
 * d(criterion)/d(affs) of a criterion the caller applied itself (the reference passes `criterion` as an
 * argument, scripts_cvppp/main.py:188-189,284-293).  `dloss` is a DEVICE f32 scalar (autograd's grad_output)
 * or NULL (= 1), so no host sync is needed.  de_other: NULL when the second operand is detached
 * (convert_consistency_flip, scripts_cvppp/data/data_consistency.py:36), otherwise receives the gradient
 * w.r.t. e_other; de may be NULL when only de_other is wanted.  Gather form, no atomics, bit-reproducible. */
int pea_affinity_bwd(const PeaDesc *desc, const void *e, const void *e_other, const float *g, const float *dloss,
                     void *de, void *de_other, void *stream);

/* The same two calls with the 1 / norm plane of e between them (new entry points; the two above forward to these with
 * NULL).  inv_norm [B,Z,Y,X] f32 = 1 / max(|e(p)|_2, eps), NEGATED where |e(p)| < eps (the clamp branch of F.normalize,
 * whose Jacobian is I / eps): four bytes per pixel that the forward knows anyway (it normalises every pixel it stages)
 * and that lets the self-loss backward of an axis-aligned in-plane stencil (every offset along y or along x only: the
 * multi_offset(neighbor=4) tables of scripts_cvppp/utils/affinity_ours.py:4-15) run as the LDS-DMA cross kernel
 * (csrc/pea_xdma.h): the channels go through LDS two at a time and sum_i g_i ehat(q_i) needs 1 / |e(q_i)| before the
 * first chunk.  pea_affinity_bwd_ex with inv_norm == NULL, a second operand, f16 storage, D != 16 or any other stencil
 * takes the kernels of pea_affinity_bwd -- same result either way.  pea_inv_norm computes the plane alone (for callers
 * that hold e but did not run the forward: the vjp of a foreign criterion).
 * With a second operand (e_other != NULL: ema_embedding_loss, loss_embedding_mse.py:79-95) inv_norm is TWO planes,
 * [2, B, Z, Y, X]: 1 / norm of e, then of e_other.  The forward then stages e_other and reads the own pixel from e; the backward
 * with de_other == NULL (the shipped, detached case) is the role-A cross kernel: sum_i g_i(p) ehat_other(p + o_i), projected on
 * e's tangent space, written to de -- or ADDED to it with PEA_FLAG_ACCUMULATE_DE in desc->flags.  2D, D = 16, f32 only; other
 * shapes fill the two planes with separate launches and take the tiled kernels. */
int pea_affinity_fwd_ex(const PeaDesc *desc, const void *e, const void *e_other, const float *target,
                        const float *weight, const uint8_t *mask, float *affs, float *g_out, float *inv_norm_out,
                        float *loss_out, void *workspace, size_t workspace_bytes, void *stream);
int pea_affinity_bwd_ex(const PeaDesc *desc, const void *e, const void *e_other, const float *g, const float *inv_norm,
                        const float *dloss, void *de, void *de_other, void *stream);
/* pea_affinity_bwd_ex with the forward's affinity map as one more input: affs [B,K,Z,Y,X] f32 must be the RAW cosine map that
 * pea_affinity_fwd(_ex) wrote for the same descriptor with NO activation flag (PEA_FLAG_RELU_AFFS ...), unmodified.  The self-loss
 * backward then knows <ehat, G> = sum_i g_i(p) a_i(p) + g_i(p - o_i) a_i(p - o_i) before it touches a channel and finishes two
 * channels per chunk (csrc/pea_xdma_pf.h): no second read of e at D > 16, three workgroups per CU for short stencils.  affs == NULL,
 * a second operand, z offsets, D = 16 (where the kernel that keeps G is the faster one) or an activation flag in desc->flags:
 * exactly pea_affinity_bwd_ex. */
int pea_affinity_bwd_ex2(const PeaDesc *desc, const void *e, const void *e_other, const float *g, const float *inv_norm,
                         const float *affs, const float *dloss, void *de, void *de_other, void *stream);
int pea_inv_norm(const PeaDesc *desc, const void *e, float *inv_norm_out, void *stream);
/* Host-only: 1 when the LDS-DMA kernels cover the descriptor (self loss, 16-byte aligned tensors assumed) for the forward
 * (backward == 0) or the backward (backward == 1, given the 1 / norm plane): the cross kernels for axis-aligned stencils, the
 * unit-box kernels (csrc/pea_box.h) for stencils with |dz|, |dy|, |dx| <= 1 such as the 26-neighbourhood; backward == 2: the cross
 * loss with a detached second operand (forward and role-A backward, given the two planes); else 0 (the tiled / direct kernels
 * run).  A caller uses it to decide whether to allocate the 1 / norm plane.  backward == 3: 1 when pea_affinity_bwd_ex2 READS the
 * raw affinity map for this descriptor (the projection-first kernels at D > 16, csrc/pea_xdma_pf.h; the z-march backward of 3D
 * volumes, csrc/pea_zmarch.h): a caller that hands `affs` over must then keep the forward's map unmodified until the backward.
 * backward == 4: the same question for the cross loss with a detached second operand (e_other != NULL, de_other == NULL): 1 at
 * D = 32 / 64, f32, 2D -- there the role-A backward is the projection-first kernel k_bwd_xdma_pfo (csrc/pea_xdma_pf.h) and runs only
 * when `affs` (the cross loss' raw map) comes along; without it the tiled kernels run.  backward == 5: 1 when
 * pea_affinity_fwd_dual_ex runs the descriptor's self + cross forward pair as one launch. */
int pea_cross_supported(const PeaDesc *desc, int backward);

/* The FORWARD of the full-resolution pair of the 2D training loops in one launch: embedding_loss(e, target, weight, mask) and
 * ema_embedding_loss(e, ema, target, weight, mask) on the SAME target / weight / mask (scripts_cvppp/main.py:284 and :293,
 * scripts_bbbc039v1/main.py likewise; scripts_cvppp/loss/loss_embedding_mse.py:18-47, 79-95).  Exactly
 *     pea_affinity_fwd_ex(desc,       e, NULL, target, weight, mask, affs, g_out,       inv_norm_out,          loss_out,       workspace, ..)
 *     pea_affinity_fwd_ex(desc_cross, e, ema,  target, weight, mask, NULL, g_cross_out, {own, inv_norm_other}, loss_cross_out, workspace_cross, ..)
 * -- every output bit-identical to those two calls -- but e, target, weight and mask are read once (csrc/pea_xdma_dual.h: 338
 * instead of 532 bytes per pixel at D = 16, K = 10).  desc_cross may differ from desc in lambda (the cross loss' affs0_weight) and in
 * the activation flags (the cross loss' map is not written); everything else must agree (PEA_E_DESC).  inv_norm_out / inv_norm_other_out:
 * [B,Z,Y,X] each, the planes pea_affinity_bwd_dual_ex takes.  workspace / workspace_cross: two DIFFERENT state blocks
 * (pea_workspace_init).  affs may be NULL.  Returns PEA_E_UNSUPPORTED -- before anything is launched -- where no fused kernel covers
 * the descriptor (pea_cross_supported(desc, 5) == 0: anything but 2D, D = 16, f32, axis-aligned stencil with K <= 10; PEA_FWD_DUAL=0)
 * or ema aliases e: the caller then makes the two calls above. */
int pea_affinity_fwd_dual_ex(const PeaDesc *desc, const PeaDesc *desc_cross, const void *e, const void *ema, const float *target,
                             const float *weight, const uint8_t *mask, float *affs, float *g_out, float *g_cross_out,
                             float *inv_norm_out, float *inv_norm_other_out, float *loss_out, float *loss_cross_out, void *workspace,
                             void *workspace_cross, size_t workspace_bytes, void *stream);

/* The backward of the full-resolution pair of the training loop in one launch: g is what pea_affinity_fwd wrote for the self
 * loss (e_other = NULL), g_cross what it wrote for the detached-EMA cross loss of the same e (e_other = ema); the stencil
 * and geometry are desc's (lambda and the normaliser are already inside g / g_cross).
 *     de = dloss * sum_i g_i d a_i / d e  +  dloss_cross * sum_i g_cross_i d a^ema_i / d e
 * Returns PEA_E_UNSUPPORTED when two pea_affinity_bwd calls (and an add) must be used instead. */
int pea_affinity_bwd_dual(const PeaDesc *desc, const void *e, const void *ema, const float *g, const float *g_cross,
                          const float *dloss, const float *dloss_cross, void *de, void *stream);
/* The same with the 1 / norm planes of the two operands ([B,Z,Y,X] each: the plane pea_affinity_fwd_ex wrote for the self loss,
 * and the SECOND plane of the pair it wrote for the cross loss): 2D, D = 16, f32, axis-aligned stencils then run as ONE launch of
 * the LDS-DMA cross kernel with a second phase (the second operand's one-sided cross, role-A pairs added into the same registers);
 * NULL planes or any other shape: PEA_E_UNSUPPORTED (two pea_affinity_bwd_ex calls and an add). */
int pea_affinity_bwd_dual_ex(const PeaDesc *desc, const void *e, const void *ema, const float *g, const float *g_cross,
                             const float *inv_norm, const float *inv_norm_other, const float *dloss, const float *dloss_cross,
                             void *de, void *stream);

/* buf[0..n) *= scale[0] in place (dtype PEA_F32 / PEA_F16; f32 buffers 16-byte aligned).  `scale` is a DEVICE scalar
 * (autograd's grad_output): the kernel reads it and returns without touching buf when it is exactly 1, which is
 * what a plain loss.backward() hands to the gradient pea_affinity_fwd_bwd_labels produced for dloss = 1. */
int pea_scale_inplace(void *buf, int dtype, size_t n, const float *scale, void *stream);
/* the same for up to 8 buffers (host arrays of device pointers / element counts) in one launch: the gradients of one loss
 * section share their grad_output */
int pea_scale_inplace_multi(void *const *bufs, const size_t *counts, int nbuf, int dtype, const float *scale, void *stream);
/* out[0] = sum_{j < n} w[j] * rows[j * stride] (device pointers; n <= 64; fixed order of additions): the total of a loss section --
 * `loss = (sum of the deep-supervision losses + loss_embedding) * self_emb + loss_embedding_cross * cross_emb`,
 * scripts_cvppp/main.py:295-306 -- from the loss_out rows of its calls in ONE small launch instead of a multiply and a reduction. */
int pea_weighted_sum(const float *rows, int stride, const float *w, int n, float *out, void *stream);

/* Caller epilogue of the 3D path, in place on affs [B,K,Z,Y,X] (scripts_ac3ac4/main.py:233-237, 296-300;
 * scripts_ac3ac4/inference.py:160-164): pred[:,0,:s] = pred[:,0,s:2s] (z), pred[:,1,:,:s] = pred[:,1,:,s:2s] (y),
 * pred[:,2,:,:,:s] = pred[:,2,:,:,s:2s] (x) for s = shift (0 = skip), then F.relu when relu != 0.  Also the plain
 * F.relu(pred) of the 2D callers (scripts_cvppp/main.py:312) with shift = 0. */
int pea_fill_border_relu(float *affs, int B, int K, int Z, int Y, int X, int shift, int relu, void *stream);

/* ---- the step feeding the path: label image -> target / mask / class-balance weight ----
 * Replaces, per batch, gen_affs_ours(labels, offsets, ignore=False, padding=...) and the per-channel
 * weight_binary_ratio(lb_affs[i]) of the reference's data providers (scripts_cvppp/utils/affinity_ours.py:17-39,
 * scripts_cvppp/data/data_segmentation.py:205-228, called at scripts_cvppp/data/data_provider.py:204-225).
 *   labels [B,Z,Y,X] int32 (0 = background);  uses desc->B, dims, K, offsets only
 *   target [B,K,Z,Y,X] f32 = 1 iff label(p) == label(p + o_i)  (PEA_TGT_BOTH_FOREGROUND: and both > 0, seg_to_aff,
 *          scripts_ac3ac4/data/data_affinity.py:53-102); neighbour outside: PEA_TGT_PADDING ? 1 : 0
 *   mask   [B,K,Z,Y,X] u8  = 1 iff p + o_i lies inside the image (nullable)
 *   weight [B,K,Z,Y,X] f32 = class balance per (b, channel) (nullable)
 * workspace: pea_targets_workspace_bytes(desc) bytes (integer counts; zeroed by the call). */
#define PEA_TGT_PADDING 1u
#define PEA_TGT_BOTH_FOREGROUND 2u
#define PEA_TGT_MASK_INSIDE 4u /* labels-in training step only: mask = [neighbour inside] (2D path); without it mask == 1 (3D path) */
#define PEA_TGT_ACCUMULATE 8u  /* labels-in training step only: de += result (e.g. the EMA cross loss on top of the self loss) */
size_t pea_targets_workspace_bytes(const PeaDesc *desc);
int pea_gen_targets(const PeaDesc *desc, const int32_t *labels, unsigned flags, float *target, uint8_t *mask,
                    float *weight, void *workspace, size_t workspace_bytes, void *stream);

/* ---- the training step from labels: no target / weight / mask tensors at all (SURVEY.md section 8f, f2 fused) ----
 * pea_label_weights: wtab [B,K,2] f32 = { weight of target-1 pixels, weight of target-0 pixels } per (image, channel), i.e.
 * weight_binary_ratio(lb_affs[i]) (scripts_cvppp/data/data_segmentation.py:205-228) as two scalars; integer counts in
 * `workspace` (pea_targets_workspace_bytes).  pea_affinity_fwd_bwd_labels: the outputs of pea_affinity_fwd (affs, nullable, and
 * loss_out[1 + K]) AND de = dloss * d loss / d e in ONE launch -- no g round trip through HBM; e_other = NULL (self loss) or
 * the DETACHED second operand; dloss = device scalar or NULL (= 1: scale later with pea_scale_inplace) -- with
 *   target_i(q) = [label(q) == label(q + o_i)] (flags as pea_gen_targets), mask_i(q) = [q + o_i inside] with
 *   PEA_TGT_MASK_INSIDE else 1, weight_i(q) = target ? wtab[b][i][0] : wtab[b][i][1]
 * evaluated inside the kernel: results equal (to rounding) to pea_gen_targets + pea_affinity_fwd + pea_affinity_bwd.
 * Returns PEA_E_UNSUPPORTED when no fused kernel covers the descriptor (then use those three).
 * Label ids: any int32 but INT32_MIN (-2^31), which the LDS-staged kernels use as their outside-the-image marker (a label image that
 * holds it gets it compared as "outside"; the Python layer's range check rejects it). */
int pea_label_weights(const PeaDesc *desc, const int32_t *labels, unsigned flags, float *wtab, void *workspace,
                      size_t workspace_bytes, void *stream);
int pea_affinity_fwd_bwd_labels(const PeaDesc *desc, const void *e, const void *e_other, const int32_t *labels,
                                const float *wtab, unsigned flags, float *affs, float *loss_out, const float *dloss, void *de,
                                void *workspace, size_t workspace_bytes, void *stream);
/* The same with a SCRATCH buffer lent by the caller (pea_labels_scratch_bytes(desc) bytes, 16-byte aligned, contents undefined
 * before and after; 0 = this descriptor has no use for it): where the LDS-DMA cross kernels cover the descriptor (self loss, 2D,
 * f32, D = 16 / 32, axis-aligned stencil of at most 10 offsets) the step then runs as TWO launches on them -- a labels-in forward
 * that writes g and the 1 / norm plane into the scratch (no target / weight / mask traffic: 4D + 8K + 8 bytes per pixel) and the
 * cross backward -- instead of the one-launch box kernel: 70 + 94 us against 202 us at B=8 x 16 x 544^2.  Same results (to
 * rounding).  scratch == NULL, a second operand, PEA_TGT_ACCUMULATE, or any other descriptor: exactly pea_affinity_fwd_bwd_labels. */
size_t pea_labels_scratch_bytes(const PeaDesc *desc);
int pea_affinity_fwd_bwd_labels_ex(const PeaDesc *desc, const void *e, const void *e_other, const int32_t *labels,
                                   const float *wtab, unsigned flags, float *affs, float *loss_out, const float *dloss, void *de,
                                   void *workspace, size_t workspace_bytes, void *scratch, size_t scratch_bytes, void *stream);

/* The full-resolution pair of the training loop in ONE launch: the self loss (desc) and the detached-EMA cross loss
 * (desc_cross: same geometry and stencil, its own lambda / normaliser) of the same embedding and labels
 * (scripts_cvppp/main.py:288,293): de = dloss * d L_self / d e + dloss_cross * d L_cross / d e; affs (nullable) is the
 * self loss' map; loss_out / loss_cross_out [1 + K] each.  workspace: 2 x pea_workspace_bytes(desc), both states initialised.  Returns
 * PEA_E_UNSUPPORTED when the two launches of pea_affinity_fwd_bwd_labels (the second with PEA_TGT_ACCUMULATE) must be
 * used instead. */
int pea_affinity_fwd_bwd_labels_dual(const PeaDesc *desc, const PeaDesc *desc_cross, const void *e, const void *ema,
                                     const int32_t *labels, const float *wtab, unsigned flags, float *affs, float *loss_out,
                                     float *loss_cross_out, const float *dloss, const float *dloss_cross, void *de,
                                     void *workspace, size_t workspace_bytes, void *stream);

/* ---- the step before the path: the embedding head ----
 * OutConv = nn.Conv2d(C, D, 1) of scripts_cvppp/model/unet2d_residual.py:67-74 (outconv_emb :307, applied :346; the same
 * class in scripts_bbbc039v1/model/unet2d_residual.py:67,235) and the 1x1x1 conv3dBlock heads out_put* of
 * scripts_ac3ac4/model/model_superhuman.py:437-441 (applied :486-490):
 *     e[b,d,p] = bias[d] + sum_c W[d,c] x[b,c,p]          x f32 [B,C,S] , W f32 [D,C] , bias f32 [D] or NULL , e f32 [B,D,S]
 * with S = H*W or Z*Y*X contiguous pixels per channel plane.  The backward takes de = d loss / d e (what pea_affinity_bwd
 * wrote) and returns dx [B,C,S] (nullable), dW [D,C] and db [D] (nullable); the sums over pixels run on the matrix cores in
 * exact f32, per-workgroup partials in `workspace` (pea_head_workspace_bytes), reduced in a fixed order.
 * Supported (C, D): every head of the reference's models -- (28|32|36|48|64|80|128|256, 16) and (32|64|128|256, 32);
 * anything else returns PEA_E_UNSUPPORTED. */
size_t pea_head_workspace_bytes(int C, int D);
int pea_head_fwd(int B, int C, int D, size_t S, const float *x, const float *W, const float *bias, float *e, void *stream);
int pea_head_bwd(int B, int C, int D, size_t S, const float *x, const float *W, const float *de, float *dx, float *dW,
                 float *db, void *workspace, size_t workspace_bytes, void *stream);

/* ---- the step after the path: 3D inference stitcher ----
 * Provider_valid.add_vol / get_results of scripts_ac3ac4/data/provider_valid.py:320-349 on the device, so a predicted
 * window never leaves HBM: out_affs [C,Z,Y,X] and weight_map [Z,Y,X] accumulate affs_vol [C,oz,oy,ox] * weight_vol
 * [oz,oy,ox] (the Gaussian blend weights of get_weight, :305-318) at (z0,y0,x0); finalize divides in place.  Separate
 * f32 multiply / add / divide (no FMA): bit-identical to the numpy statements for the same add order. */
int pea_stitch_add(float *out_affs, float *weight_map, const float *affs_vol, const float *weight_vol, int C, int Z, int Y,
                   int X, int oz, int oy, int ox, int z0, int y0, int x0, void *stream);
int pea_stitch_finalize(float *out_affs, const float *weight_map, int C, size_t voxels, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PEA_H_ */
