"""2D embedding -> affinity losses: drop-in for the reference's scripts_cvppp/loss/loss_embedding_mse.py
(byte-identical copy in scripts_bbbc039v1/loss/): same function names, argument order, defaults and
return values; the body is one fused HIP forward launch and one backward launch instead of a Python
loop of K torch.roll / mul / sum / criterion / .item() steps.

Semantics kept from the reference (file:line there):
  * circular wrap of the stencil (torch.roll, :8,:50,:69); wrapped values are returned un-masked in `affs` (:46)
  * `affs0_weight` is accepted but NOT applied by embedding_loss (:26-39); ema_embedding_loss applies it
    to the first two offsets (:90-93)
  * mode != 'ours' selects nn.CosineSimilarity(dim=1, eps=1e-6) on the un-normalised embedding (:11-13,:19-20),
    i.e. the same cosine with the norm clamp at 1e-6 instead of F.normalize's 1e-12
  * the criterion is a parameter: with this package's WeightedMSE the loss is fused into the kernel
    (normaliser B*W, loss.py:113-115); any other callable gets `criterion(affs*mask, target*mask, weightmap)`
    per offset on a differentiable affinity map, exactly like the reference
  * `all_loss` is a list of K floats; here it is filled lazily from a device tensor (no K host syncs)
  * `mask` is the 0 / 1 map gen_affs_ours produces (any dtype; the reference multiplies by mask.float(), :21): the fused path
    stores it as one byte per pixel, so values other than 0 and 1 are not supported there (pass a foreign criterion for those)
"""
import torch

from .. import _lib
from ..affinity_op import (AffinityMap, activation_flags, AffinitySpec, FusedAffinityMSE, LabelsAffinityMSE, LabelsStepUnsupported, LossList,
                           affinity_infer)


def _eps(mode):
    return 1e-12 if mode == 'ours' else 1e-6


def _spec(offsets, lam, mode, relu=False, act=0):
    return AffinitySpec(2, offsets, lam, _lib.BORDER_CIRCULAR, _lib.NORM_BX, _eps(mode), relu, act)


def _fused(criterion):
    return getattr(criterion, 'pea_fused', False)


def _foreign_criterion(embedding, ema_embedding, target, weightmap, mask, criterion, offsets, lam, mode):
    affs = AffinityMap.apply(embedding, ema_embedding, _spec(offsets, None, mode))
    mask = mask.float()
    loss = torch.zeros((), dtype=affs.dtype, device=affs.device)
    parts = []
    for i in range(len(offsets)):
        li = criterion(affs[:, i] * mask[:, i], target[:, i] * mask[:, i], weightmap[:, i])
        loss = loss + li * lam[i]
        parts.append(li.detach())
    return loss, affs.detach(), torch.stack(parts)


def embedding_loss(embedding, target, weightmap, mask, criterion, offsets, affs0_weight=1, mode='ours'):
    """-> (loss, affs [B,K,H,W], all_loss list[K]) -- reference :18-47"""
    lam = [1.0] * len(offsets)  # the reference computes affs0_weight_factor but never applies it
    if _fused(criterion):
        loss, affs, parts = FusedAffinityMSE.apply(embedding, None, target, weightmap, mask, _spec(offsets, lam, mode))
    else:
        loss, affs, parts = _foreign_criterion(embedding, None, target, weightmap, mask, criterion, offsets, lam, mode)
    return loss, affs, LossList(parts)


def embedding2affs(embedding, offsets, mode='ours', activation=None):
    """-> affs [B,K,H,W] -- reference :58-66.  activation (beyond the reference's signature): the statement the caller applies
    to the map next, fused into its store -- 'relu' (F.relu(pred), inference.py:193), 'mutex' (1 - relu(a): the map
    elf.mutex_watershed is handed, utils/seg_mutex.py:4-5), 'half' ((a + 1) / 2 = the L2 affinity 1 - d^2 / 4 of unit vectors),
    'half_clamp' (clamp((a + 1) / 2, 0, 1): the embedding2affs of scripts_cvppp/loss/loss_embedding.py:33-46, with mode='cos')."""
    return affinity_infer(embedding, None, _spec(offsets, None, mode, act=activation_flags(activation)))


def ema_embedding_loss(embedding, ema_embedding, target, weightmap, mask, criterion, offsets, affs0_weight=1, mode='ours'):
    """-> (loss, affs) with a_i(p) = <ehat(p), ehat_ema(p+o_i)> -- reference :79-95.
    Gradients flow into `ema_embedding` only if it requires grad (the shipped configs detach it,
    scripts_cvppp/data/data_consistency.py:36)."""
    lam = [float(affs0_weight) if i < 2 else 1.0 for i in range(len(offsets))]
    if _fused(criterion):
        loss, affs, _ = FusedAffinityMSE.apply(embedding, ema_embedding, target, weightmap, mask, _spec(offsets, lam, mode))
    else:
        loss, affs, _ = _foreign_criterion(embedding, ema_embedding, target, weightmap, mask, criterion, offsets, lam, mode)
    return loss, affs


# ---- the same losses straight from the label image (no target / weightmap / mask tensors) ----
_FLAGS_2D = _lib.TGT_PADDING | _lib.TGT_MASK_INSIDE  # gen_affs_ours(ignore=False, padding=True) + its mask, the shipped provider


def _from_labels(embedding, ema_embedding, labels, criterion, offsets, lam, mode, need_affs=True):
    if not _fused(criterion):
        raise NotImplementedError("the labels-in step fuses WeightedMSE; for another criterion use gen_targets + embedding_loss")
    try:
        return LabelsAffinityMSE.apply(embedding, ema_embedding, labels, _spec(offsets, lam, mode), _FLAGS_2D, need_affs)
    except LabelsStepUnsupported:  # e.g. the coarsest deep-supervision scales: targets on the GPU, then the tensor path
        from ..utils.targets import gen_targets
        t, m, w = gen_targets(labels, offsets, padding=True)
        return FusedAffinityMSE.apply(embedding, ema_embedding, t, w, m, _spec(offsets, lam, mode))


def embedding_loss_from_labels(embedding, labels, criterion, offsets, affs0_weight=1, mode='ours', need_affs=True):
    """embedding_loss(embedding, *gen_targets(labels, offsets, padding=True), criterion, offsets) without the three
    [B,K,H,W] tensors: -> (loss, affs, all_loss).  labels: int tensor [B,H,W] on the GPU.  need_affs=False skips the
    affinity-map output (an empty tensor is returned) for the calls whose map the training loop throws away."""
    loss, affs, parts = _from_labels(embedding, None, labels, criterion, offsets, [1.0] * len(offsets), mode, need_affs)
    return loss, affs, LossList(parts)


def ema_embedding_loss_from_labels(embedding, ema_embedding, labels, criterion, offsets, affs0_weight=1, mode='ours', need_affs=True):
    """ema_embedding_loss from the label image (the EMA operand must be detached, as the shipped configs have it)"""
    if ema_embedding.requires_grad:
        raise NotImplementedError("a second operand that needs its own gradient takes gen_targets + ema_embedding_loss")
    lam = [float(affs0_weight) if i < 2 else 1.0 for i in range(len(offsets))]
    loss, affs, _ = _from_labels(embedding, ema_embedding, labels, criterion, offsets, lam, mode, need_affs)
    return loss, affs
