"""3D embedding -> affinity losses: drop-in for the reference's scripts_ac3ac4/loss/loss_embedding_mse.py
(norm1, norm5 and their ema_/inf_ variants -- the ones the shipped ac3ac4.yaml selects, embedding_mode 5
plus norm1 for the four deep-supervision heads, scripts_ac3ac4/main.py:219-230).

Semantics kept from the reference (file:line there):
  * channel i compares p with p - shift along axis i % 3 of (z,y,x) on the CROPPED extent (:11-18,:143-151);
    the first `shift` slices of that axis stay 0 in `affs` (:22-25,:175-190); no mask tensor
  * WeightedMSE normaliser = B * cropped volume of that channel (pred is [B,1,Z',Y',X'], loss.py:113-115)
  * affs0_weight multiplies loss0 only in norm1 (:20) and channels 0..2 in norm5 (:181-184)
  * norm5's shift table is the literal [1,1,1,2,3,3,3,9,9,4,27,27] (:176); its `shift`/`fill` arguments are unused
  * ema_*: the EMA embedding is the shifted-from operand (:35,:241)
"""

from .. import _lib
from ..affinity_op import AffinityMap, AffinitySpec, FusedAffinityMSE, LabelsAffinityMSE, LabelsStepUnsupported, affinity_infer
from ..utils.affinity_ours import NORM5_SHIFTS, axis_offsets_3d


def _spec(shifts, affs0_weight, first):
    lam = [float(affs0_weight) if i < first else 1.0 for i in range(len(shifts))]
    return AffinitySpec(3, axis_offsets_3d(shifts), lam, _lib.BORDER_CROP_ZERO, _lib.NORM_CROPPED, 1e-12)


def _foreign(embedding, ema_embedding, target, weightmap, criterion, shifts, spec):
    affs = AffinityMap.apply(embedding, ema_embedding, spec)
    loss = 0
    for i, s in enumerate(shifts):
        ax = 2 + i % 3
        n = affs.shape[ax] - s
        li = criterion(affs[:, i:i + 1].narrow(ax, s, n), target[:, i:i + 1].narrow(ax, s, n),
                       weightmap[:, i:i + 1].narrow(ax, s, n))
        loss = loss + li * spec.lam[i]
    return loss, affs.detach()


def _run(embedding, ema_embedding, target, weightmap, criterion, shifts, affs0_weight, first):
    spec = _spec(shifts, affs0_weight, first)
    if getattr(criterion, 'pea_fused', False):
        loss, affs, _ = FusedAffinityMSE.apply(embedding, ema_embedding, target, weightmap, None, spec)
        return loss, affs
    return _foreign(embedding, ema_embedding, target, weightmap, criterion, shifts, spec)


def embedding_loss_norm1(embedding, target, weightmap, criterion, affs0_weight=1, shift=1, fill=True):
    """-> (loss, affs [B,3,Z,Y,X]) -- reference :7-27"""
    return _run(embedding, None, target, weightmap, criterion, [shift] * 3, affs0_weight, 1)


def ema_embedding_loss_norm1(embedding, ema_embedding, target, weightmap, criterion, affs0_weight=1, shift=1, fill=True):
    """reference :30-51"""
    return _run(embedding, ema_embedding, target, weightmap, criterion, [shift] * 3, affs0_weight, 1)


def inf_embedding_loss_norm1(embedding, shift=1):
    """-> affs [B,3,Z,Y,X] -- reference :54-67"""
    return affinity_infer(embedding, None, _spec([shift] * 3, 1, 1))


def embedding_loss_norm5(embedding, target, weightmap, criterion, affs0_weight=1, shift=1, fill=True):
    """-> (loss, affs [B,12,Z,Y,X]) -- reference :169-194"""
    return _run(embedding, None, target, weightmap, criterion, NORM5_SHIFTS, affs0_weight, 3)


def ema_embedding_loss_norm5(embedding, ema_embedding, target, weightmap, criterion, affs0_weight=1, shift=1, fill=True):
    """reference :263-289"""
    return _run(embedding, ema_embedding, target, weightmap, criterion, NORM5_SHIFTS, affs0_weight, 3)


def inf_embedding_loss_norm5(embedding):
    """-> affs [B,12,Z,Y,X] -- reference :212-234"""
    return affinity_infer(embedding, None, _spec(NORM5_SHIFTS, 1, 3))


# ---- norm6: generic offsets with a replicate border (reference :346-366; not selected by any shipped yaml) ----
def _spec6(offsets):
    return AffinitySpec(3, offsets, None, _lib.BORDER_REPLICATE, _lib.NORM_FULL, 1e-12)


def _run6(embedding, ema_embedding, target, weightmap, criterion, offsets):
    spec = _spec6(offsets)
    if getattr(criterion, 'pea_fused', False):
        # ONE criterion call over the [B,K,Z,Y,X] map: WeightedMSE's normaliser is B*Z*Y*X (loss.py:113-115)
        loss, affs, _ = FusedAffinityMSE.apply(embedding, ema_embedding, target, weightmap, None, spec)
        return loss, affs
    affs = AffinityMap.apply(embedding, ema_embedding, spec)
    return criterion(affs, target, weightmap), affs


def embedding_loss_norm6(embedding, target, weightmap, criterion, affs0_weight=1, shift=1, fill=True):
    """-> (loss, affs [B,K,Z,Y,X]); `shift` is the LIST of 3D offsets (the reference's naming, :348); a_i(p) =
    <ehat(p), ehat(clamp(p + o_i))> (shift_tensor with replication padding, :294-344)"""
    return _run6(embedding, None, target, weightmap, criterion, shift)


def ema_embedding_loss_norm6(embedding, ema_embedding, target, weightmap, criterion, affs0_weight=1, shift=1, fill=True):
    """reference :356-366: the EMA embedding is the shifted operand"""
    return _run6(embedding, ema_embedding, target, weightmap, criterion, shift)


# ---- the same losses straight from the segmentation (no target / weightmap tensors) ----
# targets as the 12-channel provider builds them: seg_to_aff(lb, nhood, pad='') (scripts_ac3ac4/data/data_affinity.py:53-102,
# data_provider_labeled_deep.py:253-256): 1 iff both voxels carry the same label > 0, 0 in the border slices; weights =
# weight_binary_ratio per channel over the whole volume.  The cropped-away border pairs carry no loss either way.
_FLAGS_3D = _lib.TGT_BOTH_FOREGROUND


def _run_labels(embedding, ema_embedding, labels, criterion, shifts, affs0_weight, first):
    if not getattr(criterion, 'pea_fused', False):
        raise NotImplementedError("the labels-in step fuses WeightedMSE; for another criterion use gen_targets + the tensor API")
    if ema_embedding is not None and ema_embedding.requires_grad:
        raise NotImplementedError("a second operand that needs its own gradient takes the tensor API")
    spec = _spec(shifts, affs0_weight, first)
    try:
        loss, affs, _ = LabelsAffinityMSE.apply(embedding, ema_embedding, labels, spec, _FLAGS_3D)
    except LabelsStepUnsupported:
        from ..utils.targets import gen_targets
        t, _, w = gen_targets(labels, axis_offsets_3d(shifts), padding=False, both_foreground=True, want_mask=False)
        loss, affs, _ = FusedAffinityMSE.apply(embedding, ema_embedding, t, w, None, spec)
    return loss, affs


def embedding_loss_norm1_from_labels(embedding, labels, criterion, affs0_weight=1, shift=1):
    return _run_labels(embedding, None, labels, criterion, [shift] * 3, affs0_weight, 1)


def embedding_loss_norm5_from_labels(embedding, labels, criterion, affs0_weight=1):
    return _run_labels(embedding, None, labels, criterion, NORM5_SHIFTS, affs0_weight, 3)


def ema_embedding_loss_norm5_from_labels(embedding, ema_embedding, labels, criterion, affs0_weight=1):
    return _run_labels(embedding, ema_embedding, labels, criterion, NORM5_SHIFTS, affs0_weight, 3)
