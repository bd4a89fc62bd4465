"""The loss section of the reference's training loops, call for call.

The reference's drivers are out of scope, but the ORDER and SLICING of the hot-path calls inside them is what a
drop-in has to reproduce; these two functions are that section with the reference's variable names, so a maintainer
can replace scripts_cvppp/main.py:282-312 / scripts_ac3ac4/main.py:216-238 by one call.  No host synchronisation
happens inside (the reference's K `.item()` calls per loss are gone), so the section can be captured in a HIP graph
(tests/test_gpu_parity.py::test_loss_section_graph_replay).
"""
from ..loss.loss_embedding_mse import (ema_embedding_loss, ema_embedding_loss_from_labels, embedding_loss,
                                       embedding_loss_from_labels)
from ..loss.loss_embedding_mse_3d import (ema_embedding_loss_norm1, ema_embedding_loss_norm5, embedding_loss_norm1,
                                          embedding_loss_norm5)
from ..utils.postproc import fill_border_relu_, relu_


def deep_weight_factor(deep_weight):
    """scripts_cvppp/main.py:214-219"""
    if deep_weight == 1:
        return [1.0, 1.0, 1.0, 1.0, 1.0]
    if deep_weight == 2:
        return [0.01, 0.03, 0.1, 0.3, 1.0]
    return [deep_weight, 1.0, 1.0, 1.0, 1.0]


def cvppp_loss_section(embedding, emds, ema_embedding, target, weightmap, affs_mask, downs, criterion, offsets, nb_half,
                       affs0_weight=1, dis_mode='ours', deep_weight=1, self_emb=1.0, cross_emb=1.0):
    """scripts_cvppp/main.py:284-310 (and scripts_bbbc/main.py:279-305): five self losses over the deep-supervision
    scales + the EMA cross loss at full resolution.

    emds = (emd1, emd2, emd3, emd4); downs = (down1, down2, down3, down4), each packed [B, 3k, h, w] =
    (target | weight | mask) thirds with k = nb_half * (4, 3, 2, 1) channels.  ema_embedding is the flipped-back,
    detached EMA output (convert_consistency_flip).  Returns (loss without the consistency term `loss_mask`, pred,
    parts) where pred is the full-resolution affinity map BEFORE relu (call relu_ after backward like the
    reference does at :312) and parts the individual weighted losses (device scalars)."""
    dwf = deep_weight_factor(deep_weight)
    losses = []
    for j, (emd, down) in enumerate(zip(emds, downs)):
        k = nb_half * (4 - j)
        l, _, _ = embedding_loss(emd, down[:, 0:k], down[:, k:2 * k], down[:, 2 * k:3 * k], criterion, offsets[:k],
                                 affs0_weight=affs0_weight, mode=dis_mode)
        losses.append(l)
    loss_embedding, pred, _ = embedding_loss(embedding, target, weightmap, affs_mask, criterion, offsets,
                                             affs0_weight=affs0_weight, mode=dis_mode)
    loss_embedding_cross, _ = ema_embedding_loss(embedding, ema_embedding, target, weightmap, affs_mask, criterion, offsets,
                                                 affs0_weight=affs0_weight, mode=dis_mode)
    loss_embedding = loss_embedding * dwf[0]
    loss_emd = [losses[j] * dwf[j + 1] for j in range(4)]
    loss_embedding_cross = loss_embedding_cross * dwf[0]
    loss_embedding_total = (loss_emd[0] + loss_emd[1] + loss_emd[2] + loss_emd[3] + loss_embedding) * self_emb
    loss_embedding_cross_total = loss_embedding_cross * cross_emb
    loss = loss_embedding_total + loss_embedding_cross_total
    parts = {"loss_embedding": loss_embedding, "loss_emd": loss_emd, "loss_embedding_cross": loss_embedding_cross}
    return loss, pred, parts


def ac3ac4_loss_section(embedding, emds, ema_embedding, target, weightmap, downs, criterion, embedding_mode=5, affs0_weight=1):
    """scripts_ac3ac4/main.py:219-231: full-resolution self + EMA cross loss (norm1 or norm5) and four norm1 losses on
    the deep-supervision heads; downs = (down1, .., down4) packed [B, 6, z, y, x] = (target[:3] | weight[3:]),
    paired emd1<->down4 .. emd4<->down1 as in the reference.  Returns (loss, pred before the border fill / relu);
    call finish_pred_3d_(pred) after backward (:233-237)."""
    if embedding_mode == 1:
        loss_embedding, pred = embedding_loss_norm1(embedding, target, weightmap, criterion, affs0_weight=affs0_weight)
        loss_embedding_cross, _ = ema_embedding_loss_norm1(embedding, ema_embedding, target, weightmap, criterion,
                                                           affs0_weight=affs0_weight)
    elif embedding_mode == 5:
        loss_embedding, pred = embedding_loss_norm5(embedding, target, weightmap, criterion, affs0_weight=affs0_weight)
        loss_embedding_cross, _ = ema_embedding_loss_norm5(embedding, ema_embedding, target, weightmap, criterion,
                                                           affs0_weight=affs0_weight)
    else:
        raise NotImplementedError
    loss = loss_embedding + loss_embedding_cross
    for emd, down in zip(emds, downs[::-1]):
        l, _ = embedding_loss_norm1(emd, down[:, :3], down[:, 3:], criterion, affs0_weight=affs0_weight)
        loss = loss + l
    return loss, pred


def finish_pred_3d_(pred, shift=1):
    """scripts_ac3ac4/main.py:233-237: border fill of the three shift-1 channels, then relu; in place"""
    return fill_border_relu_(pred, shift=shift, relu=True)


def finish_pred_2d_(pred):
    """scripts_cvppp/main.py:312"""
    return relu_(pred)


def cvppp_loss_section_from_labels(embedding, emds, ema_embedding, labels, label_downs, criterion, offsets, nb_half,
                                   affs0_weight=1, dis_mode='ours', deep_weight=1, self_emb=1.0, cross_emb=1.0):
    """cvppp_loss_section without any target / weight / mask tensor: `labels` [B,H,W] and `label_downs` = the four
    nearest-downsampled label images (scripts_cvppp/data/data_provider.py:199-208) replace target, weightmap, affs_mask
    and down1..down4; every loss is one labels-in launch (gen_affs_ours(padding=True) + weight_binary_ratio evaluated
    inside the kernel)."""
    dwf = deep_weight_factor(deep_weight)
    losses = []
    for j, (emd, lab) in enumerate(zip(emds, label_downs)):
        k = nb_half * (4 - j)
        l, _, _ = embedding_loss_from_labels(emd, lab, criterion, offsets[:k], affs0_weight=affs0_weight, mode=dis_mode,
                                             need_affs=False)  # the reference discards these maps (main.py:284-287)
        losses.append(l)
    loss_embedding, pred, _ = embedding_loss_from_labels(embedding, labels, criterion, offsets, affs0_weight=affs0_weight, mode=dis_mode)
    loss_embedding_cross, _ = ema_embedding_loss_from_labels(embedding, ema_embedding, labels, criterion, offsets,
                                                             affs0_weight=affs0_weight, mode=dis_mode, need_affs=False)
    loss_embedding = loss_embedding * dwf[0]
    loss_emd = [losses[j] * dwf[j + 1] for j in range(4)]
    loss_embedding_cross = loss_embedding_cross * dwf[0]
    loss = (loss_emd[0] + loss_emd[1] + loss_emd[2] + loss_emd[3] + loss_embedding) * self_emb + loss_embedding_cross * cross_emb
    return loss, pred, {"loss_embedding": loss_embedding, "loss_emd": loss_emd, "loss_embedding_cross": loss_embedding_cross}
