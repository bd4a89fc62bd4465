"""Label image -> (target, mask, weight) on the GPU: the reference's per-sample numpy / scipy target pipeline
(scripts_cvppp/data/data_provider.py:204-225) as two launches per batch, so only the uint8/int32 label image has
to cross PCIe instead of ~40 MB of f32 targets per 544^2 sample.

  gen_affs_ours(labels, offsets, ignore=False, padding=False)   scripts_cvppp/utils/affinity_ours.py:17-39
  weight_binary_ratio(label)                                    scripts_cvppp/data/data_segmentation.py:205-228
Device tensors only (no CPU fallback); the CPU restatement that checks this lives in oracle/ (tests only).
"""
import ctypes

import torch

from .. import _lib
from ..affinity_op import _labels_int32, AffinitySpec, make_desc


def gen_targets(labels, offsets, padding=True, both_foreground=False, want_mask=True, want_weight=True):
    """labels: int tensor [B,H,W] or [B,Z,Y,X] on the GPU -> (target f32, mask u8 or None, weight f32 or None),
    each [B,K,...].  One call per scale of the deep-supervision pyramid."""
    if not isinstance(labels, torch.Tensor) or not labels.is_cuda:
        raise RuntimeError("gen_targets runs on an MI355X tensor only (no CPU fallback)")
    if labels.dtype.is_floating_point:
        raise TypeError("labels must be an integer tensor")
    if labels.dim() not in (3, 4):
        raise ValueError("labels must be [B,H,W] or [B,Z,Y,X], got %s" % (tuple(labels.shape),))
    ndim = labels.dim() - 1
    lab = _labels_int32(labels)
    spec = AffinitySpec(ndim, offsets, None, _lib.BORDER_CROP_ZERO, _lib.NORM_FULL)
    # make_desc reads B / D / spatial dims off an embedding-shaped tensor: a meta tensor carries just the shape
    shape_probe = torch.empty((lab.shape[0], 1) + tuple(lab.shape[1:]), dtype=torch.float32, device="meta")
    with torch.cuda.device(lab.device):
        d = make_desc(spec, shape_probe)
        kshape = (lab.shape[0], spec.K) + tuple(lab.shape[1:])
        target = torch.empty(kshape, dtype=torch.float32, device=lab.device)
        mask = torch.empty(kshape, dtype=torch.uint8, device=lab.device) if want_mask else None
        weight = torch.empty(kshape, dtype=torch.float32, device=lab.device) if want_weight else None
        L = _lib.lib()
        wsb = L.pea_targets_workspace_bytes(ctypes.byref(d))
        work = torch.empty(max(wsb, 4) // 4, dtype=torch.int32, device=lab.device)
        flags = (_lib.TGT_PADDING if padding else 0) | (_lib.TGT_BOTH_FOREGROUND if both_foreground else 0)
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        _lib.check(L.pea_gen_targets(ctypes.byref(d), p(lab), flags, p(target), p(mask), p(weight), p(work), wsb,
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "pea_gen_targets")
    return target, mask, weight


def gen_affs_ours(labels, offsets=((-1, 0), (0, -1)), ignore=False, padding=False):
    """the reference's signature (batched, on the GPU): -> (affinities f32 [B,K,H,W], masks u8 [B,K,H,W])"""
    if ignore:
        raise NotImplementedError("ignore=True is not used by any shipped configuration")
    t, m, _ = gen_targets(labels, offsets, padding=padding, want_weight=False)
    return t, m


def seg_to_aff(seg, nhood=((-1, 0, 0), (0, -1, 0), (0, 0, -1)), pad='replicate'):
    """the reference's 3D target generator (scripts_ac3ac4/data/data_affinity.py:53-102), batched on the GPU:
    aff[b, e, p] = [seg(p) == seg(p + nhood[e])] * [both > 0], 0 where the neighbour is outside the volume; with three edges and
    pad='replicate' the first slice of channel c along axis c is (seg > 0) instead (:94-97).  seg: int tensor [B,Z,Y,X]."""
    t, _, _ = gen_targets(seg, [list(o) for o in nhood], padding=False, both_foreground=True, want_mask=False, want_weight=False)
    if len(nhood) == 3 and pad == 'replicate':
        fg = (seg > 0).to(t.dtype)
        t[:, 0, 0] = fg[:, 0]
        t[:, 1, :, 0] = fg[:, :, 0]
        t[:, 2, :, :, 0] = fg[:, :, :, 0]
    return t
