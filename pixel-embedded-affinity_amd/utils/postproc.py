"""Caller-side epilogues on the affinity map that the reference applies right after the loss call.

  * fill_border_relu_  scripts_ac3ac4/main.py:233-237, 296-300 and scripts_ac3ac4/inference.py:160-164
                       (first slice(s) of channel c along axis c copied from the next ones, then F.relu)
  * relu_              scripts_cvppp/main.py:312,395; scripts_cvppp/inference.py:193  (F.relu(pred))
Both run in place through pea_fill_border_relu (one launch, no temporaries); device tensors only.
"""
import ctypes

import torch

from .. import _lib


def fill_border_relu_(pred, shift=1, relu=True):
    """in place on pred [B,K,Z,Y,X] (f32, contiguous, on the GPU); returns pred"""
    if not isinstance(pred, torch.Tensor) or not pred.is_cuda:
        raise RuntimeError("fill_border_relu_ runs on an MI355X tensor only (no CPU fallback)")
    if pred.dtype != torch.float32 or not pred.is_contiguous():
        raise ValueError("pred must be a contiguous float32 tensor")
    if pred.dim() == 4:
        B, K, Z, Y, X = pred.shape[0], pred.shape[1], 1, pred.shape[2], pred.shape[3]
        if shift:
            raise ValueError("border fill is the 3D callers' epilogue: pass shift=0 for [B,K,H,W]")
    elif pred.dim() == 5:
        B, K, Z, Y, X = pred.shape
    else:
        raise ValueError("pred must be [B,K,H,W] or [B,K,Z,Y,X], got %s" % (tuple(pred.shape),))
    with torch.cuda.device(pred.device):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_lib.lib().pea_fill_border_relu(ctypes.c_void_p(pred.data_ptr()), B, K, Z, Y, X, int(shift),
                                                   1 if relu else 0, st), "pea_fill_border_relu")
    # the kernel writes through data_ptr(): tell autograd, so that a map saved for a backward (the raw cosines of the
    # projection-first backward, affinity_op.FusedAffinityMSE) raises "modified by an inplace operation" instead of miscomputing
    torch.autograd.graph.increment_version(pred)
    return pred


def relu_(pred):
    """F.relu(pred) in place"""
    return fill_border_relu_(pred, shift=0, relu=True)
