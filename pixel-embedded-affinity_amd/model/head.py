"""The per-pixel embedding head on the MI355X path: the 1x1 (1x1x1) convolution that turns the decoder's feature map
into the embedding the affinity kernels read.

Mirrors, with the same parameter names (so the reference's checkpoints load unchanged):
  OutConv(in_ch, out_ch)                       scripts_cvppp/model/unet2d_residual.py:67-74 (outconv_emb :307, :346;
                                               same class in scripts_bbbc039v1/model/unet2d_residual.py:67,235)
  conv3dBlock([C], [D], [(1, 1, 1)])           scripts_ac3ac4/model/basic.py:114-127, the out_put* heads of
                                               scripts_ac3ac4/model/model_superhuman.py:437-441 (applied :486-490)

Forward and backward are pea_head_fwd / pea_head_bwd (include/pea.h): hand-written streaming kernels, the weight
gradient on the matrix cores in exact f32.  The library has kernels for every head of the reference's models (2D ResUNet:
32 / 64 / 128 / 256 input channels -> 16 or 32; 3D superhuman U-Net: 28 / 36 / 48 / 64 / 80 -> 16); any other channel
pair goes through torch's own GPU convolution.  CPU tensors are refused like everywhere else in this package."""

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from ..affinity_op import _on_device, _ptr, _require_gpu, _stream


def head_supported(C, D):
    return (D == 16 and C in (28, 32, 36, 48, 64, 80, 128, 256)) or (D == 32 and C in (32, 64, 128, 256))


class EmbeddingHead(torch.autograd.Function):
    """e = conv1x1(x; weight, bias) -- x [B,C,*spatial] f32, weight [D,C,1,1(,1)], bias [D] or None"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_gpu(x, "x")
        if x.dtype != torch.float32 or weight.dtype != torch.float32:
            raise TypeError("the embedding head runs in float32 (got %s / %s)" % (x.dtype, weight.dtype))
        D, C = weight.shape[0], weight.shape[1]
        if x.dim() < 3 or x.shape[1] != C or weight.numel() != D * C:
            raise ValueError("x %s does not fit a 1x1 convolution with weight %s" % (tuple(x.shape), tuple(weight.shape)))
        xc = x.contiguous()
        wc = weight.detach().reshape(D, C).contiguous()
        bc = None if bias is None else bias.detach().contiguous()
        B, S = xc.shape[0], xc[0, 0].numel()
        with _on_device(xc.device):
            e = torch.empty((B, D) + tuple(xc.shape[2:]), dtype=torch.float32, device=xc.device)
            _lib.check(_lib.lib().pea_head_fwd(B, C, D, S, _ptr(xc), _ptr(wc), _ptr(bc), _ptr(e), _stream()), "pea_head_fwd")
        ctx.save_for_backward(xc, wc)
        ctx.has_bias = bias is not None
        ctx.wshape = tuple(weight.shape)
        return e

    @staticmethod
    def backward(ctx, de):
        xc, wc = ctx.saved_tensors
        D, C = wc.shape
        B, S = xc.shape[0], xc[0, 0].numel()
        L = _lib.lib()
        with _on_device(xc.device):
            dec = de.to(torch.float32).contiguous()
            dx = torch.empty_like(xc) if ctx.needs_input_grad[0] else None
            dW = torch.empty((D, C), dtype=torch.float32, device=xc.device)
            db = torch.empty(D, dtype=torch.float32, device=xc.device) if ctx.has_bias else None
            wsb = L.pea_head_workspace_bytes(C, D)
            work = torch.empty(wsb // 4, dtype=torch.float32, device=xc.device)
            _lib.check(L.pea_head_bwd(B, C, D, S, _ptr(xc), _ptr(wc), _ptr(dec), _ptr(dx), _ptr(dW), _ptr(db), _ptr(work), wsb,
                                      _stream()), "pea_head_bwd")
        return dx, dW.reshape(ctx.wshape), db


def _apply_head(conv, x):
    D, C = conv.weight.shape[0], conv.weight.shape[1]
    if head_supported(C, D) and x.dtype == torch.float32:
        return EmbeddingHead.apply(x, conv.weight, conv.bias)
    _require_gpu(x, "x")
    return (F.conv3d if conv.weight.dim() == 5 else F.conv2d)(x, conv.weight, conv.bias)


class OutConv(nn.Module):
    """drop-in for the reference's OutConv: same constructor, same `conv.weight` / `conv.bias` parameters"""

    def __init__(self, in_ch, out_ch):
        super(OutConv, self).__init__()
        self.conv = nn.Conv2d(in_ch, out_ch, 1)

    def forward(self, x):
        return _apply_head(self.conv, x)


class _HeadSequential(nn.Sequential):
    """nn.Sequential(Conv3d(C, D, 1)) whose forward is the HIP head: the parameter names stay `0.weight` / `0.bias`"""

    def forward(self, x):
        return _apply_head(self[0], x)


def head_conv3d_block(in_planes, out_planes, bias=True):
    """drop-in for conv3dBlock([in_planes], [out_planes], [(1, 1, 1)]) as the 3D model builds its out_put* heads
    (initialise the returned module's `[0].weight` the way the model's init_mode asks, as conv3dBlock does)"""
    return _HeadSequential(nn.Conv3d(in_planes, out_planes, kernel_size=(1, 1, 1), bias=bias))
