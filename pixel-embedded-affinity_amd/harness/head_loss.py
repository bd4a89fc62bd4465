"""The embedding head and the self loss as ONE autograd node.

The reference computes  embedding = self.outconv_emb(x)  (scripts_cvppp/model/unet2d_residual.py:346) inside the model and
embedding_loss(embedding, ...)  in the training loop (main.py:284); autograd then runs the loss' backward and the head's
backward one after the other.  Here head forward, loss forward, loss backward and head backward are one node that launches the
four kernels itself (no autograd bookkeeping between them, no zero-filled gradients of the non-differentiable outputs).

    loss, affs, all_loss, embedding = head_embedding_loss(x, head, target, weightmap, mask, criterion, offsets)

`head` is this package's OutConv (or any module with a 1x1 `conv`); `embedding` comes back as a differentiable output of the
same node, so the other losses of the section (the EMA cross loss, the consistency term) keep working on it -- whatever
gradient they send into it is added before the head's backward.

History: round 2 also had the two backwards as ONE launch (the head's dx / dW / db in the epilogue of the loss backward).
MEASURED at B=8 x 32 -> 16 x 544^2 (profiles/r2c_f1_fused_backward.txt): 345 us against 116 + 161 us for the two launches --
x loads, 512 FMAs and 32 stores per lane in the tail of a 2-workgroups-per-CU LDS kernel overlap with nothing, while the
stand-alone head kernels stream at 5.5 TB/s.  A fusion that loses is dead weight: removed in round 3."""
import ctypes

import torch

from .. import _lib
from ..affinity_op import (AffinitySpec, LossList, _affs_shape, _batch_strided, _on_device, _ptr, _require_gpu, _stream, make_desc,
                           workspace)
from ..model.head import head_supported


class HeadAffinityMSE(torch.autograd.Function):
    """(loss, affs, per_offset_losses, embedding) = f(x, weight, bias, target, weightmap, mask)"""

    @staticmethod
    def forward(ctx, x, weight, bias, target, weightmap, mask, spec):
        ctx.set_materialize_grads(False)
        _require_gpu(x, "x")
        if x.dtype != torch.float32 or weight.dtype != torch.float32:
            raise TypeError("the embedding head runs in float32 (got %s / %s)" % (x.dtype, weight.dtype))
        D, C = weight.shape[0], weight.shape[1]
        if not head_supported(C, D):
            raise ValueError("no HIP head for %d -> %d channels" % (C, D))
        xc = x.contiguous()
        wc = weight.detach().reshape(D, C).contiguous()
        bc = None if bias is None else bias.detach().contiguous()
        B, S = xc.shape[0], xc[0, 0].numel()
        L = _lib.lib()
        with _on_device(xc.device):
            e = torch.empty((B, D) + tuple(xc.shape[2:]), dtype=torch.float32, device=xc.device)
            _lib.check(L.pea_head_fwd(B, C, D, S, _ptr(xc), _ptr(wc), _ptr(bc), _ptr(e), _stream()), "pea_head_fwd")
            kshape = _affs_shape(e, spec.K)
            target, ts = _batch_strided(target, "target", torch.float32, kshape)
            weightmap, ws = _batch_strided(weightmap, "weightmap", torch.float32, kshape)
            ms = 0
            if mask is not None:
                if mask.dtype == torch.bool:
                    mask = mask.view(torch.uint8)
                mask, ms = _batch_strided(mask, "mask", torch.uint8, kshape)
            d = make_desc(spec, e, ts, ws, ms)
            affs = torch.empty(kshape, dtype=torch.float32, device=e.device)
            loss_vec = torch.empty(1 + spec.K, dtype=torch.float32, device=e.device)
            work, wsb = workspace(e.device, d)
            g = torch.empty(kshape, dtype=torch.float32, device=e.device)
            inv = torch.empty((B,) + tuple(e.shape[2:]), dtype=torch.float32, device=e.device)
            _lib.check(L.pea_affinity_fwd_ex(ctypes.byref(d), _ptr(e), None, _ptr(target), _ptr(weightmap), _ptr(mask), _ptr(affs),
                                             _ptr(g), _ptr(inv), _ptr(loss_vec), _ptr(work), wsb, _stream()), "pea_affinity_fwd_ex")
        ctx.desc, ctx.spec = d, spec
        ctx.has_bias = bias is not None
        ctx.wshape = tuple(weight.shape)
        ctx.save_for_backward(xc, wc, e, g, inv)
        loss, per_offset = loss_vec[0], loss_vec[1:]
        ctx.mark_non_differentiable(affs, per_offset)
        return loss, affs, per_offset, e

    @staticmethod
    def backward(ctx, dloss, _daffs, _dvec, de_ext):
        xc, wc, e, g, inv = ctx.saved_tensors
        D, C = wc.shape
        B, S = xc.shape[0], xc[0, 0].numel()
        L = _lib.lib()
        with _on_device(xc.device):
            dx = torch.empty_like(xc) if ctx.needs_input_grad[0] else None
            dW = torch.empty((D, C), dtype=torch.float32, device=xc.device)
            db = torch.empty(D, dtype=torch.float32, device=xc.device) if ctx.has_bias else None
            add = None if de_ext is None else de_ext.to(torch.float32).contiguous()
            if dloss is None:  # only the embedding output was used downstream: the head's backward alone
                if add is None:
                    return (None,) * 7
                de = add
            else:
                dl = dloss.to(device=xc.device, dtype=torch.float32).contiguous()
                de = torch.empty_like(e)
                _lib.check(L.pea_affinity_bwd_ex(ctypes.byref(ctx.desc), _ptr(e), None, _ptr(g), _ptr(inv), _ptr(dl), _ptr(de), None,
                                                 _stream()), "pea_affinity_bwd_ex")
                if add is not None:
                    de += add
            wsb = L.pea_head_workspace_bytes(C, D)
            work = torch.empty(wsb // 4, dtype=torch.float32, device=xc.device)
            _lib.check(L.pea_head_bwd(B, C, D, S, _ptr(xc), _ptr(wc), _ptr(de), _ptr(dx), _ptr(dW), _ptr(db), _ptr(work), wsb, _stream()),
                       "pea_head_bwd")
        return dx, dW.reshape(ctx.wshape), db, None, None, None, None


def head_embedding_loss(x, head, target, weightmap, mask, criterion, offsets, affs0_weight=1, mode='ours'):
    """-> (loss, affs [B,K,H,W], all_loss list[K], embedding [B,D,H,W]): head(x) followed by embedding_loss(...) of
    loss/loss_embedding_mse.py (reference :18-47), as one autograd node; criterion must be this package's WeightedMSE"""
    if not getattr(criterion, 'pea_fused', False):
        raise TypeError("head_embedding_loss needs the fused WeightedMSE criterion; use head(x) + embedding_loss(...) otherwise")
    conv = head.conv if hasattr(head, "conv") else head[0] if isinstance(head, torch.nn.Sequential) else head
    ndim = conv.weight.dim() - 2
    if ndim != 2:
        raise ValueError("head_embedding_loss is the 2D (CVPPP / BBBC039V1) call; 3D heads use head(x) + embedding_loss_norm*")
    spec = AffinitySpec(2, offsets, [1.0] * len(offsets), _lib.BORDER_CIRCULAR, _lib.NORM_BX, 1e-12 if mode == 'ours' else 1e-6)
    loss, affs, parts, emb = HeadAffinityMSE.apply(x, conv.weight, conv.bias, target, weightmap, mask, spec)
    return loss, affs, LossList(parts), emb
