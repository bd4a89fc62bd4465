"""CPU oracle for the embedding -> affinity path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (pixel-embedded-affinity_amd/) never does.  Three independent restatements of the
reference's algorithm live here, all pinned against the golden vectors that
tests/golden/make_golden.py produced by importing the reference's own functions:

  * C  (oracle/pea_oracle.c via ctypes)      -- the checker used by the GPU parity tests;
  * numpy (np_* below)                        -- small-case cross-check of the C code;
  * torch-CPU (torch_* below)                 -- the same op sequence the reference runs
    (F.normalize -> torch.roll / slices -> mul -> sum -> WeightedMSE -> autograd), used as the
    timed CPU baseline ("port") in bench.py because the reference's files cannot travel to the GPU box.

Reference lines followed:
  2D  scripts_cvppp/loss/loss_embedding_mse.py:7-95, scripts_cvppp/loss/loss.py:106-124
  3D  scripts_ac3ac4/loss/loss_embedding_mse.py:7-67,143-289
  offsets  scripts_cvppp/utils/affinity_ours.py:4-15, scripts_ac3ac4/loss/loss_embedding_mse.py:176
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PEA_MAX_K = 32
ABI = 2  # PEA_ABI_VERSION of include/pea.h (the oracle shares the descriptor struct)
BORDER_CIRCULAR, BORDER_CROP_ZERO, BORDER_REPLICATE = 0, 1, 2
NORM_BX, NORM_CROPPED, NORM_FULL = 0, 1, 2
FLAG_RELU = 1
NORM5_SHIFTS = [1, 1, 1, 2, 3, 3, 3, 9, 9, 4, 27, 27]  # ac34/loss/loss_embedding_mse.py:176


class PeaDesc(ctypes.Structure):
    _fields_ = [("abi", ctypes.c_int32), ("ndim", ctypes.c_int32), ("B", ctypes.c_int32),
                ("D", ctypes.c_int32), ("dims", ctypes.c_int32 * 3), ("K", ctypes.c_int32),
                ("border", ctypes.c_int32), ("dtype", ctypes.c_int32), ("norm", ctypes.c_int32),
                ("flags", ctypes.c_uint32), ("eps", ctypes.c_float),
                ("offsets", (ctypes.c_int32 * 3) * PEA_MAX_K), ("lam", ctypes.c_float * PEA_MAX_K),
                ("target_bstride", ctypes.c_int64), ("weight_bstride", ctypes.c_int64),
                ("mask_bstride", ctypes.c_int64)]


# ----------------------------------------------------------------------------------------------
# offsets (affinity_ours.py:4-15 and the 3D literal tables)
# ----------------------------------------------------------------------------------------------
def gen_offsets(shift, neighbor=4):
    assert neighbor in (4, 8), "neigbor must be 4 or 8!"
    out = [[-shift, 0], [0, -shift]]
    if neighbor == 8:
        out += [[-shift, -shift], [-shift, shift]]
    return out


def multi_offset(shifts, neighbor=4):
    return [o for s in shifts for o in gen_offsets(s, neighbor)]


def offsets3(offsets):
    """2D (dy,dx) or 3D (dz,dy,dx) list -> list of (dz,dy,dx)."""
    return [([0] * (3 - len(o)) + [int(v) for v in o]) for o in offsets]


def norm_offsets(shifts):
    """3D: channel i shifts by shifts[i] along axis i % 3 (z,y,x); neighbour = p - shift."""
    out = []
    for i, s in enumerate(shifts):
        o = [0, 0, 0]
        o[i % 3] = -int(s)
        out.append(o)
    return out


# ----------------------------------------------------------------------------------------------
# C oracle binding
# ----------------------------------------------------------------------------------------------
_lib = None


def build(force=False):
    so = os.path.join(HERE, "libpea_oracle.so")
    src = os.path.join(HERE, "pea_oracle.c")
    hdr = os.path.join(HERE, "..", "include", "pea.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", HERE, "-B", "libpea_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "libpea_oracle.so")
        if not os.path.exists(so):
            build()
        _lib = ctypes.CDLL(so)
        _lib.pea_oracle_fwd.restype = ctypes.c_int
        _lib.pea_oracle_bwd.restype = ctypes.c_int
        _lib.pea_oracle_bwd.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_float] + [ctypes.c_void_p] * 2
        _lib.pea_oracle_fwd.argtypes = [ctypes.c_void_p] * 8
    return _lib


def make_desc(B, D, dims, offs, lam=None, border=BORDER_CIRCULAR, norm=NORM_BX, eps=1e-12, flags=0, ndim=None):
    d = PeaDesc()
    dims = [1] * (3 - len(dims)) + [int(v) for v in dims]
    d.abi, d.ndim, d.B, d.D = ABI, (ndim or (2 if dims[0] == 1 else 3)), int(B), int(D)
    d.dims[:] = dims
    o3 = offsets3(offs)
    d.K = len(o3)
    assert 1 <= d.K <= PEA_MAX_K
    d.border, d.dtype, d.norm, d.flags, d.eps = border, 0, norm, flags, eps
    for i, o in enumerate(o3):
        d.offsets[i][:] = o
        d.lam[i] = 1.0 if lam is None else float(lam[i])
    return d


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def c_set_threads(n):
    """OpenMP threads of the C restatement (pea_oracle_set_threads); returns the previous maximum"""
    L = lib()
    L.pea_oracle_set_threads.restype = ctypes.c_int
    L.pea_oracle_set_threads.argtypes = [ctypes.c_int]
    return int(L.pea_oracle_set_threads(int(n)))


def c_fwd(desc, e, e_other=None, target=None, weight=None, mask=None, want_affs=True):
    """-> (affs or None, loss_vec float64[1+K] or None)"""
    e, e_other = _c(e, np.float32), _c(e_other, np.float32)
    target, weight, mask = _c(target, np.float32), _c(weight, np.float32), _c(mask, np.uint8)
    S = desc.dims[0] * desc.dims[1] * desc.dims[2]
    affs = np.empty((desc.B, desc.K, S), np.float32) if want_affs else None
    loss = np.zeros(1 + desc.K, np.float64) if target is not None else None
    rc = lib().pea_oracle_fwd(ctypes.addressof(desc), _p(e), _p(e_other), _p(target), _p(weight), _p(mask),
                              _p(affs), _p(loss))
    if rc:
        raise RuntimeError("pea_oracle_fwd rc=%d" % rc)
    if affs is not None:
        sp = [desc.dims[1], desc.dims[2]] if desc.ndim == 2 else list(desc.dims)
        affs = affs.reshape([desc.B, desc.K] + sp)
    return affs, loss


def c_bwd(desc, e, e_other, target, weight, mask, dloss=1.0, want_other=False):
    e, e_other = _c(e, np.float32), _c(e_other, np.float32)
    target, weight, mask = _c(target, np.float32), _c(weight, np.float32), _c(mask, np.uint8)
    de = np.empty_like(e)
    de_o = np.empty_like(e_other) if (want_other and e_other is not None) else None
    rc = lib().pea_oracle_bwd(ctypes.addressof(desc), _p(e), _p(e_other), _p(target), _p(weight), _p(mask),
                              ctypes.c_float(dloss), _p(de), _p(de_o))
    if rc:
        raise RuntimeError("pea_oracle_bwd rc=%d" % rc)
    return de, de_o


# reference-named wrappers over the C oracle (numpy in / numpy out) ---------------------------
def affs0_lambda_2d_self(K):
    """embedding_loss: affs0_weight is computed but NOT applied (cvppp/...mse.py:26-39)."""
    return [1.0] * K


def affs0_lambda_2d_ema(K, affs0_weight):
    """ema_embedding_loss: applied to i < 2 (cvppp/...mse.py:90-93)."""
    return [float(affs0_weight) if i < 2 else 1.0 for i in range(K)]


def affs0_lambda_3d(K, affs0_weight, first=3):
    """norm5: i < 3 (ac34/...mse.py:181-184); norm1: only loss0 (ac34/...mse.py:20)."""
    return [float(affs0_weight) if i < first else 1.0 for i in range(K)]


def desc_2d(e, offsets, lam=None, mode="ours", relu=False):
    B, D, H, W = e.shape
    return make_desc(B, D, [1, H, W], offsets, lam, BORDER_CIRCULAR, NORM_BX,
                     1e-12 if mode == "ours" else 1e-6, FLAG_RELU if relu else 0, ndim=2)


def desc_3d(e, shifts, lam=None):
    B, D, Z, Y, X = e.shape
    return make_desc(B, D, [Z, Y, X], norm_offsets(shifts), lam, BORDER_CROP_ZERO, NORM_CROPPED, 1e-12, 0, ndim=3)


def desc_3d_replicate(e, offsets):
    """embedding_loss_norm6 (scripts_ac3ac4/loss/loss_embedding_mse.py:346-354): generic offsets, replicate border, one
    WeightedMSE over the [B,K,Z,Y,X] map (normaliser B*Z*Y*X, loss.py:113-115), lambda = 1"""
    B, D, Z, Y, X = e.shape
    return make_desc(B, D, [Z, Y, X], [list(o) for o in offsets], None, BORDER_REPLICATE, NORM_FULL, 1e-12, 0, ndim=3)


# ----------------------------------------------------------------------------------------------
# numpy restatement (independent of the C code; follows the reference op by op)
# ----------------------------------------------------------------------------------------------
def np_normalize(e, eps=1e-12):
    n = np.sqrt((e * e).sum(axis=1, keepdims=True))
    return e / np.maximum(n, eps)


def np_weighted_mse(pred, target, weight):
    """loss.py:112-119: norm = prod(pred.shape[2:]) * pred.shape[0]."""
    norm = float(np.prod(pred.shape[2:])) * pred.shape[0]
    return float((weight.astype(np.float64) * (pred.astype(np.float64) - target) ** 2).sum() / norm)


def np_embedding_loss(e, target, weight, mask, offsets, ema=None, affs0_weight=1, mode="ours"):
    """2D embedding_loss / ema_embedding_loss forward: (loss, affs, all_loss)."""
    eps = 1e-12 if mode == "ours" else 1e-6
    eh = np_normalize(e.astype(np.float32), eps)
    oh = eh if ema is None else np_normalize(ema.astype(np.float32), eps)
    m = mask.astype(np.float32)
    affs = np.zeros_like(target, dtype=np.float32)
    loss, all_loss = 0.0, []
    for i, off in enumerate(offsets):
        shift_off = tuple(-x for x in off)
        rolled = np.roll(oh, shift_off, axis=(2, 3))
        a = (rolled * eh).sum(axis=1)  # [B,H,W]  -> WeightedMSE sees size()[2:] == (W,)
        li = np_weighted_mse(a * m[:, i], target[:, i] * m[:, i], weight[:, i])
        lam = (affs0_weight if i < 2 else 1.0) if ema is not None else 1.0
        loss += lam * li
        all_loss.append(li)
        affs[:, i] = a
    return loss, affs, all_loss


def np_embedding_loss_3d(e, target, weight, shifts, ema=None, affs0_weight=1, first=3):
    """3D norm1 (shifts=[s,s,s], first=1) / norm5 (NORM5_SHIFTS, first=3): (loss, affs)."""
    eh = np_normalize(e.astype(np.float32))
    oh = eh if ema is None else np_normalize(ema.astype(np.float32))
    affs = np.zeros(e.shape[:1] + (len(shifts),) + e.shape[2:], np.float32)
    loss = 0.0
    for i, s in enumerate(shifts):
        ax = 2 + i % 3
        hi = [slice(None)] * 5
        lo = [slice(None)] * 5
        hi[ax] = slice(s, None)
        lo[ax] = slice(None, e.shape[ax] - s)
        a = (eh[tuple(hi)] * oh[tuple(lo)]).sum(axis=1, keepdims=True)
        if target is not None:
            sel = list(hi)
            sel[1] = slice(i, i + 1)
            li = np_weighted_mse(a, target[tuple(sel)], weight[tuple(sel)])
            loss += (affs0_weight if i < first else 1.0) * li
        dst = list(hi)
        dst[1] = slice(i, i + 1)
        affs[tuple(dst)] = a
    return loss, affs


# ----------------------------------------------------------------------------------------------
# torch-CPU restatement: the reference's op sequence, used as the timed CPU baseline
# ----------------------------------------------------------------------------------------------
def torch_weighted_mse(pred, target, weight):
    import torch
    norm = float(np.prod(pred.shape[2:])) * pred.shape[0]
    return torch.sum(weight * (pred - target) ** 2) / norm


def torch_embedding_loss(e, target, weight, mask, offsets, ema=None, affs0_weight=1, eps=1e-12, criterion=None):
    """fwd of the 2D path on torch tensors (autograd-capable): (loss, affs, per-offset list of tensors)."""
    criterion = criterion or torch_weighted_mse
    import torch
    import torch.nn.functional as F
    eh = F.normalize(e, p=2, dim=1, eps=eps)
    oh = eh if ema is None else F.normalize(ema, p=2, dim=1, eps=eps)
    m = mask.float()
    affs = torch.zeros_like(target)
    loss = torch.zeros((), dtype=e.dtype)
    parts = []
    for i, off in enumerate(offsets):
        a = torch.sum(torch.roll(oh, shifts=(-off[0], -off[1]), dims=(2, 3)) * eh, dim=1)
        li = criterion(a * m[:, i], target[:, i] * m[:, i], weight[:, i])
        loss = loss + li * ((affs0_weight if i < 2 else 1.0) if ema is not None else 1.0)
        parts.append(li)
        affs[:, i] = a.detach()
    return loss, affs, parts


def torch_embedding_loss_3d(e, target, weight, shifts, ema=None, affs0_weight=1, first=3):
    import torch
    import torch.nn.functional as F
    eh = F.normalize(e, p=2, dim=1)
    oh = eh if ema is None else F.normalize(ema, p=2, dim=1)
    affs = torch.zeros(e.shape[:1] + (len(shifts),) + e.shape[2:], dtype=e.dtype)
    loss = torch.zeros((), dtype=e.dtype)
    for i, s in enumerate(shifts):
        ax = 2 + i % 3
        n = e.shape[ax]
        a = torch.sum(eh.narrow(ax, s, n - s) * oh.narrow(ax, 0, n - s), dim=1, keepdim=True)
        if target is not None:
            t = target[:, i:i + 1].narrow(ax, s, n - s)
            w = weight[:, i:i + 1].narrow(ax, s, n - s)
            loss = loss + torch_weighted_mse(a, t, w) * (affs0_weight if i < first else 1.0)
        affs[:, i:i + 1].narrow(ax, s, n - s).copy_(a.detach())
    return loss, affs


# ---------------------------------------------------------------------------------------------------
# label image -> target / mask / class-balance weight  (the checker of pea_gen_targets; test infrastructure)
# ---------------------------------------------------------------------------------------------------
def np_gen_targets(labels, offsets, padding=True, both_foreground=False):
    """labels [B,Z,Y,X] int -> target f32 [B,K,Z,Y,X], mask u8 [B,K,Z,Y,X].

    Follows gen_affs_ours(labels, offsets, ignore=False, padding=...) (scripts_cvppp/utils/affinity_ours.py:17-39):
    shifted = scipy shift(labels, -off, order=0) = labels(p + off), zero outside; t = 1 iff labels == shifted;
    where the neighbour is outside the image mask = 0 and t = 1 (padding) or 0.  both_foreground: t = 1 only if both
    labels are > 0 (seg_to_aff, scripts_ac3ac4/data/data_affinity.py:53-102, interior).  Pinned by
    tests/golden/gtgt_*.npz, which hold the reference's own outputs."""
    labels = np.asarray(labels)
    B, Z, Y, X = labels.shape
    offs = offsets3(offsets)
    t = np.zeros((B, len(offs), Z, Y, X), np.float32)
    m = np.zeros((B, len(offs), Z, Y, X), np.uint8)
    for i, (dz, dy, dx) in enumerate(offs):
        for z in range(Z):
            zz = z + dz
            for y in range(Y):
                yy = y + dy
                inside_zy = 0 <= zz < Z and 0 <= yy < Y
                for x in range(X):
                    xx = x + dx
                    if inside_zy and 0 <= xx < X:
                        a, b = labels[:, z, y, x], labels[:, zz, yy, xx]
                        eq = a == b
                        if both_foreground:
                            eq = eq & (a > 0) & (b > 0)
                        t[:, i, z, y, x] = eq
                        m[:, i, z, y, x] = 1
                    else:
                        t[:, i, z, y, x] = 1.0 if padding else 0.0
    return t, m


def np_weight_binary_ratio(target, alpha=1.0):
    """per (b, channel) weight_binary_ratio(lb_affs[i]) with mask=None
    (scripts_cvppp/data/data_segmentation.py:205-228, called per channel at data_provider.py:216-225): uniform 1 for a
    single-valued channel; else f = clip(mean(label != 0), 0.05, 0.99), the minority class gets max(f,1-f)/min(f,1-f),
    the majority 1; float64 arithmetic, float32 result."""
    target = np.asarray(target)
    w = np.ones(target.shape, np.float32)
    for b in range(target.shape[0]):
        for i in range(target.shape[1]):
            lab = target[b, i]
            if lab.max() == lab.min():
                continue
            lab = (lab != 0).astype(int)
            f = float(lab.sum()) / np.prod(lab.shape)
            f = np.clip(f, 5e-2, 0.99)
            if f > 0.5:
                w[b, i] = (lab + alpha * f / (1 - f) * (1 - lab)).astype(np.float32)
            else:
                w[b, i] = (alpha * (1 - f) / f * lab + (1 - lab)).astype(np.float32)
    return w


# ---------------------------------------------------------------------------------------------------
# the embedding head (the checker of pea_head_fwd / pea_head_bwd; test infrastructure)
# ---------------------------------------------------------------------------------------------------
def np_head_fwd(x, weight, bias=None):
    """x [B,C,*spatial] f32, weight [D,C], bias [D] or None -> e [B,D,*spatial] f32.

    Follows OutConv.forward = nn.Conv2d(in_ch, out_ch, 1) (scripts_cvppp/model/unet2d_residual.py:67-74) and the
    1x1x1 conv3dBlock heads (scripts_ac3ac4/model/basic.py:114-127, model_superhuman.py:437-441):
    e[b,d,p] = bias[d] + sum_c W[d,c] x[b,c,p], accumulated in float64.  Pinned by tests/golden/ghead_*.npz
    (outputs and autograd gradients of the reference modules themselves)."""
    e = np.einsum("dc,bc...->bd...", weight.astype(np.float64), x.astype(np.float64))
    if bias is not None and bias.size:
        e = e + bias.astype(np.float64).reshape((1, -1) + (1,) * (x.ndim - 2))
    return e.astype(np.float32)


def np_head_bwd(x, weight, de):
    """-> dx [B,C,*spatial], dW [D,C], db [D] for upstream de [B,D,*spatial] (float64 accumulation)."""
    B, C, D = x.shape[0], x.shape[1], de.shape[1]
    x64, de64 = x.astype(np.float64).reshape(B, C, -1), de.astype(np.float64).reshape(B, D, -1)
    dx = np.einsum("dc,bdp->bcp", weight.astype(np.float64), de64).reshape(x.shape)
    dW = np.einsum("bdp,bcp->dc", de64, x64)
    db = de64.sum(axis=(0, 2))
    return dx.astype(np.float32), dW.astype(np.float32), db.astype(np.float32)
