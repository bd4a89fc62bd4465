"""GPU (-m gpu): BASELINE.json configs[2], [3], [4] at their FULL sizes (the golden fixtures and the oracle sweeps run them at
sizes the CPU finishes in seconds).  At full size the CPU oracle checks a cropped WINDOW: every quantity the op produces
for a pixel depends only on pixels within the stencil's reach, so inside a window (margin = reach) the full-size result
must equal the oracle's result on the window alone -- with the window's per-offset weights set so that its normaliser
equals the full problem's.  Beside it, size-independent properties: bit-identical reruns, batch additivity (= the sharding
identity of SURVEY.md section 8e), and the direct (global-memory) kernels on the same window.

    configs[2]  BBBC039V1: B=8 per GPU x D=32 x 704 x 704, shifts 1,3,5,9,11 x neighbor 4 (K=10), circular, u8 mask
    configs[3]  AC3/AC4:   B=1 x D=16 x 24 x 1024 x 1024, norm5's 12 axis offsets (K=12) and the 26-neighbourhood (K=26), CROP_ZERO
    configs[4]  D=64:      B=8 x D=64 x 544 x 544, offsets[:8], f16 storage / f32 accumulate

Tolerances: affs abs 1e-5, grads rel-to-max 1e-4 (f16 storage: the stored gradient is rounded to f16, 2e-3)."""
import importlib

import numpy as np
import pytest
import torch

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu
AFFS_ATOL, GRAD_RTOL = 1e-5, 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def relmax(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def gpu_inputs(dev, B, D, dims, K, seed, f16=False, mask=True):
    g = torch.Generator(device=dev).manual_seed(seed)
    e = torch.randn([B, D] + dims, generator=g, device=dev)
    if f16:
        e = e.half()
    t = (torch.rand([B, K] + dims, generator=g, device=dev) < 0.6).float()
    w = torch.rand([B, K] + dims, generator=g, device=dev) + 0.5
    m = (torch.rand([B, K] + dims, generator=g, device=dev) < 0.9).to(torch.uint8) if mask else None
    return e, t, w, m


def run(pkg, spec, e, t, w, m, dloss=1.0):
    et = e.detach().clone().requires_grad_(True)
    loss, affs, parts = pkg.affinity_op.FusedAffinityMSE.apply(et, None, t, w, m, spec)
    (loss * dloss).backward()
    return loss.detach(), affs, parts, et.grad


def full_norms(pkg, spec, B, dims):
    """N_i of the full problem (include/pea.h)"""
    out = []
    for o in spec.offsets:
        if spec.norm == pkg._lib.NORM_BX:
            out.append(B * dims[2])
        elif spec.norm == pkg._lib.NORM_FULL:
            out.append(B * dims[0] * dims[1] * dims[2])
        else:
            out.append(B * np.prod([dims[a] - abs(o[a]) for a in range(3)]))
    return out


def window_vs_oracle(pkg, orc, spec, B, dims, e, t, w, m, affs, grad, b, lo, size, dloss, grad_tol=GRAD_RTOL):
    """oracle on the window [lo, lo + size) of image b; compare the interior (margin = the stencil's reach per axis)"""
    sl = tuple(slice(lo[a], lo[a] + size[a]) for a in range(3))
    def cut(x):
        if x is None:
            return None
        v = x[b][(slice(None),) + sl]
        return np.ascontiguousarray(v.cpu().numpy() if v.dtype == torch.uint8 else v.float().cpu().numpy())[None]

    ew, tw, ww, mw = cut(e), cut(t), cut(w), cut(m)
    S_win = float(np.prod(size))
    lam = [spec.lam[i] * S_win / n for i, n in enumerate(full_norms(pkg, spec, B, dims))]
    offs = [list(o) for o in spec.offsets]
    d = orc.make_desc(1, ew.shape[1], list(size), offs, lam, spec.border, orc.NORM_FULL, ndim=3)
    o_affs, _ = orc.c_fwd(d, ew, None, tw, ww, mw)
    o_grad, _ = orc.c_bwd(d, ew, None, tw, ww, mw, dloss=dloss)
    reach = [max(abs(o[a]) for o in offs) for a in range(3)]
    inner = tuple(slice(reach[a], size[a] - reach[a]) if size[a] > 2 * reach[a] else slice(0, size[a]) for a in range(3))
    # an axis the window covers completely (e.g. all of z) needs no margin: the window's border there IS the volume's border
    inner = tuple(slice(0, size[a]) if (lo[a] == 0 and size[a] == dims[a]) else inner[a] for a in range(3))
    a_hip = affs[b][(slice(None),) + sl].cpu().numpy()[(slice(None),) + inner]
    g_hip = grad[b][(slice(None),) + sl].float().cpu().numpy()[(slice(None),) + inner]
    a_orc = o_affs.reshape([len(offs)] + list(size))[(slice(None),) + inner]
    g_orc = o_grad.reshape([ew.shape[1]] + list(size))[(slice(None),) + inner]
    assert a_hip.size > 0 and np.abs(a_hip - a_orc).max() < AFFS_ATOL
    assert relmax(g_hip, g_orc) < grad_tol


def direct_on_window(pkg, spec, e, t, w, m, b, lo, size, monkeypatch, tol=1e-5):
    """the same window as its own small problem: tiled / cross kernels against the direct kernels (PEA_FORCE_DIRECT=1)"""
    axes = (0, 1, 2) if spec.ndim == 3 else (1, 2)
    sl = (slice(b, b + 1), slice(None)) + tuple(slice(lo[a], lo[a] + size[a]) for a in axes)
    args = [None if x is None else x[sl].contiguous() for x in (e, t, w, m)]
    monkeypatch.delenv("PEA_FORCE_DIRECT", raising=False)
    l1, a1, _, g1 = run(pkg, spec, *args)
    monkeypatch.setenv("PEA_FORCE_DIRECT", "1")
    l0, a0, _, g0 = run(pkg, spec, *args)
    monkeypatch.delenv("PEA_FORCE_DIRECT", raising=False)
    assert abs(l1.item() - l0.item()) <= 1e-5 * abs(l0.item())
    assert (a1 - a0).abs().max().item() < 2e-6
    assert relmax(g1.float().cpu().numpy(), g0.float().cpu().numpy()) < tol


def test_config2_bbbc_d32_704(pkg, dev, orc, monkeypatch):
    B, D, H, W = 8, 32, 704, 704
    offsets = pkg.multi_offset([1, 3, 5, 9, 11], 4)
    spec = pkg.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
    e, t, w, m = gpu_inputs(dev, B, D, [H, W], len(offsets), 1002)
    loss, affs, parts, grad = run(pkg, spec, e, t, w, m, dloss=0.5)
    loss2, affs2, parts2, grad2 = run(pkg, spec, e, t, w, m, dloss=0.5)
    assert torch.equal(loss, loss2) and torch.equal(affs, affs2) and torch.equal(grad, grad2) and torch.equal(parts, parts2)
    assert torch.isfinite(grad).all() and abs(loss.item() - parts.sum().item()) <= 1e-5 * abs(loss.item())
    # batch additivity: image 3 alone carries 1/8 of its share of the normaliser
    l3, a3, _, g3 = run(pkg, spec, e[3:4], t[3:4], w[3:4], m[3:4], dloss=0.5)
    assert torch.equal(a3[0], affs[3])
    assert relmax(g3[0].cpu().numpy() / B, grad[3].cpu().numpy()) < 1e-6
    e5, t5, w5, m5, a5, g5 = (x.unsqueeze(2) for x in (e, t, w, m, affs, grad))  # [B,C,1,H,W] for the window helper
    window_vs_oracle(pkg, orc, spec, B, [1, H, W], e5, t5, w5, m5, a5, g5, b=5, lo=[0, 300, 416], size=[1, 96, 160], dloss=0.5)
    window_vs_oracle(pkg, orc, spec, B, [1, H, W], e5, t5, w5, m5, a5, g5, b=0, lo=[0, 608, 544], size=[1, 96, 160], dloss=0.5)
    direct_on_window(pkg, spec, e, t, w, m, b=2, lo=[0, 128, 256], size=[1, 128, 192], monkeypatch=monkeypatch)


@pytest.mark.parametrize("stencil", ["norm5_k12", "n26"])
def test_config3_ac3ac4_24x1024x1024(pkg, dev, orc, monkeypatch, stencil):
    B, D, Z, Y, X = 1, 16, 24, 1024, 1024
    if stencil == "norm5_k12":
        offs = [list(o) for o in pkg.utils.affinity_ours.axis_offsets_3d(pkg.utils.affinity_ours.NORM5_SHIFTS)]
        lam = [2.0] * 3 + [1.0] * 9
    else:
        offs = [[dz, dy, dx] for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dz, dy, dx) != (0, 0, 0)]
        lam = [1.0 + 0.125 * (i % 4) for i in range(26)]
    K = len(offs)
    spec = pkg.AffinitySpec(3, offs, lam, pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED)
    e, t, w, _ = gpu_inputs(dev, B, D, [Z, Y, X], K, 1003 + K, mask=False)
    loss, affs, parts, grad = run(pkg, spec, e, t, w, None, dloss=2.0)
    loss2, affs2, parts2, grad2 = run(pkg, spec, e, t, w, None, dloss=2.0)
    assert torch.equal(loss, loss2) and torch.equal(grad, grad2) and torch.equal(affs, affs2)
    del affs2, grad2
    assert torch.isfinite(grad).all()
    # cropped-away neighbours: the border slices stay exactly 0
    for i, o in enumerate(offs):
        for a in range(3):
            if o[a] != 0:
                sl = [slice(None)] * 5
                sl[1] = i
                sl[2 + a] = slice(0, -o[a]) if o[a] < 0 else slice(affs.shape[2 + a] - o[a], None)
                assert (affs[tuple(sl)] == 0).all()
    # windows: an interior block over all of z, and one that touches the volume's y / x = 0 faces (the crop itself)
    window_vs_oracle(pkg, orc, spec, B, [Z, Y, X], e, t, w, None, affs, grad, b=0, lo=[0, 480, 640], size=[24, 80, 96], dloss=2.0)
    sl0 = (slice(0, 1), slice(None), slice(0, 24), slice(0, 96), slice(0, 128))
    # the corner window IS a corner of the volume on three faces: compare only where the window's own far faces do not matter
    ew, tw, ww = (np.ascontiguousarray(x[sl0].cpu().numpy()) for x in (e, t, w))
    reach = [max(abs(o[a]) for o in offs) for a in range(3)]
    norms = full_norms(pkg, spec, B, [Z, Y, X])
    lam_w = [lam[i] * (24.0 * 96 * 128) / n for i, n in enumerate(norms)]
    d = orc.make_desc(1, D, [24, 96, 128], offs, lam_w, spec.border, orc.NORM_FULL, ndim=3)
    o_affs, _ = orc.c_fwd(d, ew, None, tw, ww, None)
    o_grad, _ = orc.c_bwd(d, ew, None, tw, ww, None, dloss=2.0)
    inner = (slice(None), slice(0, 24), slice(0, 96 - reach[1]), slice(0, 128 - reach[2]))
    assert np.abs(affs[sl0][0].cpu().numpy()[inner] - o_affs.reshape(K, 24, 96, 128)[inner]).max() < AFFS_ATOL
    assert relmax(grad[sl0][0].cpu().numpy()[inner], o_grad.reshape(D, 24, 96, 128)[inner]) < GRAD_RTOL
    direct_on_window(pkg, spec, e, t, w, None, b=0, lo=[0, 256, 512], size=[24, 128, 160], monkeypatch=monkeypatch)


def test_config4_d64_f16_544(pkg, dev, orc, monkeypatch):
    B, D, H, W = 8, 64, 544, 544
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)[:8]
    spec = pkg.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
    e, t, w, m = gpu_inputs(dev, B, D, [H, W], len(offsets), 1004, f16=True)
    loss, affs, parts, grad = run(pkg, spec, e, t, w, m)
    loss2, affs2, parts2, grad2 = run(pkg, spec, e, t, w, m)
    assert grad.dtype == torch.float16
    assert torch.equal(loss, loss2) and torch.equal(affs, affs2) and torch.equal(grad, grad2)
    assert torch.isfinite(grad.float()).all()
    l6, a6, _, g6 = run(pkg, spec, e[6:7], t[6:7], w[6:7], m[6:7])
    assert torch.equal(a6[0], affs[6])
    assert relmax(g6[0].float().cpu().numpy() / B, grad[6].float().cpu().numpy()) < 2e-3  # both stored as f16
    e5, t5, w5, m5, a5, g5 = (x.unsqueeze(2) for x in (e, t, w, m, affs, grad))
    window_vs_oracle(pkg, orc, spec, B, [1, H, W], e5, t5, w5, m5, a5, g5, b=4, lo=[0, 224, 320], size=[1, 96, 128], dloss=1.0, grad_tol=2e-3)
    direct_on_window(pkg, spec, e, t, w, m, b=1, lo=[0, 64, 96], size=[1, 96, 160], monkeypatch=monkeypatch, tol=2e-3)
