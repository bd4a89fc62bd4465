"""GPU (-m gpu): full-size (B = 8 x 544 x 544, the headline batch) tests of the LDS-DMA kernel families that round 4 only tested at sizes
the oracle finishes in seconds: the EMA cross loss (k_fwd_xdma<.., OTHER>, k_bwd_xdma<.., OTHER>), the full-resolution pair's backward
(k_bwd_xdma<.., DUAL>) inside the one-node loss section, the labels-in forward (k_fwd_xdma<.., LAB>), the one-launch labels-in steps
(k_fused_labels, k_fused_labels_dual), both loss sections, and the embedding head.

Same pattern as test_gpu_fullsize.py: (1) two runs on the same inputs agree BIT FOR BIT -- the only test shape that has ever caught a
hand-off race of an LDS-DMA ring (invisible to every tolerance test); (2) a window of the full-size result
against the CPU oracle run on the window alone (margin = the stencil's reach); (3) the paths that must agree with each other do.

Tolerances: affs abs 1e-5, grads rel-to-max 1e-4 (test_gpu_parity.py)."""
import ctypes
import importlib

import numpy as np
import pytest
import torch

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu
AFFS_ATOL, GRAD_RTOL = 1e-5, 1e-4
B, D, H, W = 8, 16, 544, 544
SHIFTS = [1, 3, 5, 9, 27]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def synth():
    ge.load_package()
    return importlib.import_module(ge.PKG_NAME + ".utils.synth")


def relmax(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def npy(x):
    return np.ascontiguousarray(x.detach().float().cpu().numpy() if x.dtype != torch.uint8 else x.cpu().numpy())


def window_oracle_2d(orc, offsets, lam, nfull, b, y0, x0, hh, ww, e, eo, t, w, m, dloss):
    """oracle (affs, d loss / d e) on the window [y0, y0 + hh) x [x0, x0 + ww) of image b, with the window's per-offset weights set so
    that its normaliser equals the full problem's (N_i = nfull); CROP semantics inside a window do not matter: only the interior
    (margin = reach) is compared"""
    cut = lambda x: None if x is None else npy(x[b:b + 1, :, y0:y0 + hh, x0:x0 + ww])
    ew, ow, tw, wt, mw = cut(e), cut(eo), cut(t), cut(w), cut(m)
    K = len(offsets)
    lam_w = [lam[i] * float(hh * ww) / nfull for i in range(K)]
    d = orc.make_desc(1, ew.shape[1], [1, hh, ww], [[0, o[0], o[1]] for o in offsets], lam_w, orc.BORDER_CIRCULAR, orc.NORM_FULL, ndim=3)
    a, _ = orc.c_fwd(d, ew, ow, tw, wt, mw)
    g, _ = orc.c_bwd(d, ew, ow, tw, wt, mw, dloss=dloss)
    return a.reshape(K, hh, ww), g.reshape(ew.shape[1], hh, ww)


def inputs(dev, seed, K):
    g = torch.Generator(device=dev).manual_seed(seed)
    e = torch.randn([B, D, H, W], generator=g, device=dev)
    t = (torch.rand([B, K, H, W], generator=g, device=dev) < 0.6).float()
    w = torch.rand([B, K, H, W], generator=g, device=dev) + 0.5
    m = (torch.rand([B, K, H, W], generator=g, device=dev) < 0.9).to(torch.uint8)
    return e, t, w, m


def test_full_size_ema_cross_loss(pkg, dev, orc):
    """ema_embedding_loss (scripts_cvppp/loss/loss_embedding_mse.py:79-95, the EMA operand detached) at B = 8 x 16 x 544^2 on the role-A
    cross kernels: bit-identical reruns, two windows against the oracle"""
    offsets = pkg.multi_offset(SHIFTS, 4)
    K = len(offsets)
    lam = [2.0 if i < 2 else 1.0 for i in range(K)]  # affs0_weight = 2 on the first two offsets (:90-91)
    spec = pkg.AffinitySpec(2, offsets, lam, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
    e, t, w, m = inputs(dev, 2001, K)
    ema = torch.randn([B, D, H, W], generator=torch.Generator(device=dev).manual_seed(2002), device=dev)
    desc = pkg.affinity_op.make_desc(spec, e)
    assert pkg._lib.lib().pea_cross_supported(ctypes.byref(desc), 2) == 1, "the role-A cross kernels take the headline shape"

    def run():
        et = e.clone().requires_grad_(True)
        loss, affs, parts = pkg.affinity_op.FusedAffinityMSE.apply(et, ema, t, w, m, spec)
        (loss * 0.5).backward()
        return loss.detach().clone(), affs, parts.clone(), et.grad

    l1, a1, p1, g1 = run()
    l2, a2, p2, g2 = run()
    assert torch.equal(l1, l2) and torch.equal(a1, a2) and torch.equal(p1, p2) and torch.equal(g1, g2)
    assert torch.isfinite(g1).all()
    reach = max(SHIFTS)
    for (b, y0, x0) in ((5, 300, 64), (0, 0, 416)):  # (the second window touches the image's top border: its first rows are not compared)
        hh, ww = 96, 128
        oa, og = window_oracle_2d(orc, offsets, lam, float(B * W), b, y0, x0, hh, ww, e, ema, t, w, m, 0.5)
        inner = (slice(None), slice(reach, hh - reach), slice(reach, ww - reach))
        assert np.abs(npy(a1[b, :, y0:y0 + hh, x0:x0 + ww])[inner] - oa[inner]).max() < AFFS_ATOL
        assert relmax(npy(g1[b, :, y0:y0 + hh, x0:x0 + ww])[inner], og[inner]) < GRAD_RTOL


@pytest.mark.parametrize("Dm,f16,shifts,K", [(32, False, [1, 3, 5, 9, 11], 10), (64, True, [1, 3, 5, 9, 27], 8)])
def test_full_size_wide_ema_cross_loss(pkg, dev, orc, Dm, f16, shifts, K):
    """ema_embedding_loss at D = 32 (f32, the BBBC stencil) and D = 64 (f16 storage, offsets[:8]) at B = 8 x 544^2 -- the own-tile
    kernels of round 5 (k_fwd_xdma<.., OWNL> / k_bwd_xdma_pfo; k_fwd_xdma_h / k_bwd_xdma_h<.., OTHER>): bit-identical reruns and a
    window against the oracle"""
    offsets = pkg.multi_offset(shifts, 4)[:K]
    lam = [2.0 if i < 2 else 1.0 for i in range(K)]
    spec = pkg.AffinitySpec(2, offsets, lam, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
    g = torch.Generator(device=dev).manual_seed(3001)
    e = torch.randn([B, Dm, H, W], generator=g, device=dev)
    ema = torch.randn([B, Dm, H, W], generator=g, device=dev)
    if f16:
        e, ema = e.half(), ema.half()
    t = (torch.rand([B, K, H, W], generator=g, device=dev) < 0.6).float()
    w = torch.rand([B, K, H, W], generator=g, device=dev) + 0.5
    m = (torch.rand([B, K, H, W], generator=g, device=dev) < 0.9).to(torch.uint8)
    desc = pkg.affinity_op.make_desc(spec, e)
    L = pkg._lib.lib()
    assert L.pea_cross_supported(ctypes.byref(desc), 2) == 1 and L.pea_cross_supported(ctypes.byref(desc), 4) == 1

    def run():
        et = e.clone().requires_grad_(True)
        loss, affs, parts = pkg.affinity_op.FusedAffinityMSE.apply(et, ema, t, w, m, spec)
        (loss * 0.5).backward()
        return loss.detach().clone(), affs, et.grad

    l1, a1, g1 = run()
    l2, a2, g2 = run()
    assert torch.equal(l1, l2) and torch.equal(a1, a2) and torch.equal(g1, g2)
    assert torch.isfinite(g1.float()).all()
    reach, hh, ww = max(shifts[:(K + 1) // 2]), 96, 128
    b, y0, x0 = 6, 416, 384
    oa, og = window_oracle_2d(orc, offsets, lam, float(B * W), b, y0, x0, hh, ww, e, ema, t, w, m, 0.5)
    inner = (slice(None), slice(reach, hh - reach), slice(reach, ww - reach))
    assert np.abs(npy(a1[b, :, y0:y0 + hh, x0:x0 + ww])[inner] - oa[inner]).max() < AFFS_ATOL
    assert relmax(npy(g1[b, :, y0:y0 + hh, x0:x0 + ww])[inner], og[inner]) < (2e-3 if f16 else GRAD_RTOL)


def _section_inputs(pkg, dev, synth):
    offsets = pkg.multi_offset(SHIFTS, 4)
    nb_half = 2
    labs_np = synth.synth_labels(B, (1, H, W), 777)[:, 0]
    labs = [torch.from_numpy(np.ascontiguousarray(labs_np[:, ::2 ** j, ::2 ** j])).to(dev) for j in range(5)]
    embs = [torch.from_numpy(synth.synth_embedding((B, D, H >> j, W >> j), 800 + j)).to(dev) for j in range(5)]
    ema = torch.from_numpy(synth.synth_embedding((B, D, H, W), 900)).to(dev)
    tt, mm, ww = pkg.gen_targets(labs[0], offsets, padding=True)
    downs, small = [], []
    for j in range(1, 5):
        k = nb_half * (5 - j)
        tj, mj, wj = pkg.gen_targets(labs[j], offsets[:k], padding=True)
        downs.append(torch.cat([tj, wj, mj.float()], dim=1))
        small.append((tj, wj, mj))
    return offsets, nb_half, labs, embs, ema, tt, ww, mm, downs, small


def test_full_size_loss_sections(pkg, dev, orc, synth, monkeypatch):
    """the training loop's loss section (scripts_cvppp/main.py:284-310) at B = 8 x 544^2: the one-node tensor path (its full-resolution
    pair's backward is ONE launch, k_bwd_xdma<.., DUAL>) and the labels-in section (k_fwd_xdma<.., LAB>, k_fused_labels_dual for the
    pair where it applies): bit-identical reruns of every output; the full-resolution gradient and map in a window against the oracle
    (self + cross); the two small scales that the oracle finishes in seconds in full; the two sections against each other.  The tensor
    section's full-resolution pair runs its FORWARD as one launch too (k_fwd_xdma_dual, pea_affinity_fwd_dual_ex): the same section
    with PEA_FWD_DUAL=0 (two forward launches) must give every output bit for bit"""
    offsets, nb_half, labs, embs, ema, tt, ww, mm, downs, small = _section_inputs(pkg, dev, synth)
    crit = pkg.WeightedMSE()
    K = len(offsets)

    def run(which):
        x = [e.clone().requires_grad_(True) for e in embs]
        if which == "labels":
            loss, pred, _ = pkg.cvppp_loss_section_from_labels(x[0], x[1:], ema, labs[0], labs[1:], crit, offsets, nb_half)
        else:
            loss, pred, _ = pkg.cvppp_loss_section(x[0], x[1:], ema, tt, ww, mm, downs, crit, offsets, nb_half)
        (loss * 0.5).backward()
        return loss.detach().clone(), pred, [v.grad for v in x]

    res = {}
    for which in ("one_node", "labels"):
        l1, p1, g1 = run(which)
        l2, p2, g2 = run(which)
        assert torch.equal(l1, l2) and torch.equal(p1, p2), which
        for a, b in zip(g1, g2):
            assert torch.equal(a, b), which
        res[which] = (l1, p1, g1)
    l, pred, grads = res["one_node"]
    for sw in ("0",):  # two forward launches: the same bits as the one-launch pair
        monkeypatch.setenv("PEA_FWD_DUAL", sw)
        pkg._lib.reload_env()
        try:
            l0, p0, g0 = run("one_node")
        finally:
            monkeypatch.delenv("PEA_FWD_DUAL")
            pkg._lib.reload_env()
        assert torch.equal(l0, l) and torch.equal(p0, pred), sw
        for a, b in zip(g0, grads):
            assert torch.equal(a, b), sw
    # full resolution: d section / d embedding = 0.5 * (self gradient + cross gradient), all loss weights 1; affs0_weight 1
    reach, hh, wd = max(SHIFTS), 96, 128
    inner = (slice(None), slice(reach, hh - reach), slice(reach, wd - reach))
    for (b, y0, x0) in ((6, 200, 320), (1, 448, 0)):
        oa, og_self = window_oracle_2d(orc, offsets, [1.0] * K, float(B * W), b, y0, x0, hh, wd, embs[0], None, tt, ww, mm, 0.5)
        _, og_cross = window_oracle_2d(orc, offsets, [1.0] * K, float(B * W), b, y0, x0, hh, wd, embs[0], ema, tt, ww, mm, 0.5)
        assert np.abs(npy(pred[b, :, y0:y0 + hh, x0:x0 + wd])[inner] - oa[inner]).max() < AFFS_ATOL
        assert relmax(npy(grads[0][b, :, y0:y0 + hh, x0:x0 + wd])[inner], (og_self + og_cross)[inner]) < GRAD_RTOL
    # the two coarsest scales (68^2, K = 4 and 34^2, K = 2) whole, against the oracle
    for j in (3, 4):
        k = nb_half * (5 - j)
        tj, wj, mj = small[j - 1]
        d = orc.desc_2d(npy(embs[j]), offsets[:k])
        og, _ = orc.c_bwd(d, npy(embs[j]), None, npy(tj), npy(wj), npy(mj), dloss=0.5)
        assert relmax(npy(grads[j]), og) < GRAD_RTOL, j
    # labels-in section against the tensor section (the same targets, generated inside the kernels)
    ll, pl, gl = res["labels"]
    assert abs(ll.item() - l.item()) <= 3e-6 * abs(l.item())
    assert (pl - pred).abs().max().item() < 2e-6
    for a, b in zip(gl, grads):
        assert relmax(npy(a), npy(b)) < 1e-5


@pytest.mark.parametrize("form", ["two_launch", "one_launch", "one_launch_ema"])
def test_full_size_labels_steps(pkg, dev, synth, monkeypatch, form):
    """embedding_loss_from_labels / ema_embedding_loss_from_labels at B = 8 x 544^2: the two-launch form (k_fwd_xdma<.., LAB> + the cross
    backward) and the one-launch kernel (k_fused_labels; with the EMA operand too): bit-identical reruns, and against pea_gen_targets +
    the tensor path (window-checked against the oracle by test_gpu_parity.py / test_gpu_fullsize.py)"""
    monkeypatch.setattr(pkg.affinity_op, "LABELS_TWO_LAUNCH_MIN_PX", 0 if form == "two_launch" else 1 << 62)
    offsets = pkg.multi_offset(SHIFTS, 4)
    lab = torch.from_numpy(synth.synth_labels(B, (1, H, W), 1777)[:, 0].copy()).to(dev)
    e = torch.from_numpy(synth.synth_embedding((B, D, H, W), 1800)).to(dev)
    ema = torch.from_numpy(synth.synth_embedding((B, D, H, W), 1900)).to(dev) if form.endswith("ema") else None
    crit = pkg.WeightedMSE()
    t, m, w = pkg.gen_targets(lab, offsets, padding=True)

    def run(labels_in):
        et = e.clone().requires_grad_(True)
        if ema is not None:
            loss, affs = (pkg.ema_embedding_loss_from_labels(et, ema, lab, crit, offsets, affs0_weight=2) if labels_in else
                          pkg.ema_embedding_loss(et, ema, t, w, m, crit, offsets, affs0_weight=2))
        else:
            loss, affs, _ = (pkg.embedding_loss_from_labels(et, lab, crit, offsets) if labels_in else
                             pkg.embedding_loss(et, t, w, m, crit, offsets))
        (loss * 0.5).backward()
        return loss.detach().clone(), affs, et.grad

    l1, a1, g1 = run(True)
    l2, a2, g2 = run(True)
    assert torch.equal(l1, l2) and torch.equal(a1, a2) and torch.equal(g1, g2)
    l0, a0, g0 = run(False)
    assert abs(l1.item() - l0.item()) <= 2e-6 * abs(l0.item())
    assert (a1 - a0).abs().max().item() < 2e-6
    assert relmax(npy(g1), npy(g0)) < 1e-5


def test_full_size_head(pkg, dev, orc):
    """the embedding head (OutConv 32 -> 16, scripts_cvppp/model/unet2d_residual.py:67-74) at B = 8 x 544^2: forward, dx, dW, db twice, bit
    for bit; one image's rows against the float64 restatement"""
    L = pkg._lib.lib()
    C = 32
    g = torch.Generator(device=dev).manual_seed(31)
    X = torch.randn(B, C, H, W, generator=g, device=dev)
    Wt = torch.randn(D, C, generator=g, device=dev) * 0.2
    Bt = torch.randn(D, generator=g, device=dev)
    UP = torch.randn(B, D, H, W, generator=g, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    wsb = L.pea_head_workspace_bytes(C, D)
    outs = []
    for _ in range(2):
        E = torch.empty(B, D, H, W, device=dev)
        dX, dW, dB = torch.empty_like(X), torch.empty(D, C, device=dev), torch.empty(D, device=dev)
        work = torch.empty(wsb // 4, device=dev)
        assert L.pea_head_fwd(B, C, D, H * W, p(X), p(Wt), p(Bt), p(E), st) == 0
        assert L.pea_head_bwd(B, C, D, H * W, p(X), p(Wt), p(UP), p(dX), p(dW), p(dB), p(work), wsb, st) == 0
        torch.cuda.synchronize()
        outs.append((E, dX, dW, dB))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    E, dX, dW, dB = outs[0]
    rows = slice(200, 232)
    e_ref = orc.np_head_fwd(npy(X[3:4, :, rows]), npy(Wt), npy(Bt))
    assert np.abs(npy(E[3:4, :, rows]) - e_ref).max() <= 1e-5 * np.abs(e_ref).max()
    dx_ref, _, _ = orc.np_head_bwd(npy(X[3:4, :, rows]), npy(Wt), npy(UP[3:4, :, rows]))
    assert relmax(npy(dX[3:4, :, rows]), dx_ref) <= 1e-5
    # dW / db over the whole batch in float64 on the GPU (2.4 M products per entry)
    dw_ref = torch.einsum("bdhw,bchw->dc", UP.double(), X.double())
    db_ref = UP.double().sum(dim=(0, 2, 3))
    assert relmax(npy(dW), dw_ref.cpu().numpy()) <= 3e-5 and relmax(npy(dB), db_ref.cpu().numpy()) <= 3e-5
