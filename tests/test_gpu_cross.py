"""GPU (-m gpu): the LDS-DMA cross kernels (csrc/pea_xdma.h: k_fwd_xdma, k_bwd_xdma, k_inv_norm) against the CPU oracle,
against the tiled kernels they replace, and through the new C-ABI entry points (pea_affinity_fwd_ex / pea_affinity_bwd_ex /
pea_inv_norm).  The reference goldens sized for them (g2d_x_k10, g2d_x_k8) run in test_gpu_parity.py.

Tolerances as in test_gpu_parity.py: affs abs 1e-5, loss rel 1e-5, grads rel-to-max 1e-4."""
import ctypes
import importlib

import numpy as np
import pytest
import torch

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu
AFFS_ATOL, LOSS_RTOL, GRAD_RTOL = 1e-5, 1e-5, 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def synth():
    ge.load_package()
    return importlib.import_module(ge.PKG_NAME + ".utils.synth")


def cu(a, dev):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def relmax(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _inputs(synth, B, D, dims, K, seed, zero_px=False):
    S = int(np.prod(dims))
    e = synth.synth_embedding((B, D, S), 100 + seed).reshape([B, D] + dims)
    if zero_px:
        e[0, :, 0, 3, 5] = 0.0
        e[-1, :, 0, dims[1] - 1, dims[2] - 1] = 1e-14
    idx = np.arange(B * K * S, dtype=np.uint64)
    t = (synth.hash_uniform(idx, 200 + seed) < 0.6).astype(np.float32).reshape([B, K] + dims)
    w = (0.5 + synth.hash_uniform(idx, 300 + seed)).astype(np.float32).reshape([B, K] + dims)
    m = (synth.hash_uniform(idx, 400 + seed) < 0.9).astype(np.uint8).reshape([B, K] + dims)
    return e, t, w, m


CASES = [
    # (B, Y, X, shifts, K, border, norm, mask, relu, dloss)          border 0 CIRCULAR / 1 CROP_ZERO; norm 0 BX / 1 CROPPED / 2 FULL
    (2, 50, 100, [1, 3, 5, 9, 27], 10, 0, 0, True, False, 1.0),       # shipped stencil, ragged tiles in y and x, 64-pixel strip rows
    (1, 43, 96, [1, 3, 5, 9, 27], 10, 0, 0, False, True, 0.5),        # the smallest image the +-27 cross accepts; relu; no mask
    (3, 64, 128, [1, 3, 5, 9, 27], 10, 1, 1, True, False, 2.0),       # CROP_ZERO + cropped normaliser (the 3D path's border in 2D)
    (2, 37, 72, [1, 3, 5, 9, 11], 8, 0, 0, True, False, 1.0),         # reach 9: 32-pixel strip rows
    (2, 48, 64, [1, 3, 5, 9, 11], 10, 0, 2, True, False, 0.25),       # reach 11, minimum width, PEA_NORM_FULL
    (1, 80, 160, [2, 4, 16], 6, 1, 2, False, False, 1.0),             # even shifts, reach 16 (the widest 32-pixel strip row)
    (2, 33, 68, [1], 2, 0, 0, True, False, 1.0),                      # K = 2
    (1, 96, 200, [1, 3, 5, 9, 27], 9, 0, 0, True, False, 1.0),        # odd K: x reaches 27, y only 9 (asymmetric cross)
    (2, 48, 96, [1, 3, 5, 9, 11], 10, 0, 0, True, False, 0.5, 32),    # D = 32 (BBBC039V1 stencil): 16 chunks, own pixel re-read at the end
    (1, 50, 100, [1, 3, 5, 9, 27], 8, 0, 0, True, True, 1.0, 64),     # D = 64, offsets[:8]
    (1, 64, 128, [1, 3, 5, 9, 27], 8, 1, 1, False, False, 2.0, 32),   # D = 32, CROP_ZERO
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_cross_kernels_vs_oracle(pkg, dev, orc, synth, case):
    B, Y, X, shifts, K, border, norm, use_mask, relu, dloss = CASES[case][:10]
    D, dims = (CASES[case][10] if len(CASES[case]) > 10 else 16), [1, Y, X]
    offs = [[0] + o for o in pkg.multi_offset(shifts, 4)][:K]
    lam = [1.0 + 0.25 * (i % 3) for i in range(K)]
    e, t, w, m = _inputs(synth, B, D, dims, K, 7 * case, zero_px=(case == 0))
    op, L = pkg.affinity_op, pkg._lib.lib()
    spec = op.AffinitySpec(3, offs, lam, border, norm, relu=relu)
    et = cu(e, dev).requires_grad_(True)
    d_hip = op.make_desc(spec, et.detach())
    assert L.pea_cross_supported(ctypes.byref(d_hip), 0) == 1 and L.pea_cross_supported(ctypes.byref(d_hip), 1) == 1
    loss, affs, parts = op.FusedAffinityMSE.apply(et, None, cu(t, dev), cu(w, dev), cu(m, dev) if use_mask else None, spec)
    (loss * dloss).backward()
    d = orc.make_desc(B, D, dims, offs, lam, border, norm, flags=orc.FLAG_RELU if relu else 0, ndim=3)
    o_affs, o_loss = orc.c_fwd(d, e, None, t, w, m if use_mask else None)
    o_grad, _ = orc.c_bwd(d, e, None, t, w, m if use_mask else None, dloss=dloss)
    assert np.abs(affs.cpu().numpy().reshape(o_affs.shape) - o_affs).max() < AFFS_ATOL
    assert abs(loss.item() - o_loss[0]) <= LOSS_RTOL * max(abs(o_loss[0]), 1e-6)
    np.testing.assert_allclose(parts.cpu().numpy(), o_loss[1:], rtol=LOSS_RTOL, atol=1e-9)
    assert relmax(et.grad.cpu().numpy(), o_grad) < GRAD_RTOL
    inf = op.affinity_infer(et.detach(), None, spec)
    assert np.abs(inf.cpu().numpy().reshape(o_affs.shape) - o_affs).max() < AFFS_ATOL


def test_cross_and_tiled_kernels_agree_and_are_reproducible(pkg, dev, synth, monkeypatch):
    """the same call through the cross kernels and (PEA_FWD_XDMA=0 PEA_BWD_XDMA=0) through the tiled kernels; twice each"""
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    B, D, H, W = 3, 16, 112, 160
    e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 61)

    def run():
        et = cu(e, dev).requires_grad_(True)
        loss, affs, parts = pkg.embedding_loss(et, cu(t, dev), cu(w, dev), cu(m, dev), pkg.WeightedMSE(), offsets)
        (loss * 1.5).backward()
        return loss.item(), affs.cpu().numpy(), et.grad.cpu().numpy(), np.array(list(parts))

    monkeypatch.delenv("PEA_FWD_XDMA", raising=False)
    monkeypatch.delenv("PEA_BWD_XDMA", raising=False)
    l1, a1, g1, p1 = run()
    l1b, a1b, g1b, p1b = run()
    assert l1 == l1b and np.array_equal(a1, a1b) and np.array_equal(g1, g1b) and np.array_equal(p1, p1b)  # no atomics anywhere
    monkeypatch.setenv("PEA_FWD_XDMA", "0")
    monkeypatch.setenv("PEA_BWD_XDMA", "0")
    l0, a0, g0, p0 = run()
    assert abs(l1 - l0) <= 1e-6 * abs(l0)
    assert np.abs(a1 - a0).max() < 2e-6
    np.testing.assert_allclose(p1, p0, rtol=1e-6)
    assert relmax(g1, g0) < 1e-5


def test_inv_norm_plane_and_ex_entry_points(pkg, dev, orc, synth):
    """pea_inv_norm against numpy (sign = the clamp branch of F.normalize); the plane pea_affinity_fwd_ex writes; and
    pea_affinity_bwd_ex with and without it (cross kernel / tiled kernel) against the oracle, through the C ABI directly"""
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    B, D, H, W = 2, 16, 64, 128
    K = len(offsets)
    e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 77)
    e[0, :, 5, 7] = 0.0
    e[1, :, 63, 127] = 3e-14
    op, L = pkg.affinity_op, pkg._lib.lib()
    E, T, Wt, M = cu(e, dev), cu(t, dev), cu(w, dev), cu(m, dev)
    desc = op.make_desc(op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX), E)
    P = lambda x: ctypes.c_void_p(x.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    inv0 = torch.empty(B, H, W, device=dev)
    assert L.pea_inv_norm(ctypes.byref(desc), P(E), P(inv0), st) == 0
    nrm = np.sqrt((e.astype(np.float64) ** 2).sum(1))
    ref = np.where(nrm < 1e-12, -1e12, 1.0 / np.maximum(nrm, 1e-12))
    np.testing.assert_allclose(inv0.cpu().numpy(), ref, rtol=3e-7)
    assert (inv0.cpu().numpy() < 0).sum() == 2
    affs, g, inv1 = torch.empty(B, K, H, W, device=dev), torch.empty(B, K, H, W, device=dev), torch.full((B, H, W), 7.0, device=dev)
    lossv = torch.empty(1 + K, device=dev)
    wsb = L.pea_workspace_bytes(ctypes.byref(desc))
    work = torch.empty(max(wsb, 4) // 4, device=dev)
    assert L.pea_workspace_init(ctypes.c_void_p(work.data_ptr()), wsb, None) == 0  # the loss-state block: prepared once
    assert L.pea_affinity_fwd_ex(ctypes.byref(desc), P(E), None, P(T), P(Wt), P(M), P(affs), P(g), P(inv1), P(lossv), P(work), wsb, st) == 0
    np.testing.assert_allclose(inv1.cpu().numpy(), ref, rtol=3e-7)
    d = orc.desc_2d(e, offsets)
    o_grad, _ = orc.c_bwd(d, e, None, t, w, m, dloss=0.75)
    dl = torch.full((), 0.75, device=dev)
    de_x, de_t = torch.empty_like(E), torch.empty_like(E)
    assert L.pea_affinity_bwd_ex(ctypes.byref(desc), P(E), None, P(g), P(inv1), P(dl), P(de_x), None, st) == 0
    assert L.pea_affinity_bwd_ex(ctypes.byref(desc), P(E), None, P(g), None, P(dl), P(de_t), None, st) == 0
    assert relmax(de_x.cpu().numpy(), o_grad) < GRAD_RTOL and relmax(de_t.cpu().numpy(), o_grad) < GRAD_RTOL
    # the zero-norm pixels take the clamp branch: d ehat / d e = I / eps, no projection
    assert relmax(de_x.cpu().numpy()[0, :, 5, 7], o_grad[0, :, 5, 7]) < GRAD_RTOL


@pytest.mark.parametrize("shape", [(2, 6, 48, 96), (1, 5, 64, 132), (1, 9, 43, 96)])
def test_cross_3d_norm5_vs_oracle(pkg, dev, orc, synth, shape):
    """the AC3/AC4 stencil (embedding_loss_norm5: 12 axis offsets, z 1-4, y / x 1, 3, 9, 27; CROP_ZERO, cropped normaliser) through the
    3D instantiations of the cross kernels: in-plane offsets from LDS, z offsets gathered per chunk from the neighbouring planes"""
    B, Z, Y, X = shape
    sh = [1, 1, 1, 2, 3, 3, 3, 9, 9, 4, 27, 27]
    offs = orc.norm_offsets(sh)
    e, t, w = synth.synth_inputs_3d(B, 16, Z, Y, X, offs, 19 + Z)
    L = pkg._lib.lib()
    et = cu(e, dev).requires_grad_(True)
    spec = pkg.AffinitySpec(3, offs, orc.affs0_lambda_3d(12, 2, 3), pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED)
    d_hip = pkg.affinity_op.make_desc(spec, et.detach())
    assert L.pea_cross_supported(ctypes.byref(d_hip), 0) == 1 and L.pea_cross_supported(ctypes.byref(d_hip), 1) == 1
    loss, affs = pkg.embedding_loss_norm5(et, cu(t, dev), cu(w, dev), pkg.WeightedMSE(), affs0_weight=2)
    (loss * 0.25).backward()
    d = orc.desc_3d(e, sh, orc.affs0_lambda_3d(12, 2, 3))
    o_affs, o_loss = orc.c_fwd(d, e, None, t, w, None)
    o_grad, _ = orc.c_bwd(d, e, None, t, w, None, dloss=0.25)
    assert np.abs(affs.cpu().numpy() - o_affs).max() < AFFS_ATOL
    assert abs(loss.item() - o_loss[0]) <= LOSS_RTOL * o_loss[0]
    assert relmax(et.grad.cpu().numpy(), o_grad) < GRAD_RTOL


@pytest.mark.parametrize("shape,which", [((2, 6, 48, 96), "norm5"), ((1, 5, 64, 132), "norm5"), ((1, 9, 43, 96), "norm5"), ((2, 7, 48, 64), "norm1"),
                                         ((1, 4, 40, 72), "norm1_s2")])
def test_cross_3d_ema_vs_oracle(pkg, dev, orc, synth, monkeypatch, shape, which):
    """ema_embedding_loss_norm5 / norm1 (scripts_ac3ac4/loss/loss_embedding_mse.py:30-51, 237-289; the EMA operand is the shifted-from one,
    detached) on the role-A cross kernels' 3D instantiations (round 5: second operand staged, its z neighbours gathered per chunk, own
    pixel from e; CROP_ZERO): forward, both 1 / norm planes, backward against the oracle; and against the tiled kernels they replace"""
    B, Z, Y, X = shape
    sh = [1, 1, 1, 2, 3, 3, 3, 9, 9, 4, 27, 27] if which == "norm5" else ([2, 2, 2] if which == "norm1_s2" else [1, 1, 1])
    first = 3 if which == "norm5" else 1
    offs = orc.norm_offsets(sh)
    e, t, w = synth.synth_inputs_3d(B, 16, Z, Y, X, offs, 55 + Z)
    eo = synth.synth_embedding((B, 16, Z * Y * X), 66 + Z).reshape(e.shape)
    e[0, :, Z // 2, 3, 5] = 0.0
    eo[0, :, 1, 9, 11] = 0.0          # zero-norm pixels in both operands (the clamp branch)
    lam = orc.affs0_lambda_3d(len(sh), 2, first)
    spec = pkg.AffinitySpec(3, offs, lam, pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED)
    crit = pkg.WeightedMSE()
    E, EO, T, Wt = cu(e, dev), cu(eo, dev), cu(t, dev), cu(w, dev)
    L = pkg._lib.lib()
    d_hip = pkg.affinity_op.make_desc(spec, E)
    assert L.pea_cross_supported(ctypes.byref(d_hip), 2) == 1

    def run():
        x = E.clone().requires_grad_(True)
        if which == "norm5":
            loss, a = pkg.ema_embedding_loss_norm5(x, EO, T, Wt, crit, affs0_weight=2)
        else:
            loss, a = pkg.ema_embedding_loss_norm1(x, EO, T, Wt, crit, affs0_weight=2, shift=sh[0])
        (loss * 0.25).backward()
        return loss.item(), a.cpu().numpy(), x.grad.cpu().numpy()

    l1, a1, g1 = run()
    d = orc.desc_3d(e, sh, lam)
    o_affs, o_loss = orc.c_fwd(d, e, eo, t, w, None)
    o_grad, _ = orc.c_bwd(d, e, eo, t, w, None, dloss=0.25)
    assert np.abs(a1 - o_affs).max() < AFFS_ATOL
    assert abs(l1 - o_loss[0]) <= LOSS_RTOL * abs(o_loss[0])
    assert relmax(g1, o_grad) < GRAD_RTOL
    l1b, a1b, g1b = run()
    assert l1 == l1b and np.array_equal(a1, a1b) and np.array_equal(g1, g1b)   # bit-reproducible
    monkeypatch.setenv("PEA_FWD_XDMA", "0")
    monkeypatch.setenv("PEA_BWD_XDMA", "0")
    assert L.pea_cross_supported(ctypes.byref(pkg.affinity_op.make_desc(spec, E)), 2) == 0
    l0, a0, g0 = run()
    assert abs(l1 - l0) <= 3e-6 * abs(l0) and np.abs(a1 - a0).max() < 2e-6 and relmax(g1, g0) < 2e-5


@pytest.mark.parametrize("D,shape,shifts,K,border", [(32, (2, 48, 96), [1, 3, 5, 9, 11], 10, 0), (64, (1, 50, 100), [1, 3, 5, 9, 27], 8, 0),
                                                     (32, (1, 64, 128), [1, 3, 5, 9, 27], 10, 1), (32, (2, 43, 96), [1, 3, 5, 9, 27], 9, 0)])
def test_wide_cross_loss_with_detached_second_operand(pkg, dev, orc, synth, monkeypatch, D, shape, shifts, K, border):
    """ema_embedding_loss at D = 32 / 64 (BASELINE configs[2] and [4] train it every step, scripts_bbbc039v1/main.py:288): round 5 --
    forward k_fwd_xdma<D, .., OTHER> with the own pixel from own TILES staged beside each chunk, backward k_bwd_xdma_pfo (projection
    first from the cross loss' raw map): against the oracle, bit-reproducible, and against the tiled kernels they replace"""
    B, H, W = shape
    offsets = pkg.multi_offset(shifts, 4)[:K]
    lam = [2.0, 2.0] + [1.0] * (K - 2)
    e, t, w, m = _inputs(synth, B, D, [1, H, W], K, 41, zero_px=True)
    e, t, w, m = e[:, :, 0], t[:, :, 0], w[:, :, 0], m[:, :, 0]
    eo = synth.synth_embedding((B, D, H * W), 979).reshape(B, D, H, W)
    eo[0, :, 9, 11] = 0.0
    bmode = pkg._lib.BORDER_CROP_ZERO if border else pkg._lib.BORDER_CIRCULAR
    nmode = pkg._lib.NORM_CROPPED if border else pkg._lib.NORM_BX
    spec = pkg.AffinitySpec(2, offsets, lam, bmode, nmode)
    E, EO, T, Wt, M = cu(e, dev), cu(eo, dev), cu(t, dev), cu(w, dev), cu(m, dev)
    L = pkg._lib.lib()
    desc = pkg.affinity_op.make_desc(spec, E)
    assert L.pea_cross_supported(ctypes.byref(desc), 2) == 1 and L.pea_cross_supported(ctypes.byref(desc), 4) == 1

    def run():
        x = E.clone().requires_grad_(True)
        loss, a, parts = pkg.affinity_op.FusedAffinityMSE.apply(x, EO, T, Wt, M, spec)
        (loss * 0.75).backward()
        return loss.item(), a.cpu().numpy(), x.grad.cpu().numpy()

    l1, a1, g1 = run()
    d = orc.make_desc(B, D, [1, H, W], [[0, o[0], o[1]] for o in offsets], lam, orc.BORDER_CROP_ZERO if border else orc.BORDER_CIRCULAR,
                      orc.NORM_CROPPED if border else orc.NORM_BX, ndim=2)
    o_affs, o_loss = orc.c_fwd(d, e, eo, t, w, m)
    o_de, _ = orc.c_bwd(d, e, eo, t, w, m, dloss=0.75)
    assert np.abs(a1 - o_affs.reshape(a1.shape)).max() < AFFS_ATOL
    assert abs(l1 - o_loss[0]) <= LOSS_RTOL * abs(o_loss[0])
    assert relmax(g1, o_de.reshape(g1.shape)) < GRAD_RTOL
    l1b, a1b, g1b = run()
    assert l1 == l1b and np.array_equal(a1, a1b) and np.array_equal(g1, g1b)
    monkeypatch.setenv("PEA_FWD_XDMA", "0")
    monkeypatch.setenv("PEA_BWD_XDMA", "0")
    l0, a0, g0 = run()
    assert abs(l1 - l0) <= 3e-6 * abs(l0) and np.abs(a1 - a0).max() < 2e-6 and relmax(g1, g0) < 2e-5


@pytest.mark.parametrize("D,shape,shifts,K,border", [(64, (2, 48, 96), [1, 3, 5, 9, 27], 8, 0), (32, (1, 50, 104), [1, 3, 5, 9, 11], 10, 0),
                                                     (16, (2, 43, 96), [1, 3, 5, 9, 27], 10, 0), (64, (1, 64, 128), [1, 3, 5, 9, 27], 8, 1),
                                                     (64, (1, 400, 400), [1, 3, 5, 9, 27], 8, 0)])
def test_f16_cross_loss_with_detached_second_operand(pkg, dev, orc, synth, monkeypatch, D, shape, shifts, K, border):
    """ema_embedding_loss with f16 storage (BASELINE configs[4]: D = 64 in half precision): round 5 -- k_fwd_xdma_h<.., OTHER> (own tile
    staged beside the second operand's cross, v_dot2 gather) and k_bwd_xdma_h<.., PF, HW, OTHER>: against the oracle on the f16 values
    (gradient tolerance 2e-3: the stored gradient is rounded to f16), bit-reproducible, and against the tiled kernels"""
    B, H, W = shape
    offsets = pkg.multi_offset(shifts, 4)[:K]
    lam = [2.0, 2.0] + [1.0] * (K - 2)
    # (no zero-norm pixels here: their clamp-branch gradient, g / eps, overflows the f16 it is stored in.  The 400 x 400 case: a 20 MB
    #  tensor ends 0.5 MB before its mapping does -- where a request past the last channel faulted)
    e, t, w, m = _inputs(synth, B, D, [1, H, W], K, 43, zero_px=False)
    e, t, w, m = e[:, :, 0].astype(np.float16), t[:, :, 0], w[:, :, 0], m[:, :, 0]
    eo = synth.synth_embedding((B, D, H * W), 981).reshape(B, D, H, W).astype(np.float16)
    eo[0, :, 9, 11] = 0.0
    bmode = pkg._lib.BORDER_CROP_ZERO if border else pkg._lib.BORDER_CIRCULAR
    nmode = pkg._lib.NORM_CROPPED if border else pkg._lib.NORM_BX
    spec = pkg.AffinitySpec(2, offsets, lam, bmode, nmode)
    E, EO, T, Wt, M = cu(e, dev), cu(eo, dev), cu(t, dev), cu(w, dev), cu(m, dev)
    L = pkg._lib.lib()
    desc = pkg.affinity_op.make_desc(spec, E)
    assert L.pea_cross_supported(ctypes.byref(desc), 2) == 1 and L.pea_cross_supported(ctypes.byref(desc), 4) == 1

    def run():
        x = E.clone().requires_grad_(True)
        loss, a, parts = pkg.affinity_op.FusedAffinityMSE.apply(x, EO, T, Wt, M, spec)
        (loss * 0.75).backward()
        return loss.item(), a.cpu().numpy(), x.grad.float().cpu().numpy()

    l1, a1, g1 = run()
    ef, eof = e.astype(np.float32), eo.astype(np.float32)
    d = orc.make_desc(B, D, [1, H, W], [[0, o[0], o[1]] for o in offsets], lam, orc.BORDER_CROP_ZERO if border else orc.BORDER_CIRCULAR,
                      orc.NORM_CROPPED if border else orc.NORM_BX, ndim=2)
    o_affs, o_loss = orc.c_fwd(d, ef, eof, t, w, m)
    o_de, _ = orc.c_bwd(d, ef, eof, t, w, m, dloss=0.75)
    assert np.abs(a1 - o_affs.reshape(a1.shape)).max() < AFFS_ATOL
    assert abs(l1 - o_loss[0]) <= LOSS_RTOL * abs(o_loss[0])
    assert relmax(g1, o_de.reshape(g1.shape)) < 2e-3
    l1b, a1b, g1b = run()
    assert l1 == l1b and np.array_equal(a1, a1b) and np.array_equal(g1, g1b)
    monkeypatch.setenv("PEA_FWD_XDMA", "0")
    monkeypatch.setenv("PEA_BWD_XDMA", "0")
    assert L.pea_cross_supported(ctypes.byref(pkg.affinity_op.make_desc(spec, E)), 2) == 0
    l0, a0, g0 = run()
    assert abs(l1 - l0) <= 3e-6 * abs(l0) and np.abs(a1 - a0).max() < 2e-6 and relmax(g1, g0) < 2e-3


def test_cross_2d_crop_border_ema_vs_oracle(pkg, dev, orc, synth):
    """the role-A cross kernels with the CROP_ZERO border in 2D (round 5: the border is a template argument of the cross-loss
    instantiations too): forward + backward against the oracle"""
    B, D, H, W = 2, 16, 64, 128
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    K = len(offsets)
    e, t, w, m = _inputs(synth, B, D, [1, H, W], K, 23, zero_px=True)
    e, t, w, m = e[:, :, 0], t[:, :, 0], w[:, :, 0], m[:, :, 0]
    eo = synth.synth_embedding((B, D, H * W), 978).reshape(B, D, H, W)
    spec = pkg.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED)
    E = cu(e, dev).requires_grad_(True)
    assert pkg._lib.lib().pea_cross_supported(ctypes.byref(pkg.affinity_op.make_desc(spec, E.detach())), 2) == 1
    loss, affs, _ = pkg.affinity_op.FusedAffinityMSE.apply(E, cu(eo, dev), cu(t, dev), cu(w, dev), cu(m, dev), spec)
    (loss * 0.5).backward()
    d = orc.make_desc(B, D, [1, H, W], [[0, o[0], o[1]] for o in offsets], None, orc.BORDER_CROP_ZERO, orc.NORM_CROPPED, ndim=2)
    o_affs, o_loss = orc.c_fwd(d, e, eo, t, w, m)
    o_grad, _ = orc.c_bwd(d, e, eo, t, w, m, dloss=0.5)
    assert np.abs(affs.cpu().numpy() - o_affs.reshape(affs.shape)).max() < AFFS_ATOL
    assert abs(loss.item() - o_loss[0]) <= LOSS_RTOL * abs(o_loss[0])
    assert relmax(E.grad.cpu().numpy(), o_grad.reshape(e.shape)) < GRAD_RTOL


@pytest.mark.parametrize("with_other_loss", [False, True])
def test_head_and_loss_as_one_autograd_node(pkg, dev, orc, synth, with_other_loss):
    """f1: head(x) -> embedding_loss -> backward as ONE node (harness/head_loss.py: four launches, no autograd bookkeeping between
    them) against the two separate nodes, and against the oracle chain c_bwd -> np_head_bwd (float64 head restatement);
    with_other_loss: a second loss on the embedding output, whose gradient is added before the head's backward"""
    B, C, D, H, W = 2, 32, 16, 72, 96                       # ragged in y (4.5 tiles), K = 10 shipped stencil
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], neighbor=4)
    K = len(offsets)
    rng = np.random.default_rng(11)
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    wt = (rng.standard_normal((D, C, 1, 1)) * 0.3).astype(np.float32)
    bs = rng.standard_normal(D).astype(np.float32)
    _, t, w, m = _inputs(synth, B, D, [1, H, W], K, 5)
    t, w, m = t[:, :, 0], w[:, :, 0], m[:, :, 0]
    R = rng.standard_normal((B, D, H, W)).astype(np.float32) * 1e-3
    crit = pkg.WeightedMSE()

    def run(fused):
        head = pkg.OutConv(C, D).to(dev)
        with torch.no_grad():
            head.conv.weight.copy_(cu(wt, dev)); head.conv.bias.copy_(cu(bs, dev))
        xt = cu(x, dev).requires_grad_(True)
        if fused:
            loss, affs, parts, emb = pkg.head_embedding_loss(xt, head, cu(t, dev), cu(w, dev), cu(m, dev), crit, offsets)
        else:
            emb = head(xt)
            loss, affs, parts = pkg.embedding_loss(emb, cu(t, dev), cu(w, dev), cu(m, dev), crit, offsets)
        total = loss * 0.5
        if with_other_loss:
            total = total + (emb * cu(R, dev)).sum()
        total.backward()
        return (loss.item(), affs.cpu().numpy(), xt.grad.cpu().numpy(), head.conv.weight.grad.cpu().numpy().reshape(D, C),
                head.conv.bias.grad.cpu().numpy(), emb.detach().cpu().numpy())

    lf, af, dxf, dwf, dbf, ef = run(True)
    ls, a_s, dxs, dws, dbs, es = run(False)
    assert lf == ls and np.array_equal(af, a_s) and np.array_equal(ef, es)          # same forward kernels
    assert relmax(dxf, dxs) < 2e-6 and relmax(dwf, dws) < 1e-5 and relmax(dbf, dbs) < 1e-5
    # the oracle chain
    e_o = orc.np_head_fwd(x, wt.reshape(D, C), bs)
    d = orc.desc_2d(e_o, offsets)
    o_de, _ = orc.c_bwd(d, e_o, None, t, w, m, dloss=0.5)
    if with_other_loss:
        o_de = o_de + R
    o_dx, o_dw, o_db = orc.np_head_bwd(x, wt.reshape(D, C), o_de)
    assert relmax(dxf, o_dx) < GRAD_RTOL and relmax(dwf, o_dw) < GRAD_RTOL and relmax(dbf, o_db) < GRAD_RTOL


@pytest.mark.parametrize("shape,shifts", [((2, 50, 100), [1, 3, 5, 9, 27]), ((1, 43, 96), [1, 3, 5, 9, 27]), ((2, 37, 72), [1, 3, 5, 9, 11])])
def test_cross_loss_with_detached_second_operand_on_the_cross_kernels(pkg, dev, orc, synth, shape, shifts):
    """ema_embedding_loss with the detached EMA operand (the shipped configs): forward with e_other staged and the own pixel from
    e, the two 1 / norm planes, the role-A backward -- overwrite and PEA_FLAG_ACCUMULATE_DE -- against the oracle; and the same call
    through the Python mirror against the tiled kernels (PEA_FWD_XDMA=0 / PEA_BWD_XDMA=0 is what the old path is)"""
    B, H, W = shape
    D = 16
    offsets = pkg.multi_offset(shifts, 4)
    K = len(offsets)
    lam = [2.0, 2.0] + [1.0] * (K - 2)                      # affs0_weight on the first two offsets (reference :90-93)
    e, t, w, m = _inputs(synth, B, D, [1, H, W], K, 21, zero_px=True)
    e, t, w, m = e[:, :, 0], t[:, :, 0], w[:, :, 0], m[:, :, 0]
    eo = synth.synth_embedding((B, D, H * W), 977).reshape(B, D, H, W)
    eo[0, :, 9, 11] = 0.0                                   # a zero-norm pixel in the second operand
    op, L = pkg.affinity_op, pkg._lib.lib()
    E, EO, T, Wt, M = cu(e, dev), cu(eo, dev), cu(t, dev), cu(w, dev), cu(m, dev)
    spec = op.AffinitySpec(2, offsets, lam, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
    desc = op.make_desc(spec, E)
    assert L.pea_cross_supported(ctypes.byref(desc), 2) == 1
    P = lambda x: ctypes.c_void_p(x.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    affs, g = torch.empty(B, K, H, W, device=dev), torch.empty(B, K, H, W, device=dev)
    inv2 = torch.full((2, B, H, W), 7.0, device=dev)
    lossv = torch.empty(1 + K, device=dev)
    wsb = L.pea_workspace_bytes(ctypes.byref(desc))
    work = torch.empty(max(wsb, 4) // 4, device=dev)
    assert L.pea_workspace_init(ctypes.c_void_p(work.data_ptr()), wsb, None) == 0  # the loss-state block: prepared once
    assert L.pea_affinity_fwd_ex(ctypes.byref(desc), P(E), P(EO), P(T), P(Wt), P(M), P(affs), P(g), P(inv2), P(lossv), P(work), wsb, st) == 0
    d = orc.desc_2d(e, offsets, lam)
    o_affs, o_loss = orc.c_fwd(d, e, eo, t, w, m)
    assert np.abs(affs.cpu().numpy() - o_affs).max() < AFFS_ATOL
    assert abs(lossv[0].item() - o_loss[0]) <= LOSS_RTOL * abs(o_loss[0])
    for k, src in enumerate((e, eo)):
        nrm = np.sqrt((src.astype(np.float64) ** 2).sum(1))
        np.testing.assert_allclose(inv2[k].cpu().numpy(), np.where(nrm < 1e-12, -1e12, 1.0 / np.maximum(nrm, 1e-12)), rtol=3e-7)
    o_de, _ = orc.c_bwd(d, e, eo, t, w, m, dloss=0.75)
    dl = torch.full((), 0.75, device=dev)
    de = torch.empty_like(E)
    assert L.pea_affinity_bwd_ex(ctypes.byref(desc), P(E), P(EO), P(g), P(inv2), P(dl), P(de), None, st) == 0
    assert relmax(de.cpu().numpy(), o_de) < GRAD_RTOL
    # accumulate onto what is there
    base = synth.synth_embedding((B, D, H * W), 31).reshape(B, D, H, W)
    de2 = cu(base, dev)
    dacc = __import__("copy").copy(desc)                   # (make_desc memoises: never mutate what it returns)
    dacc.flags |= pkg._lib.FLAG_ACCUMULATE_DE
    assert desc.flags == 0
    assert L.pea_affinity_bwd_ex(ctypes.byref(dacc), P(E), P(EO), P(g), P(inv2), P(dl), P(de2), None, st) == 0
    assert relmax(de2.cpu().numpy() - base, o_de) < GRAD_RTOL
    # without the planes the flag is refused (the tiled kernels do not accumulate)
    assert L.pea_affinity_bwd_ex(ctypes.byref(dacc), P(E), P(EO), P(g), None, P(dl), P(de2), None, st) == pkg._lib.E_UNSUPPORTED
    # the Python mirror picks the same kernels; against the tiled path
    crit = pkg.WeightedMSE()

    def run():
        x = E.clone().requires_grad_(True)
        loss, a = pkg.ema_embedding_loss(x, EO, T, Wt, M, crit, offsets, affs0_weight=2)
        (loss * 0.75).backward()
        return loss.item(), a.cpu().numpy(), x.grad.cpu().numpy()

    l1, a1, g1 = run()
    assert relmax(g1, o_de) < GRAD_RTOL and np.abs(a1 - o_affs).max() < AFFS_ATOL


@pytest.mark.parametrize("shape,shifts", [((2, 50, 100), [1, 3, 5, 9, 27]), ((1, 43, 96), [1, 3, 5, 9, 27]), ((2, 37, 72), [1, 3, 5, 9, 11])])
def test_pair_backward_in_one_cross_launch(pkg, dev, orc, synth, shape, shifts):
    """pea_affinity_bwd_dual_ex: the self loss' backward and the detached-EMA cross loss' role-A backward of the same embedding as
    one launch of the cross kernel (second phase) -- against dl_self * oracle(self) + dl_cross * oracle(cross), and against the tiled
    two-phase kernel the same entry point runs without the 1 / norm planes"""
    B, H, W = shape
    D = 16
    offsets = pkg.multi_offset(shifts, 4)
    K = len(offsets)
    lamx = [3.0, 3.0] + [1.0] * (K - 2)
    e, t, w, m = _inputs(synth, B, D, [1, H, W], K, 23, zero_px=True)
    e, t, w, m = e[:, :, 0], t[:, :, 0], w[:, :, 0], m[:, :, 0]
    eo = synth.synth_embedding((B, D, H * W), 978).reshape(B, D, H, W)
    eo[0, :, 4, 6] = 0.0
    op, L = pkg.affinity_op, pkg._lib.lib()
    E, EO, T, Wt, M = cu(e, dev), cu(eo, dev), cu(t, dev), cu(w, dev), cu(m, dev)
    d0 = op.make_desc(op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX), E)
    dx = op.make_desc(op.AffinitySpec(2, offsets, lamx, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX), E)
    P = lambda x: None if x is None else ctypes.c_void_p(x.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    affs, g0, gx = (torch.empty(B, K, H, W, device=dev) for _ in range(3))
    inv0, inv2 = torch.empty(B, H, W, device=dev), torch.empty(2, B, H, W, device=dev)
    lossv = torch.empty(1 + K, device=dev)
    wsb = L.pea_workspace_bytes(ctypes.byref(d0))
    work = torch.empty(max(wsb, 4) // 4, device=dev)
    assert L.pea_workspace_init(ctypes.c_void_p(work.data_ptr()), wsb, None) == 0  # the loss-state block: prepared once
    assert L.pea_affinity_fwd_ex(ctypes.byref(d0), P(E), None, P(T), P(Wt), P(M), P(affs), P(g0), P(inv0), P(lossv), P(work), wsb, st) == 0
    assert L.pea_affinity_fwd_ex(ctypes.byref(dx), P(E), P(EO), P(T), P(Wt), P(M), None, P(gx), P(inv2), P(lossv), P(work), wsb, st) == 0
    dl0, dlx = torch.full((), 0.6, device=dev), torch.full((), 1.7, device=dev)
    de_c, de_t = torch.empty_like(E), torch.empty_like(E)
    assert L.pea_affinity_bwd_dual_ex(ctypes.byref(d0), P(E), P(EO), P(g0), P(gx), P(inv0), P(inv2[1]), P(dl0), P(dlx), P(de_c), st) == 0
    rc_t = L.pea_affinity_bwd_dual_ex(ctypes.byref(d0), P(E), P(EO), P(g0), P(gx), None, None, P(dl0), P(dlx), P(de_t), st)
    assert rc_t in (0, pkg._lib.E_UNSUPPORTED)              # (the tiled pair kernel has no plan for the smallest images)
    o_self, _ = orc.c_bwd(orc.desc_2d(e, offsets), e, None, t, w, m, dloss=0.6)
    o_cross, _ = orc.c_bwd(orc.desc_2d(e, offsets, lamx), e, eo, t, w, m, dloss=1.7)
    ref = o_self + o_cross
    assert relmax(de_c.cpu().numpy(), ref) < GRAD_RTOL
    if rc_t == 0:
        assert relmax(de_t.cpu().numpy(), ref) < GRAD_RTOL and relmax(de_c.cpu().numpy(), de_t.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("shape,shifts,K,border,relu,nb", [((2, 50, 100), [1, 3, 5, 9, 27], 10, 0, True, "1"), ((2, 50, 100), [1, 3, 5, 9, 27], 10, 0, False, "1"),
                                                        ((1, 43, 96), [1, 3, 5, 9, 27], 10, 0, False, "1"), ((3, 64, 128), [1, 3, 5, 9, 27], 10, 1, False, "1"),
                                                        ((3, 64, 128), [1, 3, 5, 9, 27], 10, 1, True, "1"), ((2, 37, 72), [1, 3, 5, 9, 11], 8, 0, False, "1"),
                                                        ((1, 96, 200), [1, 3, 5, 9, 27], 9, 0, False, "1"), ((2, 33, 68), [1], 2, 0, False, "1")])
def test_dual_forward_equals_the_two_launches(pkg, dev, orc, synth, monkeypatch, shape, shifts, K, border, relu, nb):
    """pea_affinity_fwd_dual_ex (csrc/pea_xdma_dual.h): the self loss and the detached-EMA cross loss of the same embedding on the
    same target / weight / mask as ONE forward launch -- every output BIT-identical to the two pea_affinity_fwd_ex calls it replaces
    (map, both g maps, both 1 / norm planes, both loss rows), for both borders, with and without the relu of the
    self map; the loss rows also against the oracle, and the state blocks left ready for the next call."""
    monkeypatch.setenv("PEA_FWD_DUAL", nb)
    pkg._lib.reload_env()
    try:
        B, H, W = shape
        D = 16
        offsets = pkg.multi_offset(shifts, 4)[:K]
        lamx = [3.0, 3.0] + [1.0] * (K - 2)
        e, t, w, m = _inputs(synth, B, D, [1, H, W], K, 29, zero_px=True)
        e, t, w, m = e[:, :, 0], t[:, :, 0], w[:, :, 0], m[:, :, 0]
        eo = synth.synth_embedding((B, D, H * W), 979).reshape(B, D, H, W)
        eo[0, :, 4, 6] = 0.0
        op, L = pkg.affinity_op, pkg._lib.lib()
        E, EO, T, Wt, M = cu(e, dev), cu(eo, dev), cu(t, dev), cu(w, dev), cu(m, dev)
        norm = pkg._lib.NORM_BX if border == 0 else pkg._lib.NORM_CROPPED
        d0 = op.make_desc(op.AffinitySpec(2, offsets, None, border, norm, relu=relu), E)
        dx = op.make_desc(op.AffinitySpec(2, offsets, lamx, border, norm), E)
        assert L.pea_cross_supported(ctypes.byref(d0), 5) == 1
        P = lambda x: None if x is None else ctypes.c_void_p(x.data_ptr())
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        wsb = L.pea_workspace_bytes(ctypes.byref(d0))
        work = torch.empty(2, max(wsb, 8) // 4, device=dev)
        assert L.pea_workspace_init(P(work), 2 * wsb, None) == 0
        # the two launches
        affs, g0, gx = (torch.full((B, K, H, W), 7.0, device=dev) for _ in range(3))
        inv0, inv2 = torch.empty(B, H, W, device=dev), torch.empty(2, B, H, W, device=dev)
        l0, lx = torch.empty(1 + K, device=dev), torch.empty(1 + K, device=dev)
        assert L.pea_affinity_fwd_ex(ctypes.byref(d0), P(E), None, P(T), P(Wt), P(M), P(affs), P(g0), P(inv0), P(l0), P(work[0]), wsb, st) == 0
        assert L.pea_affinity_fwd_ex(ctypes.byref(dx), P(E), P(EO), P(T), P(Wt), P(M), None, P(gx), P(inv2), P(lx), P(work[1]), wsb, st) == 0
        # the one launch, twice (the second call finds the state blocks as the first left them)
        for rep in range(2):
            affs_d, g0_d, gx_d = (torch.full((B, K, H, W), -3.0, device=dev) for _ in range(3))
            inv0_d, invo_d = torch.empty(B, H, W, device=dev), torch.empty(B, H, W, device=dev)
            l0_d, lx_d = torch.empty(1 + K, device=dev), torch.empty(1 + K, device=dev)
            rc = L.pea_affinity_fwd_dual_ex(ctypes.byref(d0), ctypes.byref(dx), P(E), P(EO), P(T), P(Wt), P(M), P(affs_d), P(g0_d), P(gx_d),
                                            P(inv0_d), P(invo_d), P(l0_d), P(lx_d), P(work[0]), P(work[1]), wsb, st)
            assert rc == 0
            torch.cuda.synchronize()
            for name, a, b in (("affs", affs_d, affs), ("g", g0_d, g0), ("g_cross", gx_d, gx), ("inv", inv0_d, inv0), ("inv_own_of_pair", inv0_d, inv2[0]),
                               ("inv_other", invo_d, inv2[1]), ("loss", l0_d, l0), ("loss_cross", lx_d, lx)):
                assert torch.equal(a, b), (name, rep)
        offs3 = [[0] + list(o) for o in offsets]
        o_loss = orc.c_fwd(orc.make_desc(B, D, [1, H, W], offs3, None, border, norm, ndim=3), e[:, :, None], None, t[:, :, None], w[:, :, None], m[:, :, None], want_affs=False)[1][0]
        o_lossx = orc.c_fwd(orc.make_desc(B, D, [1, H, W], offs3, lamx, border, norm, ndim=3), e[:, :, None], eo[:, :, None], t[:, :, None], w[:, :, None], m[:, :, None],
                            want_affs=False)[1][0]
        assert abs(l0_d[0].item() - o_loss) <= LOSS_RTOL * abs(o_loss) and abs(lx_d[0].item() - o_lossx) <= LOSS_RTOL * abs(o_lossx)
        # descriptors that disagree in more than lambda / activation are refused, and so is the same state block twice
        dbad = op.make_desc(op.AffinitySpec(2, offsets, lamx, 1 - border, norm), E)
        assert L.pea_affinity_fwd_dual_ex(ctypes.byref(d0), ctypes.byref(dbad), P(E), P(EO), P(T), P(Wt), P(M), P(affs_d), P(g0_d), P(gx_d),
                                          P(inv0_d), P(invo_d), P(l0_d), P(lx_d), P(work[0]), P(work[1]), wsb, st) == -2
        assert L.pea_affinity_fwd_dual_ex(ctypes.byref(d0), ctypes.byref(dx), P(E), P(EO), P(T), P(Wt), P(M), P(affs_d), P(g0_d), P(gx_d),
                                          P(inv0_d), P(invo_d), P(l0_d), P(lx_d), P(work[0]), P(work[0]), wsb, st) == -4
    finally:
        monkeypatch.delenv("PEA_FWD_DUAL")
        pkg._lib.reload_env()


def test_second_operand_that_aliases_the_first(pkg, dev, orc, synth):
    """ema_embedding_loss(e, e.detach(), ..) -- an EMA tensor that IS the embedding's storage: the forward has the values of the self
    loss, the backward is role A only (the second operand is detached).  Round 2 took the self path in the forward (one 1 / norm
    plane written) and the second-operand path in the backward (which reads the second plane): an uninitialised plane scaled the
    gradient.  Both planes must be filled, through the C ABI and through the Python mirror."""
    B, D, H, W = 2, 16, 50, 100
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    K = len(offsets)
    e, t, w, m = _inputs(synth, B, D, [1, H, W], K, 33)
    e, t, w, m = e[:, :, 0], t[:, :, 0], w[:, :, 0], m[:, :, 0]
    op, L = pkg.affinity_op, pkg._lib.lib()
    E, T, Wt, M = cu(e, dev), cu(t, dev), cu(w, dev), cu(m, dev)
    spec = op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
    desc = op.make_desc(spec, E)
    P = lambda x: ctypes.c_void_p(x.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    affs, g = torch.empty(B, K, H, W, device=dev), torch.empty(B, K, H, W, device=dev)
    inv2 = torch.full((2, B, H, W), float("nan"), device=dev)
    lossv = torch.empty(1 + K, device=dev)
    work, wsb = op.workspace(dev, desc)
    assert L.pea_affinity_fwd_ex(ctypes.byref(desc), P(E), P(E), P(T), P(Wt), P(M), P(affs), P(g), P(inv2), P(lossv), P(work), wsb, st) == 0
    assert torch.equal(inv2[0], inv2[1]) and bool(torch.isfinite(inv2).all())
    d = orc.desc_2d(e, offsets)
    o_affs, o_loss = orc.c_fwd(d, e, e.copy(), t, w, m)
    o_de, _ = orc.c_bwd(d, e, e.copy(), t, w, m, dloss=1.0)          # role A only
    o_self, _ = orc.c_bwd(d, e, None, t, w, m, dloss=1.0)           # both roles: a different gradient
    assert relmax(o_de, o_self) > 1e-2
    assert np.abs(affs.cpu().numpy() - o_affs).max() < AFFS_ATOL
    de = torch.empty_like(E)
    assert L.pea_affinity_bwd_ex(ctypes.byref(desc), P(E), P(E), P(g), P(inv2), None, P(de), None, st) == 0
    assert relmax(de.cpu().numpy(), o_de) < GRAD_RTOL
    x = E.clone().requires_grad_(True)
    loss, a = pkg.ema_embedding_loss(x, x.detach(), T, Wt, M, pkg.WeightedMSE(), offsets)
    loss.backward()
    assert abs(loss.item() - o_loss[0]) <= LOSS_RTOL * abs(o_loss[0])
    assert relmax(x.grad.cpu().numpy(), o_de) < GRAD_RTOL
    # PEA_FLAG_ACCUMULATE_DE has no meaning for a self loss: refused, not ignored
    dacc = __import__("copy").copy(desc)
    dacc.flags |= pkg._lib.FLAG_ACCUMULATE_DE
    assert L.pea_affinity_bwd_ex(ctypes.byref(dacc), P(E), None, P(g), P(inv2), None, P(de), None, st) == pkg._lib.E_UNSUPPORTED


def test_loss_state_contract(pkg, dev, orc, synth, monkeypatch):
    """the loss is summed in integers in a state block (csrc/pea_loss.h): a block that was never prepared gives NaN, a prepared one
    the oracle's loss, whichever forward kernel runs (3 / 2 workgroups per CU, tiled, direct), call after call on the same block"""
    B, D, H, W = 3, 16, 80, 128
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    K = len(offsets)
    e, t, w, m = _inputs(synth, B, D, [1, H, W], K, 71)
    e, t, w, m = e[:, :, 0], t[:, :, 0], w[:, :, 0], m[:, :, 0]
    op, L = pkg.affinity_op, pkg._lib.lib()
    E, T, Wt, M = cu(e, dev), cu(t, dev), cu(w, dev), cu(m, dev)
    spec = op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
    desc = op.make_desc(spec, E)
    P = lambda x: ctypes.c_void_p(x.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    wsb = L.pea_workspace_bytes(ctypes.byref(desc))
    raw = torch.full((wsb // 4,), 1.0e-3, device=dev)      # garbage, not initialised
    lossv = torch.zeros(1 + K, device=dev)
    assert L.pea_affinity_fwd_ex(ctypes.byref(desc), P(E), None, P(T), P(Wt), P(M), None, None, None, P(lossv), P(raw), wsb, st) == 0
    assert bool(torch.isnan(lossv).all())
    assert L.pea_workspace_init(P(raw), wsb, st) == 0
    d = orc.desc_2d(e, offsets)
    _, o_loss = orc.c_fwd(d, e, None, t, w, m)
    seen = []
    for switches in ({}, {"PEA_FWD_WG3": "0"}, {"PEA_FWD_XDMA": "0"}, {"PEA_FORCE_DIRECT": "1"}):
        for k_, v_ in switches.items():
            monkeypatch.setenv(k_, v_)
        for _ in range(2):
            lossv.zero_()
            assert L.pea_affinity_fwd_ex(ctypes.byref(desc), P(E), None, P(T), P(Wt), P(M), None, None, None, P(lossv), P(raw), wsb, st) == 0
            got = lossv.cpu().numpy().astype(np.float64)
            np.testing.assert_allclose(got[1:] , o_loss[1:], rtol=LOSS_RTOL)
            assert abs(got[0] - o_loss[0]) <= LOSS_RTOL * abs(o_loss[0])
            seen.append((tuple(sorted(switches)), got))
        for k_ in switches:
            monkeypatch.delenv(k_)
    # the two occupancies of the same cross kernel add the same integers: bit-identical, call after call
    for i in range(1, 4):
        assert np.array_equal(seen[0][1], seen[i][1]), seen[i][0]
