"""GPU (-m gpu): the z-march kernels (csrc/pea_zmarch.h: k_fwd_zm, k_bwd_zm) -- 3D volumes whose axis-aligned stencil steps along z
(embedding_loss_norm5 / norm1, scripts_ac3ac4/loss/loss_embedding_mse.py:7-27, 143-194) -- against the CPU oracle and against the
tile-per-plane cross kernels they replace on large volumes.  PEA_ZMARCH=2 forces the march on volumes with fewer tile columns than
CUs (the sizes the oracle finishes in seconds); PEA_ZSEG cuts the columns into segments (warm-up planes, contributor planes, drain).

Tolerances as in test_gpu_parity.py: affs abs 1e-5, loss rel 1e-5, grads rel-to-max 1e-4."""
import ctypes
import importlib

import numpy as np
import pytest
import torch

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu
AFFS_ATOL, LOSS_RTOL, GRAD_RTOL = 1e-5, 1e-5, 1e-4
NORM5 = [1, 1, 1, 2, 3, 3, 3, 9, 9, 4, 27, 27]


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


def _march_on(pkg, spec, et):
    """the backward of this descriptor reads the raw affinity map: the z-march kernel (D = 16) is what runs"""
    d = pkg.affinity_op.make_desc(spec, et.detach())
    return pkg._lib.lib().pea_cross_supported(ctypes.byref(d), 3) == 1


@pytest.mark.parametrize("shape,zseg", [((2, 6, 48, 96), 0), ((1, 5, 64, 132), 0), ((1, 9, 43, 96), 0), ((1, 13, 48, 96), 5),
                                        ((2, 11, 50, 100), 3), ((1, 24, 48, 128), 8), ((1, 7, 64, 96), 1)])
def test_zmarch_norm5_vs_oracle(pkg, dev, orc, synth, monkeypatch, shape, zseg):
    """embedding_loss_norm5 (12 axis offsets: z 1-4, y / x 1, 3, 9, 27; CROP_ZERO, cropped normaliser, affs0_weight on the first three)
    through the march: whole columns (zseg 0) and segments of 1 - 8 planes -- a segment shorter than the window (1, 3) has warm-up and
    contributor planes on both sides of every plane it owns; ragged tiles in y and x; zero-norm pixels (the clamp branch)"""
    B, Z, Y, X = shape
    monkeypatch.setenv("PEA_ZMARCH", "2")
    if zseg:
        monkeypatch.setenv("PEA_ZSEG", str(zseg))
    offs = orc.norm_offsets(NORM5)
    e, t, w = synth.synth_inputs_3d(B, 16, Z, Y, X, offs, 31 + Z)
    e[0, :, Z // 2, 3, 5] = 0.0
    e[-1, :, Z - 1, Y - 1, X - 1] = 1e-14
    e[0, :, 0, 17, 40] = 0.0
    et = cu(e, dev).requires_grad_(True)
    spec = pkg.AffinitySpec(3, offs, orc.affs0_lambda_3d(12, 2, 3), pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED)
    assert _march_on(pkg, spec, et)
    loss, affs = pkg.embedding_loss_norm5(et, cu(t, dev), cu(w, dev), pkg.WeightedMSE(), affs0_weight=2)
    (loss * 0.25).backward()
    inf = pkg.inf_embedding_loss_norm5(et.detach())
    d = orc.desc_3d(e, NORM5, orc.affs0_lambda_3d(12, 2, 3))
    o_affs, o_loss = orc.c_fwd(d, e, None, t, w, None)
    o_grad, _ = orc.c_bwd(d, e, None, t, w, None, dloss=0.25)
    assert np.abs(affs.cpu().numpy() - o_affs).max() < AFFS_ATOL
    assert np.abs(inf.cpu().numpy() - o_affs).max() < AFFS_ATOL
    assert abs(loss.item() - o_loss[0]) <= LOSS_RTOL * o_loss[0]
    assert relmax(et.grad.cpu().numpy(), o_grad) < GRAD_RTOL
    # against the tile-per-plane cross kernels (z neighbours gathered from global memory): the same arithmetic in another order
    monkeypatch.setenv("PEA_ZMARCH", "0")
    e2 = cu(e, dev).requires_grad_(True)
    assert not _march_on(pkg, spec, e2)
    loss2, affs2 = pkg.embedding_loss_norm5(e2, cu(t, dev), cu(w, dev), pkg.WeightedMSE(), affs0_weight=2)
    (loss2 * 0.25).backward()
    assert np.abs(affs.cpu().numpy() - affs2.cpu().numpy()).max() < 2e-6
    assert relmax(et.grad.cpu().numpy(), e2.grad.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("sup", [0, 1, 2, 4, 8])
@pytest.mark.parametrize("shape,zseg,blk", [((2, 6, 80, 160), 0, None), ((1, 9, 43, 96), 4, (2, 1)), ((1, 5, 272, 352), 0, (4, 2))])
def test_zmarch_walks_change_no_bit(pkg, dev, orc, synth, monkeypatch, shape, zseg, blk, sup):
    """PEA_ZM_SUP (csrc/pea_xdma.h march_tile): the eight XCDs' blocks of a round side by side as one super-block, (8 / sx) x sx, or every
    XCD on its own range of blocks (0).  The walk decides WHEN and WHERE a tile column is marched through, never what it computes: loss,
    map and gradient bit for bit against the other walk, on tile grids that are no multiple of the block / super-block (padded
    super-blocks: workgroups beyond the grid leave), with segments, and against the oracle"""
    B, Z, Y, X = shape
    monkeypatch.setenv("PEA_ZMARCH", "2")
    if zseg:
        monkeypatch.setenv("PEA_ZSEG", str(zseg))
    if blk:
        monkeypatch.setenv("PEA_ZBLK_Y", str(blk[0]))
        monkeypatch.setenv("PEA_ZBLK_X", str(blk[1]))
    offs = orc.norm_offsets(NORM5)
    e, t, w = synth.synth_inputs_3d(B, 16, Z, Y, X, offs, 91 + Z)
    spec = pkg.AffinitySpec(3, offs, None, pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED)

    def run():
        x = cu(e, dev).requires_grad_(True)
        assert _march_on(pkg, spec, x)
        loss, affs = pkg.embedding_loss_norm5(x, cu(t, dev), cu(w, dev), pkg.WeightedMSE())
        loss.backward()
        return loss.detach().clone(), affs.clone(), x.grad.clone()

    monkeypatch.setenv("PEA_ZM_SUP", str(sup))
    got = run()
    monkeypatch.setenv("PEA_ZM_SUP", "0" if sup else "8")
    ref = run()
    assert all(torch.equal(a, b) for a, b in zip(ref, got))
    d = orc.desc_3d(e, NORM5)
    o_affs, o_loss = orc.c_fwd(d, e, None, t, w, None)
    o_grad, _ = orc.c_bwd(d, e, None, t, w, None)
    assert np.abs(got[1].cpu().numpy() - o_affs).max() < AFFS_ATOL
    assert abs(got[0].item() - o_loss[0]) <= LOSS_RTOL * o_loss[0]
    assert relmax(got[2].cpu().numpy(), o_grad) < GRAD_RTOL


@pytest.mark.parametrize("shift,zseg", [(1, 0), (2, 0), (1, 4)])
def test_zmarch_norm1_vs_oracle(pkg, dev, orc, synth, monkeypatch, shift, zseg):
    """embedding_loss_norm1 (one step along z, y, x; shift 1 and 2): only one of the window's four z slots carries a coefficient"""
    B, Z, Y, X = 2, 10, 40, 72
    monkeypatch.setenv("PEA_ZMARCH", "2")
    monkeypatch.setenv("PEA_BOX", "0")  # (the unit-box backward otherwise takes shift 1)
    if zseg:
        monkeypatch.setenv("PEA_ZSEG", str(zseg))
    sh = [shift] * 3
    offs = orc.norm_offsets(sh)
    e, t, w = synth.synth_inputs_3d(B, 16, Z, Y, X, offs, 77 + shift)
    et = cu(e, dev).requires_grad_(True)
    spec = pkg.AffinitySpec(3, offs, orc.affs0_lambda_3d(3, 1.5, 1), pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED)
    assert _march_on(pkg, spec, et)
    loss, affs = pkg.embedding_loss_norm1(et, cu(t, dev), cu(w, dev), pkg.WeightedMSE(), affs0_weight=1.5, shift=shift)
    loss.backward()
    d = orc.desc_3d(e, sh, orc.affs0_lambda_3d(3, 1.5, 1))
    o_affs, o_loss = orc.c_fwd(d, e, None, t, w, None)
    o_grad, _ = orc.c_bwd(d, e, None, t, w, None)
    assert np.abs(affs.cpu().numpy() - o_affs).max() < AFFS_ATOL
    assert abs(loss.item() - o_loss[0]) <= LOSS_RTOL * o_loss[0]
    assert relmax(et.grad.cpu().numpy(), o_grad) < GRAD_RTOL


def test_zmarch_reruns_are_bit_identical_and_dloss_is_linear(pkg, dev, orc, synth, monkeypatch):
    """size-independent properties on a volume large enough for the default dispatch (no switch): two runs agree bit for bit (integer
    loss atomics, no float atomics anywhere), the gradient is linear in grad_output, and the batch is additive (the sharding identity)"""
    B, Z, Y, X = 2, 16, 256, 512            # 2 x 16 x 16 = 512 tile columns: the march runs by default
    offs = orc.norm_offsets(NORM5)
    g = torch.Generator(device=dev).manual_seed(5)
    E = torch.randn(B, 16, Z, Y, X, generator=g, device=dev)
    T = (torch.rand(B, 12, Z, Y, X, generator=g, device=dev) < 0.6).float()
    Wt = torch.rand(B, 12, Z, Y, X, generator=g, device=dev) + 0.5
    spec = pkg.AffinitySpec(3, offs, None, pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED)
    assert _march_on(pkg, spec, E)

    def run(e, t, w, scale):
        x = e.clone().requires_grad_(True)
        loss, affs = pkg.embedding_loss_norm5(x, t, w, pkg.WeightedMSE())
        (loss * scale).backward()
        return loss.item(), affs, x.grad

    l1, a1, g1 = run(E, T, Wt, 1.0)
    l2, a2, g2 = run(E, T, Wt, 1.0)
    assert l1 == l2 and torch.equal(a1, a2) and torch.equal(g1, g2)
    _, _, g3 = run(E, T, Wt, 0.5)
    assert torch.allclose(g3 * 2.0, g1, rtol=1e-6, atol=0)
    # one item alone: its loss terms carry the normaliser of a batch of one (twice the weight), its gradient likewise
    la, aa, ga = run(E[:1], T[:1], Wt[:1], 1.0)
    lb, ab, gb = run(E[1:], T[1:], Wt[1:], 1.0)
    assert torch.equal(aa, a1[:1]) and torch.equal(ab, a1[1:])
    assert abs((la + lb) * 0.5 - l1) <= 1e-6 * abs(l1)
    assert relmax(ga.cpu().numpy() * 0.5, g1[:1].cpu().numpy()) < 1e-6 and relmax(gb.cpu().numpy() * 0.5, g1[1:].cpu().numpy()) < 1e-6
    # a window of it against the oracle: the first planes (border slices), two tile rows, four tile columns
    zs, ys, xs = slice(0, 9), slice(0, 64), slice(0, 128)
    e_np = E[:1, :, zs, ys, xs].cpu().numpy().copy()
    t_np, w_np = T[:1, :, zs, ys, xs].cpu().numpy().copy(), Wt[:1, :, zs, ys, xs].cpu().numpy().copy()
    d = orc.desc_3d(e_np, NORM5)
    o_affs, _ = orc.c_fwd(d, e_np, None, t_np, w_np, None)
    # affs at a voxel depend on voxels at lower coordinates only: the window's map is the volume's map there
    assert np.abs(a1[:1, :, zs, ys, xs].cpu().numpy() - o_affs).max() < AFFS_ATOL


N26 = [[dz, dy, dx] for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dz, dy, dx) != (0, 0, 0)]


@pytest.mark.parametrize("shape,zseg,stencil", [((1, 7, 40, 72), 0, "n26"), ((2, 5, 33, 64), 2, "n26"), ((1, 9, 48, 36), 3, "n26"),
                                                ((1, 6, 40, 40), 0, "half13"), ((2, 4, 17, 36), 1, "n26"), ((1, 8, 50, 68), 5, "diag4")])
def test_box_march_backward_vs_oracle(pkg, dev, orc, synth, monkeypatch, shape, zseg, stencil):
    """the unit-box backward marching along z with all 16 channels of three planes resident in LDS (csrc/pea_boxm.h): the
    26-neighbourhood of BASELINE configs[3], the 13 offsets of its lower half, four diagonals across planes; whole columns and
    segments of 1 - 5 planes; ragged tiles; the smallest volume the box kernels take (17 x 36) -- against the C oracle and against
    the per-(z, tile) kernel it replaces on large volumes (PEA_BOXM=0)"""
    B, Z, Y, X = shape
    offs = {"n26": N26, "half13": N26[:13], "diag4": [[-1, -1, -1], [-1, 1, 1], [1, -1, 1], [0, 1, -1]]}[stencil]
    K = len(offs)
    monkeypatch.setenv("PEA_ZMARCH", "2")
    if zseg:
        monkeypatch.setenv("PEA_ZSEG", str(zseg))
    e, t, w = synth.synth_inputs_3d(B, 16, Z, Y, X, offs, 41 + Z)
    e[0, :, Z // 2, 3, 5] = 0.0          # zero-norm pixels: the clamp branch
    e[-1, :, 0, Y - 1, X - 1] = 1e-14
    op = pkg.affinity_op
    spec = op.AffinitySpec(3, offs, None, pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED)

    def run():
        et = cu(e, dev).requires_grad_(True)
        loss, affs, parts = op.FusedAffinityMSE.apply(et, None, cu(t, dev), cu(w, dev), None, spec)
        (loss * 0.5).backward()
        return loss.item(), affs.cpu().numpy(), et.grad.cpu().numpy()

    l1, a1, g1 = run()
    d = orc.make_desc(B, 16, [Z, Y, X], offs, None, orc.BORDER_CROP_ZERO, orc.NORM_CROPPED)
    o_affs, o_loss = orc.c_fwd(d, e, None, t, w, None)
    o_grad, _ = orc.c_bwd(d, e, None, t, w, None, dloss=0.5)
    assert np.abs(a1 - o_affs.reshape(a1.shape)).max() < AFFS_ATOL
    assert abs(l1 - o_loss[0]) <= LOSS_RTOL * o_loss[0]
    assert relmax(g1, o_grad.reshape(g1.shape)) < GRAD_RTOL
    monkeypatch.setenv("PEA_BOXM", "0")
    l2, a2, g2 = run()
    assert l1 == l2 and np.array_equal(a1, a2)      # the same forward kernel
    assert relmax(g1, g2) < 2e-6                    # the same sums in another order


@pytest.mark.parametrize("name", ["g3d_norm5_march", "g3d_norm5_march_b2", "g3d_norm1_march"])
def test_zmarch_matches_reference_summary(pkg, dev, monkeypatch, name):
    """the march kernels against the REFERENCE's own embedding_loss_norm5 / norm1 (tests/golden/make_golden.py `march`: the reference
    run on closed-form inputs at sizes the LDS-DMA kernels take; loss, sums and 4096 sampled values of its map and autograd gradient)"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import load_golden
    from test_oracle import march_inputs
    g = load_golden(name)
    e, t, w, shifts = march_inputs(g)
    monkeypatch.setenv("PEA_ZMARCH", "2")
    monkeypatch.setenv("PEA_BOX", "0")  # (norm1: keep the unit-box backward out of the way)
    et = cu(e, dev).requires_grad_(True)
    a0 = float(g["affs0_weight"])
    if len(shifts) == 3:
        loss, affs = pkg.embedding_loss_norm1(et, cu(t, dev), cu(w, dev), pkg.WeightedMSE(), affs0_weight=a0)
    else:
        loss, affs = pkg.embedding_loss_norm5(et, cu(t, dev), cu(w, dev), pkg.WeightedMSE(), affs0_weight=a0)
    spec_offs = [[-s if i % 3 == a else 0 for a in range(3)] for i, s in enumerate(shifts)]
    assert _march_on(pkg, pkg.AffinitySpec(3, spec_offs, None, pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED), et)
    loss.backward()
    a, gr = affs.cpu().numpy(), et.grad.cpu().numpy()
    assert abs(loss.item() - float(g["loss"])) <= LOSS_RTOL * max(1.0, abs(float(g["loss"])))
    assert np.abs(a.reshape(-1)[g["affs_idx"]] - g["affs_val"]).max() < AFFS_ATOL
    assert abs((a.astype(np.float64) ** 2).sum() / float(g["affs_sq"]) - 1) < 1e-5
    assert np.abs(gr.reshape(-1)[g["grad_idx"]] - g["grad_val"]).max() <= GRAD_RTOL * np.abs(g["grad_val"]).max()
    assert abs((gr.astype(np.float64) ** 2).sum() / float(g["grad_sq"]) - 1) < 1e-4


def test_ac3ac4_section_backward_runs_the_march(pkg, dev, orc, synth, monkeypatch):
    """round-4 advice: ac3ac4_loss_section's one-node path (_TensorSection) called the backward WITHOUT the raw affinity map, so its
    full-resolution self backward fell back to the tile-per-plane kernel.  With the map handed over the section's gradient of the
    full-resolution embedding is, bit for bit, (z-march self gradient) + (cross gradient) as the stand-alone losses compute them; with
    the march switched off the self part comes from another kernel and the bits differ."""
    crit = pkg.WeightedMSE()
    B, D = 1, 16
    shapes = [(6, 48, 96), (6, 24, 48), (6, 24, 48), (3, 12, 24), (3, 12, 24)]
    sh5, sh1 = orc.norm_offsets(NORM5), orc.norm_offsets([1, 1, 1])
    lab_t = [torch.from_numpy(synth.synth_labels(B, s, 240 + i, cell=7)).to(dev) for i, s in enumerate(shapes)]
    embs = [synth.synth_embedding((B, D) + s, 250 + i) for i, s in enumerate(shapes)]
    ema = cu(synth.synth_embedding((B, D) + shapes[0], 260), dev)
    t0, _, w0 = pkg.gen_targets(lab_t[0], sh5, padding=False, both_foreground=True, want_mask=False)
    heads = [pkg.gen_targets(lab_t[1 + j], sh1, padding=False, both_foreground=True, want_mask=False) for j in range(4)]
    downs = [torch.cat([heads[3 - k][0], heads[3 - k][2]], dim=1) for k in range(4)]

    def section_grad():
        x = [cu(e, dev).requires_grad_(True) for e in embs]
        loss, pred = pkg.ac3ac4_loss_section(x[0], x[1:], ema, t0, w0, downs, crit, embedding_mode=5, affs0_weight=2)
        loss.backward()
        return x[0].grad.clone()

    def standalone_grads():
        a = cu(embs[0], dev).requires_grad_(True)
        pkg.embedding_loss_norm5(a, t0, w0, crit, affs0_weight=2)[0].backward()
        b = cu(embs[0], dev).requires_grad_(True)
        pkg.ema_embedding_loss_norm5(b, ema, t0, w0, crit, affs0_weight=2)[0].backward()
        return a.grad, b.grad

    monkeypatch.setenv("PEA_ZMARCH", "2")
    spec = pkg.AffinitySpec(3, sh5, orc.affs0_lambda_3d(12, 2, 3), pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED)
    assert _march_on(pkg, spec, cu(embs[0], dev))
    gs = section_grad()
    g_self, g_cross = standalone_grads()
    assert torch.equal(gs, g_self + g_cross), "the section's self backward is not the z-march kernel's"
    monkeypatch.setenv("PEA_ZMARCH", "0")
    g_self0, _ = standalone_grads()
    assert not torch.equal(g_self0, g_self)  # (the check above can tell the two kernels apart)
    assert relmax(g_self0.cpu().numpy(), g_self.cpu().numpy()) < 2e-5


def test_ac3ac4_section_finished_pred_where_the_march_reads_the_raw_map(pkg, dev, orc, synth):
    """round 6: ac3ac4_loss_section(finish_pred=True) on a volume large enough for the default dispatch to march (512 tile columns): the
    z-march backward reads the forward's RAW map, so the forward must NOT clamp it -- the section then finishes the map with one fill +
    relu pass instead; every output bit for bit the unfinished section's + finish_pred_3d_, and the march is what ran (the gradient
    equals the stand-alone losses' with the march on)"""
    crit = pkg.WeightedMSE()
    B, D = 1, 16
    shapes = [(8, 256, 512), (8, 128, 256), (8, 64, 128), (8, 32, 64), (8, 16, 32)]
    sh5, sh1 = orc.norm_offsets(NORM5), orc.norm_offsets([1, 1, 1])
    g = torch.Generator(device=dev).manual_seed(77)
    embs = [torch.randn((B, D) + s, generator=g, device=dev) for s in shapes]
    ema = torch.randn((B, D) + shapes[0], generator=g, device=dev)
    t0 = (torch.rand((B, 12) + shapes[0], generator=g, device=dev) < 0.6).float()
    w0 = torch.rand((B, 12) + shapes[0], generator=g, device=dev) + 0.5
    downs = [torch.cat([(torch.rand((B, 3) + s, generator=g, device=dev) < 0.6).float(), torch.rand((B, 3) + s, generator=g, device=dev) + 0.5], dim=1)
             for s in shapes[1:]]  # down1 .. down4
    spec = pkg.AffinitySpec(3, sh5, None, pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED)
    assert _march_on(pkg, spec, embs[0])

    def run(finish):
        x = [e.clone().requires_grad_(True) for e in embs]
        loss, pred = pkg.ac3ac4_loss_section(x[0], x[1:][::-1], ema, t0, w0, downs, crit, embedding_mode=5, finish_pred=finish)
        loss.backward()
        if not finish:
            pred = pkg.finish_pred_3d_(pred.clone())
        return loss.detach(), pred, [v.grad for v in x]

    l1, p1, g1 = run(True)
    l0, p0, g0 = run(False)
    assert torch.equal(l1, l0) and torch.equal(p1, p0) and all(torch.equal(a, b) for a, b in zip(g1, g0))
    assert float(p1.min()) >= 0.0 and torch.equal(p1[:, 0, 0], p1[:, 0, 1]) and torch.equal(p1[:, 1, :, 0], p1[:, 1, :, 1])
    a = embs[0].clone().requires_grad_(True)
    pkg.embedding_loss_norm5(a, t0, w0, crit)[0].backward()
    b = embs[0].clone().requires_grad_(True)
    pkg.ema_embedding_loss_norm5(b, ema, t0, w0, crit)[0].backward()
    assert torch.equal(g1[0], a.grad + b.grad)


def test_unit_box_walks_change_no_bit(pkg, dev, orc, synth, monkeypatch):
    """round 6: k_fwd_box / k_bwd_box walk blocks of 2 x 8 tiles with the XCDs' blocks as one super-block (4 high x 2 wide) where the tile
    grid is a whole number of them (here 8 x 16 tiles = exactly one), k_bwd_boxm blocks of 2 x 8 tile columns; PEA_ZM_SUP=0 = rounds 4-5's
    walks.  Loss, map and gradient bit for bit between the walks (march and per-tile backward), and against the oracle"""
    B, D, Z, Y, X = 1, 16, 3, 128, 512
    offs = [[dz, dy, dx] for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dz, dy, dx) != (0, 0, 0)]
    e, t, w = synth.synth_inputs_3d(B, D, Z, Y, X, offs, 404)
    spec = pkg.AffinitySpec(3, offs, None, pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED)
    T, W = cu(t, dev), cu(w, dev)

    def run():
        x = cu(e, dev).requires_grad_(True)
        loss, affs, _ = pkg.affinity_op.FusedAffinityMSE.apply(x, None, T, W, None, spec)
        loss.backward()
        return loss.detach().clone(), affs.clone(), x.grad.clone()

    res = {}
    for sup, march in (("-1", "2"), ("0", "2"), ("-1", "0"), ("0", "0")):
        monkeypatch.setenv("PEA_ZM_SUP", sup)
        monkeypatch.setenv("PEA_BOXM", "1" if march == "2" else "0")
        monkeypatch.setenv("PEA_ZMARCH", "2")
        res[(sup, march)] = run()
    for k in (("0", "2"), ("-1", "0"), ("0", "0")):
        a, b = res[("-1", "2")], res[k]
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), k
    assert torch.equal(res[("-1", "2")][2], res[("0", "2")][2]) and torch.equal(res[("-1", "0")][2], res[("0", "0")][2])
    d = orc.make_desc(B, D, [Z, Y, X], offs, None, pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED, ndim=3)
    o_affs, o_loss = orc.c_fwd(d, e, None, t, w, None)
    o_grad, _ = orc.c_bwd(d, e, None, t, w, None)
    got = res[("-1", "2")]
    assert np.abs(got[1].cpu().numpy() - o_affs).max() < AFFS_ATOL
    assert abs(got[0].item() - o_loss[0]) <= LOSS_RTOL * o_loss[0]
    assert relmax(got[2].cpu().numpy(), o_grad) < GRAD_RTOL
    assert relmax(res[("-1", "0")][2].cpu().numpy(), o_grad) < GRAD_RTOL
