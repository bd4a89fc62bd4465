"""GPU (-m gpu): the HIP path, called through the C ABI, against the reference's golden vectors, the
CPU oracle, and size-independent properties at BASELINE.json's full sizes.

Tolerances (north_star: affinity maps within 1e-4 of the reference; fp32 everywhere in the reference):
    affs   abs 1e-5  (10x tighter than required)      loss  rel 1e-5      grads  rel-to-max 1e-4
"""
import ctypes
import importlib

import os

import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from conftest import golden_names, lam_for, load_golden, shifts_for

pytestmark = pytest.mark.gpu

AFFS_ATOL, LOSS_RTOL, GRAD_RTOL = 1e-5, 1e-5, 1e-4
G2D = [n for n in golden_names("g2d_") if "summary" not in n]
G3D = [n for n in golden_names("g3d_") if "march" not in n]  # (the *_march fixtures are summaries: test_gpu_zmarch.py, test_oracle.py below)


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


def run_2d(pkg, g, dev, dloss=1.0):
    e = cu(g["e"], dev).requires_grad_(True)
    crit = pkg.WeightedMSE()
    offsets = g["offsets"].tolist()
    mode = str(g["mode"]) if "mode" in g else "ours"
    if str(g["kind"]) == "2d_ema":
        ema = cu(g["ema"], dev).requires_grad_(not bool(g["detach"]))
        loss, affs = pkg.ema_embedding_loss(e, ema, cu(g["target"], dev), cu(g["weight"], dev), cu(g["mask"], dev), crit,
                                            offsets, affs0_weight=float(g["affs0_weight"]))
        all_loss = None
    else:
        ema = None
        loss, affs, all_loss = pkg.embedding_loss(e, cu(g["target"], dev), cu(g["weight"], dev), cu(g["mask"], dev), crit,
                                                  offsets, affs0_weight=1, mode=mode)
    (loss * dloss).backward()
    return loss, affs, all_loss, e.grad, (ema.grad if ema is not None and ema.requires_grad else None)


@pytest.mark.parametrize("name", G2D)
def test_2d_matches_reference_golden(pkg, dev, name):
    g = load_golden(name)
    loss, affs, all_loss, grad, grad_ema = run_2d(pkg, g, dev)
    assert np.abs(affs.cpu().numpy() - g["affs"]).max() < AFFS_ATOL
    assert abs(loss.item() - float(g["loss"])) <= LOSS_RTOL * max(1.0, abs(float(g["loss"])))
    if all_loss is not None:
        np.testing.assert_allclose(np.array(list(all_loss)), g["all_loss"], rtol=LOSS_RTOL, atol=1e-9)
    assert relmax(grad.cpu().numpy(), g["grad"]) < GRAD_RTOL
    if "grad_ema" in g:
        assert relmax(grad_ema.cpu().numpy(), g["grad_ema"]) < GRAD_RTOL
    if "affs_infer" in g:
        mode = str(g["mode"]) if "mode" in g else "ours"
        inf = pkg.embedding2affs(cu(g["e"], dev), g["offsets"].tolist(), mode=mode)
        assert np.abs(inf.cpu().numpy() - g["affs_infer"]).max() < AFFS_ATOL


def test_generic_d_forward(pkg, dev):
    g = load_golden("g2d_d5_generic")
    offsets = g["offsets"].tolist()
    inf = pkg.embedding2affs(cu(g["e"], dev), offsets)
    assert np.abs(inf.cpu().numpy() - g["affs_infer"]).max() < AFFS_ATOL
    loss, affs, all_loss = pkg.embedding_loss(cu(g["e"], dev), cu(g["target"], dev), cu(g["weight"], dev), cu(g["mask"], dev),
                                              pkg.WeightedMSE(), offsets)
    assert abs(loss.item() - float(g["loss"])) <= LOSS_RTOL * max(1.0, abs(float(g["loss"])))
    # any width trains: D = 5 goes through the runtime-D backward (k_bwd_direct_anyd)
    e = cu(g["e"], dev).requires_grad_(True)
    loss, _, _ = pkg.embedding_loss(e, cu(g["target"], dev), cu(g["weight"], dev), cu(g["mask"], dev), pkg.WeightedMSE(), offsets)
    loss.backward()
    assert relmax(e.grad.cpu().numpy(), g["grad"]) < GRAD_RTOL
    e16 = cu(g["e"].astype(np.float16), dev).requires_grad_(True)
    loss16, _, _ = pkg.embedding_loss(e16, cu(g["target"], dev), cu(g["weight"], dev), cu(g["mask"], dev), pkg.WeightedMSE(), offsets)
    loss16.backward()
    assert relmax(e16.grad.float().cpu().numpy(), g["grad"]) < 5e-3


@pytest.mark.parametrize("name", G3D)
def test_3d_matches_reference_golden(pkg, dev, name):
    g = load_golden(name)
    kind = str(g["kind"])
    e = cu(g["e"], dev).requires_grad_(True)
    t, w = cu(g["target"], dev), cu(g["weight"], dev)
    crit = pkg.WeightedMSE()
    a0, sh = float(g["affs0_weight"]), int(g["shift"])
    if kind == "3d_norm1":
        loss, affs = pkg.embedding_loss_norm1(e, t, w, crit, affs0_weight=a0, shift=sh)
        inf = pkg.inf_embedding_loss_norm1(e.detach(), shift=sh)
    elif kind == "3d_norm5":
        loss, affs = pkg.embedding_loss_norm5(e, t, w, crit, affs0_weight=a0)
        inf = pkg.inf_embedding_loss_norm5(e.detach())
    elif kind == "3d_norm1_ema":
        loss, affs = pkg.ema_embedding_loss_norm1(e, cu(g["ema"], dev), t, w, crit, affs0_weight=a0, shift=sh)
        inf = None
    else:
        loss, affs = pkg.ema_embedding_loss_norm5(e, cu(g["ema"], dev), t, w, crit, affs0_weight=a0)
        inf = None
    loss.backward()
    assert np.abs(affs.cpu().numpy() - g["affs"]).max() < AFFS_ATOL
    assert abs(loss.item() - float(g["loss"])) <= LOSS_RTOL * max(1.0, abs(float(g["loss"])))
    assert relmax(e.grad.cpu().numpy(), g["grad"]) < GRAD_RTOL
    if inf is not None:
        assert np.abs(inf.cpu().numpy() - g["affs_infer"]).max() < AFFS_ATOL
    # the border slices the reference leaves at zero
    a = affs.cpu().numpy()
    for i, s in enumerate(shifts_for(g)):
        sl = [slice(None)] * 5
        sl[1], sl[2 + i % 3] = i, slice(0, s)
        assert np.all(a[tuple(sl)] == 0)


def test_full_size_cvppp_against_reference_summary(pkg, dev, synth):
    """2 x 16 x 544 x 544, K=10 (BASELINE configs[0] shape): HIP vs the reference's summary statistics"""
    g = load_golden("g2d_full544_summary")
    B, D, H, W = [int(v) for v in g["shape"]]
    offsets = g["offsets"].tolist()
    e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, int(g["seed"]))
    et = cu(e, dev).requires_grad_(True)
    loss, affs, all_loss = pkg.embedding_loss(et, cu(t, dev), cu(w, dev), cu(m, dev), pkg.WeightedMSE(), offsets)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= LOSS_RTOL * float(g["loss"])
    np.testing.assert_allclose(np.array(list(all_loss)), g["all_loss"], rtol=LOSS_RTOL)
    a = affs.cpu().numpy()
    assert np.abs(a.reshape(-1)[g["affs_idx"]] - g["affs_val"]).max() < AFFS_ATOL
    assert abs(a.astype(np.float64).sum() - float(g["affs_sum"])) < 1e-5 * a.size ** 0.5 * 10
    assert abs((a.astype(np.float64) ** 2).sum() / float(g["affs_sq"]) - 1) < 1e-5
    gr = et.grad.cpu().numpy()
    assert np.abs(gr.reshape(-1)[g["grad_idx"]] - g["grad_val"]).max() <= GRAD_RTOL * np.abs(g["grad_val"]).max()
    assert abs((gr.astype(np.float64) ** 2).sum() / float(g["grad_sq"]) - 1) < 1e-4


@pytest.mark.parametrize("K", [8, 10])
def test_baseline_config0_single_image(pkg, dev, orc, synth, K):
    """BASELINE configs[0] taken literally: ONE 544 x 544 image (530 x 500 padded), 16-dim embedding, the first 8 affinity offsets
    (`offsets[:8]`) and the shipped 10, forward + backward -- the whole image against the C oracle (one image is
    a quarter of a second of CPU), and inference against the same map"""
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)[:K]
    B, D, H, W = 1, 16, 544, 544
    e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 77)
    et = cu(e, dev).requires_grad_(True)
    loss, affs, all_loss = pkg.embedding_loss(et, cu(t, dev), cu(w, dev), cu(m, dev), pkg.WeightedMSE(), offsets)
    loss.backward()
    inf = pkg.embedding2affs(et.detach(), offsets)
    d = orc.desc_2d(e, offsets)
    o_affs, o_loss = orc.c_fwd(d, e, None, t, w, m)
    o_grad, _ = orc.c_bwd(d, e, None, t, w, m)
    assert np.abs(affs.cpu().numpy() - o_affs.reshape(affs.shape)).max() < AFFS_ATOL
    assert np.abs(inf.cpu().numpy() - o_affs.reshape(affs.shape)).max() < AFFS_ATOL
    assert abs(loss.item() - o_loss[0]) <= LOSS_RTOL * o_loss[0]
    np.testing.assert_allclose(np.array(list(all_loss)), o_loss[1:], rtol=LOSS_RTOL)
    assert relmax(et.grad.cpu().numpy(), o_grad.reshape(e.shape)) < GRAD_RTOL


def test_baseline_config_b8_properties(pkg, dev, synth):
    """BASELINE configs[1]: B=8, D=16, 544x544, K=10 -- size-independent properties instead of a CPU re-run:
    run-to-run bit reproducibility, linearity of the gradient in dloss, batch additivity of the loss
    (what the multi-GPU sharding relies on), and roll-equivariance of the circular stencil."""
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    B, D, H, W = 8, 16, 544, 544
    e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 555)
    E, T, Wt, M = cu(e, dev), cu(t, dev), cu(w, dev), cu(m, dev)
    crit = pkg.WeightedMSE()

    def run(Ein, scale=1.0, sl=slice(None)):
        x = Ein[sl].clone().requires_grad_(True)
        loss, affs, parts = pkg.embedding_loss(x, T[sl], Wt[sl], M[sl], crit, offsets)
        (loss * scale).backward()
        return loss.detach(), affs, parts.tensor.clone(), x.grad

    l1, a1, p1, g1 = run(E)
    l2, a2, p2, g2 = run(E)
    assert torch.equal(l1, l2) and torch.equal(a1, a2) and torch.equal(p1, p2) and torch.equal(g1, g2)
    _, _, _, g3 = run(E, scale=3.0)
    assert relmax(g3.cpu().numpy(), 3.0 * g1.cpu().numpy()) < 1e-6
    # batch additivity: loss(B=8) == mean over 4 shards of loss(B=2) (normaliser 1/(B*W)), grads scale by 1/4
    shard_losses = []
    for r in range(4):
        sl = slice(2 * r, 2 * r + 2)
        ls, as_, _, gs = run(E, sl=sl)
        shard_losses.append(ls.item())
        assert torch.equal(as_, a1[sl])
        assert relmax(gs.cpu().numpy() / 4.0, g1[sl].cpu().numpy()) < 1e-5
    assert abs(np.mean(shard_losses) - l1.item()) < 1e-6 * abs(l1.item())
    # roll equivariance of embedding2affs (circular border): rolling e rolls affs
    inf = pkg.embedding2affs(E[:1], offsets)
    inf_r = pkg.embedding2affs(torch.roll(E[:1], shifts=(13, -29), dims=(2, 3)), offsets)
    assert torch.equal(torch.roll(inf, shifts=(13, -29), dims=(2, 3)), inf_r)
    # inference runs k_fwd_tiled (normalise, then dot), training k_fwd_xdma (raw dot, then the two norms): same map to rounding
    assert float((inf - a1[:1]).abs().max()) < 2e-6
    # |a| <= 1 (cosine) and the zero offset would be exactly ~1
    assert float(a1.abs().max()) <= 1.0 + 1e-5


def test_graphed_step_equals_eager(pkg, dev, synth):
    """pea.graphed: embedding_loss + backward of BASELINE configs[0] (one 544 x 544 image, K = 10) captured in a HIP graph through the
    PUBLIC API -- every replay equals the eager call bit for bit, also after the static inputs were refilled"""
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    crit = pkg.WeightedMSE()
    e, t, w, m = synth.synth_inputs_2d(1, 16, 544, 544, offsets, seed=71)
    e2 = synth.synth_embedding((1, 16, 544, 544), 72)
    E = cu(e, dev).requires_grad_(True)
    T, W, M = cu(t, dev), cu(w, dev), cu(m, dev)

    def step(E, T, W, M):
        E.grad = None
        loss, affs, parts = pkg.embedding_loss(E, T, W, M, crit, offsets)
        pkg.backward(loss)
        return loss, affs, E.grad

    def eager(ev):
        x = cu(ev, dev).requires_grad_(True)
        loss, affs, _ = pkg.embedding_loss(x, T, W, M, crit, offsets)
        pkg.backward(loss)
        return loss.detach().clone(), affs.clone(), x.grad.clone()

    g = pkg.graphed(step, E, T, W, M)
    for ev in (e, e2, e):
        with torch.no_grad():
            E.copy_(cu(ev, dev))
        loss, affs, grad = g.replay()
        l0, a0, g0 = eager(ev)
        assert torch.equal(loss, l0) and torch.equal(affs, a0) and torch.equal(grad, g0)
    # and the replay is what it claims to be: one graph launch, no autograd graph left behind
    loss, affs, grad = g()
    assert grad.shape == E.shape and torch.isfinite(grad).all()


@pytest.mark.parametrize("switch,value", [("PEA_BWD_REV", "0"), ("PEA_FWD_WG3", "0")])
def test_walk_and_placement_switches_change_no_bit(pkg, dev, synth, monkeypatch, switch, value):
    """The tile walk of the backward (last tile first / first tile first, csrc/pea_xdma.h xdma_tile) and the forward's workgroups per CU
    decide WHEN and WHERE a tile is worked on, never what it computes: loss, map and gradient bit for bit, B=4 x 16 x 272 x 320.
    (Round 6: the walk / placement experiments that lost -- PEA_SKEW, PEA_WALK2D, PEA_XCD_STAGGER -- left the library.)"""
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    B, D, H, W = 4, 16, 272, 320
    e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 77)
    E, T, Wt, M = cu(e, dev), cu(t, dev), cu(w, dev), cu(m, dev)
    crit = pkg.WeightedMSE()

    def run():
        x = E.clone().requires_grad_(True)
        loss, affs, parts = pkg.embedding_loss(x, T, Wt, M, crit, offsets)
        loss.backward()
        return loss.detach(), affs, x.grad

    ref = run()
    monkeypatch.setenv(switch, value)
    got = run()
    assert all(torch.equal(a, b) for a, b in zip(ref, got))


def test_packed_down_tensor_slices_need_no_copy(pkg, dev, orc, synth):
    """scripts_cvppp/main.py:284: target/weight/mask are channel slices of one packed `down` tensor"""
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)[:8]
    B, D, H, W = 3, 16, 34, 40
    e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 7)
    down = torch.from_numpy(np.concatenate([t, w, m.astype(np.float32)], axis=1)).to(dev)
    et = cu(e, dev).requires_grad_(True)
    loss, affs, _ = pkg.embedding_loss(et, down[:, 0:8], down[:, 8:16], down[:, 16:24], pkg.WeightedMSE(), offsets)
    loss.backward()
    d = orc.desc_2d(e, offsets)
    o_affs, o_loss = orc.c_fwd(d, e, None, t, w, m)
    o_grad, _ = orc.c_bwd(d, e, None, t, w, m)
    assert np.abs(affs.cpu().numpy() - o_affs).max() < AFFS_ATOL
    assert abs(loss.item() - o_loss[0]) <= LOSS_RTOL * o_loss[0]
    assert relmax(et.grad.cpu().numpy(), o_grad) < GRAD_RTOL


def test_foreign_criterion_uses_vjp(pkg, dev, orc, synth):
    """criterion is a parameter of the reference API: a non-fused criterion runs on a differentiable
    affinity map (pea_affinity_bwd with an upstream d_affs) and must match torch autograd of the reference op sequence"""
    offsets = pkg.multi_offset([1, 3, 9], 8)
    B, D, H, W = 2, 16, 30, 44
    e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 11)
    crit = lambda pred, tgt, wt: torch.sum(wt * torch.abs(pred - tgt) ** 3) / pred.shape[0]
    et = cu(e, dev).requires_grad_(True)
    loss, affs, all_loss = pkg.embedding_loss(et, cu(t, dev), cu(w, dev), cu(m, dev), crit, offsets)
    loss.backward()
    ec = torch.from_numpy(e).requires_grad_(True)
    l_ref, a_ref, _ = orc.torch_embedding_loss(ec, torch.from_numpy(t), torch.from_numpy(w), torch.from_numpy(m), offsets, criterion=crit)
    l_ref.backward()
    assert abs(loss.item() - l_ref.item()) <= 1e-5 * abs(l_ref.item())
    assert np.abs(affs.cpu().numpy() - a_ref.numpy()).max() < AFFS_ATOL
    assert relmax(et.grad.cpu().numpy(), ec.grad.numpy()) < GRAD_RTOL
    assert len(list(all_loss)) == len(offsets)


def test_relu_epilogue_and_c_abi_direct(pkg, dev, orc, synth):
    """direct ctypes call of pea_affinity_infer with the RELU flag (F.relu(pred), scripts_cvppp/inference.py:193)"""
    import ctypes
    op = pkg.affinity_op
    offsets = pkg.multi_offset([1, 3], 4)
    e, _, _, _ = synth.synth_inputs_2d(1, 16, 20, 24, offsets, 3)
    E = cu(e, dev)
    spec = op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX, relu=True)
    d = op.make_desc(spec, E)
    out = torch.full((1, 4, 20, 24), -7.0, device=dev)
    rc = pkg._lib.lib().pea_affinity_infer(ctypes.byref(d), ctypes.c_void_p(E.data_ptr()), None, ctypes.c_void_p(out.data_ptr()),
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    o_affs, _ = orc.c_fwd(orc.desc_2d(e, offsets, relu=True), e)
    assert np.abs(out.cpu().numpy() - o_affs).max() < AFFS_ATOL
    assert float(out.min()) >= 0.0


def test_fp16_storage_f32_accumulate(pkg, dev, orc, synth):
    """BASELINE configs[4] flavour: f16 embedding storage, f32 arithmetic; D=64, K=8"""
    offsets = pkg.multi_offset([1, 3, 5, 9], 4)
    B, D, H, W = 1, 64, 24, 40
    e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 5)
    e16 = e.astype(np.float16)
    et = torch.from_numpy(e16).to(dev).requires_grad_(True)
    loss, affs, _ = pkg.embedding_loss(et, cu(t, dev), cu(w, dev), cu(m, dev), pkg.WeightedMSE(), offsets)
    loss.backward()
    e_r = e16.astype(np.float32)  # the oracle sees exactly the rounded inputs
    d = orc.desc_2d(e_r, offsets)
    o_affs, o_loss = orc.c_fwd(d, e_r, None, t, w, m)
    o_grad, _ = orc.c_bwd(d, e_r, None, t, w, m)
    assert affs.dtype == torch.float32 and et.grad.dtype == torch.float16
    assert np.abs(affs.cpu().numpy() - o_affs).max() < AFFS_ATOL
    assert abs(loss.item() - o_loss[0]) <= LOSS_RTOL * o_loss[0]
    assert relmax(et.grad.float().cpu().numpy(), o_grad) < 2e-3  # f16 rounding of the stored gradient


def test_3d_26_neighbourhood_vs_oracle(pkg, dev, orc, synth):
    """BASELINE configs[3] stencil: full 3x3x3 neighbourhood minus centre, CROP_ZERO border"""
    op = pkg.affinity_op
    offs = [[dz, dy, dx] for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dz, dy, dx) != (0, 0, 0)]
    B, D, Z, Y, X = 1, 16, 5, 21, 37
    e, t, w = synth.synth_inputs_3d(B, D, Z, Y, X, offs, 9)
    spec = op.AffinitySpec(3, offs, None, pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED)
    et = cu(e, dev).requires_grad_(True)
    loss, affs, parts = op.FusedAffinityMSE.apply(et, None, cu(t, dev), cu(w, dev), None, spec)
    (loss * 2.0).backward()
    d = orc.make_desc(B, D, [Z, Y, X], offs, None, orc.BORDER_CROP_ZERO, orc.NORM_CROPPED)
    o_affs, o_loss = orc.c_fwd(d, e, None, t, w, None)
    o_grad, _ = orc.c_bwd(d, e, None, t, w, None, dloss=2.0)
    assert np.abs(affs.cpu().numpy() - o_affs).max() < AFFS_ATOL
    np.testing.assert_allclose(parts.cpu().numpy(), o_loss[1:], rtol=LOSS_RTOL)
    assert relmax(et.grad.cpu().numpy(), o_grad) < GRAD_RTOL


def test_wide_label_ids_are_range_checked_without_a_host_sync(pkg, dev):
    """GPU labels wider than int32: the range check runs on the device and is read back later (check_label_ranges); a bad id
    raises there, a good tensor passes through unchanged; uint64 ids >= 2^63 are caught as well"""
    op = pkg.affinity_op
    pkg.check_label_ranges()
    good = torch.tensor([[0, 5, 2 ** 31 - 1, -2 ** 31 + 1]], dtype=torch.int64, device=dev)
    assert op._labels_int32(good).tolist() == [[0, 5, 2 ** 31 - 1, -2 ** 31 + 1]]
    pkg.check_label_ranges()
    op._labels_int32(torch.tensor([[3, -2 ** 31]], dtype=torch.int64, device=dev))  # -2^31: the LDS-staged kernels' outside-the-image marker
    with pytest.raises(ValueError, match="outside marker"):
        pkg.check_label_ranges()
    op._labels_int32(torch.tensor([[0, 2 ** 31]], dtype=torch.int64, device=dev))  # does not raise here ...
    with pytest.raises(ValueError, match="fit int32"):
        pkg.check_label_ranges()                                                  # ... but here
    op._labels_int32(torch.tensor([[1, -2 ** 31 - 1]], dtype=torch.int64, device=dev))
    with pytest.raises(ValueError, match="fit int32"):
        pkg.check_label_ranges()
    pkg.check_label_ranges()  # the queue is empty again


@pytest.mark.parametrize("case", ["n26_crop", "n26_circular", "subset_3d", "diag_2d_mask", "z1_volume"])
def test_unit_box_stencils_vs_oracle(pkg, dev, orc, synth, monkeypatch, case):
    """stencils inside the unit box on the LDS-DMA box kernels (csrc/pea_box.h: X % 4 == 0, D = 16): the 26-neighbourhood with both
    borders, subsets with one sign only, a 2D diagonal stencil with a mask, partial tiles in y and x; against the C oracle, against
    the tiled kernels (PEA_BOX=0) and the inference entry point"""
    op, lib = pkg.affinity_op, pkg._lib.lib()
    n26 = [[dz, dy, dx] for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dz, dy, dx) != (0, 0, 0)]
    if case == "n26_crop":
        ndim, B, dims, offs, border, norm = 3, 2, [5, 37, 72], n26, pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED
    elif case == "n26_circular":
        ndim, B, dims, offs, border, norm = 3, 1, [4, 33, 68], n26, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_FULL
    elif case == "subset_3d":
        ndim, B, dims, offs = 3, 2, [3, 20, 40], [[-1, 0, 0], [0, -1, 1], [1, 1, -1], [0, 0, -1], [-1, -1, -1], [1, 0, 1], [0, 1, 0]]
        border, norm = pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED
    elif case == "diag_2d_mask":
        ndim, B, dims, offs = 2, 3, [1, 50, 100], [[0, -1, -1], [0, -1, 1], [0, 1, 1], [0, 0, -1], [0, -1, 0]]
        border, norm = pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX
    else:  # a volume of one plane: the z neighbours do not exist
        ndim, B, dims, offs = 3, 1, [1, 48, 64], [[0, dy, dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dy, dx) != (0, 0)]
        border, norm = pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED
    D, K = 16, len(offs)
    Z, Y, X = dims
    lam = [1.0 + 0.25 * (i % 3) for i in range(K)]
    e, t, w = synth.synth_inputs_3d(B, D, Z, Y, X, offs, 31 + K)
    m = None
    if case == "diag_2d_mask":
        m = (synth.hash_uniform(np.arange(B * K * Z * Y * X, dtype=np.uint64), 77) < 0.85).astype(np.uint8).reshape(B, K, Z, Y, X)
    shp = (lambda a: a) if ndim == 3 else (lambda a: None if a is None else a.reshape(a.shape[0], a.shape[1], Y, X))
    spec = op.AffinitySpec(ndim, [o[3 - ndim:] for o in offs], lam, border, norm)
    et = cu(shp(e), dev).requires_grad_(True)
    d_hip = op.make_desc(spec, et)
    assert lib.pea_cross_supported(ctypes.byref(d_hip), 0) == 1 and lib.pea_cross_supported(ctypes.byref(d_hip), 1) == 1
    mt = None if m is None else cu(shp(m), dev)
    loss, affs, parts = op.FusedAffinityMSE.apply(et, None, cu(shp(t), dev), cu(shp(w), dev), mt, spec)
    (loss * 0.5).backward()
    d = orc.make_desc(B, D, dims, offs, lam, border, norm, ndim=ndim)
    o_affs, o_loss = orc.c_fwd(d, e, None, t, w, m)
    o_grad, _ = orc.c_bwd(d, e, None, t, w, m, dloss=0.5)
    assert np.abs(affs.cpu().numpy().reshape(o_affs.shape) - o_affs).max() < AFFS_ATOL
    np.testing.assert_allclose(parts.cpu().numpy(), o_loss[1:], rtol=LOSS_RTOL)
    assert abs(loss.item() - o_loss[0]) <= LOSS_RTOL * o_loss[0]
    assert relmax(et.grad.cpu().numpy().reshape(o_grad.shape), o_grad) < GRAD_RTOL
    inf = op.affinity_infer(et.detach(), None, spec)
    assert torch.equal(inf, affs)
    # the tiled kernels on the same inputs: the dispatcher really took a different path above
    monkeypatch.setenv("PEA_BOX", "0")
    pkg._lib.reload_env()
    assert lib.pea_cross_supported(ctypes.byref(d_hip), 1) == 0
    et2 = cu(shp(e), dev).requires_grad_(True)
    loss2, affs2, _ = op.FusedAffinityMSE.apply(et2, None, cu(shp(t), dev), cu(shp(w), dev), mt, spec)
    (loss2 * 0.5).backward()
    assert np.abs((affs2 - affs).cpu().numpy()).max() < AFFS_ATOL
    assert relmax(et2.grad.cpu().numpy(), et.grad.cpu().numpy()) < GRAD_RTOL


def test_norm1_cross_forward_with_box_backward(pkg, dev, orc, synth, monkeypatch):
    """the reference's norm1 table (one step along z, y, x) is axis-aligned AND inside the unit box: its forward runs on the 3D cross
    kernel, its backward on the unit-box kernel (the 1 / norm plane of the one feeds the other); against the C oracle, and against
    the cross backward (PEA_BOX=0)"""
    op = pkg.affinity_op
    offs = [[-1, 0, 0], [0, -1, 0], [0, 0, -1]]
    B, D, Z, Y, X = 2, 16, 6, 40, 72
    e, t, w = synth.synth_inputs_3d(B, D, Z, Y, X, offs, 61)
    spec = op.AffinitySpec(3, offs, [2.0, 1.0, 1.0], pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED)
    d = orc.make_desc(B, D, [Z, Y, X], offs, [2.0, 1.0, 1.0], orc.BORDER_CROP_ZERO, orc.NORM_CROPPED)
    o_affs, o_loss = orc.c_fwd(d, e, None, t, w, None)
    o_grad, _ = orc.c_bwd(d, e, None, t, w, None, dloss=1.25)
    grads = []
    for box in ("1", "0"):
        monkeypatch.setenv("PEA_BOX", box)
        et = cu(e, dev).requires_grad_(True)
        loss, affs, parts = op.FusedAffinityMSE.apply(et, None, cu(t, dev), cu(w, dev), None, spec)
        (loss * 1.25).backward()
        assert np.abs(affs.cpu().numpy() - o_affs).max() < AFFS_ATOL
        np.testing.assert_allclose(parts.cpu().numpy(), o_loss[1:], rtol=LOSS_RTOL)
        assert relmax(et.grad.cpu().numpy(), o_grad) < GRAD_RTOL, box
        grads.append(et.grad.cpu().numpy())
    assert relmax(grads[0], grads[1]) < 1e-5


def test_random_unit_box_stencils_vs_oracle(pkg, dev, orc, synth):
    """seeded sweep over the unit-box kernels' domain: random subsets of the 26 displacements (both signs, duplicates excluded),
    ragged Z / Y and X % 4 == 0 widths (partial tiles, one-tile images), both borders, all three normalisers, batch, mask"""
    op, lib = pkg.affinity_op, pkg._lib.lib()
    rng = np.random.default_rng(3026)
    n26 = [[dz, dy, dx] for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dz, dy, dx) != (0, 0, 0)]
    for it in range(10):
        B = int(rng.integers(1, 3))
        Z, Y, X = int(rng.integers(2, 7)), int(rng.integers(17, 60)), 4 * int(rng.integers(9, 30))
        K = int(rng.integers(1, 27))
        offs = [n26[i] for i in rng.permutation(26)[:K]]
        border = int(rng.integers(0, 2))
        norm = int(rng.integers(0, 3))
        lam = [float(v) for v in rng.uniform(0.25, 2.0, K)]
        S = Z * Y * X
        e = synth.synth_embedding((B, 16, S), 700 + it).reshape(B, 16, Z, Y, X)
        t = (synth.hash_uniform(np.arange(B * K * S, dtype=np.uint64), 800 + it) < 0.6).astype(np.float32).reshape(B, K, Z, Y, X)
        w = (0.5 + synth.hash_uniform(np.arange(B * K * S, dtype=np.uint64), 900 + it)).astype(np.float32).reshape(B, K, Z, Y, X)
        m = (synth.hash_uniform(np.arange(B * K * S, dtype=np.uint64), 1000 + it) < 0.9).astype(np.uint8).reshape(B, K, Z, Y, X) if it % 2 else None
        spec = op.AffinitySpec(3, offs, lam, border, norm)
        et = cu(e, dev).requires_grad_(True)
        d_hip = op.make_desc(spec, et)
        assert lib.pea_cross_supported(ctypes.byref(d_hip), 1) == 1, (it, offs)
        loss, affs, parts = op.FusedAffinityMSE.apply(et, None, cu(t, dev), cu(w, dev), cu(m, dev), spec)
        (loss * 1.5).backward()
        d = orc.make_desc(B, 16, [Z, Y, X], offs, lam, border, norm, ndim=3)
        o_affs, o_loss = orc.c_fwd(d, e, None, t, w, m)
        o_grad, _ = orc.c_bwd(d, e, None, t, w, m, dloss=1.5)
        ctx = "case %d: B=%d %dx%dx%d K=%d border=%d norm=%d mask=%s" % (it, B, Z, Y, X, K, border, norm, m is not None)
        assert np.abs(affs.cpu().numpy() - o_affs).max() < AFFS_ATOL, ctx
        np.testing.assert_allclose(parts.cpu().numpy(), o_loss[1:], rtol=LOSS_RTOL, atol=1e-7, err_msg=ctx)
        assert relmax(et.grad.cpu().numpy(), o_grad) < GRAD_RTOL, ctx


def test_random_shapes_and_stencils_vs_oracle(pkg, dev, orc, synth):
    """seeded sweep over ragged sizes, offset lists (both signs), borders, batch and D"""
    op = pkg.affinity_op
    rng = np.random.default_rng(2024)
    for it in range(24):
        three_d = it % 3 == 2
        D = int(rng.choice([4, 8, 16, 32]))
        B = int(rng.integers(1, 4))
        if three_d:
            dims = [int(rng.integers(2, 7)), int(rng.integers(3, 40)), int(rng.integers(3, 70))]
        else:
            dims = [1, int(rng.integers(2, 60)), int(rng.integers(2, 300))]
        K = int(rng.integers(1, 9))
        offs = []
        for _ in range(K):
            offs.append([int(rng.integers(-(d - 1), d)) if d > 1 else 0 for d in dims])
        border = int(rng.integers(0, 2))
        norm = int(rng.integers(0, 3))
        lam = [float(v) for v in rng.uniform(0.25, 2.0, K)]
        S = dims[0] * dims[1] * dims[2]
        e = synth.synth_embedding((B, D, S), 100 + it).reshape([B, D] + dims)
        t = (synth.hash_uniform(np.arange(B * K * S, dtype=np.uint64), 200 + it) < 0.6).astype(np.float32).reshape([B, K] + dims)
        w = (0.5 + synth.hash_uniform(np.arange(B * K * S, dtype=np.uint64), 300 + it)).astype(np.float32).reshape([B, K] + dims)
        m = (synth.hash_uniform(np.arange(B * K * S, dtype=np.uint64), 400 + it) < 0.9).astype(np.uint8).reshape([B, K] + dims)
        use_mask = bool(it % 2)
        ema = synth.synth_embedding((B, D, S), 500 + it).reshape([B, D] + dims) if it % 4 == 1 else None
        spec = op.AffinitySpec(3, offs, lam, border, norm)
        et = cu(e, dev).requires_grad_(True)
        em = cu(ema, dev).requires_grad_(True) if ema is not None else None
        loss, affs, parts = op.FusedAffinityMSE.apply(et, em, cu(t, dev), cu(w, dev), cu(m, dev) if use_mask else None, spec)
        loss.backward()
        d = orc.make_desc(B, D, dims, offs, lam, border, norm, ndim=3)
        o_affs, o_loss = orc.c_fwd(d, e, ema, t, w, m if use_mask else None)
        o_grad, o_grad_e = orc.c_bwd(d, e, ema, t, w, m if use_mask else None, want_other=True)
        ctx = "case %d dims=%s offs=%s border=%d norm=%d" % (it, dims, offs, border, norm)
        assert np.abs(affs.cpu().numpy().reshape(o_affs.shape) - o_affs).max() < AFFS_ATOL, ctx
        assert abs(loss.item() - o_loss[0]) <= LOSS_RTOL * max(abs(o_loss[0]), 1e-6), ctx
        assert relmax(et.grad.cpu().numpy(), o_grad) < GRAD_RTOL, ctx
        if ema is not None:
            assert relmax(em.grad.cpu().numpy(), o_grad_e) < GRAD_RTOL, ctx


def test_tiled_crop_3d_norm5_and_ema_vs_oracle(pkg, dev, orc, synth):
    """3D shapes wide enough for the LDS-tiled CROP_ZERO kernels (the golden 3D fixtures are narrower than a tile
    and take the direct kernels): norm5 (z offsets and +-27 are 'far', the rest 'near'), self and EMA."""
    B, D, Z, Y, X = 2, 16, 5, 40, 64
    sh = [1, 1, 1, 2, 3, 3, 3, 9, 9, 4, 27, 27]
    offs = orc.norm_offsets(sh)
    e, t, w = synth.synth_inputs_3d(B, D, Z, Y, X, offs, 17)
    ema = synth.synth_embedding((B, D, Z, Y, X), 18)
    crit = pkg.WeightedMSE()
    for use_ema in (False, True):
        et = cu(e, dev).requires_grad_(True)
        if use_ema:
            loss, affs = pkg.ema_embedding_loss_norm5(et, cu(ema, dev), cu(t, dev), cu(w, dev), crit, affs0_weight=2)
        else:
            loss, affs = pkg.embedding_loss_norm5(et, cu(t, dev), cu(w, dev), crit, affs0_weight=2)
        (loss * 0.25).backward()
        d = orc.desc_3d(e, sh, orc.affs0_lambda_3d(12, 2, 3))
        o_affs, o_loss = orc.c_fwd(d, e, ema if use_ema else None, t, w, None)
        o_grad, _ = orc.c_bwd(d, e, ema if use_ema else None, t, w, None, dloss=0.25)
        assert np.abs(affs.cpu().numpy() - o_affs).max() < AFFS_ATOL
        assert abs(loss.item() - o_loss[0]) <= LOSS_RTOL * o_loss[0]
        assert relmax(et.grad.cpu().numpy(), o_grad) < GRAD_RTOL
    inf = pkg.inf_embedding_loss_norm5(cu(e, dev))
    d = orc.desc_3d(e, sh)
    assert np.abs(inf.cpu().numpy() - orc.c_fwd(d, e)[0]).max() < AFFS_ATOL


def test_tiled_fp16_d16_vs_oracle(pkg, dev, orc, synth):
    """f16 storage through the LDS-tiled kernels (D=16): f32 arithmetic on exactly the rounded inputs"""
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    B, D, H, W = 2, 16, 48, 96
    e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 23)
    e16 = e.astype(np.float16)
    et = torch.from_numpy(e16).to(dev).requires_grad_(True)
    loss, affs, _ = pkg.embedding_loss(et, cu(t, dev), cu(w, dev), cu(m, dev), pkg.WeightedMSE(), offsets)
    loss.backward()
    e_r = e16.astype(np.float32)
    d = orc.desc_2d(e_r, offsets)
    o_affs, o_loss = orc.c_fwd(d, e_r, None, t, w, m)
    o_grad, _ = orc.c_bwd(d, e_r, None, t, w, m)
    assert np.abs(affs.cpu().numpy() - o_affs).max() < AFFS_ATOL
    assert abs(loss.item() - o_loss[0]) <= LOSS_RTOL * o_loss[0]
    assert relmax(et.grad.float().cpu().numpy(), o_grad) < 2e-3  # f16 rounding of the stored gradient
    inf = pkg.embedding2affs(et.detach(), offsets)
    assert np.abs(inf.cpu().numpy() - o_affs).max() < AFFS_ATOL


def test_tiled_and_direct_kernels_agree(pkg, dev, synth, monkeypatch):
    """the LDS-tiled kernels against the global-memory kernels on the same inputs (PEA_FORCE_DIRECT switches)"""
    offsets = pkg.multi_offset([1, 3, 9], 8)  # diagonal and mixed-sign offsets: two-sided halos in x
    B, D, H, W = 2, 16, 70, 132
    e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 29)

    def run():
        et = cu(e, dev).requires_grad_(True)
        loss, affs, parts = pkg.embedding_loss(et, cu(t, dev), cu(w, dev), cu(m, dev), pkg.WeightedMSE(), offsets)
        loss.backward()
        return loss.item(), affs.cpu().numpy(), et.grad.cpu().numpy()

    monkeypatch.delenv("PEA_FORCE_DIRECT", raising=False)
    l1, a1, g1 = run()
    monkeypatch.setenv("PEA_FORCE_DIRECT", "1")
    l0, a0, g0 = run()
    assert abs(l1 - l0) <= 1e-6 * abs(l0)
    assert np.abs(a1 - a0).max() < 2e-6
    assert relmax(g1, g0) < 1e-5


def test_second_backward_over_retained_graph(pkg, dev, synth):
    """a second backward over a retained graph reuses the saved g (and 1 / norm plane): twice the grad_output, twice the gradient"""
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    e, t, w, m = synth.synth_inputs_2d(2, 16, 64, 128, offsets, 47)
    et = cu(e, dev).requires_grad_(True)
    loss, affs, _ = pkg.embedding_loss(et, cu(t, dev), cu(w, dev), cu(m, dev), pkg.WeightedMSE(), offsets)
    (g1,) = torch.autograd.grad(loss, et, retain_graph=True)
    g1 = g1.clone()
    (g2,) = torch.autograd.grad(loss * 2.0, et)
    assert relmax(g2.cpu().numpy(), 2.0 * g1.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("shape", [(64, 128), (544, 544)])
def test_labels_step_second_backward_over_retained_graph(pkg, dev, synth, shape):
    """the labels-in step hands its gradient buffer to the first backward (scaled in place); a second backward over a retained graph
    runs the step again on the saved inputs: twice the grad_output, twice the gradient, and the first result is left alone
    (one-launch kernel on the small shape, the two-launch cross path on the large one)"""
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    H, W = shape
    B = 1 if H > 100 else 2
    e = synth.synth_embedding((B, 16, H, W), 48)
    lab = torch.from_numpy(synth.synth_labels(B, (1, H, W), 49)[:, 0].copy()).to(dev)
    et = cu(e, dev).requires_grad_(True)
    loss, affs, _ = pkg.embedding_loss_from_labels(et, lab, pkg.WeightedMSE(), offsets)
    (g1,) = torch.autograd.grad(loss * 0.5, et, retain_graph=True)
    keep = g1.clone()
    (g2,) = torch.autograd.grad(loss, et)
    assert torch.equal(g1, keep)
    assert relmax(g2.cpu().numpy(), 2.0 * keep.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("ema", [False, True])
def test_tiled_d32_vs_oracle(pkg, dev, orc, synth, ema):
    """D = 32 (BBBC039V1 backbone) through the LDS-tiled kernels: 16x32 tiles, 128 B of LDS per
    region pixel, shifts 1,3,5,9,11 (scripts_bbbc/config/bbbc039v1.yaml) on a shape wider than tile + halo"""
    offsets = pkg.multi_offset([1, 3, 5, 9, 11], 4)
    B, D, H, W = 2, 32, 72, 104
    e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 53)
    other = synth.synth_embedding((B, D, H, W), 54) if ema else None
    et = cu(e, dev).requires_grad_(True)
    crit = pkg.WeightedMSE()
    if ema:
        loss, affs = pkg.ema_embedding_loss(et, cu(other, dev), cu(t, dev), cu(w, dev), cu(m, dev), crit, offsets)
    else:
        loss, affs, _ = pkg.embedding_loss(et, cu(t, dev), cu(w, dev), cu(m, dev), crit, offsets)
    (loss * 1.5).backward()
    d = orc.desc_2d(e, offsets)
    o_affs, o_loss = orc.c_fwd(d, e, other, t, w, m)
    o_grad, _ = orc.c_bwd(d, e, other, t, w, m, dloss=1.5)
    assert np.abs(affs.cpu().numpy() - o_affs).max() < AFFS_ATOL
    assert abs(loss.item() - o_loss[0]) <= LOSS_RTOL * o_loss[0]
    assert relmax(et.grad.cpu().numpy(), o_grad) < GRAD_RTOL
    inf = pkg.embedding2affs(et.detach(), offsets) if not ema else None
    if inf is not None:
        assert np.abs(inf.cpu().numpy() - o_affs).max() < AFFS_ATOL


def test_fill_border_relu_matches_reference_statements(pkg, dev):
    """scripts_ac3ac4/main.py:233-237 verbatim in torch against pea_fill_border_relu (in place, one launch)"""
    torch.manual_seed(3)
    pred = torch.randn(2, 12, 6, 20, 24, device=dev)
    ref = pred.clone()
    shift = 1
    ref[:, 1, :, :shift, :] = ref[:, 1, :, shift:shift * 2, :]
    ref[:, 2, :, :, :shift] = ref[:, 2, :, :, shift:shift * 2]
    ref[:, 0, :shift, :, :] = ref[:, 0, shift:shift * 2, :, :]
    ref = torch.nn.functional.relu(ref)
    out = pkg.fill_border_relu_(pred, shift=1, relu=True)
    assert out is pred and torch.equal(pred, ref)
    p2 = torch.randn(2, 10, 40, 48, device=dev)
    r2 = torch.nn.functional.relu(p2)
    assert torch.equal(pkg.relu_(p2), r2)
    with pytest.raises(RuntimeError):
        pkg.relu_(torch.zeros(1, 2, 4, 4))


@pytest.mark.parametrize("shape,shift,relu", [((2, 12, 6, 20, 24), 1, True), ((2, 12, 6, 20, 24), 2, True), ((1, 3, 5, 9, 16), 1, True),
                                              ((2, 12, 6, 20, 24), 1, False), ((1, 12, 7, 12, 20), 3, False), ((2, 3, 4, 10, 22), 1, True),
                                              ((1, 2, 4, 6, 8), 1, False), ((1, 1, 4, 6, 8), 2, True)])
def test_fill_border_relu_kernel_forms(pkg, dev, shape, shift, relu):
    """round 6: the three forms of pea_fill_border_relu -- four voxels per lane (X % 4 == 0), the border slices alone (relu = 0: what a
    map clamped by the forward still needs), one voxel per lane (any X) -- against scripts_ac3ac4/main.py:233-237 in torch, for the
    channel counts < 3 the statements still make sense for"""
    torch.manual_seed(sum(shape) + shift)
    pred = torch.randn(*shape, device=dev)
    ref = pred.clone()
    K = shape[1]
    if K > 1:
        ref[:, 1, :, :shift, :] = ref[:, 1, :, shift:shift * 2, :]
    if K > 2:
        ref[:, 2, :, :, :shift] = ref[:, 2, :, :, shift:shift * 2]
    ref[:, 0, :shift, :, :] = ref[:, 0, shift:shift * 2, :, :]
    if relu:
        ref = torch.nn.functional.relu(ref)
    assert torch.equal(pkg.fill_border_relu_(pred, shift=shift, relu=relu), ref)


def _section_inputs(synth, offsets, nb_half, B, D, H, W, seed):
    e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, seed)
    ema = synth.synth_embedding((B, D, H, W), seed + 1)
    emds, downs = [], []
    for j in range(4):
        k = nb_half * (4 - j)
        h, ww = H >> (j + 1), W >> (j + 1)
        ej, tj, wj, mj = synth.synth_inputs_2d(B, D, h, ww, offsets[:k], seed + 2 + j)
        emds.append(ej)
        downs.append(np.concatenate([tj, wj, mj.astype(np.float32)], axis=1))  # packed thirds, mask as float like the reference
    return e, ema, t, w, m, emds, downs


def test_cvppp_loss_section_matches_oracle(pkg, dev, orc, synth):
    """the six calls of scripts_cvppp/main.py:284-293 with their slicing and weighting, against the oracle per call"""
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    nb_half, B, D, H, W = 2, 2, 16, 96, 96
    e, ema, t, w, m, emds, downs = _section_inputs(synth, offsets, nb_half, B, D, H, W, 61)
    et = cu(e, dev).requires_grad_(True)
    emd_t = [cu(x, dev).requires_grad_(True) for x in emds]
    down_t = [cu(x, dev) for x in downs]
    loss, pred, parts = pkg.cvppp_loss_section(et, emd_t, cu(ema, dev), cu(t, dev), cu(w, dev), cu(m, dev), down_t,
                                               pkg.WeightedMSE(), offsets, nb_half, deep_weight=2, self_emb=0.7, cross_emb=1.3)
    loss.backward()
    pkg.finish_pred_2d_(pred)
    dwf = pkg.deep_weight_factor(2)
    d = orc.desc_2d(e, offsets)
    a_self, l_self = orc.c_fwd(d, e, None, t, w, m)
    _, l_cross = orc.c_fwd(d, e, ema, t, w, m)
    g_self, _ = orc.c_bwd(d, e, None, t, w, m, dloss=dwf[0] * 0.7)
    g_cross, _ = orc.c_bwd(d, e, ema, t, w, m, dloss=dwf[0] * 1.3)
    want = dwf[0] * 0.7 * l_self[0] + dwf[0] * 1.3 * l_cross[0]
    for j in range(4):
        k = nb_half * (4 - j)
        dj = orc.desc_2d(emds[j], offsets[:k])
        tj, wj, mj = downs[j][:, :k], downs[j][:, k:2 * k], downs[j][:, 2 * k:].astype(np.uint8)
        _, lj = orc.c_fwd(dj, emds[j], None, np.ascontiguousarray(tj), np.ascontiguousarray(wj), np.ascontiguousarray(mj))
        gj, _ = orc.c_bwd(dj, emds[j], None, np.ascontiguousarray(tj), np.ascontiguousarray(wj), np.ascontiguousarray(mj),
                          dloss=dwf[j + 1] * 0.7)
        want += dwf[j + 1] * 0.7 * lj[0]
        assert relmax(emd_t[j].grad.cpu().numpy(), gj) < GRAD_RTOL
    assert abs(loss.item() - want) <= LOSS_RTOL * want
    assert relmax(et.grad.cpu().numpy(), g_self + g_cross) < GRAD_RTOL
    assert np.abs(pred.cpu().numpy() - np.maximum(a_self, 0)).max() < AFFS_ATOL


def test_loss_section_graph_replay(pkg, dev, synth):
    """no host synchronisation inside the loss section: capture it (forward + backward) in a HIP graph, replay it on
    new data in the same buffers, compare with an eager run"""
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    nb_half, B, D, H, W = 2, 2, 16, 96, 96
    crit = pkg.WeightedMSE()

    def tensors(seed):
        e, ema, t, w, m, emds, downs = _section_inputs(synth, offsets, nb_half, B, D, H, W, seed)
        return [cu(e, dev)] + [cu(x, dev) for x in emds] + [cu(ema, dev), cu(t, dev), cu(w, dev), cu(m, dev)] + [cu(x, dev) for x in downs]

    def section(bufs):
        et = bufs[0].detach().requires_grad_(True)
        emd_t = [b.detach().requires_grad_(True) for b in bufs[1:5]]
        loss, pred, _ = pkg.cvppp_loss_section(et, emd_t, bufs[5], bufs[6], bufs[7], bufs[8], bufs[9:13], crit, offsets, nb_half)
        grads = torch.autograd.grad(loss, [et] + emd_t)
        return loss, pred, grads

    static = tensors(71)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            section(static)  # warm-up: kernel attributes, plan cache, allocator pools
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g_loss, g_pred, g_grads = section(static)
    fresh = tensors(73)
    for dst, src in zip(static, fresh):
        dst.copy_(src)
    graph.replay()
    torch.cuda.synchronize()
    e_loss, e_pred, e_grads = section(fresh)
    assert abs(g_loss.item() - e_loss.item()) <= 1e-6 * abs(e_loss.item())
    assert torch.equal(g_pred, e_pred)
    for a, b in zip(g_grads, e_grads):
        assert torch.equal(a, b)


@pytest.mark.parametrize("name", ["gtgt_2d_nb4", "gtgt_2d_nb8"])
def test_gpu_target_generation_matches_reference_golden(pkg, dev, name):
    """pea_gen_targets against the reference's gen_affs_ours outputs (and the restated weight_binary_ratio): bit-exact"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    lab = torch.from_numpy(g["labels"]).to(dev)
    offs = [list(o) for o in g["offsets"]]
    for padding, tag in ((True, "pad"), (False, "nopad")):
        t, m, w = pkg.gen_targets(lab, offs, padding=padding)
        assert np.array_equal(t.cpu().numpy(), g["target_" + tag])
        assert np.array_equal(m.cpu().numpy(), g["mask_" + tag])
        assert np.array_equal(w.cpu().numpy(), g["weight_" + tag])
    t2, m2 = pkg.gen_affs_ours(lab, offs, padding=True)
    assert np.array_equal(t2.cpu().numpy(), g["target_pad"]) and np.array_equal(m2.cpu().numpy(), g["mask_pad"])


def test_gpu_target_generation_3d_and_full_size(pkg, dev, orc, synth):
    """3D both-foreground targets (seg_to_aff semantics) against the oracle on a small volume, and the CVPPP bench
    size against the host generator bench.py feeds the loss with (synth.affinity_targets / class_balance_weights)"""
    offs3 = orc.norm_offsets([1, 1, 1, 2, 3, 3, 3, 9, 9, 4])
    lab = synth.synth_labels(2, (7, 21, 26), 77, cell=5)
    o_t, o_m = orc.np_gen_targets(lab, offs3, padding=False, both_foreground=True)
    t, m, w = pkg.gen_targets(torch.from_numpy(lab).to(dev), offs3, padding=False, both_foreground=True)
    assert np.array_equal(t.cpu().numpy(), o_t) and np.array_equal(m.cpu().numpy(), o_m)
    assert np.array_equal(w.cpu().numpy(), orc.np_weight_binary_ratio(o_t.reshape(2, len(offs3), -1)).reshape(o_t.shape))
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    B, H, W = 4, 544, 544
    labf = synth.synth_labels(B, (1, H, W), 555)
    ht, hm = synth.affinity_targets(labf, [[0] + list(o) for o in offsets], padding=True)
    hw = synth.class_balance_weights(ht)
    t, m, w = pkg.gen_targets(torch.from_numpy(labf[:, 0]).to(dev), offsets, padding=True)
    assert np.array_equal(t.cpu().numpy(), ht[:, :, 0]) and np.array_equal(m.cpu().numpy(), hm[:, :, 0])
    assert np.array_equal(w.cpu().numpy(), hw[:, :, 0])


def test_volume_stitcher_matches_reference_statements(pkg, dev):
    """overlapping windows blended with the reference's Gaussian weights: numpy statements of
    scripts_ac3ac4/data/provider_valid.py:320-349 against pea_stitch_add / pea_stitch_finalize, bit for bit"""
    stitch = __import__("importlib").import_module(ge.PKG_NAME + ".harness.stitch")
    C, shape, out_size, stride, vp = 3, (26, 56, 56), (18, 40, 40), (8, 16, 16), (4, 8, 8)
    rng = np.random.default_rng(5)
    w = stitch.get_weight(out_size)
    out_np = np.zeros((C,) + shape, np.float32)
    wm_np = np.zeros((1,) + shape, np.float32)
    st = pkg.VolumeStitcher(C, shape, out_size, dev)
    for fz in range(0, shape[0] - out_size[0] + 1, stride[0]):
        for fy in range(0, shape[1] - out_size[1] + 1, stride[1]):
            for fx in range(0, shape[2] - out_size[2] + 1, stride[2]):
                vol = rng.standard_normal((C,) + out_size).astype(np.float32)
                out_np[:, fz:fz + out_size[0], fy:fy + out_size[1], fx:fx + out_size[2]] += vol * w
                wm_np[:, fz:fz + out_size[0], fy:fy + out_size[1], fx:fx + out_size[2]] += w
                st.add_vol(torch.from_numpy(vol).to(dev), (fz, fy, fx))
    res_np = (out_np / wm_np)[:, vp[0]:-vp[0], vp[1]:-vp[1], vp[2]:-vp[2]]
    res = st.get_results(vp)
    assert np.array_equal(st.weight_map.cpu().numpy(), wm_np)
    assert np.array_equal(res.cpu().numpy(), res_np)


@pytest.mark.parametrize("case", ["self_nb4", "self_nb8", "ema", "f16", "d32", "d32_ema"])
@pytest.mark.parametrize("two_launch", [False, True])
def test_labels_step_matches_targets_path(pkg, dev, synth, case, two_launch, monkeypatch):
    """the labels-in training step (pea_label_weights + pea_affinity_fwd_bwd_labels[_ex]) against pea_gen_targets +
    embedding_loss / ema_embedding_loss on the same label images: loss, per-offset losses, affs, gradient.
    two_launch: the scratch-lending form at every size (on the cross kernels where they cover the case: f32 self losses with the
    axis-aligned stencils -- the labels-in forward k_fwd_xdma<.., LAB> + the cross backward; else it is the one-launch kernel)"""
    monkeypatch.setattr(pkg.affinity_op, "LABELS_TWO_LAUNCH_MIN_PX", 0 if two_launch else 1 << 62)
    nb = 8 if case == "self_nb8" else 4
    offsets = pkg.multi_offset(([1, 3, 5, 9, 11] if case.startswith("d32") else [1, 3, 5, 9, 27]) if nb == 4 else [1, 3, 9], nb)
    B, D, H, W = 3, (32 if case.startswith("d32") else 16), 80, 136
    lab = synth.synth_labels(B, (1, H, W), 91, cell=11)[:, 0]
    lab_t = torch.from_numpy(lab).to(dev)
    e = synth.synth_embedding((B, D, H, W), 92)
    if case == "f16":
        e = e.astype(np.float16)
    ema = torch.from_numpy(synth.synth_embedding((B, D, H, W), 93).astype(e.dtype)).to(dev) if case.endswith("ema") else None
    crit = pkg.WeightedMSE()
    t, m, w = pkg.gen_targets(lab_t, offsets, padding=True)

    def run(labels_in):
        et = torch.from_numpy(e).to(dev).requires_grad_(True)
        if ema is not None:
            if labels_in:
                loss, affs = pkg.ema_embedding_loss_from_labels(et, ema, lab_t, crit, offsets, affs0_weight=2)
            else:
                loss, affs = pkg.ema_embedding_loss(et, ema, t, w, m, crit, offsets, affs0_weight=2)
            parts = None
        elif labels_in:
            loss, affs, parts = pkg.embedding_loss_from_labels(et, lab_t, crit, offsets)
        else:
            loss, affs, parts = pkg.embedding_loss(et, t, w, m, crit, offsets)
        (loss * 0.5).backward()
        return loss.item(), affs.cpu().numpy(), et.grad.float().cpu().numpy(), (None if parts is None else list(parts))

    l0, a0, g0, p0 = run(False)
    l1, a1, g1, p1 = run(True)
    assert abs(l1 - l0) <= 2e-6 * abs(l0)
    assert np.abs(a1 - a0).max() < 2e-6
    assert relmax(g1, g0) < (2e-3 if case == "f16" else 1e-5)
    if p0 is not None:
        np.testing.assert_allclose(p1, p0, rtol=2e-6)
    # which cases the two-launch form really covers (the others silently ran the one-launch kernel, as documented)
    spec = pkg.affinity_op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
    sb = pkg._lib.lib().pea_labels_scratch_bytes(ctypes.byref(pkg.affinity_op.make_desc(spec, torch.from_numpy(e).to(dev))))
    assert (sb > 0) == (case in ("self_nb4", "ema", "d32", "d32_ema"))   # (the descriptor qualifies; a second operand opts out per call)


@pytest.mark.parametrize("shape,shifts,nb,flags", [((3, 80, 136), [1, 3, 5, 9, 27], 4, 5), ((2, 37, 100), [1, 3, 5, 9, 27], 4, 1), ((1, 544, 544), [1, 3, 5, 9, 27], 4, 5),
                                                   ((2, 33, 68), [1, 3, 9], 8, 7), ((1, 64, 64), [1], 4, 4), ((2, 50, 72), [1, 28], 4, 3)])
def test_label_weights_lds_counts_equal_the_global_memory_counts(pkg, dev, synth, monkeypatch, shape, shifts, nb, flags):
    """pea_label_weights: the LDS-staged count kernel (k_label_counts_lds: 32 x 64 label tiles + a halo of 28 through LDS) against the
    one-dword-loads kernel it replaces for 2D tables (PEA_FORCE_DIRECT=1 selects the latter): the counts are integers, so the weight
    tables must be EQUAL -- ragged tiles, every target flag, a diagonal table, the largest reach the halo takes (28)"""
    B, H, W = shape
    offsets = pkg.multi_offset(shifts, nb)
    lab = torch.from_numpy(synth.synth_labels(B, (1, H, W), 171, cell=9)[:, 0].copy()).to(dev)
    op, L = pkg.affinity_op, pkg._lib.lib()
    E = torch.zeros(B, 16, H, W, device=dev)
    d = op.make_desc(op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX), E)
    K = len(offsets)
    cb = L.pea_targets_workspace_bytes(ctypes.byref(d))
    P = lambda x: ctypes.c_void_p(x.data_ptr())
    out = []
    for force in ("0", "1"):
        monkeypatch.setenv("PEA_FORCE_DIRECT", force)
        pkg._lib.reload_env()
        try:
            wtab = torch.full((B * K * 2,), -1.0, device=dev)
            cnt = torch.empty(cb // 4, dtype=torch.int32, device=dev)
            assert L.pea_label_weights(ctypes.byref(d), P(lab), flags, P(wtab), P(cnt), cb, None) == 0
            torch.cuda.synchronize()
            out.append(wtab)
        finally:
            monkeypatch.delenv("PEA_FORCE_DIRECT")
            pkg._lib.reload_env()
    assert torch.equal(out[0], out[1])
    assert (out[0] >= 1.0).all() and (out[0] > 1.0).any()  # (real tables: some channel is re-weighted)


def test_labels_step_3d_and_section(pkg, dev, orc, synth):
    """3D labels-in losses (seg_to_aff(pad='') targets, both-foreground, no mask, cropped border) against gen_targets +
    the tensor API; and the multi-scale 2D loss section from label images against the tensor section"""
    crit = pkg.WeightedMSE()
    B, D, Z, Y, X = 2, 16, 6, 72, 76
    shifts = [1, 1, 1, 2, 3, 3, 3, 9, 9, 4, 27, 27]
    offs = orc.norm_offsets(shifts)
    lab = synth.synth_labels(B, (Z, Y, X), 101, cell=9)
    lab_t = torch.from_numpy(lab).to(dev)
    e = synth.synth_embedding((B, D, Z, Y, X), 102)
    ema = cu(synth.synth_embedding((B, D, Z, Y, X), 103), dev)
    t, _, w = pkg.gen_targets(lab_t, offs, padding=False, both_foreground=True, want_mask=False)
    for use_ema in (False, True):
        res = []
        for labels_in in (False, True):
            et = cu(e, dev).requires_grad_(True)
            if labels_in:
                loss, affs = (pkg.ema_embedding_loss_norm5_from_labels(et, ema, lab_t, crit, affs0_weight=2) if use_ema
                              else pkg.embedding_loss_norm5_from_labels(et, lab_t, crit, affs0_weight=2))
            else:
                loss, affs = (pkg.ema_embedding_loss_norm5(et, ema, t, w, crit, affs0_weight=2) if use_ema
                              else pkg.embedding_loss_norm5(et, t, w, crit, affs0_weight=2))
            (loss * 0.5).backward()
            res.append((loss.item(), affs.cpu().numpy(), et.grad.cpu().numpy()))
        (l0, a0, g0), (l1, a1, g1) = res
        assert abs(l1 - l0) <= 2e-6 * abs(l0)
        assert np.abs(a1 - a0).max() < 2e-6
        assert relmax(g1, g0) < 1e-5
    # multi-scale section
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    nb_half, B, D, H, W = 2, 2, 16, 96, 96
    labf = synth.synth_labels(B, (1, H, W), 105, cell=12)[:, 0]
    lab_downs = [np.ascontiguousarray(labf[:, ::2 ** (j + 1), ::2 ** (j + 1)]) for j in range(4)]
    e0 = synth.synth_embedding((B, D, H, W), 106)
    emds = [synth.synth_embedding((B, D, H >> (j + 1), W >> (j + 1)), 107 + j) for j in range(4)]
    ema2 = cu(synth.synth_embedding((B, D, H, W), 111), dev)
    labf_t, lab_downs_t = torch.from_numpy(labf).to(dev), [torch.from_numpy(x).to(dev) for x in lab_downs]
    out = []
    for labels_in in (False, True):
        et = cu(e0, dev).requires_grad_(True)
        emd_t = [cu(x, dev).requires_grad_(True) for x in emds]
        if labels_in:
            loss, pred, _ = pkg.cvppp_loss_section_from_labels(et, emd_t, ema2, labf_t, lab_downs_t, crit, offsets, nb_half, deep_weight=2)
        else:
            tt, mm, ww = pkg.gen_targets(labf_t, offsets, padding=True)
            downs = []
            for j in range(4):
                k = nb_half * (4 - j)
                tj, mj, wj = pkg.gen_targets(lab_downs_t[j], offsets[:k], padding=True)
                downs.append(torch.cat([tj, wj, mj.float()], dim=1))
            loss, pred, _ = pkg.cvppp_loss_section(et, emd_t, ema2, tt, ww, mm, downs, crit, offsets, nb_half, deep_weight=2)
        loss.backward()
        out.append((loss.item(), pred.cpu().numpy(), [et.grad.cpu().numpy()] + [x.grad.cpu().numpy() for x in emd_t]))
    assert abs(out[1][0] - out[0][0]) <= 2e-6 * abs(out[0][0])
    assert np.abs(out[1][1] - out[0][1]).max() < 2e-6
    for a, b in zip(out[1][2], out[0][2]):
        assert relmax(a, b) < 1e-5


@pytest.mark.parametrize("name", ["g3r_norm6", "g3r_norm6_ema", "g3r_norm6_ema_both"])
def test_replicate_border_norm6_matches_reference_golden(pkg, dev, name):
    """PEA_BORDER_REPLICATE (row a-15): embedding_loss_norm6 / ema_embedding_loss_norm6 against the reference's outputs,
    including the border pixels whose clamped neighbour collects several pairs, and a non-detached EMA operand"""
    g = load_golden(name)
    offs = [list(map(int, o)) for o in g["offsets"]]
    et = cu(g["e"], dev).requires_grad_(True)
    crit = pkg.WeightedMSE()
    if "ema" in g:
        mt = cu(g["ema"], dev).requires_grad_("grad_ema" in g)
        loss, affs = pkg.ema_embedding_loss_norm6(et, mt, cu(g["target"], dev), cu(g["weight"], dev), crit, shift=offs)
    else:
        loss, affs = pkg.embedding_loss_norm6(et, cu(g["target"], dev), cu(g["weight"], dev), crit, shift=offs)
    loss.backward()
    assert np.abs(affs.detach().cpu().numpy() - g["affs"]).max() < AFFS_ATOL
    assert abs(loss.item() - float(g["loss"])) <= LOSS_RTOL * float(g["loss"])
    assert relmax(et.grad.cpu().numpy(), g["grad"]) < GRAD_RTOL
    if "grad_ema" in g:
        assert relmax(mt.grad.cpu().numpy(), g["grad_ema"]) < GRAD_RTOL


@pytest.mark.parametrize("which,shape", [("self", (2, 5, 40, 72)), ("ema", (1, 4, 50, 64)), ("ema_both", (1, 3, 37, 68)), ("self", (1, 2, 33, 100))])
def test_replicate_border_on_the_tiled_kernels(pkg, dev, orc, synth, monkeypatch, which, shape):
    """PEA_BORDER_REPLICATE (row a-15) on the LDS-tiled kernels (k_fwd_tiled_v / k_fwd_tiled / k_bwd_tiled with the clamped region,
    csrc/pea_tiled.h wrap1r) at shapes their plan takes: embedding_loss_norm6 / ema_embedding_loss_norm6 against the CPU oracle --
    near offsets served from LDS (in-plane, radius <= 9), far ones (z steps, diagonals through planes, reach 27) from global memory,
    border pixels collecting the pre-images the clamp folds onto them -- and against the global-memory kernels (PEA_FORCE_DIRECT=1)"""
    B, Z, Y, X = shape
    D = 16
    offs = [[-1, 0, 0], [0, -1, 0], [0, 0, -1], [-1, -1, -1], [-1, 1, 1], [0, -9, 0], [0, 0, -9], [0, -9, -4], [0, 4, -9], [0, 9, -4],
            [0, -27, 0], [0, 0, -27], [1, 2, -2], [0, -3, 3], [0, 3, 0], [0, 0, 2]]
    K, S = len(offs), Z * Y * X
    e = synth.synth_embedding((B, D, S), 4100 + Z).reshape(B, D, Z, Y, X)
    e[0, :, 0, 0, 0] = 0.0               # a zero pixel in the corner that collects the largest pre-image
    eo = synth.synth_embedding((B, D, S), 4200 + Z).reshape(B, D, Z, Y, X)
    idx = np.arange(B * K * S, dtype=np.uint64)
    t = (synth.hash_uniform(idx, 4300) < 0.6).astype(np.float32).reshape(B, K, Z, Y, X)
    w = (0.5 + synth.hash_uniform(idx, 4400)).astype(np.float32).reshape(B, K, Z, Y, X)
    crit = pkg.WeightedMSE()
    d = orc.make_desc(B, D, [Z, Y, X], offs, None, orc.BORDER_REPLICATE, orc.NORM_FULL, ndim=3)
    other = None if which == "self" else eo
    o_affs, o_loss = orc.c_fwd(d, e, other, t, w, None)
    o_grad, o_grad_o = orc.c_bwd(d, e, other, t, w, None, dloss=0.5, want_other=(which == "ema_both"))

    def run():
        et = cu(e, dev).requires_grad_(True)
        if which == "self":
            loss, affs = pkg.embedding_loss_norm6(et, cu(t, dev), cu(w, dev), crit, shift=offs)
            mt = None
        else:
            mt = cu(eo, dev).requires_grad_(which == "ema_both")
            loss, affs = pkg.ema_embedding_loss_norm6(et, mt, cu(t, dev), cu(w, dev), crit, shift=offs)
        (loss * 0.5).backward()
        return loss.item(), affs.detach().cpu().numpy(), et.grad.cpu().numpy(), None if (mt is None or mt.grad is None) else mt.grad.cpu().numpy()

    res = {}
    for force in ("0", "1"):
        monkeypatch.setenv("PEA_FORCE_DIRECT", force)
        pkg._lib.reload_env()
        try:
            res[force] = run()
        finally:
            monkeypatch.delenv("PEA_FORCE_DIRECT")
            pkg._lib.reload_env()
        loss, affs, grad, grad_o = res[force]
        assert np.abs(affs.reshape(o_affs.shape) - o_affs).max() < AFFS_ATOL, force
        assert abs(loss - o_loss[0]) <= LOSS_RTOL * abs(o_loss[0]), force
        assert relmax(grad.reshape(o_grad.shape), o_grad) < GRAD_RTOL, force
        if which == "ema_both":
            assert relmax(grad_o.reshape(o_grad_o.shape), o_grad_o) < GRAD_RTOL, force
    # (different kernels: the sums run in different orders -- identical bits would mean the switch did nothing)
    assert not np.array_equal(res["0"][2], res["1"][2])


def test_loss_section_node_equals_composed(pkg, dev, synth):
    """cvppp_loss_section as one autograd node (weights folded into the launches) against the statement-for-statement
    composition from embedding_loss / ema_embedding_loss, with non-trivial weights and an outer factor on the loss"""
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    nb_half, B, D, H, W = 2, 2, 16, 96, 96
    e, ema, t, w, m, emds, downs = _section_inputs(synth, offsets, nb_half, B, D, H, W, 131)
    crit = pkg.WeightedMSE()
    res = []
    for fn in (pkg.cvppp_loss_section_composed, pkg.cvppp_loss_section):
        et = cu(e, dev).requires_grad_(True)
        emd_t = [cu(x, dev).requires_grad_(True) for x in emds]
        loss, pred, parts = fn(et, emd_t, cu(ema, dev), cu(t, dev), cu(w, dev), cu(m, dev), [cu(x, dev) for x in downs], crit,
                               offsets, nb_half, affs0_weight=2, deep_weight=2, self_emb=0.7, cross_emb=1.3)
        (loss * 0.5).backward()
        res.append((loss.item(), pred.cpu().numpy(), [et.grad.cpu().numpy()] + [x.grad.cpu().numpy() for x in emd_t],
                    float(parts["loss_embedding_cross"].detach()), [float(v.detach()) for v in parts["loss_emd"]]))
    assert abs(res[1][0] - res[0][0]) <= 2e-6 * abs(res[0][0])
    assert np.abs(res[1][1] - res[0][1]).max() < 2e-6
    for a, b in zip(res[1][2], res[0][2]):
        assert relmax(a, b) < 1e-5
    assert abs(res[1][3] - res[0][3]) <= 1e-5 * abs(res[0][3])
    np.testing.assert_allclose(res[1][4], res[0][4], rtol=1e-5)


def test_loss_section_second_backward_over_retained_graph(pkg, dev, synth):
    """the one-node sections (tensor path and labels-in) hand their gradient buffers to the first backward; a second backward over
    a retained graph computes the section again: twice the grad_output, twice every gradient, the first results untouched"""
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    nb_half, B, D, H, W = 2, 2, 16, 96, 96
    e, ema, t, w, m, emds, downs = _section_inputs(synth, offsets, nb_half, B, D, H, W, 137)
    crit = pkg.WeightedMSE()
    lab = synth.synth_labels(B, (1, H, W), 138)[:, 0]
    labs = [torch.from_numpy(np.ascontiguousarray(lab[:, ::2 ** j, ::2 ** j])).to(dev) for j in range(5)]
    for which in ("tensor", "labels"):
        et = cu(e, dev).requires_grad_(True)
        emd_t = [cu(x, dev).requires_grad_(True) for x in emds]
        if which == "tensor":
            loss, _, _ = pkg.cvppp_loss_section(et, emd_t, cu(ema, dev), cu(t, dev), cu(w, dev), cu(m, dev), [cu(x, dev) for x in downs],
                                                crit, offsets, nb_half)
        else:
            loss, _, _ = pkg.cvppp_loss_section_from_labels(et, emd_t, cu(ema, dev), labs[0], labs[1:], crit, offsets, nb_half)
        leaves = [et] + emd_t
        g1 = torch.autograd.grad(loss * 0.5, leaves, retain_graph=True)
        keep = [g.clone() for g in g1]
        g2 = torch.autograd.grad(loss, leaves)
        for a, k, b in zip(g1, keep, g2):
            assert torch.equal(a, k), which
            assert relmax(b.cpu().numpy(), 2.0 * k.cpu().numpy()) < 1e-6, which


def test_3d_loss_section_variants_agree(pkg, dev, orc, synth):
    """scripts_ac3ac4/main.py:219-231 (norm5 self + EMA cross + four norm1 heads): call-by-call composition, one autograd
    node on the tensor path, and the labels-in section must agree"""
    crit = pkg.WeightedMSE()
    B, D = 1, 16
    shapes = [(6, 72, 76), (6, 36, 38), (6, 36, 38), (3, 18, 20), (3, 18, 20)]  # full, emd1..emd4 (as the 3D U-Net's heads)
    sh5 = orc.norm_offsets([1, 1, 1, 2, 3, 3, 3, 9, 9, 4, 27, 27])
    sh1 = orc.norm_offsets([1, 1, 1])
    labs = [synth.synth_labels(B, s, 140 + i, cell=7) for i, s in enumerate(shapes)]
    lab_t = [torch.from_numpy(x).to(dev) for x in labs]
    embs = [synth.synth_embedding((B, D) + s, 150 + i) for i, s in enumerate(shapes)]
    ema = cu(synth.synth_embedding((B, D) + shapes[0], 160), dev)
    t0, _, w0 = pkg.gen_targets(lab_t[0], sh5, padding=False, both_foreground=True, want_mask=False)
    heads = [pkg.gen_targets(lab_t[1 + j], sh1, padding=False, both_foreground=True, want_mask=False) for j in range(4)]
    # the reference pairs emd1 with down4, ..., emd4 with down1: downs[k] belongs to emd(4-k)
    downs = [torch.cat([heads[3 - k][0], heads[3 - k][2]], dim=1) for k in range(4)]
    label_downs = [lab_t[1 + (3 - k)] for k in range(4)]
    res = []
    for which in ("composed", "node", "labels"):
        x = [cu(e, dev).requires_grad_(True) for e in embs]
        if which == "labels":
            loss, pred = pkg.ac3ac4_loss_section_from_labels(x[0], x[1:], ema, lab_t[0], label_downs, crit, embedding_mode=5, affs0_weight=2)
        else:
            fn = pkg.ac3ac4_loss_section_composed if which == "composed" else pkg.ac3ac4_loss_section
            loss, pred = fn(x[0], x[1:], ema, t0, w0, downs, crit, embedding_mode=5, affs0_weight=2)
        (loss * 0.5).backward()
        pkg.finish_pred_3d_(pred)
        res.append((loss.item(), pred.cpu().numpy(), [v.grad.cpu().numpy() for v in x]))
    for k in (1, 2):
        assert abs(res[k][0] - res[0][0]) <= 3e-6 * abs(res[0][0])
        assert np.abs(res[k][1] - res[0][1]).max() < 2e-6
        for a, b in zip(res[k][2], res[0][2]):
            assert relmax(a, b) < 1e-5


@pytest.mark.parametrize("name", ["ghead_2d_c32_d16", "ghead_2d_c64_d32", "ghead_3d_c28_d16"])
def test_head_matches_reference_golden(pkg, dev, name):
    """pea_head_fwd / pea_head_bwd (through the drop-in modules, parameters loaded the way a checkpoint would be) against
    the reference's OutConv / conv3dBlock outputs and autograd gradients.  Tolerance 1e-5 of the largest magnitude
    (f32, other summation order)."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    D, C = g["weight"].shape
    three_d = g["x"].ndim == 5
    head = (pkg.head_conv3d_block(C, D, bias=bool(g["bias"].size)) if three_d else pkg.OutConv(C, D)).to(dev)
    wkey, bkey = ("0.weight", "0.bias") if three_d else ("conv.weight", "conv.bias")
    state = {wkey: torch.from_numpy(g["weight"].reshape((D, C) + (1,) * (g["x"].ndim - 2)))}
    if g["bias"].size:
        state[bkey] = torch.from_numpy(g["bias"])
    head.load_state_dict(state)  # the reference's parameter names
    x = cu(g["x"], dev).requires_grad_(True)
    e = head(x)
    assert np.abs(e.detach().cpu().numpy() - g["e"]).max() <= 1e-5 * np.abs(g["e"]).max()
    (e * cu(g["upstream"], dev)).sum().backward()
    conv = head[0] if three_d else head.conv
    assert relmax(x.grad.cpu().numpy(), g["dx"]) <= 1e-5
    assert relmax(conv.weight.grad.cpu().numpy().reshape(D, C), g["dW"]) <= 1e-5
    if g["bias"].size:
        assert relmax(conv.bias.grad.cpu().numpy(), g["db"]) <= 1e-5


def test_head_full_size_vs_oracle_and_into_the_loss(pkg, dev, orc, synth):
    """the head at a CVPPP-sized batch against the float64 restatement (ragged pixel count: 2 x 530 x 500 is not a
    multiple of the 256-pixel chunks), dx-less backward, C ABI error codes, and head -> embedding_loss -> backward as one
    autograd graph"""
    import ctypes
    L = pkg._lib.lib()
    rng = np.random.default_rng(5)
    B, C, D, H, W = 2, 32, 16, 530, 500
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((D, C)) * 0.2).astype(np.float32)
    b = rng.standard_normal(D).astype(np.float32)
    up = rng.standard_normal((B, D, H, W)).astype(np.float32)
    X, Wt, Bt, UP = cu(x, dev), cu(w, dev), cu(b, dev), cu(up, dev)
    E = torch.empty(B, D, H, W, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.pea_head_fwd(B, C, D, H * W, p(X), p(Wt), p(Bt), p(E), st) == 0
    e_ref = orc.np_head_fwd(x, w, b)
    assert np.abs(E.cpu().numpy() - e_ref).max() <= 1e-5 * np.abs(e_ref).max()
    wsb = L.pea_head_workspace_bytes(C, D)
    work = torch.empty(wsb // 4, device=dev)
    dX, dW, dB = torch.empty_like(X), torch.empty(D, C, device=dev), torch.empty(D, device=dev)
    assert L.pea_head_bwd(B, C, D, H * W, p(X), p(Wt), p(UP), p(dX), p(dW), p(dB), p(work), wsb, st) == 0
    dx_ref, dw_ref, db_ref = orc.np_head_bwd(x, w, up)
    assert relmax(dX.cpu().numpy(), dx_ref) <= 1e-5
    # 530,000 products per entry: the f32 chains are 130 pixels per wave pass, then per-workgroup partials
    assert relmax(dW.cpu().numpy(), dw_ref) <= 2e-5 and relmax(dB.cpu().numpy(), db_ref) <= 2e-5
    dW2 = torch.empty_like(dW)
    assert L.pea_head_bwd(B, C, D, H * W, p(X), p(Wt), p(UP), None, p(dW2), None, p(work), wsb, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(dW2, dW)  # deterministic, and independent of the optional outputs
    assert L.pea_head_fwd(B, 33, D, H * W, p(X), p(Wt), p(Bt), p(E), st) == pkg._lib.E_UNSUPPORTED
    assert L.pea_head_bwd(B, C, D, H * W, p(X), p(Wt), p(UP), p(dX), p(dW), p(dB), p(work), wsb - 4, st) == -4  # PEA_E_WORKSPACE
    # ---- as a module in front of the loss
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    e0, t, wm, m = synth.synth_inputs_2d(1, D, 64, 96, offsets, 9)
    head = pkg.OutConv(C, D).to(dev)
    xs = cu(rng.standard_normal((1, C, 64, 96)).astype(np.float32), dev)
    loss, _, _ = pkg.embedding_loss(head(xs), cu(t, dev), cu(wm, dev), cu(m, dev), pkg.WeightedMSE(), offsets)
    loss.backward()
    e_np = orc.np_head_fwd(xs.cpu().numpy(), head.conv.weight.detach().cpu().numpy().reshape(D, C), head.conv.bias.detach().cpu().numpy())
    d = orc.desc_2d(e_np, offsets)
    de, _ = orc.c_bwd(d, e_np, None, t, wm, m)
    _, dw_o, db_o = orc.np_head_bwd(xs.cpu().numpy(), head.conv.weight.detach().cpu().numpy().reshape(D, C), de)
    assert relmax(head.conv.weight.grad.cpu().numpy().reshape(D, C), dw_o) <= 5e-5
    assert relmax(head.conv.bias.grad.cpu().numpy(), db_o) <= 5e-5
    with pytest.raises(RuntimeError):
        pkg.OutConv(C, D)(torch.zeros(1, C, 8, 8))  # CPU tensors are refused, no fallback


@pytest.mark.parametrize("C,D,sp", [(256, 16, (34, 34)), (128, 32, (17, 40)), (80, 16, (3, 20, 20)), (36, 16, (5, 16, 24)),
                                    (48, 16, (40, 40)), (256, 32, (9, 11))])
def test_head_wide_and_3d_channel_pairs_vs_oracle(pkg, dev, orc, C, D, sp):
    """the coarse-scale heads (channel-chunked kernels: C = 80 / 128 / 256) and the remaining 3D pairs against the
    float64 restatement; 1e-5 of the largest magnitude (2e-5 for the sums over pixels)"""
    rng = np.random.default_rng(C + D)
    B = 2
    x = rng.standard_normal((B, C) + sp).astype(np.float32)
    w = (rng.standard_normal((D, C)) * 0.1).astype(np.float32)
    b = rng.standard_normal(D).astype(np.float32)
    up = rng.standard_normal((B, D) + sp).astype(np.float32)
    head = (pkg.head_conv3d_block(C, D) if len(sp) == 3 else pkg.OutConv(C, D)).to(dev)
    conv = head[0] if len(sp) == 3 else head.conv
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(w.reshape(conv.weight.shape)))
        conv.bias.copy_(torch.from_numpy(b))
    assert pkg.model.head.head_supported(C, D)
    xt = cu(x, dev).requires_grad_(True)
    e = head(xt)
    (e * cu(up, dev)).sum().backward()
    assert relmax(e.detach().cpu().numpy(), orc.np_head_fwd(x, w, b)) <= 1e-5
    dx, dW, db = orc.np_head_bwd(x, w, up)
    assert relmax(xt.grad.cpu().numpy(), dx) <= 1e-5
    assert relmax(conv.weight.grad.cpu().numpy().reshape(D, C), dW) <= 2e-5
    assert relmax(conv.bias.grad.cpu().numpy(), db) <= 2e-5


def test_loss_section_relu_pred_option(pkg, dev, synth):
    """relu_pred=True: the section's affinity map comes back already clamped (the reference's `pred = F.relu(pred)`,
    scripts_cvppp/main.py:312, applied by the kernel that writes it); loss and gradients are untouched"""
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    nb_half, B, D, H, W = 2, 2, 16, 96, 96
    e, ema, t, w, m, emds, downs = _section_inputs(synth, offsets, nb_half, B, D, H, W, 77)
    crit = pkg.WeightedMSE()
    labs = synth.synth_labels(B, (1, H, W), 78)[:, 0]
    lab_t = [torch.from_numpy(np.ascontiguousarray(labs[:, ::2 ** j, ::2 ** j])).to(dev) for j in range(5)]
    for path in ("tensor", "labels"):
        res = []
        for flag in (False, True):
            et = cu(e, dev).requires_grad_(True)
            emd_t = [cu(x, dev).requires_grad_(True) for x in emds]
            if path == "tensor":
                loss, pred, _ = pkg.cvppp_loss_section(et, emd_t, cu(ema, dev), cu(t, dev), cu(w, dev), cu(m, dev),
                                                       [cu(x, dev) for x in downs], crit, offsets, nb_half, relu_pred=flag)
            else:
                loss, pred, _ = pkg.cvppp_loss_section_from_labels(et, emd_t, cu(ema, dev), lab_t[0], lab_t[1:], crit, offsets, nb_half,
                                                                   relu_pred=flag)
            loss.backward()
            res.append((loss.item(), pred.clone(), et.grad.clone()))
        assert res[0][0] == res[1][0] and torch.equal(res[0][2], res[1][2])
        assert float(res[0][1].min()) < 0.0 and torch.equal(torch.relu(res[0][1]), res[1][1])
    # the class-balance tables computed ahead of the section: bit-identical results
    tabs = pkg.cvppp_label_weight_tables(lab_t[0], lab_t[1:], offsets, nb_half)
    et = cu(e, dev).requires_grad_(True)
    emd_t = [cu(x, dev).requires_grad_(True) for x in emds]
    loss, pred, _ = pkg.cvppp_loss_section_from_labels(et, emd_t, cu(ema, dev), lab_t[0], lab_t[1:], crit, offsets, nb_half,
                                                       relu_pred=True, weight_tables=tabs)
    loss.backward()
    assert loss.item() == res[1][0] and torch.equal(pred, res[1][1]) and torch.equal(et.grad, res[1][2])


@pytest.mark.parametrize("case", ["self_f32", "ema_f32", "self_f16", "zero_px"])
def test_chunked_d64_forward_vs_oracle(pkg, dev, orc, synth, monkeypatch, case):
    """D = 64 (BASELINE config 5) through the channel-chunked tiled forward (two chunks of 32 channels staged raw, norms
    applied to the raw dot products): training forward + inference against the C oracle, with the CVPPP stencil (two far
    offsets), and against the direct kernels it replaces (PEA_FORCE_DIRECT=1)"""
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    B, D, H, W = 2, 64, 70, 100
    e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 63)
    if case == "zero_px":
        e[:, :, 10:14, 20:60] = 0.0  # zero-norm pixels: ehat = 0, no NaN
    f16 = case == "self_f16"
    if f16:
        e = e.astype(np.float16).astype(np.float32)  # the oracle sees the values the kernel loads
    other = synth.synth_embedding((B, D, H, W), 64) if case == "ema_f32" else None
    crit = pkg.WeightedMSE()

    def run():
        et = cu(e, dev)
        et = (et.half() if f16 else et).requires_grad_(True)
        if other is not None:
            loss, affs = pkg.ema_embedding_loss(et, cu(other, dev), cu(t, dev), cu(w, dev), cu(m, dev), crit, offsets)
        else:
            loss, affs, _ = pkg.embedding_loss(et, cu(t, dev), cu(w, dev), cu(m, dev), crit, offsets)
        loss.backward()
        inf = pkg.embedding2affs(et.detach(), offsets) if other is None else affs
        return loss.item(), affs.cpu().numpy(), inf.cpu().numpy(), et.grad.float().cpu().numpy()

    got = run()
    d = orc.desc_2d(e, offsets)
    o_affs, o_loss = orc.c_fwd(d, e, other, t, w, m)
    o_grad, _ = orc.c_bwd(d, e, other, t, w, m)
    assert np.isfinite(got[1]).all()
    assert np.abs(got[1] - o_affs).max() < AFFS_ATOL and np.abs(got[2] - o_affs).max() < AFFS_ATOL
    assert abs(got[0] - o_loss[0]) <= LOSS_RTOL * o_loss[0]
    assert relmax(got[3], o_grad) < (2e-3 if f16 else GRAD_RTOL)  # f16: the gradient itself is stored in half precision
    monkeypatch.setenv("PEA_FORCE_DIRECT", "1")
    ref = run()
    assert np.abs(got[1] - ref[1]).max() < 2e-6 and abs(got[0] - ref[0]) <= 2e-6 * abs(ref[0])


@pytest.mark.parametrize("D,shape,border", [(64, (72, 104), "circular"), (32, (50, 72), "crop"), (16, (37, 64), "circular"),
                                            (64, (50, 136), "crop"), (32, (90, 128), "circular"), (64, (44, 200), "circular"),
                                            (32, (41, 168), "crop")])
def test_f16_cross_kernels_both_working_buffers(pkg, dev, orc, synth, monkeypatch, D, shape, border):
    """f16 storage on the LDS-DMA cross kernels (csrc/pea_xdma_h16.h): the half-precision working buffer (forward on v_dot2_f32_f16,
    backward on v_fma_mix_f32) with the backward on producer / consumer waves where the image is wide enough for its 8 x 64 tiles
    (csrc/pea_xdma_hq.h: PEA_H16_HW=2, the default), the same on the LDS-DMA backward (=1) and the f32 working buffer (=0), each against
    the C oracle on the rounded inputs, and against each other (2 and 1 bit for bit: the same arithmetic in the same order); training
    forward, projection-first backward (D >= 32) and inference"""
    H, W = shape
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)[:8]
    B = 2
    e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 90 + D)
    e = e.astype(np.float16).astype(np.float32)
    op = pkg.affinity_op
    bd = pkg._lib.BORDER_CIRCULAR if border == "circular" else pkg._lib.BORDER_CROP_ZERO
    spec = op.AffinitySpec(2, offsets, None, bd, pkg._lib.NORM_BX)

    def run():
        et = cu(e, dev).half().requires_grad_(True)
        d_hip = op.make_desc(spec, et)
        assert pkg._lib.lib().pea_cross_supported(ctypes.byref(d_hip), 0) == 1
        loss, affs, _ = op.FusedAffinityMSE.apply(et, None, cu(t, dev), cu(w, dev), cu(m, dev), spec)
        (loss * 0.75).backward()
        inf = op.affinity_infer(et.detach(), None, spec)
        return loss.item(), affs.cpu().numpy(), inf.cpu().numpy(), et.grad.float().cpu().numpy()

    d = orc.make_desc(B, D, [1, H, W], offsets, None, bd, pkg._lib.NORM_BX, ndim=2)
    o_affs, o_loss = orc.c_fwd(d, e, None, t, w, m)
    o_grad, _ = orc.c_bwd(d, e, None, t, w, m, dloss=0.75)
    res = {}
    for hw in ("2", "1", "0"):
        monkeypatch.setenv("PEA_H16_HW", hw)
        got = res[hw] = run()
        assert np.abs(got[1].reshape(o_affs.shape) - o_affs).max() < AFFS_ATOL, hw
        assert np.abs(got[2].reshape(o_affs.shape) - o_affs).max() < AFFS_ATOL, hw
        assert abs(got[0] - o_loss[0]) <= LOSS_RTOL * o_loss[0], hw
        assert relmax(got[3].reshape(o_grad.shape), o_grad) < 2e-3, hw  # the gradient is stored in half precision
    assert np.abs(res["1"][1] - res["0"][1]).max() < 2e-6
    assert relmax(res["1"][3], res["0"][3]) < 2e-3
    assert np.array_equal(res["2"][3], res["1"][3]) and np.array_equal(res["2"][1], res["1"][1])


@pytest.mark.parametrize("scale", [1e-3, 3e-5, 1e-6])
def test_f16_denormal_embeddings(pkg, dev, orc, synth, scale):
    """f16 storage with embeddings so small that most or all halves are DENORMAL numbers: the cosine does not depend on the scale,
    and the v_dot2_f32_f16 forward must not flush them (it does not: measured 2e-7 against the oracle at every scale)"""
    op = pkg.affinity_op
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)[:8]
    B, D, H, W = 1, 64, 48, 72
    e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, 5)
    e16 = (e * scale).astype(np.float16)
    ef = e16.astype(np.float32)
    assert ((np.abs(ef) < 6.1e-5) & (ef != 0)).sum() > (0.04 if scale > 1e-4 else 0.9) * ef.size
    spec = op.AffinitySpec(2, offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
    d = orc.make_desc(B, D, [1, H, W], offsets, None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX, ndim=2)
    o_affs, o_loss = orc.c_fwd(d, ef, None, t, w, m)
    et = torch.from_numpy(e16).to(dev).requires_grad_(True)
    loss, affs, _ = op.FusedAffinityMSE.apply(et, None, cu(t, dev), cu(w, dev), cu(m, dev), spec)
    assert np.abs(affs.cpu().numpy().reshape(o_affs.shape) - o_affs).max() < AFFS_ATOL
    assert abs(loss.item() - o_loss[0]) <= LOSS_RTOL * o_loss[0]
    inf = op.affinity_infer(et.detach(), None, spec)
    assert np.abs(inf.cpu().numpy().reshape(o_affs.shape) - o_affs).max() < AFFS_ATOL


def test_plain_c_consumer_of_the_abi(tmp_path):
    """examples/abi_demo.c: the library driven from plain C (gcc, HIP runtime for memory, no Python / torch in the process)
    against a scalar double-precision restatement of the reference's op sequence written out in the program"""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("needs gcc and the ROCm headers")
    csrc = os.path.join(root, ge.PKG_DIR if hasattr(ge, "PKG_DIR") else "pixel-embedded-affinity_amd", "csrc")
    exe = str(tmp_path / "abi_demo")
    subprocess.check_call(["gcc", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", "abi_demo.c"), "-L" + csrc, "-lpea_hip", "-L/opt/rocm/lib", "-lamdhip64",
                           "-lm", "-o", exe])
    env = dict(os.environ, LD_LIBRARY_PATH=csrc + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "abi_demo: OK" in out.stdout, out.stdout + out.stderr


def test_randomised_tiled_vs_direct_sweep():
    """tests/fuzz/fuzz_tiled_vs_direct.py: 80 random configurations (D 16 / 32 / 64, f32 / f16, self / EMA, 2D / 3D, random
    stencils, masks, normalisers) on shapes wide enough for the LDS-tiled kernels, default dispatch against the direct
    kernels, the oracle as referee on a disagreement"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz", "fuzz_tiled_vs_direct.py"), "80", "23"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_randomised_alternative_paths_sweep():
    """tests/fuzz/fuzz_paths.py: 40 random configurations of the paths that must give the same numbers -- the labels-in step
    against gen_targets + the tensor path, the one-launch step against forward + backward, the embedding head against
    torch's GPU convolution (every supported channel pair, ragged pixel counts)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz", "fuzz_paths.py"), "40", "31"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_end_to_end_training_step_demo():
    """examples/train_step_demo.py: stand-in backbone -> HIP embedding heads -> labels-in loss section (one node, second
    stream, tables computed ahead) -> backward -> Adam -> EMA teacher, a few steps; the loss must fall"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "train_step_demo.py"), "6"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and "train_step_demo: OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_validation_section_matches_call_by_call(pkg, dev, orc, synth):
    """cvppp_validation_section (scripts_cvppp/inference.py:179-193): the five losses summed unweighted and relu(pred)
    against the same calls made one by one, and against the oracle for the full-resolution loss; test mode = embedding2affs"""
    offsets = pkg.multi_offset([1, 3, 5, 9, 27], 4)
    nb_half, B, D, H, W = 2, 2, 16, 96, 128
    e, _, t, w, m, emds, downs = _section_inputs(synth, offsets, nb_half, B, D, H, W, 91)
    crit = pkg.WeightedMSE()
    E, T, Wt, M = cu(e, dev), cu(t, dev), cu(w, dev), cu(m, dev)
    emd_t, down_t = [cu(x, dev) for x in emds], [cu(x, dev) for x in downs]
    loss, pred = pkg.cvppp_validation_section(E, emd_t, T, Wt, M, down_t, crit, offsets, nb_half)
    ref = 0.0
    for j in range(4):
        k = nb_half * (4 - j)
        l, _, _ = pkg.embedding_loss(emd_t[j], down_t[j][:, 0:k], down_t[j][:, k:2 * k], down_t[j][:, 2 * k:3 * k], crit, offsets[:k])
        ref += l.item()
    l0, a0, _ = pkg.embedding_loss(E, T, Wt, M, crit, offsets)
    assert abs(loss.item() - (ref + l0.item())) <= 1e-6 * abs(ref + l0.item())
    assert torch.equal(pred, torch.relu(a0)) and not loss.requires_grad
    o_affs, o_loss = orc.c_fwd(orc.desc_2d(e, offsets), e, None, t, w, m)
    assert abs(l0.item() - o_loss[0]) <= LOSS_RTOL * o_loss[0]
    none, pred_t = pkg.cvppp_validation_section(E, emd_t, T, Wt, M, down_t, crit, offsets, nb_half, test_mode=True)
    assert none is None and np.abs(pred_t.cpu().numpy() - np.maximum(o_affs, 0)).max() < AFFS_ATOL


def test_randomised_data_format_paths_sweep():
    """tests/fuzz/fuzz_formats.py: 30 random configurations of the data-format paths either side of the loss -- target
    generation bit-exact against the numpy restatement, the 3D labels-in losses against targets + tensor functions, the
    replicate-border variant against the C oracle, the device stitcher bit-exact against the reference's numpy statements"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz", "fuzz_formats.py"), "30", "17"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_seg_to_aff_matches_reference_golden(pkg, dev):
    """pkg.seg_to_aff / gen_targets(both_foreground) on the GPU, bit-exact against the reference's seg_to_aff outputs"""
    g = load_golden("gseg2aff_3d")
    seg = cu(g["seg"][None], dev)
    a3 = pkg.seg_to_aff(seg)
    assert np.array_equal(a3[0].cpu().numpy(), g["aff3_replicate"])
    nh = lambda a, b, c: [[-a, 0, 0], [0, -b, 0], [0, 0, -c]]
    a12 = torch.cat([pkg.seg_to_aff(seg, pad=''), pkg.seg_to_aff(seg, nh(2, 3, 3), pad=''), pkg.seg_to_aff(seg, nh(3, 9, 9), pad=''),
                     pkg.seg_to_aff(seg, nh(4, 27, 27), pad='')], dim=1)
    assert np.array_equal(a12[0].cpu().numpy(), g["aff12_nopad"])


@pytest.mark.parametrize("path", ["one_node", "composed", "labels"])
def test_cvppp_loss_section_matches_reference_golden(pkg, dev, path):
    """the six-loss section of the training loop against the reference's own functions called in scripts_cvppp/main.py:284-293's
    order (tests/golden/gsection_cvppp.npz): per-loss values, total, relu(pred) and the five gradients"""
    g = load_golden("gsection_cvppp")
    offsets = g["offsets"].tolist()
    nb_half = 2
    crit = pkg.WeightedMSE()
    embs = [cu(g["emb%d" % j], dev).requires_grad_(True) for j in range(5)]
    ema = cu(g["ema"], dev)
    if path == "labels":
        labs = [cu(g["lab%d" % j], dev) for j in range(5)]
        loss, pred, parts = pkg.cvppp_loss_section_from_labels(embs[0], embs[1:], ema, labs[0], labs[1:], crit, offsets, nb_half, relu_pred=True)
    else:
        downs = [torch.cat([cu(g["t%d" % j], dev), cu(g["w%d" % j], dev), cu(g["m%d" % j], dev).float()], dim=1) for j in range(1, 5)]
        fn = pkg.cvppp_loss_section if path == "one_node" else pkg.cvppp_loss_section_composed
        kw = {"relu_pred": True} if path == "one_node" else {}
        loss, pred, parts = fn(embs[0], embs[1:], ema, cu(g["t0"], dev), cu(g["w0"], dev), cu(g["m0"], dev), downs, crit, offsets, nb_half, **kw)
    loss.backward()
    if path == "composed":
        pkg.finish_pred_2d_(pred)
    assert abs(loss.item() - float(g["total"])) <= 1e-5 * abs(float(g["total"]))
    assert np.abs(pred.cpu().numpy() - g["pred"]).max() < AFFS_ATOL
    for j in range(5):
        assert relmax(embs[j].grad.cpu().numpy(), g["grad%d" % j]) < GRAD_RTOL, j


def test_affs_activations_match_reference_golden(pkg, dev):
    """the activation flags of the affs output (include/pea.h PEA_FLAG_*) against what the reference computes: relu and 1 - relu
    of the shipped map (inference.py:193, seg_mutex.py:5) and the (a + 1) / 2 + clamp map of loss_embedding.py's embedding2affs"""
    g = load_golden("gact_2d")
    e, offsets = cu(g["e"], dev), g["offsets"].tolist()
    assert np.abs(pkg.embedding2affs(e, offsets, activation="relu").cpu().numpy() - g["relu_ours"]).max() < AFFS_ATOL
    assert np.abs(pkg.embedding2affs(e, offsets, activation="mutex").cpu().numpy() - g["mutex_ours"]).max() < AFFS_ATOL
    assert np.abs(pkg.embedding2affs(e, offsets, mode="cos", activation="half_clamp").cpu().numpy() - g["half_clamp_cos"]).max() < AFFS_ATOL
    half = pkg.embedding2affs(e, offsets, activation="half").cpu().numpy()
    raw = pkg.embedding2affs(e, offsets).cpu().numpy()
    assert np.abs(half - (raw + 1) / 2).max() < 1e-6
    # the hand-off: one pinned [N, K, H, W] array in the layout of affs.hdf, 1 - affs for elf
    ho = __import__("importlib").import_module(ge.PKG_NAME + ".harness.handoff")
    col = ho.AffsCollector(2, len(offsets), 40, 72)
    col.add(pkg.embedding2affs(e, offsets, activation="relu"))
    col.add(pkg.embedding2affs(e, offsets, activation="relu")[0])
    a = col.numpy()
    assert a.shape == (2, len(offsets), 40, 72) and a.flags["C_CONTIGUOUS"] and np.abs(a[1] - g["relu_ours"][0]).max() < AFFS_ATOL
    assert np.abs(col.mutex_input(0) - g["mutex_ours"][0]).max() < AFFS_ATOL


def test_degenerate_inputs_vs_oracle(pkg, dev, orc, synth):
    """the corners of the domain: an all-zero mask (no pixel carries loss: loss and gradient exactly 0), all-zero weights, a single
    offset, an offset as long as the image (torch.roll folds it to 0: a = 1 wherever the embedding is not zero), the smallest
    images (every kernel family has to refuse them down to the direct kernels), an all-zero embedding (clamp branch everywhere)"""
    crit = pkg.WeightedMSE()

    def run(e, t, w, m, offsets):
        x = cu(e, dev).requires_grad_(True)
        loss, affs, parts = pkg.embedding_loss(x, cu(t, dev), cu(w, dev), cu(m, dev), crit, offsets)
        loss.backward()
        d = orc.desc_2d(e, offsets)
        o_affs, o_loss = orc.c_fwd(d, e, None, t, w, m)
        o_grad, _ = orc.c_bwd(d, e, None, t, w, m)
        return loss.item(), affs.cpu().numpy(), x.grad.cpu().numpy(), o_loss[0], o_affs, o_grad

    offsets = pkg.multi_offset([1, 3, 5, 9, 27], neighbor=4)
    B, D, H, W = 2, 16, 64, 96
    e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, seed=808)
    # all-zero mask / all-zero weights: nothing to learn from
    for tt, ww, mm in ((t, w, np.zeros_like(m)), (t, np.zeros_like(w), m)):
        l, a, g, ol, oa, og = run(e, tt, ww, mm, offsets)
        assert l == 0.0 and ol == 0.0 and not g.any() and not og.any() and np.abs(a - oa).max() < AFFS_ATOL
    # all-zero embedding: every pixel takes the clamp branch of F.normalize; affs are 0, the gradient is finite
    l, a, g, ol, oa, og = run(np.zeros_like(e), t, w, m, offsets)
    assert not a.any() and np.isfinite(g).all() and abs(l - ol) <= LOSS_RTOL * abs(ol) and np.abs(g - og).max() <= GRAD_RTOL * np.abs(og).max() + 1e-6
    # a single offset, and one as long as the image (folds to the zero offset)
    for offs in ([[-1, 0]], [[0, -W]], [[-H, 0], [0, -1]]):
        K = len(offs)
        l, a, g, ol, oa, og = run(e, t[:, :K], w[:, :K], m[:, :K], offs)
        assert np.abs(a - oa).max() < AFFS_ATOL and abs(l - ol) <= LOSS_RTOL * abs(ol) + 1e-12
        assert np.abs(g - og).max() <= GRAD_RTOL * np.abs(og).max() + 1e-6   # (+ floor: a folded offset's gradient is rounding noise)
    assert np.abs(run(e, t[:, :1], w[:, :1], m[:, :1], [[0, -W]])[1] - 1.0).max() < 1e-5
    # the smallest images
    for (h, wd) in ((2, 4), (3, 5), (1, 8), (8, 1)):
        offs = [[-1, 0], [0, -1]]
        e2, t2, w2, m2 = synth.synth_inputs_2d(1, D, h, wd, offs, seed=900 + h)
        l, a, g, ol, oa, og = run(e2, t2, w2, m2, offs)
        assert np.abs(a - oa).max() < AFFS_ATOL and abs(l - ol) <= LOSS_RTOL * abs(ol) + 1e-12
        assert np.abs(g - og).max() <= GRAD_RTOL * np.abs(og).max() + 1e-6   # (+ floor: a folded offset's gradient is rounding noise)


@pytest.mark.parametrize("name,path", [("gsection_ac3ac4_norm5", "one_node"), ("gsection_ac3ac4_norm5", "composed"),
                                       ("gsection_ac3ac4_norm1", "one_node"), ("gsection_ac3ac4_norm1", "composed")])
def test_ac3ac4_loss_section_matches_reference_golden(pkg, dev, name, path):
    """the loss section of the 3D training loop against the reference's own functions called in scripts_ac3ac4/main.py:219-237's
    order (tests/golden/make_golden.py case_section_3d): total, the border-filled relu'd pred, the five gradients -- pins the
    emd1 <-> down4 .. emd4 <-> down1 pairing to a run of the reference"""
    g = load_golden(name)
    crit = pkg.WeightedMSE()
    emb = cu(g["emb"], dev).requires_grad_(True)
    emds = [cu(g["emd%d" % j], dev).requires_grad_(True) for j in range(1, 5)]
    downs = [cu(g["down%d" % j], dev) for j in range(1, 5)]
    fn = pkg.ac3ac4_loss_section if path == "one_node" else pkg.ac3ac4_loss_section_composed
    loss, pred = fn(emb, emds, cu(g["ema"], dev), cu(g["target"], dev), cu(g["weight"], dev), downs, crit,
                    embedding_mode=int(g["mode"]), affs0_weight=1)
    loss.backward()
    pred = pkg.finish_pred_3d_(pred.clone())
    assert abs(loss.item() - float(g["total"])) <= 1e-5 * abs(float(g["total"]))
    assert np.abs(pred.cpu().numpy() - g["pred"]).max() < AFFS_ATOL
    assert relmax(emb.grad.cpu().numpy(), g["grad_emb"]) < GRAD_RTOL
    for j in range(1, 5):
        assert relmax(emds[j - 1].grad.cpu().numpy(), g["grad_emd%d" % j]) < GRAD_RTOL, j


@pytest.mark.parametrize("name", ["gsection_ac3ac4_norm1", "gsection_ac3ac4_norm5"])
def test_ac3ac4_loss_section_finished_pred(pkg, dev, name):
    """round 6: ac3ac4_loss_section(finish_pred=True) -- the forward clamps the map, a border-only launch fills the three slices, the
    cross loss' gradient is ADDED by its kernel (no second buffer) -- against the reference's run (scripts_ac3ac4/main.py:219-237)
    and, bit for bit, against the unfinished section + finish_pred_3d_"""
    g = load_golden(name)
    crit = pkg.WeightedMSE()
    downs = [cu(g["down%d" % j], dev) for j in range(1, 5)]

    def run(finish):
        emb = cu(g["emb"], dev).requires_grad_(True)
        emds = [cu(g["emd%d" % j], dev).requires_grad_(True) for j in range(1, 5)]
        loss, pred = pkg.ac3ac4_loss_section(emb, emds, cu(g["ema"], dev), cu(g["target"], dev), cu(g["weight"], dev), downs, crit,
                                             embedding_mode=int(g["mode"]), affs0_weight=1, finish_pred=finish)
        loss.backward()
        if not finish:
            pred = pkg.finish_pred_3d_(pred.clone())
        return loss.detach(), pred, emb.grad, [e.grad for e in emds]

    l1, p1, g1, s1 = run(True)
    l0, p0, g0, s0 = run(False)
    assert torch.equal(l1, l0) and torch.equal(p1, p0) and torch.equal(g1, g0) and all(torch.equal(a, b) for a, b in zip(s1, s0))
    assert abs(l1.item() - float(g["total"])) <= 1e-5 * abs(float(g["total"]))
    assert np.abs(p1.cpu().numpy() - g["pred"]).max() < AFFS_ATOL
    assert relmax(g1.cpu().numpy(), g["grad_emb"]) < GRAD_RTOL


def test_validation_section_matches_reference_golden(pkg, dev):
    """cvppp_validation_section (scripts_cvppp/inference.py:179-193) on the inputs of gsection_cvppp: the unweighted sum of the five
    self losses the reference's functions produced, and relu(pred); the test-mode branch = relu(embedding2affs)"""
    g = load_golden("gsection_cvppp")
    offsets = g["offsets"].tolist()
    crit = pkg.WeightedMSE()
    embs = [cu(g["emb%d" % j], dev) for j in range(5)]
    downs = [torch.cat([cu(g["t%d" % j], dev), cu(g["w%d" % j], dev), cu(g["m%d" % j], dev).float()], dim=1) for j in range(1, 5)]
    loss, pred = pkg.cvppp_validation_section(embs[0], embs[1:], cu(g["t0"], dev), cu(g["w0"], dev), cu(g["m0"], dev), downs, crit, offsets, 2)
    want = float(g["losses"][:5].sum())
    assert abs(loss.item() - want) <= 1e-5 * abs(want)
    assert np.abs(pred.cpu().numpy() - g["pred"]).max() < AFFS_ATOL
    none, pred_t = pkg.cvppp_validation_section(embs[0], embs[1:], None, None, None, None, crit, offsets, 2, test_mode=True)
    assert none is None and np.abs(pred_t.cpu().numpy() - g["pred"]).max() < AFFS_ATOL
