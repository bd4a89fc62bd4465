"""CPU: pin the oracle (C, numpy and torch restatements) against the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_names, lam_for, load_golden, shifts_for

G2D = [n for n in golden_names("g2d_") if "summary" not in n]
G3D = [n for n in golden_names("g3d_") if "march" not in n]  # (the *_march fixtures are summaries: test_gpu_zmarch.py, test_oracle.py below)


def _desc(orc, g):
    kind = str(g["kind"])
    e = g["e"]
    if kind.startswith("2d"):
        K = len(g["offsets"])
        mode = str(g["mode"]) if "mode" in g else "ours"
        return orc.desc_2d(e, g["offsets"].tolist(), lam_for(g, K), mode=mode), K
    sh = shifts_for(g)
    return orc.desc_3d(e, sh, lam_for(g, len(sh))), len(sh)


@pytest.mark.parametrize("name", G2D + G3D)
def test_c_oracle_forward_matches_reference(orc, name):
    g = load_golden(name)
    d, K = _desc(orc, g)
    affs, loss = orc.c_fwd(d, g["e"], g.get("ema"), g["target"], g["weight"], g.get("mask"))
    assert np.abs(affs - g["affs"]).max() < 2e-6  # fp32 rounding only (different summation order)
    assert abs(loss[0] - float(g["loss"])) <= 2e-6 * max(1.0, abs(float(g["loss"])))
    if "all_loss" in g:
        np.testing.assert_allclose(loss[1:], g["all_loss"], rtol=3e-6, atol=1e-9)
    if "affs_infer" in g:
        inf, _ = orc.c_fwd(d, g["e"], None)
        assert np.abs(inf - g["affs_infer"]).max() < 2e-6


@pytest.mark.parametrize("name", G2D + G3D)
def test_c_oracle_backward_matches_reference_autograd(orc, name):
    g = load_golden(name)
    d, K = _desc(orc, g)
    want_other = "grad_ema" in g
    de, de_o = orc.c_bwd(d, g["e"], g.get("ema"), g["target"], g["weight"], g.get("mask"), 1.0, want_other)
    scale = np.abs(g["grad"]).max()
    assert np.abs(de - g["grad"]).max() <= 2e-5 * scale + 1e-12
    if want_other:
        assert np.abs(de_o - g["grad_ema"]).max() <= 2e-5 * np.abs(g["grad_ema"]).max()


def test_zero_norm_pixels_follow_clamp(orc):
    """F.normalize's eps clamp: the fixture has an all-zero pixel and a 1e-14 pixel."""
    g = load_golden("g2d_k8_zero")
    assert np.all(g["e"][0, :, 3, 4] == 0)
    d, _ = _desc(orc, g)
    de, _ = orc.c_bwd(d, g["e"], None, g["target"], g["weight"], g["mask"])
    ref = g["grad"][0, :, 3, 4]
    assert np.abs(ref).max() > 0  # the reference does propagate G/eps there
    np.testing.assert_allclose(de[0, :, 3, 4], ref, rtol=2e-4)


@pytest.mark.parametrize("name", G2D)
def test_numpy_restatement_2d(orc, name):
    g = load_golden(name)
    mode = str(g["mode"]) if "mode" in g else "ours"
    a0 = float(g["affs0_weight"]) if "affs0_weight" in g else 1
    loss, affs, all_loss = orc.np_embedding_loss(g["e"], g["target"], g["weight"], g["mask"], g["offsets"].tolist(),
                                                 ema=g.get("ema"), affs0_weight=a0, mode=mode)
    assert np.abs(affs - g["affs"]).max() < 2e-6
    assert abs(loss - float(g["loss"])) <= 3e-6 * max(1.0, abs(float(g["loss"])))


@pytest.mark.parametrize("name", G3D)
def test_numpy_restatement_3d(orc, name):
    g = load_golden(name)
    sh = shifts_for(g)
    first = 1 if "norm1" in str(g["kind"]) else 3
    loss, affs = orc.np_embedding_loss_3d(g["e"], g["target"], g["weight"], sh, ema=g.get("ema"),
                                          affs0_weight=float(g["affs0_weight"]), first=first)
    assert np.abs(affs - g["affs"]).max() < 2e-6
    assert abs(loss - float(g["loss"])) <= 3e-6 * max(1.0, abs(float(g["loss"])))


@pytest.mark.parametrize("name", ["g2d_cvppp_k10", "g2d_ema_detach", "g3d_norm5", "g3d_norm1_ema"])
def test_torch_cpu_baseline_restatement(orc, name):
    """the op sequence timed as bench.py's cpu_baseline reproduces the reference (values and autograd)"""
    g = load_golden(name)
    e = torch.from_numpy(g["e"]).requires_grad_(True)
    ema = torch.from_numpy(g["ema"]) if "ema" in g else None
    a0 = float(g["affs0_weight"]) if "affs0_weight" in g else 1
    if str(g["kind"]).startswith("2d"):
        loss, affs, _ = orc.torch_embedding_loss(e, torch.from_numpy(g["target"]), torch.from_numpy(g["weight"]),
                                                 torch.from_numpy(g["mask"]), g["offsets"].tolist(), ema=ema, affs0_weight=a0)
    else:
        first = 1 if "norm1" in str(g["kind"]) else 3
        loss, affs = orc.torch_embedding_loss_3d(e, torch.from_numpy(g["target"]), torch.from_numpy(g["weight"]),
                                                 shifts_for(g), ema=ema, affs0_weight=a0, first=first)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 1e-6 * max(1.0, abs(float(g["loss"])))
    assert np.abs(affs.numpy() - g["affs"]).max() < 1e-6
    assert np.abs(e.grad.numpy() - g["grad"]).max() <= 1e-5 * np.abs(g["grad"]).max()


def test_offset_tables(orc, pkg):
    assert orc.multi_offset([1, 3, 5, 9, 27], 4) == [[-1, 0], [0, -1], [-3, 0], [0, -3], [-5, 0], [0, -5], [-9, 0], [0, -9], [-27, 0], [0, -27]]
    assert pkg.multi_offset([1, 3], 8) == [[-1, 0], [0, -1], [-1, -1], [-1, 1], [-3, 0], [0, -3], [-3, -3], [-3, 3]]
    assert pkg.multi_offset([1, 3, 5, 9, 27], 4) == orc.multi_offset([1, 3, 5, 9, 27], 4)
    g = load_golden("g2d_nb8_k12")
    assert pkg.multi_offset([1, 3, 9], 8) == g["offsets"].tolist()
    with pytest.raises(AssertionError):
        pkg.gen_offsets(1, neighbor=6)


def test_full_size_summary_fixture_against_oracle(orc, pkg):
    """full CVPPP-size case (2 x 16 x 544 x 544, K=10): oracle vs the reference's summary statistics"""
    import importlib
    import __graft_entry__ as ge
    synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
    g = load_golden("g2d_full544_summary")
    B, D, H, W = [int(v) for v in g["shape"]]
    offsets = g["offsets"].tolist()
    e, t, w, m = synth.synth_inputs_2d(B, D, H, W, offsets, int(g["seed"]))
    d = orc.desc_2d(e, offsets)
    affs, loss = orc.c_fwd(d, e, None, t, w, m)
    assert abs(loss[0] - float(g["loss"])) <= 3e-6 * float(g["loss"])
    np.testing.assert_allclose(loss[1:], g["all_loss"], rtol=5e-6)
    assert np.abs(affs.reshape(-1)[g["affs_idx"]] - g["affs_val"]).max() < 2e-6
    assert abs(affs.astype(np.float64).sum() - float(g["affs_sum"])) < 1e-6 * affs.size ** 0.5 * 10
    de, _ = orc.c_bwd(d, e, None, t, w, m)
    assert np.abs(de.reshape(-1)[g["grad_idx"]] - g["grad_val"]).max() <= 2e-5 * np.abs(g["grad_val"]).max()


@pytest.mark.parametrize("name", ["gtgt_2d_nb4", "gtgt_2d_nb8"])
def test_target_generation_restatement_matches_reference(name):
    """oracle np_gen_targets / np_weight_binary_ratio against what the reference's gen_affs_ours produced
    (tests/golden/make_golden.py::case_targets): bit-exact (integer / byte work)"""
    import oracle.pea_oracle as orc
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    lab = g["labels"][:, None]
    offs = [list(o) for o in g["offsets"]]
    for padding, tag in ((True, "pad"), (False, "nopad")):
        t, m = orc.np_gen_targets(lab, offs, padding=padding)
        assert np.array_equal(t[:, :, 0], g["target_" + tag])
        assert np.array_equal(m[:, :, 0], g["mask_" + tag])
        assert np.array_equal(orc.np_weight_binary_ratio(t[:, :, 0]), g["weight_" + tag])


@pytest.mark.parametrize("name", ["g3r_norm6", "g3r_norm6_ema", "g3r_norm6_ema_both"])
def test_replicate_border_restatement_matches_reference(name):
    """embedding_loss_norm6 / ema_embedding_loss_norm6 (replicate-padded shifts, one criterion over all channels): the C
    restatement (scatter-form backward) against the reference's loss, affinity map and autograd gradients"""
    import oracle.pea_oracle as orc
    g = load_golden(name)
    e, ema = g["e"], g.get("ema")
    d = orc.desc_3d_replicate(e, g["offsets"])
    affs, loss = orc.c_fwd(d, e, ema, g["target"], g["weight"], None)
    assert np.abs(affs - g["affs"]).max() < 2e-6
    assert abs(loss[0] - float(g["loss"])) <= 3e-6 * float(g["loss"])
    de, de_o = orc.c_bwd(d, e, ema, g["target"], g["weight"], None, want_other="grad_ema" in g)
    assert np.abs(de - g["grad"]).max() <= 2e-5 * np.abs(g["grad"]).max()
    if "grad_ema" in g:
        assert np.abs(de_o - g["grad_ema"]).max() <= 2e-5 * np.abs(g["grad_ema"]).max()


@pytest.mark.parametrize("name", ["ghead_2d_c32_d16", "ghead_2d_c64_d32", "ghead_3d_c28_d16"])
def test_head_restatement_matches_reference(name):
    """np_head_fwd / np_head_bwd against the outputs and autograd gradients of the reference's OutConv / conv3dBlock
    (tests/golden/make_golden.py::case_head).  Tolerance: f32 sums of 28-64 products (forward, dx) and of a few hundred
    to a few thousand (dW, db) in another order: 1e-5 of the largest magnitude."""
    import oracle.pea_oracle as orc
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    bias = g["bias"] if g["bias"].size else None
    e = orc.np_head_fwd(g["x"], g["weight"], bias)
    assert e.shape == g["e"].shape
    assert np.abs(e - g["e"]).max() <= 1e-5 * np.abs(g["e"]).max()
    dx, dW, db = orc.np_head_bwd(g["x"], g["weight"], g["upstream"])
    assert np.abs(dx - g["dx"]).max() <= 1e-5 * np.abs(g["dx"]).max()
    assert np.abs(dW - g["dW"]).max() <= 1e-5 * np.abs(g["dW"]).max()
    if bias is not None:
        assert np.abs(db - g["db"]).max() <= 1e-5 * np.abs(g["db"]).max()


def test_oracle_target_generation_matches_reference_seg_to_aff(orc):
    """np_gen_targets(both_foreground, no padding) against the reference's seg_to_aff run on the same segmentation
    (tests/golden/gseg2aff_3d.npz: the twelve norm5 channels with pad='' and the 3-edge graph with pad='replicate')"""
    g = load_golden("gseg2aff_3d")
    seg = g["seg"][None]
    offs12 = orc.norm_offsets([1, 1, 1, 2, 3, 3, 3, 9, 9, 4, 27, 27])
    t12, _ = orc.np_gen_targets(seg, offs12, padding=False, both_foreground=True)
    assert np.array_equal(t12[0], g["aff12_nopad"])
    t3, _ = orc.np_gen_targets(seg, offs12[:3], padding=False, both_foreground=True)
    t3 = t3[0].copy()
    fg = (seg[0] > 0).astype(np.float32)
    t3[0, 0], t3[1, :, 0], t3[2, :, :, 0] = fg[0], fg[:, 0], fg[:, :, 0]
    assert np.array_equal(t3, g["aff3_replicate"])


MARCH = golden_names("g3d_norm5_march") + golden_names("g3d_norm1_march")


def march_inputs(g):
    """the inputs of a *_march summary fixture: a closed-form function of the index (utils/synth.py), regenerated, not stored"""
    import importlib

    import __graft_entry__ as ge
    ge.load_package()
    synth = importlib.import_module(ge.PKG_NAME + ".utils.synth")
    B, D, Z, Y, X = [int(v) for v in g["shape"]]
    shifts = [1, 1, 1] if "norm1" in str(g["kind"]) else [1, 1, 1, 2, 3, 3, 3, 9, 9, 4, 27, 27]
    offs = [[-s if i % 3 == a else 0 for a in range(3)] for i, s in enumerate(shifts)]
    e, t, w = synth.synth_inputs_3d(B, D, Z, Y, X, offs, int(g["seed"]))
    return e, t, w, shifts


@pytest.mark.parametrize("name", MARCH)
def test_c_oracle_matches_reference_summary_at_march_size(orc, name):
    """the fixtures sized for the z-march kernels (too big to store whole: loss, sums and 4096 sampled values of the reference's
    affinity map and autograd gradient): the C oracle, which checks those kernels on the GPU, against the reference itself"""
    g = load_golden(name)
    e, t, w, shifts = march_inputs(g)
    first = 1 if len(shifts) == 3 else 3
    d = orc.desc_3d(e, shifts, orc.affs0_lambda_3d(len(shifts), float(g["affs0_weight"]), first))
    affs, loss = orc.c_fwd(d, e, None, t, w, None)
    grad, _ = orc.c_bwd(d, e, None, t, w, None)
    assert abs(loss[0] - float(g["loss"])) <= 3e-6 * max(1.0, abs(float(g["loss"])))
    assert np.abs(affs.reshape(-1)[g["affs_idx"]] - g["affs_val"]).max() < 2e-6
    assert abs((affs.astype(np.float64) ** 2).sum() / float(g["affs_sq"]) - 1) < 1e-5
    assert np.abs(grad.reshape(-1)[g["grad_idx"]] - g["grad_val"]).max() <= 2e-5 * np.abs(g["grad_val"]).max()
    assert abs((grad.astype(np.float64) ** 2).sum() / float(g["grad_sq"]) - 1) < 1e-4
