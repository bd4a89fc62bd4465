"""CPU: the re-declared ResidualUNet2D_deep (model/unet2d_residual.py) against the reference's own module -- fixtures made by
tests/golden/make_golden.py `model` from scripts_cvppp/model/unet2d_residual.py:279-353: the state_dict layout (what checkpoints
are saved in, scripts_cvppp/main.py:453-456) loads with strict=True and the six outputs agree in training mode."""
import importlib

import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from conftest import load_golden


@pytest.mark.parametrize("name", ["gmodel_resunet2d", "gmodel_resunet2d_odd"])
def test_backbone_layout_and_outputs_match_reference(pkg, name):
    g = load_golden(name)
    mod = importlib.import_module(ge.PKG_NAME + ".model.unet2d_residual")
    net = mod.ResidualUNet2D_deep(in_channels=3, out_channels=2, nfeatures=g["nfeatures"].tolist(), emd=int(g["emd"]), hip_heads=False)
    keys = [str(k) for k in g["keys"]]
    assert list(net.state_dict().keys()) == keys  # same names, same order
    net.load_state_dict({k: torch.from_numpy(g["sd/" + k]) for k in keys}, strict=True)
    outs = net(torch.from_numpy(g["x"]))
    assert len(outs) == 6
    for i, o in enumerate(outs):
        ref = g["out%d" % i]
        assert tuple(o.shape) == ref.shape
        np.testing.assert_allclose(o.detach().numpy(), ref, rtol=2e-5, atol=2e-6)


def test_default_backbone_is_the_shipped_4p7m_parameter_net(pkg):
    mod = importlib.import_module(ge.PKG_NAME + ".model.unet2d_residual")
    net = mod.ResidualUNet2D_deep(hip_heads=False)
    n = sum(p.numel() for p in net.parameters())
    assert 4.6e6 < n < 4.9e6  # "79.01 GMac, 4.7M" (comment at scripts_cvppp/model/unet2d_residual.py:365)
    e16, e8, e4, e2, e1, mask = net(torch.zeros(1, 3, 64, 96))
    assert [tuple(t.shape[1:]) for t in (e16, e8, e4, e2, e1, mask)] == [(16, 4, 6), (16, 8, 12), (16, 16, 24), (16, 32, 48), (16, 64, 96), (2, 64, 96)]


def test_label_pyramid_follows_the_nearest_resize_rule(pkg):
    """label_pyramid against a restatement of cv2.resize(.., fx=1/2^j, INTER_NEAREST) as the provider calls it
    (scripts_cvppp/data/data_provider.py:199-208): dst extent cvRound(n * fx), src index floor(dst / fx); cv2 itself is not in this image"""
    import numpy as np
    import torch
    rng = np.random.default_rng(3)
    for (h, w) in ((544, 544), (530, 500), (72, 88)):
        lab = rng.integers(0, 9, size=(2, h, w)).astype(np.int32)
        pyr = pkg.label_pyramid(torch.from_numpy(lab))
        for j, got in enumerate(pyr, start=1):
            f = 2 ** j
            ny, nx = int(np.rint(h / f)), int(np.rint(w / f))          # np.rint: round half to even, as cvRound
            iy = np.minimum(np.floor(np.arange(ny) * f).astype(int), h - 1)
            ix = np.minimum(np.floor(np.arange(nx) * f).astype(int), w - 1)
            assert np.array_equal(got.numpy(), lab[:, iy][:, :, ix]), (h, w, j)


def test_convert_consistency_flip_matches_reference_golden(pkg):
    """harness/train_step.convert_consistency_flip against the reference's function run on all eight rule combinations
    (tests/golden/make_golden.py case_flip: scripts_cvppp/data/data_consistency.py:34-45); the result is detached"""
    import numpy as np
    import torch
    from conftest import load_golden
    g = load_golden("gflip_rules")
    x = torch.from_numpy(g["gt"]).requires_grad_(True)
    for rules in (torch.from_numpy(g["rules"]), g["rules"].tolist()):
        out = pkg.convert_consistency_flip(x, rules)
        assert not out.requires_grad and np.array_equal(out.numpy(), g["out"])
    assert np.array_equal(pkg.convert_consistency_flip(x, None).numpy(), g["gt"])
