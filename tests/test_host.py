"""CPU: host logic of the Python mirror (argument checking, descriptor building, loud failure without a GPU)."""
import ctypes
import inspect

import pytest
import torch


def test_reference_signatures(pkg):
    """same names, argument order and defaults as the reference's functions"""
    sig = lambda f: [(p.name, p.default) for p in inspect.signature(f).parameters.values()]
    E = inspect.Parameter.empty
    assert sig(pkg.embedding_loss) == [("embedding", E), ("target", E), ("weightmap", E), ("mask", E), ("criterion", E),
                                       ("offsets", E), ("affs0_weight", 1), ("mode", "ours")]
    assert sig(pkg.ema_embedding_loss) == [("embedding", E), ("ema_embedding", E), ("target", E), ("weightmap", E), ("mask", E),
                                           ("criterion", E), ("offsets", E), ("affs0_weight", 1), ("mode", "ours")]
    # the reference's three arguments first; `activation` (include/pea.h PEA_FLAG_*) is this package's trailing extension
    assert sig(pkg.embedding2affs) == [("embedding", E), ("offsets", E), ("mode", "ours"), ("activation", None)]
    for f in (pkg.embedding_loss_norm1, pkg.embedding_loss_norm5):
        assert sig(f) == [("embedding", E), ("target", E), ("weightmap", E), ("criterion", E), ("affs0_weight", 1),
                          ("shift", 1), ("fill", True)]
    for f in (pkg.ema_embedding_loss_norm1, pkg.ema_embedding_loss_norm5):
        assert sig(f)[:2] == [("embedding", E), ("ema_embedding", E)]
    assert sig(pkg.inf_embedding_loss_norm1) == [("embedding", E), ("shift", 1)]
    assert sig(pkg.inf_embedding_loss_norm5) == [("embedding", E)]


def test_no_cpu_fallback(pkg):
    e = torch.randn(1, 16, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.embedding2affs(e, [[-1, 0]])
    crit = pkg.WeightedMSE()
    t = torch.zeros(1, 1, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.embedding_loss(e, t, t, t.to(torch.uint8), crit, [[-1, 0]])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.inf_embedding_loss_norm5(torch.randn(1, 16, 6, 30, 30))


def test_missing_library_fails_loudly(pkg, monkeypatch, tmp_path):
    monkeypatch.setattr(pkg._lib, "_lib", None)
    monkeypatch.setattr(pkg._lib, "SO_PATH", str(tmp_path / "libpea_hip.so"))
    with pytest.raises(pkg.PeaLibraryError, match="no CPU fallback"):
        pkg._lib.lib()


def test_product_never_imports_oracle(pkg):
    import os
    import re
    root = os.path.dirname(pkg.__file__)
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle|pea_oracle", src, flags=re.M), f


def test_oracle_is_used_by_the_checkers_only():
    """outside tests/ the oracle may be touched by __graft_entry__.smoke() and bench.py's cpu_baseline leg only: the
    profiling scripts and the examples must not import it"""
    import os
    import re
    from conftest import ROOT
    for sub in ("profiles", "examples"):
        for dp, _, fs in os.walk(os.path.join(ROOT, sub)):
            for f in fs:
                if f.endswith((".py", ".sh", ".c", ".hip")):
                    src = open(os.path.join(dp, f)).read()
                    assert not re.search(r"load_oracle|pea_oracle|^\s*(from|import)\s+oracle", src, flags=re.M), os.path.join(dp, f)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"load_oracle", bench)]
    lo, hi = bench.index("def cpu_baseline"), bench.index("def main")
    assert uses and all(lo < u < hi for u in uses), "bench.py may use the oracle inside cpu_baseline only"


def test_weighted_mse_module_formula(pkg):
    crit = pkg.WeightedMSE()
    pred, tgt, w = torch.rand(2, 5, 7), torch.rand(2, 5, 7), torch.rand(2, 5, 7)
    # [B,H,W] -> normaliser B*W (the 2D quirk); [B,1,Z,Y,X] -> B*Z*Y*X
    assert torch.allclose(crit(pred, tgt, w), (w * (pred - tgt) ** 2).sum() / (2 * 7))
    p5 = torch.rand(2, 1, 3, 4, 5)
    assert torch.allclose(crit(p5, p5 * 0, None), (p5 ** 2).sum() / (2 * 3 * 4 * 5))
    assert crit.pea_fused


def test_descriptor_building(pkg):
    op = pkg.affinity_op
    spec = op.AffinitySpec(2, [[-1, 0], [0, -27], [-3, 3]], [2, 1, 1], pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
    e = torch.empty(2, 16, 40, 56)
    d = op.make_desc(spec, e, tstride=123)
    assert (d.B, d.D, list(d.dims), d.K, d.ndim) == (2, 16, [1, 40, 56], 3, 2)
    assert [list(d.offsets[i]) for i in range(3)] == [[0, -1, 0], [0, 0, -27], [0, -3, 3]]
    assert d.target_bstride == 123 and d.lam[0] == 2.0
    # torch.roll is modular: a circular offset beyond the extent folds back
    d2 = op.make_desc(op.AffinitySpec(2, [[-41, 60]], None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX), e)
    assert list(d2.offsets[0]) == [0, -1, 4]
    # cropped slices cannot exceed the extent
    with pytest.raises(ValueError):
        op.make_desc(op.AffinitySpec(3, [[-6, 0, 0]], None, pkg._lib.BORDER_CROP_ZERO, pkg._lib.NORM_CROPPED), torch.empty(1, 16, 6, 8, 8))
    with pytest.raises(ValueError):
        op.AffinitySpec(2, [], None, 0, 0)
    with pytest.raises(ValueError):
        op.make_desc(spec, torch.empty(2, 16, 4, 40, 56))
    d16 = op.make_desc(spec, torch.empty(1, 16, 40, 56, dtype=torch.float16))
    assert d16.dtype == pkg._lib.F16


def test_loss_list_is_lazy(pkg):
    ll = pkg.affinity_op.LossList(torch.tensor([0.5, 0.25, 0.125]))
    assert len(ll) == 3 and ll._vals is None
    assert ll[1] == 0.25 and list(ll) == [0.5, 0.25, 0.125]


def test_loss_list_behaves_like_the_reference_list(pkg):
    """all_loss is a list of K floats in the reference (loss_embedding_mse.py:41); the lazy stand-in must survive what callers do
    with such a list: len, indexing, iteration, == / + against lists, copy, pickle"""
    import copy
    import pickle
    import torch
    ll = pkg.affinity_op.LossList(torch.tensor([0.5, 1.5, 2.0]))
    assert len(ll) == 3 and ll[1] == 1.5 and list(ll) == [0.5, 1.5, 2.0] and sum(ll) == 4.0
    assert ll == [0.5, 1.5, 2.0] and ll + [7.0] == [0.5, 1.5, 2.0, 7.0] and [7.0] + ll == [7.0, 0.5, 1.5, 2.0]
    assert copy.copy(ll) == [0.5, 1.5, 2.0] and pickle.loads(pickle.dumps(ll)) == [0.5, 1.5, 2.0]
    assert repr(ll) == "[0.5, 1.5, 2.0]" and 1.5 in ll


def test_label_ids_are_range_checked_before_the_int32_cast(pkg):
    import torch
    ok = pkg.affinity_op._labels_int32(torch.tensor([[0, 5, 2 ** 31 - 1]], dtype=torch.int64))
    assert ok.dtype == torch.int32 and ok.tolist() == [[0, 5, 2 ** 31 - 1]]
    with pytest.raises(ValueError, match="fit int32"):
        pkg.affinity_op._labels_int32(torch.tensor([[0, 2 ** 31]], dtype=torch.int64))
    with pytest.raises(ValueError, match="outside marker"):  # -2^31: what the LDS-staged label kernels mark "outside the image" with
        pkg.affinity_op._labels_int32(torch.tensor([[0, -2 ** 31]], dtype=torch.int64))
    assert pkg.affinity_op._labels_int32(torch.tensor([[-2 ** 31 + 1]], dtype=torch.int64)).tolist() == [[-2 ** 31 + 1]]


def test_labels_in_path_refuses_offsets_as_long_as_the_image(pkg):
    import torch
    spec = pkg.AffinitySpec(2, [[-1, 0], [0, -40]], None, pkg._lib.BORDER_CIRCULAR, pkg._lib.NORM_BX)
    assert not pkg.affinity_op.labels_offsets_in_range(spec, torch.zeros(1, 16, 64, 34))
    assert pkg.affinity_op.labels_offsets_in_range(spec, torch.zeros(1, 16, 64, 48))


def test_stitcher_blend_weights_match_the_stored_reference_array(pkg):
    """harness/stitch.get_weight against an array computed OUTSIDE the product (tests/golden/make_golden.py restates
    Provider_valid.get_weight, scripts_ac3ac4/data/provider_valid.py:306-318, whose module needs cv2 / h5py)"""
    import importlib
    import numpy as np
    import __graft_entry__ as ge
    from conftest import load_golden
    g = load_golden("gstitch_weight_18x160x160")
    st = importlib.import_module(ge.PKG_NAME + ".harness.stitch")
    w = st.get_weight(tuple(int(v) for v in g["out_size"]))
    assert w.dtype == np.float32 and np.array_equal(w, g["weight"])


def test_activation_flags_and_spec(pkg):
    L, op = pkg._lib, pkg.affinity_op
    assert op.activation_flags(None) == 0 and op.activation_flags("relu") == L.FLAG_RELU_AFFS
    assert op.activation_flags("mutex") == L.FLAG_RELU_AFFS | L.FLAG_ONE_MINUS
    assert op.activation_flags("half_clamp") == L.FLAG_HALF_SHIFT | L.FLAG_CLAMP01
    with pytest.raises(ValueError):
        op.activation_flags("sigmoid")
    spec = op.AffinitySpec(2, [[-1, 0]], None, L.BORDER_CIRCULAR, L.NORM_BX, act=op.activation_flags("mutex"))
    assert spec.relu and op.make_desc(spec, torch.empty(1, 16, 8, 8)).flags & (L.FLAG_RELU_AFFS | L.FLAG_ONE_MINUS) == 3
