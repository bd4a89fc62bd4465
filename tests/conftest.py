import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return ge.load_package()


@pytest.fixture(autouse=True)
def _pea_switches(monkeypatch):
    """The library reads its PEA_* switches once; tests flip them with monkeypatch.setenv / delenv.  Make those two calls tell
    the library (and the Python layer's memo of pea_cross_supported) to re-read, and restore + re-read at teardown."""
    lib_mod = sys.modules.get(ge.PKG_NAME + "._lib")
    real_set, real_del = monkeypatch.setenv, monkeypatch.delenv

    def reload(name):
        mod = sys.modules.get(ge.PKG_NAME + "._lib")
        if name.startswith("PEA_") and mod is not None:
            mod.reload_env()

    def setenv(name, value, *a, **k):
        real_set(name, value, *a, **k)
        reload(name)

    def delenv(name, *a, **k):
        real_del(name, *a, **k)
        reload(name)

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield
    monkeypatch.undo()
    reload("PEA_")


@pytest.fixture(scope="session")
def orc():
    o = ge.load_oracle()
    o.build()
    return o


def golden_names(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def lam_for(g, K):
    """per-offset loss weights the reference applies for this fixture kind"""
    kind = str(g["kind"])
    a0 = float(g["affs0_weight"]) if "affs0_weight" in g else 1.0
    if kind == "2d_self":
        return [1.0] * K
    if kind == "2d_ema":
        return [a0 if i < 2 else 1.0 for i in range(K)]
    first = 1 if "norm1" in kind else 3
    return [a0 if i < first else 1.0 for i in range(K)]


def shifts_for(g):
    kind = str(g["kind"])
    if "norm1" in kind:
        return [int(g["shift"])] * 3
    return [1, 1, 1, 2, 3, 3, 3, 9, 9, 4, 27, 27]
