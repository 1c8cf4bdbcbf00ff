"""CPU-only: the C-ABI shared library loads, exports every symbol include/loik_amd.h and
include/loik_amd_models.h declare, and fails loudly (no CPU fallback) when no GPU is present."""
import ctypes
import os
import re

import numpy as np
import pytest

import loik_amd
from loik_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in ("loik_amd.h", "loik_amd_models.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names.update(re.findall(r"\b(loikb_[a-z_0-9]+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol():
    L = loik_amd.lib()
    decl = declared_symbols()
    assert len(decl) >= 24
    for name in decl:
        assert hasattr(L, name), "libloik_amd.so does not export %s" % name
    assert decl == set(capi.EXPORTED_SYMBOLS), decl ^ set(capi.EXPORTED_SYMBOLS)
    assert L.loikb_version() == loik_amd.capi.ABI_VERSION == 602


def test_builtin_models():
    for name, nj in [("panda7", 8), ("panda9", 10), ("talos32", 33)]:
        m = loik_amd.builtin_model(name)
        assert m.njoints == nj and m.nv == nj - 1
        assert all(m.parents[i] < i for i in range(1, nj))
        for i in range(nj):
            R = m.placement[i, :9].reshape(3, 3)
            assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and abs(np.linalg.det(R) - 1) < 1e-14
    t = loik_amd.builtin_model("talos32")
    assert t.names[-1] == "head_2_joint" and t.getJointId("arm_left_7_joint") == 21
    with pytest.raises(KeyError):
        loik_amd.builtin_model("nope")


def test_status_strings_match_reference_messages():
    L = loik_amd.lib()
    assert b"equality constraint dimension is not 6" in L.loikb_status_string(-1)
    assert b"number of equality constraints doesn't match initialization" in L.loikb_status_string(-2)
    assert b"inequality constraint dimension has changed" in L.loikb_status_string(-3)
    assert b"constraint doesn't yet exist at link 'c_id'" in L.loikb_status_string(-4)


def test_no_silent_cpu_fallback():
    """without a GPU the product refuses to create a solver; with one it must succeed"""
    m = loik_amd.builtin_model("panda7")
    if loik_amd.device_count() == 0:
        with pytest.raises(loik_amd.LoikError) as e:
            loik_amd.BatchedLoik(m, 4, max_iter=10)
        assert e.value.code == -22
    else:
        loik_amd.BatchedLoik(m, 4, max_iter=10).close()


def test_create_argument_errors():
    m = loik_amd.builtin_model("panda7")
    with pytest.raises(loik_amd.LoikError) as e:
        loik_amd.BatchedLoik(m, 4, max_iter=10, eq_c_dim=3)
    assert e.value.code == -1  # thrown before any device work, like the reference ctor (hpp:41-44)
    bad = loik_amd.Model([0, 0, 3, 1], [0, 3, 3, 3], np.zeros((4, 3)), np.tile(np.r_[np.eye(3).ravel(), 0, 0, 0], (4, 1)))
    with pytest.raises(loik_amd.LoikError) as e:
        loik_amd.BatchedLoik(bad, 4, max_iter=10)
    assert e.value.code == -7


def test_product_never_imports_oracle():
    """the product path must not route through the CPU oracle (or any CPU fallback)"""
    pat = re.compile(r"(import\s+oracle|from\s+oracle|from\s+\.\.?oracle|loik_ref|libloik_ref)")
    for sub in ("loik_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".c", ".h", ".cpp")):
                    assert not pat.search(open(os.path.join(dirpath, f)).read()), os.path.join(dirpath, f)
