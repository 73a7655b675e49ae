"""The C-ABI library builds/loads and exports exactly what include/ovo_hip.h declares (no GPU needed)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = []
    for f in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if f.endswith(".h"):
            text = open(os.path.join(ROOT, "include", f)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names += re.findall(r"\b(ovo_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from ovo_amd import _lib, build
    build.build(verbose=False)
    lib = _lib.load()
    declared = _declared()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported"
    assert set(_lib.exported_symbols()) == set(declared), "ctypes binding and header disagree"
    assert lib.ovo_hip_abi_version() == _lib.ABI_VERSION


def test_argument_errors_are_reported_without_a_gpu():
    from ovo_amd import _lib
    lib = _lib.load()
    rc = lib.ovo_depth_filter(None, 0, 0, 7, 2.5, 0.05, None, None)
    assert rc == -1
    assert b"ovo_depth_filter" in lib.ovo_hip_last_error()
    with pytest.raises(_lib.OvoHipError):
        _lib.check(rc)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ovo_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(base, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports oracle"


def test_cpu_tensors_are_rejected_loudly():
    import torch
    from ovo_amd import _lib
    from ovo_amd.utils import geometry_utils as G
    with pytest.raises(_lib.OvoHipError):
        G.depth_filter(torch.zeros(8, 8))
