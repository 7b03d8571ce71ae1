"""The C-ABI shared library loads and exports every symbol include/hypmi.h declares (no compute
calls: those need a GPU and live in the gpu-marked tests)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from hyperion_amd import _lib


def header_symbols():
    text = open(os.path.join(ROOT, "include", "hypmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hypmi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build libhypmi.so first (make -C hyperion_amd/csrc)"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 20
    for name in syms:
        assert hasattr(lib, name), "%s declared in hypmi.h but not exported" % name


def test_bindings_cover_the_header():
    assert sorted(_lib.SIGNATURES) == header_symbols()


def test_error_string_and_version():
    lib = _lib.load()
    assert lib.hypmi_version() >= 1
    assert lib.hypmi_packed_width(150) == 152
    assert lib.hypmi_packed_width(20) == 32
    assert lib.hypmi_packed_width(510) == 512
    # argument validation happens before any device work
    rc = lib.hypmi_score(None, 1, None, 1, 152, None, 1, None)
    assert rc == -1
    assert b"null" in lib.hypmi_last_error()
    with pytest.raises(_lib.HypmiError):
        _lib.check(rc)


def test_no_cpu_fallback():
    import numpy as np
    import torch

    from hyperion_amd.pdfs.plda import SPLDA

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    m = SPLDA(mu=np.zeros(4), V=np.ones((2, 4)), W=np.eye(4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.llr_1vs1(np.zeros((2, 4)), np.zeros((3, 4)))
