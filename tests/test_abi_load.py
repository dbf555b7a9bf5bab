"""CPU: the C-ABI libraries load and export every symbol include/*.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import REPO


def _declared(header):
    src = open(os.path.join(REPO, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(stem_[a-z0-9_]+)\s*\(", src)))


def test_hip_library_exports_header_symbols():
    from spatiotemporalentropymodel_amd import _lib
    names = _declared("stem_hip.h")
    assert len(names) > 30
    lib = ctypes.CDLL(_lib.HIP_SO)
    for n in names:
        assert hasattr(lib, n), f"libstem_hip.so does not export {n}"
    assert set(names) == set(_lib.declared_hip_symbols()), set(names) ^ set(_lib.declared_hip_symbols())
    assert _lib.hip().stem_abi_version() == 2


def test_rans_library_exports_header_symbols():
    from spatiotemporalentropymodel_amd import _lib
    names = _declared("stem_rans.h")
    lib = ctypes.CDLL(_lib.RANS_SO)
    for n in names:
        assert hasattr(lib, n), f"libstem_rans.so does not export {n}"
    assert set(names) == set(_lib.declared_rans_symbols())


def test_ops_fail_loudly_without_gpu():
    """The product path must not fall back to CPU: CPU tensors are rejected."""
    import pytest
    import torch
    from spatiotemporalentropymodel_amd import functional as F
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    with pytest.raises(RuntimeError):
        F.to_nhwc(torch.zeros(1, 4, 2, 2))
