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
    assert _lib.hip().stem_abi_version() == 5


def test_rans_library_exports_header_symbols():
    from spatiotemporalentropymodel_amd import _lib
    names = _declared("stem_rans.h")
    lib = ctypes.CDLL(_lib.RANS_SO)
    for n in names:
        assert hasattr(lib, n), f"libstem_rans.so does not export {n}"
    assert set(names) == set(_lib.declared_rans_symbols())


def test_dp_library_exports_header_symbols():
    """libstem_dp.so (native RCCL issue path of a data-parallel rank) loads without a GPU and exports what include/stem_dp.h declares"""
    from spatiotemporalentropymodel_amd import _lib
    names = _declared("stem_dp.h")
    assert len(names) == 11
    lib = ctypes.CDLL(_lib.DP_SO)
    for n in names:
        assert hasattr(lib, n), f"libstem_dp.so does not export {n}"
    assert set(names) == set(_lib.declared_dp_symbols())
    d = _lib.dp()
    assert d.stem_dp_submit(None, None, 0, None, 0) != 0 and b"stem_dp_submit" in d.stem_dp_last_error()      # argument checks need no device
    assert d.stem_dp_connect(None, None, 1, 0) != 0 and d.stem_dp_nranks(None) < 0 and d.stem_dp_abort(None, -1, None) != 0


def test_dp_prepare_fails_cleanly_without_a_device():
    """stem_dp_prepare is the LOCAL half of the construction: on a box without a GPU it returns an error with a message and leaves
    the handle NULL -- which is what lets the ranks agree (distributed._NativeIssuer) before any of them enters ncclCommInitRank"""
    import pytest
    import torch
    from spatiotemporalentropymodel_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    d = _lib.dp()
    h = ctypes.c_void_p(1234)
    assert d.stem_dp_prepare(ctypes.byref(h), 0) != 0
    assert not h.value and b"stem_dp_prepare" in d.stem_dp_last_error()


def test_ops_fail_loudly_without_gpu():
    """The product path must not fall back to CPU: CPU tensors are rejected."""
    import pytest
    import torch
    from spatiotemporalentropymodel_amd import functional as F
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    with pytest.raises(RuntimeError):
        F.to_nhwc(torch.zeros(1, 4, 2, 2))


def test_shipped_library_has_no_result_changing_switches():
    """The contract of libstem_hip.so is fp32-class products everywhere: ablated / instrumented kernel variants exist only in
    `make experiments` builds (-DSTEM_EXPERIMENTS -> libstem_hip_exper.so), no product-count switch exists at all.  The shipped binary reports
    so, does not contain the switches' names, and its plan selectors go through stem_tuning_set (validated, no environment)."""
    from spatiotemporalentropymodel_amd import _lib
    lib = _lib.hip()
    assert lib.stem_built_with_experiments() == 0
    blob = open(_lib.HIP_SO, "rb").read()
    for name in (b"STEM_BF16_PRODUCTS", b"STEM_GA_BF16_PRODUCTS", b"STEM_BF16_PRODUCTS_DYN", b"STEM_IGEMM_EXPER", b"STEM_FX3_EXPER",
                 b"STEM_FX3_SPLIT", b"STEM_WG3_SPLIT"):
        assert name not in blob, name
    assert lib.stem_tuning_get(b"fx3_split") == 0 and lib.stem_tuning_get(b"no_such") == -1
    assert lib.stem_tuning_set(b"fx3_split", 3) == 0 and lib.stem_tuning_get(b"fx3_split") == 3
    assert lib.stem_tuning_set(b"fx3_split", 0) == 0
    assert lib.stem_tuning_set(b"fx3_tile", 96) != 0 and b"fx3_tile" in lib.stem_last_error()
    assert lib.stem_tuning_set(b"products", 4) != 0
