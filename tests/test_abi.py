"""The C-ABI shared library: it loads, and exports exactly what include/colddiff.h declares."""
import ctypes
import os
import re
import subprocess

from colddiff import _lib


def test_header_parses_and_library_exports_every_symbol():
    protos = _lib.parse_header()
    assert len(protos) >= 45 and "cdf_conv_gemm" in protos and "cdf_blur_chain" in protos
    assert os.path.exists(_lib.LIB_PATH), "build first: python __graft_entry__.py"
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), name
    lib = _lib.Lib(_lib.LIB_PATH)
    assert lib.cdf_abi_version() == 1 and lib.cdf_is_device_build() == 1
    assert lib.cdf_last_error() is not None


def test_no_undeclared_exports():
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (cdf_\w+)", out))
    declared = set(_lib.parse_header())
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))


def test_bad_arguments_return_status_not_abort():
    lib = _lib.Lib(_lib.LIB_PATH)
    try:
        lib.cdf_blur_chain(0, 0, 0, 0, 0, 0, 1, 1, 8, 8, 3, 0, 0, 0, -1, 0, 0)
        assert False, "expected CdfError"
    except _lib.CdfError as e:
        assert "null pointer" in str(e)
    assert lib._dll.cdf_blur_chain(0, 0, 0, 0, 0, 0, 1, 1, 8, 8, 4, 0, 0, 0, -1, 0, 0) == -1      # CDF_E_INVALID
