"""Helpers for the CPU test-suite: build + load the SIMT-simulator build of the kernel library.

TEST INFRASTRUCTURE ONLY — the simulator executes the same kernel sources on the host so that
indexing logic can be checked without a GPU; the product never loads it.
"""
import importlib.util
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "cold-diffusion-models_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)

_emu = None


def _build_module():
    spec = importlib.util.spec_from_file_location("cdf_build", os.path.join(PKG, "csrc", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def emu_lib():
    global _emu
    if _emu is None:
        from colddiff import _lib
        path = _build_module().build_emu()
        _emu = _lib.Lib(path)
        assert _emu.cdf_is_device_build() == 0
    return _emu


def P(t):
    """device/host pointer of a tensor (0 for None)."""
    return 0 if t is None else t.data_ptr()


def install_emu():
    """Route the colddiff Python layer to the simulator build (CPU tensors). Tests only."""
    from colddiff import runtime
    runtime._lib_override = emu_lib()
    return runtime._lib_override


def uninstall_emu():
    from colddiff import runtime
    runtime._lib_override = None
