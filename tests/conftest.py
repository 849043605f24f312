import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "cold-diffusion-models_amd"), os.path.join(REPO, "tests"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a machine without a GPU skips the gpu-marked items instead of failing them (on the GPU box nothing is
    skipped: a missing device there must fail loudly)."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this machine (gpu-marked tests run on the MI355X box with -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Backend:
    """Where a kernel-level test runs.

    'hip' : the product library libcolddiff_hip.so on cuda:0 (tests marked gpu)
    'emu' : the same kernel sources compiled against the host SIMT simulator (tests/emu) on CPU
            tensors — CPU-only test infrastructure for the kernels' indexing logic.
    """

    def __init__(self, kind):
        import torch
        self.kind = kind
        self._keep = []
        if kind == "hip":
            from colddiff import _lib
            self.L = _lib.get()
            assert self.L.cdf_is_device_build() == 1
            self.device = torch.device("cuda:0")
        else:
            from emu_util import emu_lib
            self.L = emu_lib()
            self.device = torch.device("cpu")
        from colddiff import _lib as _l
        self.tune = _l.GemmTuning(self.L)          # the explicit tuning argument of the pre-split GEMM entry points (defaults; tests set fields)

    def to(self, t):
        """Device copy of t; kept alive until the end of the test (tests pass raw pointers inline)."""
        out = t.detach().to(self.device).contiguous().clone()
        self._keep.append(out)
        return out

    def empty(self, *shape):
        import torch
        return torch.empty(*shape, device=self.device, dtype=torch.float32)

    def zeros(self, *shape):
        import torch
        return torch.zeros(*shape, device=self.device, dtype=torch.float32)

    def stream(self):
        import torch
        return torch.cuda.current_stream().cuda_stream if self.kind == "hip" else 0


@pytest.fixture(scope="session", params=[pytest.param("emu"), pytest.param("hip", marks=pytest.mark.gpu)])
def be(request):
    return Backend(request.param)


@pytest.fixture(autouse=True)
def _release_kept_tensors(request):
    yield
    if "be" in request.fixturenames:
        import torch
        b = request.getfixturevalue("be")
        if b.kind == "hip":
            torch.cuda.synchronize()
        b._keep.clear()
