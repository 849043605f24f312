import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "cold-diffusion-models_amd"), os.path.join(REPO, "tests"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`) runs the kernels on a single-threaded host simulator: ~13 minutes in one process.  When pytest-xdist is
    there, spread it over worker processes (COLDDIFF_TEST_WORKERS=0 keeps one process; an explicit -n wins).  Never for the GPU suite: those
    tests share one device and the harness looks at the process that loads the HIP library."""
    if "PYTEST_XDIST_WORKER" in os.environ or os.environ.get("COLDDIFF_TEST_WORKERS", "") == "0":
        return None
    opt = config.option
    if "".join(getattr(opt, "markexpr", "").split()) != "notgpu" or not hasattr(opt, "numprocesses") or opt.numprocesses:
        return None
    if getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False):
        return None
    n = int(os.environ.get("COLDDIFF_TEST_WORKERS", "0") or 0) or max(1, min(6, (os.cpu_count() or 2) - 2))
    if n < 2:
        return None
    try:                                                     # the simulator library is built once, here, not by six racing workers
        from emu_util import _build_module
        _build_module().build_emu()
    except Exception:
        return None
    os.environ.setdefault("OMP_NUM_THREADS", "2")            # (inherited by the workers: torch's CPU ops would otherwise each take every core)
    os.environ.setdefault("MKL_NUM_THREADS", "2")
    opt.numprocesses = n                                     # (xdist's own hook, which runs after this one, turns it into worker specs)
    return None


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
