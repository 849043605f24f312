#!/usr/bin/env python3
"""Host-side cost of one training step: the Python layer runs against a NULL kernel library (compute entry points return 0
at once, size queries go to the real simulator build), so what is timed is autograd bookkeeping, plan / cache look-ups, tensor
allocation and ctypes marshalling -- the part that has to stay below the GPU's step time.  CPU only; no numbers are computed.

  python tools/hostprof.py [--profile]
"""
import contextlib
import cProfile
import io
import os
import pstats
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "cold-diffusion-models_amd"), os.path.join(REPO, "tests"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402
from colddiff import _lib, runtime  # noqa: E402
from emu_util import emu_lib  # noqa: E402


class NullLib:
    calls = 0

    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        if _lib._QUERY.search(name) or name in ("protos", "cdf_gemm_tuning_default", "structs"):
            return getattr(self._real, name)

        def f(*a):
            NullLib.calls += 1
            return 0
        setattr(self, name, f)
        return f


def main():
    runtime._lib_override = NullLib(emu_lib())
    from denoising_diffusion_pytorch import GaussianDiffusion, Trainer, Unet
    with contextlib.redirect_stdout(io.StringIO()):
        net = Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3)
        d = GaussianDiffusion(net, image_size=128, channels=3, timesteps=200, loss_type='l1')
        tr = Trainer(d, None, image_size=128, train_batch_size=2, dataset='synthetic', results_folder='/tmp/hostprof_res')
    for _ in range(2):
        tr.train_step(); tr.step += 1
    NullLib.calls = 0
    n = 5
    t0 = time.perf_counter()
    if "--profile" in sys.argv:
        pr = cProfile.Profile(); pr.enable()
    for _ in range(n):
        tr.train_step(); tr.step += 1
    if "--profile" in sys.argv:
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime" if "--tottime" in sys.argv else "cumulative").print_stats(45)
    dt = (time.perf_counter() - t0) / n
    print(f"host side: {1000 * dt:.1f} ms per optimizer step, {NullLib.calls // n} library calls per step, {1e6 * dt / (NullLib.calls / n):.1f} us per call")


if __name__ == "__main__":
    main()
