"""The 200-step sampler's UNet forward at the sample batch (BASELINE config 3: gen_sample, x0_step_down), a few reverse steps:
   python tools/sample_prof.py [--batch 16] [--steps 20]            one timing (ms per UNet forward, ms per image for T = 200)
   rocprofv3 --kernel-trace -d out -- python tools/sample_prof.py   + tools/prof_summary.py out  -> which kernels the sampler spends its time in
   python tools/sample_prof.py --sweep                              the tuning hooks of include/colddiff.h one by one, same process (A/B rule)
"""
import argparse
import contextlib
import io
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "cold-diffusion-models_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--precision", default=None)
    a = ap.parse_args()
    from colddiff import runtime
    from denoising_diffusion_pytorch import GaussianDiffusion, Unet
    if a.precision:
        runtime.set_precision(a.precision)
    dev = torch.device("cuda:0")
    torch.manual_seed(123457)
    with contextlib.redirect_stdout(io.StringIO()):
        net = Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3).to(dev)
    d = GaussianDiffusion(net, image_size=128, channels=3, timesteps=200, loss_type='l1', sampling_routine='x0_step_down').to(dev)
    L = runtime.lib()
    TUNE = runtime.tuning()                                   # the process's cdf_gemm_tuning argument (Python-side; the library has no setters)

    def run(batch):
        noise = torch.randn(batch, 3, 128, 128, device=dev)
        with torch.no_grad():
            d.gen_sample(batch_size=batch, img=noise, t=3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d.gen_sample(batch_size=batch, img=noise, t=a.steps)
            torch.cuda.synchronize()
        ms = 1000 * (time.perf_counter() - t0) / a.steps
        return ms

    if not a.sweep:
        ms = run(a.batch)
        print(f"batch {a.batch}: {ms:.3f} ms per reverse step = {ms * 200 / a.batch:.2f} ms per image at T = 200")
        return
    from colddiff import unet as U

    def tba(v):
        U._TIME_BIAS_ALL = v

    variants = [("default", lambda: None, lambda: None),
                ("small_n64=0 (128-wide N tiles on small grids)", lambda: TUNE.set(small_n64=0), lambda: TUNE.set(small_n64=1)),
                ("time-bias linears one by one", lambda: tba(False), lambda: tba(True)),
                ("splitk=0", lambda: TUNE.set(splitk=0), lambda: TUNE.set(splitk=1)),
                ("halo_bm=128", lambda: TUNE.set(halo_bm=128), lambda: TUNE.set(halo_bm=0)),
                ("max_bm=128", lambda: TUNE.set(max_bm=128), lambda: TUNE.set(max_bm=0)),
                ("deep=0", lambda: TUNE.set(deep=0), lambda: TUNE.set(deep=1)),
                ("tile 128x64", lambda: TUNE.set(tile_bm=128, tile_bn=64), lambda: TUNE.set(tile_bm=0, tile_bn=0)),
                ("halo=0 (generic gather kernel)", lambda: TUNE.set(halo=0), lambda: TUNE.set(halo=47)),
                ("halo min_tiles=200", lambda: TUNE.set(halo_min_tiles=200), lambda: TUNE.set(halo_min_tiles=1)),
                ("default again", lambda: None, lambda: None)]
    for batch in (a.batch, 64):
        for name, on, off in variants:
            on()
            ms = min(run(batch) for _ in range(2))
            off()
            print(f"batch {batch:3d}  {name:48s} {ms:8.3f} ms/step  {ms * 200 / batch:7.2f} ms/img", flush=True)


if __name__ == "__main__":
    main()
