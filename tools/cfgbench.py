#!/usr/bin/env python3
"""Train-step throughput of the other BASELINE.json configurations on one MI355X (synthetic data, random-init
weights).  bench.py stays the judged line (config 3); this records the remaining configs for DESIGN.md.

  python tools/cfgbench.py [cfg ...]      cfg in {1, 2, 4, 5r, 5f}; default: all
"""
import contextlib
import io
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "cold-diffusion-models_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402


def build(cfg, device):
    """(diffusion, image_size, batch, description) as the reference scripts configure them (README.md:72-74 and the
    *_train.py scripts)."""
    with contextlib.redirect_stdout(io.StringIO()):
        if cfg == "1":      # MNIST deblurring: mnist_train.py (Unet dim 64, channels 1, 32x32 padded images)
            from deblurring_diffusion_pytorch import GaussianDiffusion, Unet
            net = Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=1).to(device)
            d = GaussianDiffusion(net, image_size=32, device_of_kernel='cuda', channels=1, timesteps=20, loss_type='l1',
                                  kernel_std=7.0, kernel_size=11, blur_routine='Constant', train_routine='Final',
                                  sampling_routine='x0_step_down').to(device)
            return d, 32, 32, "MNIST 32x32 deblurring T=20 k=11 std=7 Constant, Unet(64,(1,2,4,8),ch=1), batch 32"
        if cfg == "2":      # CIFAR-10 deblurring: cifar10_train.py (Model ch 128, (1,2,2,2), 2 res blocks, attn at 16)
            from deblurring_diffusion_pytorch import GaussianDiffusion, Model
            net = Model(resolution=32, in_channels=3, out_ch=3, ch=128, ch_mult=(1, 2, 2, 2), num_res_blocks=2,
                        attn_resolutions=(16,), dropout=0.1).to(device)
            d = GaussianDiffusion(net, image_size=32, device_of_kernel='cuda', channels=3, timesteps=50, loss_type='l1',
                                  kernel_std=0.1, kernel_size=11, blur_routine='Special_6_routine', train_routine='Final',
                                  sampling_routine='x0_step_down').to(device)
            return d, 32, 128, "CIFAR-10 32x32 deblurring T=50 Special_6_routine, Model(ch=128,(1,2,2,2)), batch 128"
        if cfg == "4":      # CelebA-128 deblurring: celebA_128.py
            from deblurring_diffusion_pytorch import GaussianDiffusion, Unet
            net = Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3).to(device)
            d = GaussianDiffusion(net, image_size=128, device_of_kernel='cuda', channels=3, timesteps=200, loss_type='l1',
                                  kernel_std=0.01, kernel_size=15, blur_routine='Exponential_reflect', train_routine='Final',
                                  sampling_routine='x0_step_down').to(device)
            return d, 128, 32, "CelebA 128x128 deblurring T=200 k=15 Exponential_reflect, Unet(64,(1,2,4,8)), batch 32"
        if cfg == "5r":     # AFHQ resolution diffusion
            from resolution_diffusion_pytorch import GaussianDiffusion, Unet
            net = Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3).to(device)
            d = GaussianDiffusion(net, image_size=128, device_of_kernel='cuda', channels=3, timesteps=4, loss_type='l1',
                                  resolution_routine='Incremental_factor_2', train_routine='Final',
                                  sampling_routine='x0_step_down').to(device)
            return d, 128, 32, "AFHQ 128x128 resolution T=4 Incremental_factor_2, Unet(64,(1,2,4,8)), batch 32"
        if cfg == "5f":     # AFHQ defading
            from defading_diffusion_pytorch import GaussianDiffusion, Unet
            net = Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3).to(device)
            d = GaussianDiffusion(net, image_size=128, device_of_kernel='cuda', channels=3, timesteps=100, loss_type='l1',
                                  kernel_std=0.2, initial_mask=1, fade_routine='Incremental', sampling_routine='x0_step_down').to(device)
            return d, 128, 32, "AFHQ 128x128 defading T=100 Incremental kernel_std=0.2 initial_mask=1 (README.md:127,133), Unet(64,(1,2,4,8)), batch 32"
    raise SystemExit("unknown cfg " + cfg)


def main():
    from colddiff.trainer import Trainer
    cfgs = sys.argv[1:] or ["1", "2", "4", "5r", "5f"]
    device = torch.device("cuda:0")
    for cfg in cfgs:
        torch.manual_seed(123457)
        diffusion, size, batch, desc = build(cfg, device)
        with contextlib.redirect_stdout(io.StringIO()):
            tr = Trainer(diffusion, None, image_size=size, train_batch_size=batch, train_lr=2e-5, train_num_steps=10 ** 9,
                         gradient_accumulate_every=2, ema_decay=0.995, fp16=False, dataset='synthetic',
                         results_folder=os.path.join(REPO, "gpurun_out", "cfgbench_results"))
        tr.quiet = True
        loss = None
        for _ in range(3):
            loss = tr.train_step()
            tr.step += 1
        torch.cuda.synchronize()
        steps = 10
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = tr.train_step()
            tr.step += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"cfg": cfg, "workload": desc, "ms_per_step": round(1000 * dt / steps, 2),
                          "img_per_s": round(steps * batch * 2 / dt, 1), "last_loss": round(float(loss), 5)}),
              flush=True)
        del tr, diffusion
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
