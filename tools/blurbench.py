#!/usr/bin/env python3
"""Time the LDS-resident blur chains (dense k x k vs separable) on 128x128 planes: us per blur step."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cold-diffusion-models_amd"))
import torch
from colddiff import degrade as D
dev = torch.device("cuda:0")
B, C, H, k, T = 32, 3, 128, 15, 200
taps = torch.stack([torch.stack([D.gaussian_kernel2d(k, 1.0 + 0.03 * i)] * C) for i in range(T)]).to(dev)
t1 = D.separable_taps(taps)
x = torch.randn(B, C, H, H, device=dev)
for name, kw in (("dense", {}), ("separable", {"taps1d": t1})):
    for steps in (50, 200):
        f = lambda: D.blur_chain(x, taps, k, 1, step_lo=0, step_hi=steps - 1, **kw)
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"{name:10s} {steps:4d} steps: {dt*1e3:8.3f} ms  = {dt/steps*1e6:7.2f} us/step (96 planes in parallel)", flush=True)
