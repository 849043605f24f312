"""Depthwise 7x7 forward / data-gradient / weight-gradient launch times (HIP events) at every (channels, size) the CelebA-128 Unet runs them at, B = DW_B (32).
A/B: run again with COLDDIFF_LIB=tools/_ablate/ab/lib_prev.so (tools/ab_lib.py)."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cold-diffusion-models_amd"))
from colddiff import ops
dev = torch.device("cuda:0")
B = int(os.environ.get("DW_B", "32")); iters = int(os.environ.get("DW_ITERS", "20"))
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
# (C, H, how many ConvNeXt blocks of the net run it)
for C, H, n in ((64, 128, 2), (64, 64, 2), (128, 64, 1), (256, 64, 1), (128, 32, 2), (256, 32, 1), (512, 32, 1), (256, 16, 2), (512, 16, 3), (1024, 16, 1)):
    x = torch.randn(B, H, H, C, device=dev); dy = torch.randn_like(x); y = torch.empty_like(x)
    w = torch.nn.Parameter(torch.randn(C, 1, 7, 7, device=dev)); b = torch.zeros(C, device=dev); tb = torch.randn(B, C, device=dev)
    wp = ops.packed(w, "dw")
    bp = torch.nn.Parameter(b.clone())
    for what, fn in (("fwd", lambda: ops.dwconv7(x, wp, b, tb, y=y)), ("dgrad", lambda: ops.dwconv7(dy, wp, None, None, flip=1, y=y, res=x)),
                     ("wgrad", lambda: ops.dwconv7_wgrad(x, dy, w, bp, True))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        us = 1000 * e0.elapsed_time(e1) / iters
        nbytes = x.numel() * 4 * (2 if what == "fwd" else 3)
        tot[what] += n * us
        print(f"{what:5s} C={C:4d} @{H:3d}: {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s", flush=True)
print("per forward pass of the net: fwd %.1f us, dgrad %.1f us, wgrad %.1f us" % (tot["fwd"], tot["dgrad"], tot["wgrad"]))
