#!/usr/bin/env python3
"""One pre-split wgrad shape in a loop, for `rocprofv3 --pmc ...` (see tools/pmc_summary.py)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cold-diffusion-models_amd"))
import torch
from colddiff import ops, convdesc as cd
dev = torch.device("cuda:0")
Cin, Cout, HW, B = [int(v) for v in os.environ.get("PMC_SHAPE", "256,512,32,32").split(",")]
x = torch.randn(B, HW, HW, Cin, device=dev)
dy = torch.randn(B, HW, HW, Cout, device=dev)
w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05)
xs, ds = ops.split_bf16(x), ops.split_bf16(dy)
wp = cd.conv_wgrad(HW, HW, 3, 3, 1, 1, 1, 1, 1)
for _ in range(8):
    ops.wgrad_into(ops.grad_of(w), wp, x, Cin, dy, Cout, 1, 9, Cin * 9, xa_s=xs, xb_s=ds)
torch.cuda.synchronize()
