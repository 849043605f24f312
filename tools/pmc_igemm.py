#!/usr/bin/env python3
"""One pre-split igemm shape in a loop, for `rocprofv3 --pmc ...` (counters are summed per kernel by tools/pmc_summary)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cold-diffusion-models_amd"))
import torch
from colddiff import ops, functions as F_
dev = torch.device("cuda:0")
Cin, Cout, HW, B = [int(v) for v in os.environ.get("PMC_SHAPE", "512,1024,16,32").split(",")]
x = torch.randn(B, HW, HW, Cin, device=dev)
w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05)
xs = ops.split_bf16(x)
for _ in range(8):
    F_.conv_forward(x, Cin, w, None, xs=xs)
torch.cuda.synchronize()
