"""depthwise 7x7 kernels alone (for rocprofv3 --pmc passes): DW_C, DW_H, DW_B, DW_WHAT=fwd|dgrad|wgrad|all"""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cold-diffusion-models_amd"))
from colddiff import ops
dev = torch.device("cuda:0")
C, H, B = int(os.environ.get("DW_C", "64")), int(os.environ.get("DW_H", "128")), int(os.environ.get("DW_B", "32"))
what = os.environ.get("DW_WHAT", "all")
x = torch.randn(B, H, H, C, device=dev); dy = torch.randn_like(x); y = torch.empty_like(x)
w = torch.nn.Parameter(torch.randn(C, 1, 7, 7, device=dev)); b = torch.nn.Parameter(torch.zeros(C, device=dev))
tb = torch.randn(B, C, device=dev)
wp = ops.packed(w, "dw")
for _ in range(int(os.environ.get("DW_ITERS", "6"))):
    if what in ("fwd", "all"): ops.dwconv7(x, wp, b.detach(), tb, y=y)
    if what in ("dgrad", "all"): ops.dwconv7(dy, wp, None, None, flip=1, y=y, res=x)
    if what in ("wgrad", "all"): ops.dwconv7_wgrad(x, dy, w, b, True)
torch.cuda.synchronize()
print("done")
