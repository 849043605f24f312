"""The HBM-bound kernels of the ConvNeXt block at the CelebA-128 shapes, through the package's own wrappers (colddiff.ops), B = 32:
depthwise 7x7 forward / data gradient / weight gradient, channel LayerNorm forward / backward, the operand split.
One line per (kernel, shape): ms, effective GB/s on the algorithmic bytes (tensor read + tensor written)."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cold-diffusion-models_amd"))
from colddiff import ops, runtime as rt
dev = torch.device("cuda:0")
B = int(os.environ.get("KB_B", "32"))


def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def line(name, ms, nbytes):
    print(f"{name:44s} {ms * 1000:8.1f} us  {nbytes / ms / 1e6:8.0f} GB/s", flush=True)


shapes = [(64, 128), (128, 128), (128, 64), (256, 64), (256, 32), (512, 32), (512, 16), (1024, 16)]
if os.environ.get("HB_SHAPES"):
    shapes = [tuple(int(v) for v in t.split("-")) for t in os.environ["HB_SHAPES"].split(",")]
only_dw = os.environ.get("HB_ONLY_DW", "0") == "1"
for (C, H) in shapes:
    x = torch.randn(B, H, H, C, device=dev); dy = torch.randn_like(x)
    w = torch.nn.Parameter(torch.randn(C, 1, 7, 7, device=dev)); b = torch.nn.Parameter(torch.zeros(C, device=dev))
    tb = torch.randn(B, C, device=dev)
    wp = ops.packed(w, "dw")
    n = x.numel()
    y = torch.empty_like(x)
    line(f"dwconv7 fwd  C={C} @{H}", timeit(lambda: ops.dwconv7(x, wp, b.detach(), tb, y=y)), 8.0 * n)
    line(f"dwconv7 dgrad C={C} @{H} (+res)", timeit(lambda: ops.dwconv7(dy, wp, None, None, flip=1, y=y, res=x)), 12.0 * n)
    line(f"dwconv7 wgrad C={C} @{H}", timeit(lambda: ops.dwconv7_wgrad(x, dy, w, b, True)), 8.0 * n)
    if only_dw:
        continue
    g, bb = torch.nn.Parameter(torch.ones(1, C, 1, 1, device=dev)), torch.nn.Parameter(torch.zeros(1, C, 1, 1, device=dev))
    hn, mean, rstd, hs = ops.layernorm_fwd(x, g, bb, 1e-5, True, split_out=True, planes_only=True)
    line(f"layernorm fwd (planes only) C={C} @{H}", timeit(lambda: ops.layernorm_fwd(x, g, bb, 1e-5, True, split_out=True, planes_only=True)), 8.0 * n)
    line(f"layernorm bwd C={C} @{H}", timeit(lambda: ops.layernorm_bwd(dy, x, g, bb, mean, rstd)), 12.0 * n)
    line(f"split_bf16 C={C} @{H}", timeit(lambda: ops.split_bf16(x)), 8.0 * n)
