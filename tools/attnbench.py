"""Time Residual(PreNorm(LinearAttention)) forward + backward at the CelebA-128 levels (B = KB_B, default 32):
   python tools/attnbench.py            # per-level fwd / bwd ms
   rocprofv3 --kernel-trace -d out -- python tools/attnbench.py   # per-kernel table through tools/prof_summary.py
KB_LEVELS="64-128,128-64" picks (dim-size) pairs."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cold-diffusion-models_amd"))
from colddiff import unet as U
dev = torch.device("cuda:0")
B = int(os.environ.get("KB_B", "32")); iters = int(os.environ.get("KB_ITERS", "5"))
levels = [(64, 128), (128, 64), (256, 32), (512, 16)]
if os.environ.get("KB_LEVELS"):
    levels = [tuple(int(v) for v in t.split("-")) for t in os.environ["KB_LEVELS"].split(",")]
for dim, H in levels:
    torch.manual_seed(0)
    blk = U.Residual(U.PreNorm(dim, U.LinearAttention(dim))).to(dev)
    x = torch.randn(B, H, H, dim, device=dev, requires_grad=True)
    dy = torch.randn(B, H, H, dim, device=dev)
    def fwd():
        return blk(x)
    def both():
        y = blk(x); y.backward(dy)
    for fn, name in ((fwd, "fwd"), (both, "fwd+bwd")):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"dim {dim:4d} @{H:3d} {name:8s}: {e0.elapsed_time(e1) / iters:8.3f} ms", flush=True)
