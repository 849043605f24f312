"""Time cdf_linattn_kvctx alone at the CelebA-128 attention levels (COLDDIFF_LIB selects a probe build)."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cold-diffusion-models_amd"))
from colddiff import ops, unet as U
dev = torch.device("cuda:0")
B = int(os.environ.get("KB_B", "32"))
for dim, H in ((64, 128), (128, 64)):
    att = U.LinearAttention(dim).to(dev)
    xn = torch.randn(B, H, H, dim, device=dev)
    f = lambda: ops.linattn_kvctx(xn, dim, att.to_qkv.weight, 4, att.scale)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{os.environ.get('COLDDIFF_LIB', 'product')[-12:]:>12s} dim {dim:3d} @{H:3d}: {e0.elapsed_time(e1) / 10 * 1000:7.1f} us", flush=True)
