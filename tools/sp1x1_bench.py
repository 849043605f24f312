"""In-kernel-split 1x1 GEMM (cdf_conv_gemm_bf16: attention projections) and its weight gradient (cdf_conv_wgrad_bf16) at the deep
attention shapes, back-to-back launches between two events (no host gaps).  S1_SHAPES=Cin-Cout-HW,..."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cold-diffusion-models_amd"))
from colddiff import _lib, convdesc as cd
L = _lib.get(); dev = torch.device("cuda:0")
S = lambda: torch.cuda.current_stream().cuda_stream
P = lambda t: 0 if t is None else t.data_ptr()
B = int(os.environ.get("KB_B", "32"))
shapes = [tuple(int(v) for v in t.split("-")) for t in os.environ.get("S1_SHAPES", "128-512-16,512-384-16,128-256-32,256-384-32,128-256-64,64-256-128").split(",")]


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000


for Cin, Cout, H in shapes:
    x = torch.randn(B, H, H, Cin, device=dev); y = torch.empty(B, H, H, Cout, device=dev); res = torch.randn_like(y); bias = torch.randn(Cout, device=dev)
    ldk = (Cin + 31) // 32 * 32
    hi = torch.zeros(1, Cout, ldk, dtype=torch.int16, device=dev); lo = torch.zeros_like(hi)
    w = torch.randn(Cout, Cin, 1, 1, device=dev) * 0.05
    L.cdf_pack_weight_bf16(P(w), P(hi), P(lo), 1, Cout, Cin, ldk, 1, Cin, 1, S())
    p = cd.conv_fwd(H, H, 1, 1, 1, 0, 0, 0, 0)
    fl = 2.0 * B * H * H * Cin * Cout
    byts = 4.0 * B * H * H * (Cin + Cout)
    us = timeit(lambda: L.cdf_conv_gemm_bf16(P(x), Cin, P(hi), P(lo), ldk, P(y), Cout, B, H, H, Cin, H, H, Cout, H, H, 1, 1, 1, p.desc, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, S()))
    print(f"sp 1x1 fwd  {Cin:4d}->{Cout:4d} @{H:3d}: {us:7.1f} us  {fl / us / 1e6:6.1f} TF  {byts / us / 1e3:6.0f} GB/s (x + y)", flush=True)
    us = timeit(lambda: L.cdf_conv_gemm_bf16(P(x), Cin, P(hi), P(lo), ldk, P(y), Cout, B, H, H, Cin, H, H, Cout, H, H, 1, 1, 1, p.desc, P(bias), 0, 0, P(res), Cout, 0, 0, 0, 0, 0, 0, 0, 3, S()))
    print(f"sp 1x1 fwd+bias+res              : {us:7.1f} us", flush=True)
    # fp32-MFMA kernel on the same shape
    wk = torch.randn(1, Cin, Cout, device=dev) * 0.05
    us = timeit(lambda: L.cdf_conv_gemm(P(x), Cin, P(wk), Cout, P(y), Cout, B, H, H, Cin, H, H, Cout, H, H, 1, 1, 1, p.desc, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, S()))
    print(f"f32 1x1 fwd                      : {us:7.1f} us  {fl / us / 1e6:6.1f} TF", flush=True)
    wg = cd.conv_wgrad(H, H, 1, 1, 1, 0, 0, 0, 0); M = B * H * H
    gy = torch.randn(B, H, H, Cout, device=dev)
    for minpix in (512, 128, 64):
        tiles = ((Cin + 127) // 128) * ((Cout + 127) // 128)
        hi_ns = max(1, M // minpix)
        cost = [-(-tiles * ns // 512) / ns for ns in range(1, min(hi_ns, 256) + 1)]
        best = min(cost); ns = next(i for i, c in enumerate(cost, 1) if c <= best * 1.03)
        ws = torch.empty(ns, 1, Cin, Cout, device=dev)
        us = timeit(lambda: L.cdf_conv_wgrad_bf16(P(x), Cin, P(gy), Cout, P(ws), Cout, B, H, H, H, H, 1, H, H, 1, Cin, Cout, 1, wg.desc, ns, 0, S()))
        print(f"sp 1x1 wgrad min {minpix:3d} px/split (ns={ns:3d}): {us:7.1f} us  {fl / us / 1e6:6.1f} TF", flush=True)
