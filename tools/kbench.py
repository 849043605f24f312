"""Micro-benchmarks of the individual HIP kernels at the CelebA-128 Unet shapes (run on the GPU box).
Prints one line per kernel: time, achieved TFLOP/s or GB/s.  Writes gpurun_out/kbench.json."""
import json
import math
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cold-diffusion-models_amd"))
from colddiff import _lib, convdesc as cd  # noqa: E402

L = _lib.get()
dev = torch.device("cuda:0")
S = lambda: torch.cuda.current_stream().cuda_stream
P = lambda t: 0 if t is None else t.data_ptr()
r4 = lambda c: (c + 3) // 4 * 4
results = []


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


def rec(name, ms, flops=None, bytes_=None):
    r = {"kernel": name, "ms": round(ms, 4)}
    if flops:
        r["TFLOPs"] = round(flops / ms / 1e9, 2)
    if bytes_:
        r["GBs"] = round(bytes_ / ms / 1e6, 1)
    results.append(r)
    print(json.dumps(r), flush=True)


def conv_bench(B, Cin, Cout, H, k, tag):
    x = torch.randn(B, H, H, r4(Cin), device=dev)
    w = torch.randn(k * k, Cin, r4(Cout), device=dev) * 0.05
    y = torch.empty(B, H, H, r4(Cout), device=dev)
    p = cd.conv_fwd(H, H, k, k, 1, k // 2, k // 2, k // 2, k // 2)
    fn = lambda: L.cdf_conv_gemm(P(x), x.shape[-1], P(w), w.shape[-1], P(y), y.shape[-1], B, H, H, Cin, H, H, Cout, H, H, 1, 1, 1, p.desc,
                                 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, S())
    ms = timeit(fn)
    fl = 2.0 * B * H * H * Cin * Cout * k * k
    rec(f"conv{k}x{k}_fwd_{tag}_B{B}_{Cin}->{Cout}@{H}", ms, flops=fl, bytes_=4.0 * B * H * H * (Cin + Cout))
    # wgrad
    wg = cd.conv_wgrad(H, H, k, k, 1, k // 2, k // 2, k // 2, k // 2)
    M = B * H * H
    ns = L.cdf_wgrad_nsplit(M, Cin, Cout, k * k)
    ws = torch.empty(ns, k * k, Cin, r4(Cout), device=dev)
    fn2 = lambda: L.cdf_conv_wgrad(P(x), x.shape[-1], P(y), y.shape[-1], P(ws), r4(Cout), B, H, H, H, H, 1, H, H, 1, Cin, Cout, k * k,
                                   wg.desc, ns, 1, 0, 0, 0, 0, S())
    ms = timeit(fn2)
    rec(f"conv{k}x{k}_wgrad_{tag}_B{B}_{Cin}->{Cout}@{H}_ns{ns}", ms, flops=fl)


B = int(os.environ.get("KB_B", "16"))
conv_bench(B, 64, 128, 128, 3, "s0a")
conv_bench(B, 128, 64, 128, 3, "s0b")
conv_bench(B, 128, 256, 64, 3, "s1a")
conv_bench(B, 256, 128, 64, 3, "s1b")
conv_bench(B, 256, 512, 32, 3, "s2a")
conv_bench(B, 512, 1024, 16, 3, "s3a")
conv_bench(B, 1024, 512, 16, 3, "s3b")
conv_bench(B, 64, 384, 128, 1, "qkv0")
conv_bench(B, 3, 128, 128, 3, "first")
conv_bench(B, 64, 3, 128, 1, "final")

# depthwise 7x7
for (C, H) in [(64, 128), (128, 64), (256, 32), (512, 16)]:
    x = torch.randn(B, H, H, C, device=dev)
    w = torch.randn(49, C, device=dev)
    y = torch.empty_like(x)
    ms = timeit(lambda: L.cdf_dwconv7(P(x), C, P(w), C, 0, 0, 0, P(y), C, B, H, H, C, 0, 0, 0, 0, S()))
    rec(f"dwconv7_B{B}_C{C}@{H}", ms, bytes_=8.0 * x.numel())
    nch = L.cdf_dwconv7_wgrad_nchunk(H)
    ws = torch.empty(B * nch * 50 * C, device=dev)
    dw, dbias, dsb = torch.zeros(C, 1, 7, 7, device=dev), torch.zeros(C, device=dev), torch.zeros(B, C, device=dev)
    ms = timeit(lambda: L.cdf_dwconv7_wgrad(P(x), C, P(y), C, P(dw), P(dbias), P(dsb), C, P(ws), B, H, H, C, 0, S()))
    rec(f"dwconv7_wgrad_B{B}_C{C}@{H}", ms, bytes_=8.0 * x.numel())

# layernorm
for (C, H) in [(64, 128), (256, 32), (1024, 16)]:
    M = B * H * H
    x = torch.randn(M, C, device=dev)
    g, b_ = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    y, mo, ro = torch.empty_like(x), torch.empty(M, device=dev), torch.empty(M, device=dev)
    ms = timeit(lambda: L.cdf_layernorm_c_fwd(P(x), C, P(y), C, P(g), P(b_), P(mo), P(ro), M, C, 1e-5, 0, 0, 0, S()))
    rec(f"layernorm_fwd_M{M}_C{C}", ms, bytes_=8.0 * x.numel())
    nb = L.cdf_layernorm_blocks(M, C)
    part, dx, dg, db = torch.empty(nb * 2 * C, device=dev), torch.empty_like(x), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ms = timeit(lambda: L.cdf_layernorm_c_bwd(P(y), C, P(x), C, P(g), P(mo), P(ro), P(dx), C, 0, 0, P(dg), P(db), P(part), M, C, 0, 0, S()))
    rec(f"layernorm_bwd_M{M}_C{C}", ms, bytes_=12.0 * x.numel())

# linear attention
for H in (128, 32):
    n = H * H
    qkv = torch.randn(B, n, 384, device=dev)
    out, ctx, kmax, ksum = torch.empty(B, n, 128, device=dev), torch.empty(B, 4, 32, 32, device=dev), torch.empty(B, 128, device=dev), torch.empty(B, 128, device=dev)
    ws = torch.empty(L.cdf_linattn_ws_floats(B, n, 4), device=dev)
    ms = timeit(lambda: L.cdf_linattn_fwd(P(qkv), 384, P(out), 128, P(ctx), P(kmax), P(ksum), P(ws), B, n, 4, 32 ** -0.5, S()))
    rec(f"linattn_fwd_B{B}_n{n}", ms, bytes_=4.0 * (qkv.numel() * 4 / 3 + out.numel()))
    dqkv, dctx, rv = torch.empty_like(qkv), torch.empty_like(ctx), torch.empty(B, 128, device=dev)
    ms = timeit(lambda: L.cdf_linattn_bwd(P(qkv), 384, P(out), 128, P(ctx), P(kmax), P(ksum), P(dqkv), 384, P(dctx), P(rv), P(ws), B, n, 4, 32 ** -0.5, S()))
    rec(f"linattn_bwd_B{B}_n{n}", ms, bytes_=4.0 * (2 * qkv.numel() * 4 / 3 + 2 * out.numel()))

# blur chain: CelebA config (T=200, k=15, reflect), t uniform
Bb = 64
x = torch.rand(Bb, 3, 128, 128, device=dev) * 2 - 1
taps = torch.rand(200, 3, 15, 15, device=dev)
taps /= taps.sum((2, 3), keepdim=True)
t = torch.randint(0, 200, (Bb,), device=dev)
y = torch.empty_like(x)
ms = timeit(lambda: L.cdf_blur_chain(P(x), P(y), 0, 0, P(taps), P(t), Bb, 3, 128, 128, 15, 0, 0, 1, -1, 0, S()), iters=5)
steps = float((t + 1).sum().item())
rec(f"blur_chain_B{Bb}_T200_k15 (sum steps {int(steps)})", ms, flops=2.0 * 225 * 3 * 128 * 128 * steps, bytes_=steps * 3 * 128 * 128 * 8.0)
# adam
n = 56_615_708
p, g, m, v = (torch.randn(n, device=dev) for _ in range(4))
v.abs_()
ms = timeit(lambda: L.cdf_adam_step(P(p), P(g), P(m), P(v), n, 2e-5, 0.9, 0.999, 1e-8, 3, S()))
rec("adam_56.6M", ms, bytes_=28.0 * n)

os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(results, open(os.path.join(REPO, "gpurun_out", "kbench.json"), "w"), indent=1)
