"""Split-K factor of the pre-split weight gradient: time of cdf_conv_wgrad_bf16x + cdf_unpack_reduce per ns (tuning aid)."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cold-diffusion-models_amd"))
from colddiff import _lib, convdesc as cd, ops
L = _lib.get(); dev = torch.device("cuda:0")
S = lambda: torch.cuda.current_stream().cuda_stream
P = lambda t: 0 if t is None else t.data_ptr()
r4 = lambda c: (c + 3) // 4 * 4
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
zero = torch.zeros(64, device=dev)
def split(t):
    C = t.shape[-1]; hi = torch.empty(t.shape, dtype=torch.int16, device=dev); lo = torch.empty_like(hi)
    L.cdf_split_bf16(P(t), C, P(hi), P(lo), C, t.numel()//C, C, S()); return hi, lo
B, k = 32, 3
for (Cin, Cout, H) in [(512, 1024, 16), (1024, 512, 16), (256, 512, 32), (512, 256, 32), (128, 256, 64), (256, 128, 64), (64, 128, 128), (128, 64, 128)]:
    x = torch.randn(B, H, H, Cin, device=dev); gy = torch.randn(B, H, H, Cout, device=dev)
    xs, gs = split(x), split(gy)
    wg = cd.conv_wgrad(H, H, k, k, 1, 1, 1, 1, 1); M = B * H * H
    dw = torch.zeros(Cout, Cin, k, k, device=dev)
    tiles = (1 if Cin <= 64 else (Cin + 127) // 128) * (1 if Cout <= 64 else (Cout + 127) // 128) * 3
    auto = ops.best_nsplit(tiles, 256, M // 512)
    res = []
    for ns in sorted(set([1, 2, 3, 4, 6, 8, 12, 16, 32, 64, auto])):
        if ns > M // 512: continue
        ws = torch.empty(ns, k * k, Cin, r4(Cout), device=dev)
        f1 = lambda: L.cdf_conv_wgrad_bf16x(P(xs[0]), P(xs[1]), Cin, P(gs[0]), P(gs[1]), Cout, P(zero), P(ws), r4(Cout), B, H, H, H, H, 1, H, H, 1, Cin, Cout, k * k, wg.desc, ns, 0, 0, S())
        f2 = lambda: L.cdf_unpack_reduce(P(ws), P(dw), ns, k * k, Cin, Cout, r4(Cout), 1, k * k, Cin * k * k, 1, 1, S())
        t1, t2 = timeit(f1), timeit(f2)
        res.append((ns, t1, t2))
    print(f"{Cin}->{Cout}@{H} (auto ns={auto}): " + "  ".join(f"ns{ns}{'*' if ns == auto else ''}: {1000*t1:.0f}+{1000*t2:.0f}={1000*(t1+t2):.0f}us" for ns, t1, t2 in res), flush=True)
