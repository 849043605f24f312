"""A/B of cdf_gemm_tuning settings on ONE pre-split 3x3 GEMM shape, same process, interleaved rounds (the A/B rule of tools/README.md):
   GA_SHAPE=Cin-Cout-HW  GA_B=32  GA_VARIANTS="halo=47;halo=175"  python tools/gemm_ab.py
prints ms per launch (min / median over rounds), TF-equivalent, and the max |difference| of every variant's output to the first's."""
import os, sys, statistics, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cold-diffusion-models_amd"))
from colddiff import _lib, convdesc as cd
L = _lib.get(); dev = torch.device("cuda:0")
S = lambda: torch.cuda.current_stream().cuda_stream
P = lambda t: 0 if t is None else t.data_ptr()
Cin, Cout, H = (int(v) for v in os.environ.get("GA_SHAPE", "128-64-128").split("-"))
B, k = int(os.environ.get("GA_B", "32")), 3
variants = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in v.split(",") if kv) for v in os.environ.get("GA_VARIANTS", "halo=47;halo=175").split(";")]
epi = os.environ.get("GA_EPI", "plain")            # plain | gelu (bias + GELU, pre-activation + planes out) | gelun (the same without pre: no-grad forward)
                                                   # | res (bias + residual) | mulg (x GELU'(pre) -> planes) | acc (y += .)
torch.manual_seed(0)
x = torch.randn(B, H, H, Cin, device=dev)
w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
ldk = (Cin + 31) // 32 * 32
hi = torch.zeros(k * k, Cout, ldk, dtype=torch.int16, device=dev); lo = torch.zeros_like(hi)
L.cdf_pack_weight_bf16(P(w), P(hi), P(lo), k * k, Cout, Cin, ldk, 1, Cin * k * k, k * k, S())
xh = torch.empty(x.shape, dtype=torch.int16, device=dev); xl = torch.empty_like(xh)
L.cdf_split_bf16(P(x), Cin, P(xh), P(xl), Cin, x.numel() // Cin, Cin, S())
zero = torch.zeros(64, device=dev)
p = cd.conv_fwd(H, H, k, k, 1, 1, 1, 1, 1)
bias = torch.randn(Cout, device=dev); res = torch.randn(B, H, H, Cout, device=dev)
y = torch.empty(B, H, H, Cout, device=dev); pre = torch.empty_like(y)
yh = torch.empty(B, H, H, Cout, dtype=torch.int16, device=dev); yl = torch.empty_like(yh)
tunes = [_lib.GemmTuning(L).set(**v) for v in variants]


def launch(t):
    if epi == "gelu":
        L.cdf_conv_gemm_bf16x(P(xh), P(xl), Cin, P(zero), P(hi), P(lo), ldk, 0, Cout, B, H, H, Cin, H, H, Cout, H, H, 1, 1, 1, p.desc, P(bias), 0, 0, 0, 0,
                              P(pre), Cout, 0, 0, 1, 0, 0, P(yh), P(yl), Cout, 0, 0, t.ptr, S())
    elif epi == "gelun":
        L.cdf_conv_gemm_bf16x(P(xh), P(xl), Cin, P(zero), P(hi), P(lo), ldk, 0, Cout, B, H, H, Cin, H, H, Cout, H, H, 1, 1, 1, p.desc, P(bias), 0, 0, 0, 0,
                              0, 0, 0, 0, 1, 0, 0, P(yh), P(yl), Cout, 0, 0, t.ptr, S())
    elif epi == "mulg":
        L.cdf_conv_gemm_bf16x(P(xh), P(xl), Cin, P(zero), P(hi), P(lo), ldk, 0, Cout, B, H, H, Cin, H, H, Cout, H, H, 1, 1, 1, p.desc, 0, 0, 0, 0, 0,
                              0, 0, P(res), Cout, 0, 1, 0, P(yh), P(yl), Cout, 0, 0, t.ptr, S())
    elif epi == "acc":
        L.cdf_conv_gemm_bf16x(P(xh), P(xl), Cin, P(zero), P(hi), P(lo), ldk, P(y), Cout, B, H, H, Cin, H, H, Cout, H, H, 1, 1, 1, p.desc, 0, 0, 0, 0, 0,
                              0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, t.ptr, S())
    elif epi == "res":
        L.cdf_conv_gemm_bf16x(P(xh), P(xl), Cin, P(zero), P(hi), P(lo), ldk, P(y), Cout, B, H, H, Cin, H, H, Cout, H, H, 1, 1, 1, p.desc, P(bias), 0, 0, P(res), Cout,
                              0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, t.ptr, S())
    else:
        L.cdf_conv_gemm_bf16x(P(xh), P(xl), Cin, P(zero), P(hi), P(lo), ldk, P(y), Cout, B, H, H, Cin, H, H, Cout, H, H, 1, 1, 1, p.desc, 0, 0, 0, 0, 0,
                              0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, t.ptr, S())


outs = []
for t in tunes:
    y.zero_(); pre.zero_()
    launch(t); torch.cuda.synchronize()
    outs.append((pre if epi == "gelu" else (yh.float() if epi in ("gelun", "mulg") else y)).clone())
    sums = [float(o.double().abs().sum()) for o in (y, pre, yh.float(), yl.float())]
ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2)[:2], w).permute(0, 2, 3, 1) if False else None
times = [[] for _ in tunes]
for rnd in range(int(os.environ.get("GA_ROUNDS", "5"))):
    for i, t in enumerate(tunes):
        for _ in range(2): launch(t)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): launch(t)
        e1.record(); torch.cuda.synchronize()
        times[i].append(e0.elapsed_time(e1) / 10)
fl = 2.0 * B * H * H * Cin * Cout * k * k
for v, ts, o in zip(variants, times, outs):
    print(f"{Cin}->{Cout} @{H} B={B} epi={epi} {str(v):40s} min {min(ts):.4f} ms  median {statistics.median(ts):.4f} ms  {fl / min(ts) / 1e9:6.1f} TF   max|diff vs first| {float((o - outs[0]).abs().max()):.2e}  |y|max {float(o.abs().max()):.2f}  sums {sums}", flush=True)
