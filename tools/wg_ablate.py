#!/usr/bin/env python3
"""Ablation timing of the row-of-taps weight-gradient kernel (tuning aid, not part of the product).

  python tools/wg_ablate.py build     # here: library variants with parts of conv_wgrad_row3_kernel removed -> tools/_ablate/wg_*.so
  python tools/wg_ablate.py run       # on the GPU: time each variant on the step's 3 x 3 weight-gradient shapes

Variant = (CDF_WG_ABLATE, CDF_ABLATE): WG bits 1 no global loads in the loop, 4 no slab stores, 8 no LDS stores in the loop; CDF_ABLATE 2 = no MFMAs.
Results are wrong by construction; only the time matters."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "cold-diffusion-models_amd", "csrc")
OUT = os.path.join(REPO, "tools", "_ablate")
VARIANTS = [(0, 0), (1, 0), (9, 0), (4, 0), (0, 2), (9, 2), (13, 0), (13, 2)]
SHAPES = [(64, 128, 128), (128, 128, 128), (128, 256, 64), (256, 256, 64), (256, 512, 32), (512, 1024, 16), (1024, 1024, 16)]   # Cin, Cout, HW


def build():
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, CSRC)
    import build as B
    B.build_device()
    objs = [os.path.join(CSRC, "_obj", s[:-4] + ".o") for s in B.SOURCES if s != "k_conv_sp.hip"]
    for wg, ab in VARIANTS:
        o = os.path.join(OUT, "k_conv_sp_wg.o")
        subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result",
                               "-I", CSRC, "-I", os.path.join(REPO, "include"), "-DCDF_WG_ABLATE=%d" % wg, "-DCDF_ABLATE=%d" % ab, "-c",
                               os.path.join(CSRC, "k_conv_sp.hip"), "-o", o])
        subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, "wg_%d_%d.so" % (wg, ab))] + objs + [o])
        os.remove(o)
        print("built variant", wg, ab)


def run_one():
    sys.path.insert(0, os.path.join(REPO, "cold-diffusion-models_amd"))
    import torch
    from colddiff import _lib, convdesc as cd
    L = _lib.get()
    dev = torch.device("cuda:0")
    S = lambda: torch.cuda.current_stream().cuda_stream
    P = lambda t: t.data_ptr()
    B = int(os.environ.get("WG_B", "32"))
    zero = torch.zeros(64, device=dev)
    row = []
    for Cin, Cout, H in SHAPES:
        x = torch.randn(B, H, H, Cin, device=dev)
        gy = torch.randn(B, H, H, Cout, device=dev)
        def split(t):
            h = torch.empty(t.shape, dtype=torch.int16, device=dev); l = torch.empty_like(h)
            L.cdf_split_bf16(P(t), t.shape[-1], P(h), P(l), t.shape[-1], t.numel() // t.shape[-1], t.shape[-1], S())
            return h, l
        xs, gs = split(x), split(gy)
        wg = cd.conv_wgrad(H, H, 3, 3, 1, 1, 1, 1, 1)
        M = B * H * H
        tiles = ((Cin + 127) // 128) * ((Cout + 127) // 128) * 3
        best, bc = 1, None
        for ns_ in range(1, min(M // 512, 256) + 1):
            c_ = -(-tiles * ns_ // 256) / ns_
            if bc is None or c_ < bc - 1e-9:
                best, bc = ns_, c_
        ns = int(os.environ.get("WG_NS", "0")) or best
        ws = torch.empty(ns, 9, Cin, Cout, device=dev)
        f = lambda: L.cdf_conv_wgrad_bf16x(P(xs[0]), P(xs[1]), Cin, P(gs[0]), P(gs[1]), Cout, P(zero), P(ws), Cout, B, H, H, H, H, 1, H, H, 1, Cin, Cout, 9,
                                           wg.desc, ns, 0, 0, S())
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        row.append((e0.elapsed_time(e1) / 10, ns))
    print("wg %-6s: " % os.environ["CDF_VARIANT"] + "  ".join("%6.3f ms/%-3d" % t for t in row), flush=True)


def run():
    print("shapes (Cin,Cout,HW):", SHAPES, " (ms / splits)")
    for wg, ab in VARIANTS:
        env = dict(os.environ, COLDDIFF_LIB=os.path.join(OUT, "wg_%d_%d.so" % (wg, ab)), CDF_VARIANT="%d,%d" % (wg, ab))
        subprocess.call([sys.executable, os.path.abspath(__file__), "one"], env=env)


if __name__ == "__main__":
    {"build": build, "run": run, "one": run_one}[sys.argv[1]]()
