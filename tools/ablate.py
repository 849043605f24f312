#!/usr/bin/env python3
"""Ablation timing of the pre-split bf16x3 conv GEMM kernel (tuning aid, not part of the product).

  python tools/ablate.py build        # here: libcolddiff variants with parts of the kernel removed -> tools/_ablate/
  python tools/ablate.py run          # on the GPU: time each variant on the step's dominant shapes

CDF_ABLATE bits: 1 no operand DMA in the K loop, 2 no MFMAs, 4 no epilogue (return before the stores), 8 no fragment LDS reads (generic kernel only).
Results are wrong by construction; only the time matters."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "cold-diffusion-models_amd", "csrc")
OUT = os.path.join(REPO, "tools", "_ablate")
VARIANTS = [0, 1, 2, 4, 3, 5, 6, 7]
SHAPES = [(64, 128, 128, 3, 32), (128, 64, 128, 3, 32), (128, 256, 64, 3, 32), (512, 1024, 16, 3, 32)]   # Cin, Cout, HW, k, B


def build():
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, CSRC)
    import build as B
    B.build_device()
    objs = [os.path.join(CSRC, "_obj", s[:-4] + ".o") for s in B.SOURCES if s != "k_conv_sp.hip"]
    for v in VARIANTS:
        o = os.path.join(OUT, "k_conv_sp_%d.o" % v)
        subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result",
                               "-I", CSRC, "-I", os.path.join(REPO, "include"), "-DCDF_ABLATE=%d" % v, "-c",
                               os.path.join(CSRC, "k_conv_sp.hip"), "-o", o])
        subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, "lib_%d.so" % v)] + objs + [o])
        os.remove(o)
        print("built variant", v)


def run_one():
    sys.path.insert(0, os.path.join(REPO, "cold-diffusion-models_amd"))
    import torch
    from colddiff import ops, functions as F_
    v = os.environ["CDF_VARIANT"]
    dev = torch.device("cuda:0")
    row = []
    for Cin, Cout, HW, k, B in SHAPES:
        x = torch.randn(B, HW, HW, Cin, device=dev)
        w = torch.nn.Parameter(torch.randn(Cout, Cin, k, k, device=dev) * 0.05)
        xs = ops.split_bf16(x)
        f = lambda: F_.conv_forward(x, Cin, w, None, xs=xs)
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 10)
    print("ablate %3s: " % v + "  ".join("%7.3f ms" % t for t in row), flush=True)


def run():
    print("shapes (Cin,Cout,HW,k,B):", SHAPES)
    for v in VARIANTS:
        env = dict(os.environ, COLDDIFF_LIB=os.path.join(OUT, "lib_%d.so" % v), CDF_VARIANT=str(v))
        subprocess.call([sys.executable, os.path.abspath(__file__), "one"], env=env)


if __name__ == "__main__":
    {"build": build, "run": run, "one": run_one}[sys.argv[1]]()
