#!/usr/bin/env python3
"""Build a SECOND library for same-call A/Bs: the current objects with some sources taken from another git revision.

    python tools/ab_lib.py <rev> k_dwconv.hip [more.hip ...]   ->  tools/_ablate/ab/lib_prev.so

The result rides along to the GPU box (tools/_ablate/ is git-ignored, not gpurun-ignored); a tool then runs once with the product
library and once with COLDDIFF_LIB=tools/_ablate/ab/lib_prev.so (both must share the ABI version of include/colddiff.h)."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "cold-diffusion-models_amd", "csrc")
HIPCC = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")


def main():
    rev, names = sys.argv[1], sys.argv[2:]
    out_dir = os.path.join(REPO, "tools", "_ablate", "ab")
    os.makedirs(out_dir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result", "-I", CSRC, "-I", os.path.join(REPO, "include")]
    objs = []
    for f in sorted(x for x in os.listdir(CSRC) if x.endswith(".hip")):
        if f in names:
            src = os.path.join(out_dir, f)
            open(src, "w").write(subprocess.run(["git", "-C", REPO, "show", f"{rev}:cold-diffusion-models_amd/csrc/{f}"], capture_output=True, text=True, check=True).stdout)
            obj = os.path.join(out_dir, f[:-4] + ".o")
            subprocess.run([HIPCC] + flags + ["-c", src, "-o", obj], check=True)
            objs.append(obj)
        else:
            objs.append(os.path.join(CSRC, "_obj", f[:-4] + ".o"))
    out = os.path.join(out_dir, "lib_prev.so")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
    print("built", out)


if __name__ == "__main__":
    main()
