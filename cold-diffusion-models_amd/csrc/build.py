#!/usr/bin/env python3
"""Build the colddiff HIP library.

  python build.py            -> libcolddiff_hip.so   (hipcc --offload-arch=gfx950; the product)
  python build.py --emu      -> tests/emu/_build/libcolddiff_emu.so (host clang++ against the
                                fiber SIMT simulator; CPU test infrastructure only)

Objects are cached per source by mtime so incremental rebuilds take seconds.
"""
import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
HOSTCXX = os.path.join(ROCM, "lib", "llvm", "bin", "clang++")

SOURCES = sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))
HEADERS = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")] + [os.path.join(REPO, "include", "colddiff.h")]


def _newer(src_list, out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in src_list)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError("build failed: " + cmd[-1])
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


def build_device(verbose=False):
    objdir = os.path.join(HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    out = os.path.join(HERE, "libcolddiff_hip.so")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result",
             "-I", HERE, "-I", os.path.join(REPO, "include")]
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(HERE, s)
        obj = os.path.join(objdir, s[:-4] + ".o")
        objs.append(obj)
        if _newer([src] + HEADERS + [os.path.abspath(__file__)], obj):
            jobs.append([HIPCC] + flags + ["-c", src, "-o", obj])
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(_run, jobs))
    if jobs or not os.path.exists(out):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    if verbose:
        print("built", out)
    return out


def build_emu(verbose=False):
    emudir = os.path.join(REPO, "tests", "emu")
    objdir = os.path.join(emudir, "_build")
    os.makedirs(objdir, exist_ok=True)
    out = os.path.join(objdir, "libcolddiff_emu.so")
    flags = ["-x", "c++", "-DCDF_EMU", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value",
             "-Wno-unknown-pragmas", "-Wno-pass-failed", "-I", HERE, "-I", emudir, "-I", os.path.join(REPO, "include")]
    emu_hdr = [os.path.join(emudir, "hipemu.h")]
    jobs = []
    objs = []
    for s in SOURCES + ["hipemu.cpp"]:
        src = os.path.join(emudir if s == "hipemu.cpp" else HERE, s)
        obj = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if _newer([src] + HEADERS + emu_hdr + [os.path.abspath(__file__)], obj):
            jobs.append([HOSTCXX] + flags + ["-c", src, "-o", obj])
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(_run, jobs))
    if jobs or not os.path.exists(out):
        _run([HOSTCXX, "-shared", "-fPIC", "-o", out] + objs)
    if verbose:
        print("built", out)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--emu", action="store_true")
    ap.add_argument("--both", action="store_true")
    a = ap.parse_args()
    if a.emu or a.both:
        build_emu(verbose=True)
    if not a.emu or a.both:
        build_device(verbose=True)
