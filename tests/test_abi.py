"""The C-ABI shared library: it loads, and exports exactly what include/colddiff.h declares."""
import ctypes
import os
import re
import subprocess

from colddiff import _lib


def test_header_parses_and_library_exports_every_symbol():
    protos = _lib.parse_header()
    assert len(protos) >= 45 and "cdf_conv_gemm" in protos and "cdf_blur_chain" in protos
    assert os.path.exists(_lib.LIB_PATH), "build first: python __graft_entry__.py"
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), name
    lib = _lib.Lib(_lib.LIB_PATH)
    assert lib.cdf_abi_version() == lib.abi_version >= 4 and lib.cdf_is_device_build() == 1
    assert lib.cdf_last_error() is not None


def test_no_undeclared_exports():
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (cdf_\w+)", out))
    declared = set(_lib.parse_header())
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))


def test_bad_arguments_return_status_not_abort():
    lib = _lib.Lib(_lib.LIB_PATH)
    try:
        lib.cdf_blur_chain(0, 0, 0, 0, 0, 0, 1, 1, 8, 8, 3, 0, 0, 0, -1, 0, 0)
        assert False, "expected CdfError"
    except _lib.CdfError as e:
        assert "null pointer" in str(e)
    assert lib._dll.cdf_blur_chain(0, 0, 0, 0, 0, 0, 1, 1, 8, 8, 4, 0, 0, 0, -1, 0, 0) == -1      # CDF_E_INVALID


def test_hot_kernels_use_no_scratch():
    """hipcc silently parks arrays of HIP vector structs and over-hoisted loads in scratch memory (each cost 1.3-2x
    when it happened); the hot kernels must compile to zero scratch.  Needs hipcc (cross-compiles without a GPU)."""
    import shutil
    import pytest
    hipcc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
    if not shutil.which(hipcc):
        pytest.skip("hipcc not available")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(repo, "cold-diffusion-models_amd", "csrc")
    hot = {"k_conv_sp.hip": ("conv_igemm_spx_kernel", "conv_igemm_halo_kernel", "conv_igemm_rowhalo_stream_kernel", "conv_wgrad_spx_kernel", "conv_wgrad_row3_kernel", "conv_igemm_sp_kernel",
                             "conv_wgrad_sp_kernel", "split_bf16_kernel"),
           "k_conv.hip": ("conv_igemm_kernel", "unpack_reduce_kernel"),
           "k_dwconv.hip": ("dwconv7_kernel", "dwconv7_wgrad_partial_kernel"),
           "k_norm.hip": ("layernorm_c_fwd_kernel", "layernorm_c_bwd_kernel")}
    from concurrent.futures import ThreadPoolExecutor

    def compile_one(src):
        return subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", csrc, "-I", os.path.join(repo, "include"),
                               "-c", os.path.join(csrc, src), "-o", os.devnull, "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"],
                              capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        results = dict(zip(hot, ex.map(compile_one, hot)))
    for src, names in hot.items():
        r = results[src]
        assert r.returncode == 0, r.stderr[-2000:]
        cur, seen = None, set()
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = m.group(1)
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            if m and cur and any(n in cur for n in names):
                seen.add(cur)
                assert int(m.group(1)) == 0, (cur, line)
        assert seen, "no resource-usage remarks parsed for " + src


def test_argument_checks_of_the_gemm_and_blend_entry_points():
    """Contract of include/colddiff.h: a bad argument is a status + message, never an abort, never a launch (host-side checks:
    they run without a GPU on the device build too)."""
    import torch
    lib = _lib.Lib(_lib.LIB_PATH)
    buf = torch.zeros(4096)
    p = buf.data_ptr()
    desc = (ctypes.c_int * 8)(0, 0, 1, 0, 0, 0, 0, 0)

    def expect(fn, args, text):
        try:
            fn(*args)
        except _lib.CdfError as e:
            assert text in str(e), str(e)
        else:
            raise AssertionError("expected CdfError containing %r" % text)

    # pre-split GEMM: unaligned operand, channel count not a multiple of 8, split planes without the vectorised epilogue layout
    gemm = [p, p, 8, p, p, p, 32, p, 8, 1, 4, 4, 8, 4, 4, 8, 4, 4, 1, 1, 1, desc, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]
    bad = list(gemm); bad[0] = p + 2
    expect(lib.cdf_conv_gemm_bf16x, bad, "16B aligned")
    bad = list(gemm); bad[12] = 6
    expect(lib.cdf_conv_gemm_bf16x, bad, "multiples of 8")
    bad = list(gemm); bad[34], bad[35], bad[36] = p, p, 6
    expect(lib.cdf_conv_gemm_bf16x, bad, "output planes need")
    # weight gradient: tap count out of range
    expect(lib.cdf_conv_wgrad_bf16x, [p, p, 8, p, p, 8, p, p, 8, 1, 4, 4, 4, 4, 1, 4, 4, 1, 8, 8, 0, desc, 1, 0, 0, 0], "tap / split count")
    # depthwise 7 x 7: the kernel's per-image element offsets are 24 x 24-bit products kept in 32 bits -- larger images are refused, not wrapped
    expect(lib.cdf_dwconv7, [p, 4096, p, 4096, 0, 0, 0, p, 4096, 1, 1024, 1024, 4096, 0, 0, 0, 0, 0], "32-bit per-image offsets")
    # fused GroupNorm tail: dropout probability out of range, output planes without a hi plane
    gn = [p, 32, p, 32, p, p, p, p, p, 1, 16, 32, 32, 1e-6, 1, 0.0, 0, 0, 0, 0, 0]
    bad = list(gn); bad[15] = 1.0
    expect(lib.cdf_groupnorm_fwd_ex, bad, "dropout probability")
    bad = list(gn); bad[18] = p
    expect(lib.cdf_groupnorm_fwd_ex, bad, "lo plane without hi plane")
    # per-pixel blend: t = 0 is not a reverse step
    expect(lib.cdf_blend_step, [p, p, p, p, p, 0, p, 16, 48, 0], "bad args")
    expect(lib.cdf_blend_qsample, [p, p, p, p, 0, p, 1, 3, 16, 0], "bad args")
    # the optional tuning argument is validated: a struct of the wrong size (another header version) or with impossible tiles is refused;
    # the library has no setters and no tuning state of its own
    tune = _lib.GemmTuning(lib)
    assert tune.get("halo") == 47 and tune.get("dephase") == 1 and tune.get("splitk") == 1 and tune.get("small_n64") == 1
    bad = list(gemm); bad[-2] = tune.set(tile_bm=96).ptr
    expect(lib.cdf_conv_gemm_bf16x, bad, "bad cdf_gemm_tuning")
    bad = list(gemm); bad[-2] = tune.set(tile_bm=0, size=8).ptr
    expect(lib.cdf_conv_gemm_bf16x, bad, "bad cdf_gemm_tuning")
    assert not [n for n in lib.protos if n.endswith(("_tile", "_halo", "_halo_bm", "_dephase", "_deep", "_splitk", "_taprot", "_waves", "_row3",
                                                     "_swizzle", "_stack", "_onepass", "_slots", "_tiled", "_max_bm", "_small_n64")) and n != "cdf_conv_wgrad_bf16x_is_row3"]


def test_bench_quotes_profiles_only_with_matching_provenance(tmp_path):
    """bench.py's traffic / per-class figures come from committed rocprofv3 summaries: a summary stamped with another digest of the
    kernel sources (or with none, like the round-1..3 files) must not be quoted."""
    import importlib.util
    import json
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("cdf_bench", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sys.path.insert(0, os.path.join(repo, "tools"))
    import provenance
    now = provenance.csrc_sha16()
    assert len(now) == 16 and provenance.stamp()["csrc_sha16"] == now
    p = str(tmp_path / "x.json")
    assert bench._provenance(p, {"provenance": {"git_head": "abc", "csrc_sha16": now}})["match"] is True
    assert bench._provenance(p, {"_provenance": {"git_head": "abc", "csrc_sha16": now}})["match"] is True      # (kernel-trace summaries)
    assert bench._provenance(p, {"provenance": {"git_head": "abc", "csrc_sha16": "0" * 16}})["match"] is False
    assert bench._provenance(p, {})["match"] is False                                                            # unstamped: never quoted
    # the digest follows the sources: any byte of a kernel file changes it
    src = os.path.join(repo, "cold-diffusion-models_amd", "csrc", "k_misc.hip")
    data, st = open(src, "rb").read(), os.stat(src)
    try:
        open(src, "ab").write(b"\n// x\n")
        assert provenance.csrc_sha16() != now
    finally:
        open(src, "wb").write(data)
        os.utime(src, ns=(st.st_atime_ns, st.st_mtime_ns))       # (the build caches objects by mtime: leave no trace)
    assert provenance.csrc_sha16() == now
