"""ctypes binding of libcolddiff_hip.so.

The C header ``include/colddiff.h`` is the single source of truth for the ABI: its prototypes
are parsed here into ctypes signatures, so a symbol that is declared but not exported (or the
other way round) fails at load time.  There is NO fallback: if the gfx950 library cannot be
loaded, :func:`get` raises and every operator of the package is unusable.
"""
import ctypes
import os
import re

# torch must be imported BEFORE libcolddiff_hip.so is dlopen'ed: the wheel bundles its own HIP runtime
# (torch/lib/libamdhip64.so, soname libamdhip64.so.7) and our library's NEEDED libamdhip64.so.7 then binds
# to that already-loaded copy.  Loaded the other way round the process ends up with two HIP runtimes and
# every launch from this library fails with "no ROCm-capable device is detected".
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG_ROOT = os.path.dirname(_HERE)
_REPO = os.path.dirname(_PKG_ROOT)
HEADER = os.path.join(_REPO, "include", "colddiff.h")
LIB_PATH = os.environ.get("COLDDIFF_LIB") or os.path.join(_PKG_ROOT, "csrc", "libcolddiff_hip.so")   # env: an alternative gfx950 build

_CTYPES = {
    "int": ctypes.c_int,
    "long long": ctypes.c_longlong,
    "int64_t": ctypes.c_int64,
    "size_t": ctypes.c_size_t,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
}


def _ctype_of(decl):
    decl = decl.strip()
    if "*" in decl:
        return ctypes.c_char_p if decl.replace(" ", "") == "constchar*" else ctypes.c_void_p
    decl = re.sub(r"\bconst\b", "", decl).strip()
    # drop the parameter name
    for name, ct in sorted(_CTYPES.items(), key=lambda kv: -len(kv[0])):
        if decl == name or decl.startswith(name + " "):
            return ct
    raise ValueError("colddiff.h: unsupported C type in %r" % decl)


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(cdf_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        restype = _ctype_of(ret + " x") if "*" not in ret else _ctype_of(ret)
        argtypes = [] if args in ("", "void") else [_ctype_of(a) for a in args.split(",")]
        protos[name] = (restype, argtypes)
    return protos


# int-returning entry points that are pure host-side queries (sizes / counts), not status codes
_QUERY = re.compile(r"(_blocks|_nchunk|_nsplit|_abi_version|_is_device_build|_lds_bytes|_is_row3|_ssim_tiles|_pack_entry_bytes|_pack_blocks|_kvctx_parts|_bf16x_ksplit)$")


class CdfError(RuntimeError):
    pass


class Lib:
    """A loaded colddiff kernel library with checked call wrappers."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise CdfError(
                "colddiff: %s not found — build it with `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
        self.path = path
        self._dll = ctypes.CDLL(path)
        self.protos = parse_header()
        for name, (restype, argtypes) in self.protos.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError:
                raise CdfError("colddiff: %s does not export %s declared in colddiff.h" % (path, name))
            fn.restype = restype
            fn.argtypes = argtypes
            if restype is ctypes.c_int and not _QUERY.search(name):
                setattr(self, name, self._checked(name, fn))
            else:
                setattr(self, name, fn)

    def _checked(self, name, fn):
        last_error = self._dll.cdf_last_error
        last_error.restype = ctypes.c_char_p

        def call(*args):
            rc = fn(*args)
            if rc != 0:
                raise CdfError("%s failed (%d): %s" % (name, rc, last_error().decode()))
            return rc

        call.__name__ = name
        return call


_instance = None


def get():
    """The process-wide library handle (loads on first use; raises if the .so is missing)."""
    global _instance
    if _instance is None:
        _instance = Lib(LIB_PATH)
        # tuning hooks of include/colddiff.h from the environment (results do not depend on them)
        tile, waves = os.environ.get("COLDDIFF_SPX_TILE"), os.environ.get("COLDDIFF_SPX_WAVES")
        if tile:
            _instance.cdf_conv_gemm_bf16x_tile(*[int(v) for v in tile.split("x")])
        if waves:
            _instance.cdf_conv_gemm_bf16x_waves(int(waves))
        if os.environ.get("COLDDIFF_SPX_SPLITK"):
            _instance.cdf_conv_gemm_bf16x_splitk(int(os.environ["COLDDIFF_SPX_SPLITK"]))
        if os.environ.get("COLDDIFF_SPX_DEEP"):
            _instance.cdf_conv_gemm_bf16x_deep(int(os.environ["COLDDIFF_SPX_DEEP"]))
        if os.environ.get("COLDDIFF_KVCTX_SLOTS"):
            _instance.cdf_linattn_kvctx_slots(int(os.environ["COLDDIFF_KVCTX_SLOTS"]))
        if os.environ.get("COLDDIFF_LINATTN_ONEPASS"):
            _instance.cdf_linattn_onepass(int(os.environ["COLDDIFF_LINATTN_ONEPASS"]))
        if os.environ.get("COLDDIFF_UNPACK_TILED"):
            _instance.cdf_unpack_reduce_tiled(int(os.environ["COLDDIFF_UNPACK_TILED"]))
        if os.environ.get("COLDDIFF_WGRAD_ROW3"):
            _instance.cdf_conv_wgrad_bf16x_row3(int(os.environ["COLDDIFF_WGRAD_ROW3"]))
        if os.environ.get("COLDDIFF_WGRAD_SWIZZLE"):
            _instance.cdf_conv_wgrad_bf16x_swizzle(int(os.environ["COLDDIFF_WGRAD_SWIZZLE"]))
        if os.environ.get("COLDDIFF_SPX_TAPROT"):
            _instance.cdf_conv_gemm_bf16x_taprot(int(os.environ["COLDDIFF_SPX_TAPROT"]))
        if os.environ.get("COLDDIFF_SPX_HALO"):
            v = [int(x) for x in os.environ["COLDDIFF_SPX_HALO"].split(",")]
            _instance.cdf_conv_gemm_bf16x_halo(v[0], v[1] if len(v) > 1 else 1)
        if os.environ.get("COLDDIFF_SPX_HALO_BM"):
            _instance.cdf_conv_gemm_bf16x_halo_bm(int(os.environ["COLDDIFF_SPX_HALO_BM"]))
        if os.environ.get("COLDDIFF_SPX_MAX_BM"):
            _instance.cdf_conv_gemm_bf16x_max_bm(int(os.environ["COLDDIFF_SPX_MAX_BM"]))
    return _instance
