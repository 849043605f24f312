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


def parse_struct(name, path=HEADER):
    """Field names of `typedef struct <name> { int a; int b, c; ... } <name>;` in the header (all fields are C ints)."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    m = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), src, flags=re.S)
    if not m:
        raise ValueError("colddiff.h: struct %s not found" % name)
    fields = []
    for decl in m.group(1).split(";"):
        decl = decl.strip()
        if not decl:
            continue
        assert decl.startswith("int "), "colddiff.h: %s has a non-int field: %r" % (name, decl)
        fields += [f.strip() for f in decl[4:].split(",")]
    return fields


class GemmTuning:
    """`cdf_gemm_tuning` of include/colddiff.h as a ctypes structure (fields read from the header): the OPTIONAL tuning argument of the
    pre-split GEMM entry points.  The library holds no tuning state; whoever calls it passes one of these (or NULL for the defaults).
    `from_env()` reads the COLDDIFF_SPX_* / COLDDIFF_WGRAD_* variables the measurement tools use -- in Python only."""
    _cls = None

    def __init__(self, lib):
        if GemmTuning._cls is None:
            names = parse_struct("cdf_gemm_tuning")
            GemmTuning._cls = type("cdf_gemm_tuning", (ctypes.Structure,), {"_fields_": [(n, ctypes.c_int) for n in names]})
        self.c = GemmTuning._cls()
        lib.cdf_gemm_tuning_default(ctypes.byref(self.c))
        assert self.c.size == ctypes.sizeof(self.c), "cdf_gemm_tuning: header and library disagree"
        self.ptr = ctypes.addressof(self.c)

    def set(self, **kw):
        for k, v in kw.items():
            assert hasattr(self.c, k), "cdf_gemm_tuning has no field %r" % k
            setattr(self.c, k, int(v))
        return self

    def get(self, k):
        return getattr(self.c, k)

    def from_env(self):
        env = os.environ.get
        if env("COLDDIFF_SPX_TILE"):
            bm, bn = (int(v) for v in env("COLDDIFF_SPX_TILE").split("x"))
            self.set(tile_bm=bm, tile_bn=bn)
        if env("COLDDIFF_SPX_HALO"):
            v = [int(x) for x in env("COLDDIFF_SPX_HALO").split(",")]
            self.set(halo=v[0], halo_min_tiles=v[1] if len(v) > 1 else 1)
        for var, field in (("COLDDIFF_SPX_SPLITK", "splitk"), ("COLDDIFF_SPX_DEEP", "deep"), ("COLDDIFF_SPX_HALO_BM", "halo_bm"),
                           ("COLDDIFF_SPX_MAX_BM", "max_bm"), ("COLDDIFF_SPX_DEPHASE", "dephase"), ("COLDDIFF_SPX_SMALL_N64", "small_n64"),
                           ("COLDDIFF_WGRAD_ROW3", "wgrad_row3"), ("COLDDIFF_ROWHALO_STREAM", "rowhalo_stream"), ("COLDDIFF_WGRAD_SWIZZLE", "wgrad_swizzle"), ("COLDDIFF_WGRAD_STACK", "wgrad_stack"), ("COLDDIFF_RESIDENT_RESERVE", "resident_reserve"), ("COLDDIFF_EPILOGUE", "epilogue")):
            if env(var):
                v = int(env(var))
                ok = self._ALLOWED.get(field)
                if ok is not None and v not in ok:             # named here: the library only says `bad cdf_gemm_tuning` (on every GEMM call)
                    raise ValueError(f"{var}={v}: cdf_gemm_tuning.{field} takes " + (f"{ok.start}..{ok.stop - 1}" if isinstance(ok, range) else "/".join(map(str, sorted(ok)))))
                self.set(**{field: v})
        return self

    # the values cdf_tune_ok (csrc/k_conv_sp.hip) accepts, field by field
    _ALLOWED = {"rowhalo_stream": (0, 1), "epilogue": (0, 1), "resident_reserve": range(0, 249), "halo_bm": (0, 128, 256), "max_bm": (0, 128, 256),
                "tile_bm": (0, 64, 128, 256), "tile_bn": (0, 64, 128), "halo": range(0, 128)}


# int-returning entry points that are pure host-side queries (sizes / counts), not status codes
_QUERY = re.compile(r"(_blocks|_nchunk|_nsplit|_abi_version|_is_device_build|_lds_bytes|_is_row3|_ssim_tiles|_pack_entry_bytes|_pack_blocks|_kvctx_parts|_bf16x_ksplit|_lnbwd_ok)$")


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
        # header and library must be the same ABI revision BEFORE any call: a library built from another revision of colddiff.h would
        # read shifted arguments (e.g. the stream pointer as a flag)
        want = int(re.search(r"^\s*#\s*define\s+CDF_ABI_VERSION\s+(\d+)", open(HEADER).read(), flags=re.M).group(1))
        try:
            ver = self._dll.cdf_abi_version
        except AttributeError:
            raise CdfError("colddiff: %s does not export cdf_abi_version" % path)
        ver.restype, ver.argtypes = ctypes.c_int, []
        if ver() != want:
            raise CdfError("colddiff: %s is ABI version %d, include/colddiff.h is version %d -- rebuild the library from this tree "
                           "(python __graft_entry__.py)" % (path, ver(), want))
        self.abi_version = want
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
    return _instance
