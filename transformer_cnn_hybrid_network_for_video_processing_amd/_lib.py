"""ctypes binding of libhybrid_hip.so.  Prototypes are parsed from include/hybrid_hip.h so the
Python side cannot drift from the C ABI.  There is NO fallback: if the shared object is missing or
a call fails, a RuntimeError is raised."""
import ctypes
import os
import re

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
HEADER = os.path.join(ROOT, "include", "hybrid_hip.h")
LIB_PATH = os.path.join(PKG, "libhybrid_hip.so")
LIB_X3_PATH = os.path.join(PKG, "libhybrid_hip_x3.so")

# dtype codes of the C ABI (include/hybrid_hip.h) -- and HYB_F32X3, which exists on the host side only: the SAME ABI served by the second
# build of the library (libhybrid_hip_x3.so, -DHYB_F32_X3), where HYB_F32 means "fp32 storage, products from split-bf16 MFMAs"
HYB_F32, HYB_BF16, HYB_F32X3 = 0, 1, 2
HYB_H_BF16 = 0x100           # flag on the dtype of hyb_temporal_*: the pooled map / its gradient are bf16 (include/hybrid_hip.h)

_CTYPES = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "long long": ctypes.c_longlong,
    "unsigned long long": ctypes.c_ulonglong,
    "size_t": ctypes.c_size_t,
}


DTYPE_FIRST = set()          # entry points whose first parameter is `int dtype`
BOTH_BUILDS = ("hyb_profile_set", "hyb_profile_clear")      # no dtype argument, but state in each of the two libraries


def parse_header(path=HEADER):
    """-> {name: (restype_str, [argtype_str, ...])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    DTYPE_FIRST.clear()
    for m in re.finditer(r"\b(int|size_t|long long)\s+(hyb_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append("ptr")
                else:
                    a = re.sub(r"\bconst\b", "", a).strip()
                    a = re.sub(r"\s+\w+$", "", a).strip()      # drop the parameter name
                    argtypes.append(a)
        protos[name] = (ret, argtypes)
        if re.match(r"\s*int\s+dtype\b", args):
            DTYPE_FIRST.add(name)
    return protos


class _Lib:
    def __init__(self, path=LIB_PATH):
        self._dll = None
        self._path = path
        self.protos = parse_header()

    def _load(self):
        if self._dll is not None:
            return self._dll
        if not os.path.exists(self._path):
            raise RuntimeError(
                f"{self._path} is missing: the HIP extension has not been built. Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `python -m transformer_cnn_hybrid_network_for_video_processing_amd.build`). "
                "There is no CPU/eager fallback for this path.")
        dll = ctypes.CDLL(self._path)
        for name, (ret, args) in self.protos.items():
            fn = getattr(dll, name)            # AttributeError if the .so does not export a declared symbol
            fn.restype = _CTYPES[ret]
            fn.argtypes = [ctypes.c_void_p if a == "ptr" else _CTYPES[a] for a in args]
        self._dll = dll
        return dll

    def raw(self, name):
        return getattr(self._load(), name)

    def call(self, name, *args):
        """Call an int-returning entry point; raise on a non-zero status."""
        rc = getattr(self._load(), name)(*args)
        if rc != 0:
            kind = "argument check" if rc == -1 else "workspace too small" if rc == -2 else f"hipError_t {rc}"
            raise RuntimeError(f"{name} failed: {kind} (status {rc})")

    def query(self, name, *args):
        """Call a size/count-returning entry point."""
        return getattr(self._load(), name)(*args)


class _Mux(_Lib):
    """The library as the host code sees it: calls whose first argument is the dtype code HYB_F32X3 go to the split-bf16 build with
    HYB_F32 in its place; everything else to the main build."""

    def __init__(self):
        super().__init__(LIB_PATH)
        self.x3 = _Lib(LIB_X3_PATH)

    @staticmethod
    def _x3(name, args):
        return bool(args) and name in DTYPE_FIRST and isinstance(args[0], int) and not isinstance(args[0], bool) and (args[0] & 0xff) == HYB_F32X3

    def call(self, name, *args):
        if self._x3(name, args):
            return self.x3.call(name, HYB_F32 | (args[0] & ~0xff), *args[1:])
        if name in BOTH_BUILDS:            # process-global state that each build keeps for itself (the measurement hooks)
            self.x3.call(name, *args)
        return super().call(name, *args)

    def query(self, name, *args):
        if self._x3(name, args):
            return self.x3.query(name, HYB_F32 | (args[0] & ~0xff), *args[1:])
        return super().query(name, *args)


lib = _Mux()


def ptr_array(ptrs):
    """Host array of device pointers for `const float* const*` parameters."""
    return (ctypes.c_void_p * len(ptrs))(*ptrs)
