"""ctypes binding of libdana_hip.so (the C-ABI drop-in boundary, include/dana_hip.h).

The prototypes are parsed from the header itself, so the Python side can never drift from the
C ABI. There is NO fallback: if the shared library is missing the import of any op raises, and
every op refuses non-CUDA tensors -- the product path is the HIP path or nothing.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(_ROOT, "include", "dana_hip.h")
# debug / tuning switches: bound for tools/ and the bit-identity tests, not part of the drop-in ABI
DEBUG_HEADER = os.path.join(_ROOT, "include", "dana_hip_debug.h")
LIB_PATH = os.environ.get("DANA_LIB_PATH") or os.path.join(_HERE, "libdana_hip.so")  # (override: same-box A/B of two builds)

_CTYPES = {
    "int": ctypes.c_int,
    "long": ctypes.c_long,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "unsigned long long": ctypes.c_ulonglong,
    "size_t": ctypes.c_size_t,
    "dana_stream_t": ctypes.c_void_p,
    "unsigned int": ctypes.c_uint,
}


def parse_header(path=HEADER):
    """-> {name: (restype, [(argtype, argname), ...])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    protos = {}
    for m in re.finditer(r"(const char\*|int|size_t)\s+(dana_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        alist = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                ty, an = mm.group(1).strip(), mm.group(2)
                alist.append((ty, an))
        protos[name] = (ret, alist)
    return protos


def _ctype(ty):
    if ty.endswith("*"):
        return ctypes.c_void_p
    return _CTYPES[ty.replace("const ", "")]


class DanaError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libdana_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C %s/csrc`. There is no CPU fallback." % (LIB_PATH, _HERE))
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        self.debug_protos = parse_header(DEBUG_HEADER) if os.path.exists(DEBUG_HEADER) else {}
        self.fn = {}  # name -> bound foreign function (one dict lookup per launch on the hot path)
        for name, (ret, args) in list(self.protos.items()) + list(self.debug_protos.items()):
            fn = getattr(self.cdll, name)
            fn.argtypes = [_ctype(t) for t, _ in args]
            fn.restype = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "const char*": ctypes.c_char_p}[ret]
            self.fn[name] = fn

    def call(self, name, *args):
        """Call an `int dana_*` entry point; raise DanaError with dana_last_error() on failure."""
        fn = self.fn[name]
        rc = fn(*args)
        if rc != 0:
            raise DanaError("%s failed (%d): %s" % (name, rc, self.cdll.dana_last_error().decode()))
        if RECORDER is not None:  # program.LaunchProgram: the eager step's launch sequence, recorded for replay
            RECORDER.add_call(fn, name, args)

    def query(self, name, *args):
        return self.fn[name](*args)


_lib = None
RECORDER = None  # set by program.LaunchProgram.recording(): every successful call is appended to the program


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
