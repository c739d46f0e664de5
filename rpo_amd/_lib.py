"""ctypes binding of librpo_hip.so (the C ABI declared in include/rpo_hip.h).

There is no CPU fallback: if the library is missing or fails to load, every use raises.  The prototypes are parsed
from the header itself so that the binding cannot drift from the declaration.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "rpo_hip.h")
# (RPO_HIP_LIBRARY: another build of the SAME ABI, e.g. the sanitizer build csrc/librpo_hip_asan.so of tests/test_sanitizers.py)
LIBRARY = os.environ.get("RPO_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "librpo_hip.so")

_CTYPES = (
    ("unsigned long long", ctypes.c_ulonglong),
    ("long long", ctypes.c_longlong),
    ("unsigned", ctypes.c_uint),
    ("float", ctypes.c_float),
    ("int", ctypes.c_int),
)


class RpoHipError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> ({name: [ctypes argtypes]}, {macro: int}) for every `int rpo_*(...)` prototype / integer #define."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(rpo_\w+)\s*\(([^)]*)\)\s*;", text):
        name, args = m.group(1), m.group(2).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                    continue
                for key, ct in _CTYPES:
                    if re.match(r"(const\s+)?%s\b" % key, a):
                        argtypes.append(ct)
                        break
                else:
                    raise RpoHipError("cannot map C parameter %r of %s" % (a, name))
        protos[name] = argtypes
    macros = {m.group(1): int(m.group(2)) for m in
              re.finditer(r"#define\s+(RPO_\w+)\s+\(?(-?\d+)\)?\s*(?:/\*.*)?$", open(path).read(), flags=re.M)}
    return protos, macros


PROTOTYPES, CONST = parse_header()
_lib = None


def _bind_to_torch_hip_runtime():
    """PyTorch-ROCm wheels carry their own libamdhip64.so; the streams and device pointers handed to the kernels belong
    to THAT runtime instance.  Put it into the global symbol scope before librpo_hip.so is loaded so that the library's
    HIP calls bind to it instead of opening a second runtime from /opt/rocm (which fails with hipErrorNoDevice on the
    first launch).  A non-Python host simply links the library against its own HIP runtime."""
    import torch
    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)


def load():
    """Load (once) and return the ctypes library with typed entry points.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBRARY):
        raise RpoHipError(
            "%s is missing: build it with `python rpo_amd/csrc/build.py` (hipcc --offload-arch=gfx950). "
            "rpo_amd has no CPU fallback." % LIBRARY)
    _bind_to_torch_hip_runtime()
    lib = ctypes.CDLL(LIBRARY)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError = symbol missing: loud by construction
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    if lib.rpo_abi_version() != CONST["RPO_ABI_VERSION"]:
        raise RpoHipError("librpo_hip.so ABI %d != header ABI %d; rebuild" %
                          (lib.rpo_abi_version(), CONST["RPO_ABI_VERSION"]))
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        kind = {CONST["RPO_ERR_ARG"]: "invalid argument", CONST["RPO_ERR_NULL"]: "null pointer"}.get(
            code, "hipError_t %d" % code)
        raise RpoHipError("%s failed: %s" % (what, kind))
