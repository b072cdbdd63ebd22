"""ctypes binding of libmidihip.so.  The prototypes are read from include/midihip.h (the single source
of truth for the C-ABI), so a symbol the header declares but the library lacks is an import-time error.

There is NO fallback: if the shared object is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "midihip.h")
LIB_PATH = os.environ.get("MH_LIB_PATH") or os.path.join(HERE, "libmidihip.so")  # (MH_LIB_PATH: A/B runs of two builds)

# the A/B test library (build.py: the same sources with -DMH_AB_BUILDS); loaded by tests / tools through use_ab() only
AB_LIB_PATH = os.path.join(HERE, "libmidihip_ab.so")

MH_F32, MH_BF16 = 0, 1

_CTYPES = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "int32_t": ctypes.c_int32}


def parse_header(path: str = HEADER) -> Dict[str, Tuple[str, List[Tuple[str, str]]]]:
    """-> {name: (return_type, [(ctype, argname), ...])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int)\s+(mh_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        parsed = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                parsed.append((mm.group(1).strip(), mm.group(2)))
        protos[name] = ("str" if "char" in ret else "int", parsed)
    return protos


def _to_ctype(t: str):
    if "*" in t:
        return ctypes.c_char_p if "char" in t else ctypes.c_void_p
    t = t.replace("const", "").strip()
    return _CTYPES[t]


class _Lib:
    def __init__(self, path: str = "") -> None:
        path = path or LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: the HIP kernels are the only implementation of this path. "
                "Build them with `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc).")
        self.path = path
        # ONE HIP runtime per process.  PyTorch-ROCm carries its own libamdhip64 / ROCr; were this library loaded first, the
        # loader would bind it to /opt/rocm's copies, `import torch` would bring a second runtime, and whichever of the two
        # initialises second finds no device ("no ROCm-capable device is detected" at the first launch).  With torch imported
        # first the library's DT_NEEDED sonames resolve to the copies torch already holds.  (A host without torch links
        # /opt/rocm's runtime and has nothing to order: INTEGRATION.md.)
        import torch  # noqa: F401
        self.cdll = ctypes.CDLL(path)
        self.protos = parse_header()
        for name, (ret, args) in self.protos.items():
            try:
                fn = getattr(self.cdll, name)
            except AttributeError as e:
                if os.environ.get("MH_LIB_PATH") and path == os.environ["MH_LIB_PATH"]:
                    continue  # (an OLDER build named for a same-box A/B run, tools/*_lib_once.py: entry points added since are absent)
                raise RuntimeError(f"libmidihip.so does not export {name} declared in include/midihip.h") from e
            fn.argtypes = [_to_ctype(t) for t, _ in args]
            fn.restype = ctypes.c_char_p if ret == "str" else ctypes.c_int

    profile = None  # bench.py: set to a list to collect (entry point, start event, end event) around EVERY C-ABI call

    def call(self, name: str, *args) -> None:
        prof = self.profile
        if prof is not None:
            import torch  # (HIP events on torch's current stream == the stream every wrapper in ops.py launches on)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rc = getattr(self.cdll, name)(*args)
        if prof is not None:
            e1.record()
            prof.append((name, e0, e1))
        if rc != 0:
            msg = self.cdll.mh_last_error()
            raise RuntimeError(f"{name} failed ({rc}): {msg.decode() if msg else '?'}")


_lib = None
_ab_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


class use_ab:
    """``with use_ab():`` -- tests and measurement tools only: every C-ABI call of this process goes to the A/B library
    (libmidihip_ab.so: the production sources + the kernel forms kept for comparisons) while the block runs.  The package
    itself never enters it."""

    def __enter__(self):
        global _lib, _ab_lib
        if _ab_lib is None:
            _ab_lib = _Lib(AB_LIB_PATH)
            if _ab_lib.cdll.mh_ab_builds() != 1:
                raise RuntimeError(f"{AB_LIB_PATH} was not built with -DMH_AB_BUILDS")
        self.prev = lib()
        _ab_lib.profile = self.prev.profile
        _lib = _ab_lib
        return _ab_lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


def available() -> bool:
    return os.path.exists(LIB_PATH)
