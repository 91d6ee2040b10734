"""Build + load libvunet_hip.so (the C-ABI HIP library, include/vunet_hip.h).

The library is built in-tree with hipcc for gfx950 only and loaded with ctypes.  There is no CPU
fallback: if the shared object is missing, or no GPU is present when a kernel is requested, the
product path raises.
"""
from __future__ import annotations

import ctypes
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libvunet_hip.so")
INCLUDE_DIR = os.path.join(os.path.dirname(PKG_DIR), "include")
HEADER = os.path.join(INCLUDE_DIR, "vunet_hip.h")
ARCH = "gfx950"


def headers() -> list:
    """Every public header of the C ABI (include/*.h): vunet_hip.h and the per-family headers beside it."""
    return sorted(os.path.join(INCLUDE_DIR, f) for f in os.listdir(INCLUDE_DIR) if f.endswith(".h"))


def _deps_of(src: str, seen=None) -> list:
    """``src`` and the project headers it includes (transitively; ``#include "..."`` resolved beside the includer)."""
    seen = set() if seen is None else seen
    if src in seen or not os.path.isfile(src):
        return []
    seen.add(src)
    out = [src]
    for inc in re.findall(r'^\s*#\s*include\s*"([^"]+)"', open(src).read(), flags=re.M):
        out += _deps_of(os.path.normpath(os.path.join(os.path.dirname(src), inc)), seen)
    return out


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = _sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + headers()
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 every csrc/*.hip and link libvunet_hip.so (cross-compiles without a GPU)."""
    if not force and not _needs_build():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(PKG_DIR, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        deps = _deps_of(src)   # the source and the headers it actually includes: a new entry point rebuilds its own file only
        if not force and os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in deps):
            return obj
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, _sources()))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB_PATH


def _header_text() -> str:
    txt = "\n".join(open(h).read() for h in headers())
    return re.sub(r"/\*.*?\*/", "", txt, flags=re.S)


def declared_symbols() -> list:
    """Function names declared in include/*.h."""
    txt = _header_text()
    return sorted(set(re.findall(r"\bint\s+(vunet_[a-z0-9_]+)\s*\(", txt)))


_CTYPE = {"int32_t": ctypes.c_int32, "uint32_t": ctypes.c_uint32, "int64_t": ctypes.c_int64,
          "float": ctypes.c_float, "double": ctypes.c_double, "int": ctypes.c_int}


def declared_prototypes() -> dict:
    """name -> list of ctypes argument types, parsed from include/*.h (single source of truth)."""
    txt = _header_text()
    protos = {}
    for m in re.finditer(r"\bint\s+(vunet_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", txt, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    types.append(ctypes.c_void_p)
                else:
                    types.append(_CTYPE[a.replace("const ", "").split(" ")[0]])
        protos[name] = types
    return protos


_LIB = None


def lib() -> ctypes.CDLL:
    """The loaded library.  Raises if it has not been built -- there is no fallback."""
    global _LIB
    if _LIB is None:
        # PyTorch ships its own libamdhip64: it must be the HIP runtime this library binds to (same
        # streams / allocations), so torch is imported -- and its runtime loaded -- before the dlopen.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the HIP extension is mandatory, there is no CPU fallback)")
        # VUNET_HIP_LIB: an alternative build of the same library (kernel tuning A/B runs: tools/ab_build.sh)
        _LIB = ctypes.CDLL(os.environ.get("VUNET_HIP_LIB", LIB_PATH))
        for name, argtypes in declared_prototypes().items():
            fn = getattr(_LIB, name)  # AttributeError if the .so is stale
            fn.restype = ctypes.c_int
            fn.argtypes = argtypes
    return _LIB
