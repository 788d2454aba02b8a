"""ctypes binding of the C-ABI HIP extension (include/rg_gesture.h).

The product path has no CPU or PyTorch fallback: if librg_gesture.so is missing or fails
to load, or no GPU is present, every op raises.
"""
import ctypes
import os
import re

import torch

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "librg_gesture_diag.so" if os.environ.get("RG_DIAG") == "1" else "librg_gesture.so")
HEADER_PATH = os.path.join(os.path.dirname(PKG_DIR), "include", "rg_gesture.h")

_lib = None


class RgError(RuntimeError):
    pass


class RgConfigError(RgError, AssertionError):
    """An argument / configuration check that the reference states as an `assert` (e.g. the inference_kwargs compatibility
    rules, diffusion_architecture.py:227-241): still an AssertionError for callers that catch one, but raised explicitly,
    so it does not disappear under `python -O`."""


def require(cond, msg="unsupported configuration"):
    if not cond:
        raise RgConfigError(msg)


def header_symbols():
    """Every entry point include/rg_gesture.h declares."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rg_[a-z0-9_]+)\s*\(", text)))


_SCALARS = {"int": ctypes.c_int, "unsigned": ctypes.c_uint, "unsigned int": ctypes.c_uint, "int64_t": ctypes.c_int64,
            "float": ctypes.c_float, "double": ctypes.c_double}


def header_prototypes():
    """name -> (restype, [argtypes]) for every function include/rg_gesture.h declares: pointers (device or host) are
    void*, scalars keep their C width, so ctypes converts and range-checks every argument instead of guessing."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)   # struct bodies hold no prototypes
    protos = {}
    for ret, name, args in re.findall(r"\b(int|void|const\s+char\s*\*)\s+(rg_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        argtypes = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                    continue
                ty = " ".join(a.replace("const ", "").split()[:-1])
                if ty not in _SCALARS:
                    raise RgError("include/rg_gesture.h: cannot bind argument %r of %s" % (a, name))
                argtypes.append(_SCALARS[ty])
        restype = None if ret == "void" else (ctypes.c_char_p if "char" in ret else ctypes.c_int)
        protos[name] = (restype, argtypes)
    return protos


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RgError("HIP extension not built: %s is missing (run __graft_entry__.build())" % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in header_prototypes().items():
            fn = getattr(_lib, name, None)
            if fn is not None:
                fn.restype, fn.argtypes = restype, argtypes
        _lib.rg_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
    return _lib


def _convert(a):
    """Tensors -> device pointers; everything else is converted (and range-checked) by the prototype's argtypes."""
    if isinstance(a, torch.Tensor):
        if not a.is_cuda:
            raise RgError("device tensor expected, got a CPU tensor")
        if not a.is_contiguous():
            raise RgError("contiguous tensor expected")
        return a.data_ptr()
    if isinstance(a, bool):
        return int(a)
    return a


class I64(int):
    """Kept for callers that mark int64_t arguments; the width now comes from the header's prototype."""


class Handle:
    """One rg_handle per device; calls on a handle are serialised by the caller."""

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise RgError("no GPU visible: the HIP path cannot run (there is no CPU fallback)")
        lib = load_library()
        self.lib = lib
        self.device = torch.cuda.current_device() if device is None else int(device)
        h = ctypes.c_void_p()
        rc = lib.rg_create(ctypes.byref(h), self.device)
        if rc != 0:
            raise RgError("rg_create failed with %d" % rc)
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.rg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def call(self, name, *args, stream=None):
        """Invoke rg_<name>(handle, *args, stream) and raise on a non-zero status."""
        fn = getattr(self.lib, "rg_" + name)
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        if fn.argtypes is None or len(fn.argtypes) != len(args) + 2:
            raise RgError("rg_%s takes %s arguments besides handle and stream, got %d"
                          % (name, "?" if fn.argtypes is None else len(fn.argtypes) - 2, len(args)))
        try:
            rc = fn(self._h, *[_convert(a) for a in args], s)
        except ctypes.ArgumentError as e:
            raise RgError("rg_%s: %s" % (name, e))
        if rc != 0:
            raise RgError("rg_%s failed (%d): %s" % (name, rc, self.lib.rg_last_error(self._h).decode()))


_handles = {}


def get_handle(device=None):
    d = torch.cuda.current_device() if device is None else int(device)
    if d not in _handles:
        _handles[d] = Handle(d)
    return _handles[d]
