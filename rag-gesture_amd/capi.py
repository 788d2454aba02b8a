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


def header_symbols():
    """Every entry point include/rg_gesture.h declares."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rg_[a-z0-9_]+)\s*\(", text)))


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RgError("HIP extension not built: %s is missing (run __graft_entry__.build())" % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.rg_last_error.restype = ctypes.c_char_p
        _lib.rg_last_error.argtypes = [ctypes.c_void_p]
        _lib.rg_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
        _lib.rg_destroy.argtypes = [ctypes.c_void_p]
        _lib.rg_destroy.restype = None
    return _lib


def _convert(a):
    if isinstance(a, torch.Tensor):
        if not a.is_cuda:
            raise RgError("device tensor expected, got a CPU tensor")
        if not a.is_contiguous():
            raise RgError("contiguous tensor expected")
        return ctypes.c_void_p(a.data_ptr())
    if a is None:
        return ctypes.c_void_p(0)
    if isinstance(a, bool):
        return ctypes.c_int(int(a))
    if isinstance(a, int):
        return ctypes.c_int64(a) if abs(a) >= 2 ** 31 else ctypes.c_int(a)
    if isinstance(a, float):
        return ctypes.c_float(a)
    return a


class I64(int):
    """Marks an integer argument that the C prototype declares as int64_t."""


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
        cargs = [ctypes.c_int64(int(a)) if isinstance(a, I64) else _convert(a) for a in args]
        rc = fn(self._h, *cargs, ctypes.c_void_p(s))
        if rc != 0:
            raise RgError("rg_%s failed (%d): %s" % (name, rc, self.lib.rg_last_error(self._h).decode()))


_handles = {}


def get_handle(device=None):
    d = torch.cuda.current_device() if device is None else int(device)
    if d not in _handles:
        _handles[d] = Handle(d)
    return _handles[d]
