"""ctypes binding of the C-ABI HIP extension (include/rg_gesture.h).

The product path has no CPU or PyTorch fallback: if librg_gesture.so is missing or fails
to load, or no GPU is present, every op raises.
"""
import contextlib
import ctypes
import gc
import os
import re

import torch

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_TAG = "diag" if os.environ.get("RG_DIAG") == "1" else os.environ.get("RG_LIB_TAG", "")    # (see build.py: diagnostic / experiment builds)
LIB_PATH = os.path.join(PKG_DIR, "librg_gesture%s.so" % ("_" + _TAG if _TAG else ""))
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


def header_version():
    """RG_VERSION of include/rg_gesture.h."""
    with open(HEADER_PATH) as f:
        m = re.search(r"^#define\s+RG_VERSION\s+(\d+)", f.read(), flags=re.M)
    if not m:
        raise RgError("include/rg_gesture.h defines no RG_VERSION")
    return int(m.group(1))


_SCALARS = {"int": ctypes.c_int, "unsigned": ctypes.c_uint, "unsigned int": ctypes.c_uint, "int64_t": ctypes.c_int64,
            "long long": ctypes.c_longlong,
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
        _lib.rg_version.restype = ctypes.c_int
        want, got = header_version(), _lib.rg_version()
        if want != got:      # (argument blocks are passed by pointer: a library built against another header would misread them silently)
            _lib = None
            raise RgError("%s is ABI version %d, include/rg_gesture.h is %d: rebuild (__graft_entry__.build())" % (LIB_PATH, got, want))
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
        self.recorder = None      # an OpRecorder while launches are being recorded instead of issued (vae.py)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.rg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def call(self, name, *args, stream=None, keep=None):
        """Invoke rg_<name>(handle, *args, stream) and raise on a non-zero status.  keep: objects (argument blocks, tensors an
        argument block points into) that must stay alive until the launch has been issued (recording)."""
        if self.recorder is not None and stream is None:
            self.recorder.add(("call", name, args, keep))   # (the tensors in args / keep stay alive until the recorder has issued)
            return
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


@contextlib.contextmanager
def capture(graph):
    """torch.cuda.graph(graph) with Python's cyclic garbage collector off: a collection that starts in the middle of a
    capture can finalise objects that own device resources (graphs, events, streams of an earlier model), and a HIP call
    from a finaliser aborts a capturing process.  torch's context collects once before the capture begins."""
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph):
            yield
    finally:
        if was:
            gc.enable()


class OpRecorder:
    """Launch sequences of up to 4 independent, structurally identical jobs (the four body-part VAEs: same layers,
    different weights and rows), recorded job by job and issued position by position: where the jobs' i-th launches are
    the same operation on the same shapes they go out as ONE grouped launch (rg_gemm_grouped, rg_layernorm_grouped, ...),
    otherwise one by one in job order.  Control flow never depends on device data, so recording is exact; recorded
    arguments keep their tensors alive until `issue` (the allocator must not reuse a block of job 0 for job 1 while job 0's
    later launches, issued later, still need it)."""
    # name -> (indices of the per-job pointer arguments, grouped entry point, argument order of the grouped call)
    GROUPED = {"layernorm": ((0, 1, 2, 3, 6), "layernorm_grouped", ("P0", "P1", "P2", "P3", 4, 5, "P6")),
               "add_rows": ((0, 1, 2), "add_rows_grouped", ("P0", "P1", "P2", 3, 4)),
               "copy_rows": ((0, 1), "copy_rows_grouped", ("P0", "P1", 2, 3, 4, 5, 6, 7, 8))}

    def __init__(self):
        self.jobs, self.cur = [], None

    def begin_job(self):
        self.cur = []
        self.jobs.append(self.cur)

    def add(self, op):
        self.cur.append(op)

    @staticmethod
    def _ptr(a):
        return None if a is None else (_convert(a) if isinstance(a, torch.Tensor) else a)

    def _single(self, h, op, s):
        if op[0] == "call":
            h.call(op[1], *op[2], stream=s)
        elif op[0] == "gemm":
            rc = h.lib.rg_gemm(h._h, ctypes.byref(op[1]), ctypes.c_void_p(s))
            if rc != 0:
                raise RgError("rg_gemm failed (%d): %s" % (rc, h.lib.rg_last_error(h._h).decode()))
        else:   # ("mha", fast, args)
            fn = h.lib.rg_mha_bf16 if op[1] else h.lib.rg_mha
            a = [self._ptr(x) for x in op[2]]
            if fn(h._h, *a, s) != 0:
                raise RgError("rg_mha failed: %s" % h.lib.rg_last_error(h._h).decode())

    def issue(self, h):
        """Launch everything that was recorded (on the current stream) and forget it."""
        jobs, self.jobs, self.cur = self.jobs, [], None
        s = torch.cuda.current_stream().cuda_stream
        n = len(jobs)
        if n == 0:
            return
        if n > 4 or len({len(j) for j in jobs}) != 1:
            for j in jobs:
                for op in j:
                    self._single(h, op, s)
            return
        arr = lambda ptrs: (ctypes.c_void_p * n)(*ptrs)
        for ops in zip(*jobs):
            kinds = {(op[0], op[1] if op[0] != "gemm" else None) for op in ops}
            done = False
            if n > 1 and len(kinds) == 1:
                kind = ops[0][0]
                if kind == "gemm":
                    descs = (type(ops[0][1]) * n)(*[op[1] for op in ops])
                    rc = h.lib.rg_gemm_grouped(h._h, descs, n, ctypes.c_void_p(s))
                    if rc != 0:
                        raise RgError("rg_gemm_grouped failed (%d): %s" % (rc, h.lib.rg_last_error(h._h).decode()))
                    done = True
                elif kind == "call" and ops[0][1] in ("venc_forward", "vdec_step"):
                    # the parts' fused encoder stacks / decoder steps: one launch, gridDim.y = part (argument blocks by value)
                    blocks = [op[2][0]._obj for op in ops]
                    arr_t = type(blocks[0]) * n
                    h.call(ops[0][1] + "_grouped", ctypes.byref(arr_t(*blocks)), n, stream=s)
                    done = True
                elif kind == "call" and ops[0][1] in self.GROUPED:
                    pidx, entry, order = self.GROUPED[ops[0][1]]
                    shared = [tuple(a for i, a in enumerate(op[2]) if i not in pidx) for op in ops]
                    nulls = [tuple(op[2][i] is None for i in pidx) for op in ops]
                    if len(set(shared)) == 1 and len(set(nulls)) == 1:
                        args = [arr([self._ptr(op[2][int(o[1:])]) for op in ops]) if isinstance(o, str) else ops[0][2][o] for o in order]
                        h.call(entry, n, *args, stream=s)
                        done = True
                elif kind == "mha" and ops[0][1]:
                    a0 = ops[0][2]
                    if all(tuple(op[2][i] for i in (1, 3, 5, 7, 8, 9, 10, 11, 12, 13)) == tuple(a0[i] for i in (1, 3, 5, 7, 8, 9, 10, 11, 12, 13)) for op in ops):
                        q, k, v, o = (arr([self._ptr(op[2][i]) for op in ops]) for i in (0, 2, 4, 6))
                        h.call("mha_bf16_grouped", n, q, a0[1], k, a0[3], v, a0[5], o, *a0[7:], stream=s)
                        done = True
            if not done:
                for op in ops:
                    self._single(h, op, s)


_handles = {}


def get_handle(device=None):
    d = torch.cuda.current_device() if device is None else int(device)
    if d not in _handles:
        _handles[d] = Handle(d)
    return _handles[d]
