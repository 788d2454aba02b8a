"""Host-side descriptor builder for rg_gemm (include/rg_gesture.h: rg_gemm_desc)."""
import ctypes

import torch

from . import capi

A_IDENT, A_LN, A_STYL = 0, 1, 2
STATS_COLS = 128  # rg_gemm emits one (sum, sumsq) pair per 128 output columns
MAX_SEG = 4
_vp = ctypes.c_void_p


class ASegment(ctypes.Structure):
    _fields_ = [("src", _vp), ("ld", ctypes.c_int), ("mode", ctypes.c_int), ("stats", _vp),
                ("nparts", ctypes.c_int), ("pad_", ctypes.c_int), ("gamma", _vp), ("beta", _vp),
                ("scale_shift", _vp)]


class GemmDesc(ctypes.Structure):
    _fields_ = [("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int), ("a_is_bf16", ctypes.c_int),
                ("A", _vp), ("lda", ctypes.c_int), ("a_row_mod", ctypes.c_int), ("seg_len", ctypes.c_int),
                ("nseg", ctypes.c_int), ("seg", ASegment * MAX_SEG), ("gb_group", ctypes.c_int),
                ("gb_stride", ctypes.c_int), ("W", _vp), ("ldw", ctypes.c_int), ("act", ctypes.c_int),
                ("bias", _vp), ("tbias", _vp), ("tb_period", ctypes.c_int), ("softmax_cols", ctypes.c_int),
                ("residual", _vp), ("ldr", ctypes.c_int), ("out_bf16", ctypes.c_int), ("out", _vp),
                ("ldo", ctypes.c_int), ("ldo2", ctypes.c_int), ("stats_out", _vp), ("W_lo", _vp), ("out2", _vp), ("ln_stats", _vp), ("ln_c1", _vp),
                ("ln_nparts", ctypes.c_int), ("split_col", ctypes.c_int), ("tile_n", ctypes.c_int), ("pad4_", ctypes.c_int)]


assert ctypes.sizeof(ASegment) == 56 and ctypes.sizeof(GemmDesc) == 400


def _p(t, dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise capi.RgError("device tensor expected")
    if dtype is not None and t.dtype != dtype:
        raise capi.RgError("expected %s, got %s" % (dtype, t.dtype))
    return t.data_ptr()


class PackedWeight:
    """bf16 [ceil64(N), ceil64(K)] zero-padded GEMM operand; `lo` holds bf16(W - float(hi)) when the
    weight was packed for the precise (bf16x3) mode."""

    def __init__(self, hi, lo=None):
        self.hi, self.lo = hi, lo


def pack_weight(w, device=None, split=False):
    """fp32 [N,K] (nn.Linear layout) -> PackedWeight on device."""
    n, k = w.shape
    np_, kp = (n + 127) // 128 * 128, (k + 63) // 64 * 64  # rows: the kernel's BN = 128 tile
    dev = device or w.device
    wf = w.to(dev).float()
    hi = torch.zeros(np_, kp, dtype=torch.bfloat16, device=dev)
    hi[:n, :k] = wf.to(torch.bfloat16)
    lo = None
    if split:
        lo = torch.zeros(np_, kp, dtype=torch.bfloat16, device=dev)
        lo[:n, :k] = (wf - hi[:n, :k].float()).to(torch.bfloat16)
    return PackedWeight(hi, lo)


class Seg:
    """One fp32 K-segment of the A operand."""

    def __init__(self, src, ld=None, mode=A_IDENT, stats=None, gamma=None, beta=None, scale_shift=None,
                 col_offset=0):
        self.src, self.mode, self.stats, self.gamma, self.beta, self.ss = src, mode, stats, gamma, beta, scale_shift
        self.ld = (src.stride(-2) if src is not None else 0) if ld is None else ld
        self.col_offset = col_offset


def make_desc(*, M, N, K, W, out, A=None, segs=None, seg_len=None, bias=None, tbias=None, tb_period=0,
              residual=None, act=0, softmax_cols=0, stats_out=None, a_row_mod=0, gb_group=0, gb_stride=0,
              ldo=None, ldr=None, out2=None, ln_stats=None, ln_c1=None, lda=None, split_col=0, tile_n=0, a_styl=None):
    """a_styl: a Seg (src None; stats, gamma = gamma (1 + scale), beta = beta (1 + scale) + shift) that makes the kernel
    stylize the bf16 rows of A in LDS: operand = SiLU((A - mean) * rstd * gamma + beta)."""
    d = GemmDesc()
    d.M, d.N, d.K = M, N, K
    if A is not None:
        d.a_is_bf16 = 1
        d.A = _p(A, torch.bfloat16)
        d.lda = A.stride(-2) if lda is None else lda
        if a_styl is not None:
            e = d.seg[0]
            d.nseg, d.seg_len, e.mode = 1, K, A_STYL
            e.stats, e.nparts = _p(a_styl.stats, torch.float32), a_styl.stats.shape[-2]
            e.gamma, e.beta = _p(a_styl.gamma, torch.float32), _p(a_styl.beta, torch.float32)   # folded gain / offset
    else:
        d.a_is_bf16 = 0
        d.nseg = len(segs)
        d.seg_len = seg_len if seg_len is not None else ((K + 63) // 64 * 64 if len(segs) == 1 else 512)
        for i, s in enumerate(segs):
            e = d.seg[i]
            e.src = _p(s.src, torch.float32) + 4 * s.col_offset
            e.ld, e.mode = s.ld, s.mode
            if s.mode != A_IDENT:
                e.stats = _p(s.stats, torch.float32)
                e.nparts = s.stats.shape[-2]
                e.gamma, e.beta = _p(s.gamma, torch.float32), _p(s.beta, torch.float32)
            if s.mode == A_STYL:
                e.scale_shift = _p(s.ss, torch.float32)
    d.a_row_mod, d.gb_group, d.gb_stride = a_row_mod, gb_group, gb_stride
    d.W, d.ldw = _p(W.hi, torch.bfloat16), W.hi.stride(0)
    d.W_lo = _p(W.lo, torch.bfloat16) if (W.lo is not None and A is None) else None
    d.act, d.softmax_cols = act, softmax_cols
    d.bias = _p(bias, torch.float32)
    d.tbias, d.tb_period = _p(tbias, torch.float32), tb_period
    d.residual = _p(residual, torch.float32)
    d.ldr = (residual.stride(-2) if ldr is None else ldr) if residual is not None else 0
    d.out_bf16 = 1 if out.dtype == torch.bfloat16 else 0
    d.out = _p(out)
    d.ldo = out.stride(-2) if ldo is None else ldo
    d.stats_out = _p(stats_out, torch.float32)
    if ln_stats is not None:  # LayerNorm folded into the epilogue (A = bf16 copy of the un-normalised rows)
        d.ln_stats, d.ln_nparts, d.ln_c1 = _p(ln_stats, torch.float32), ln_stats.shape[-2], _p(ln_c1, torch.float32)
    d.split_col = split_col
    d.tile_n = tile_n
    if out2 is not None:  # bf16 copy of the output for the next GEMM's A operand
        d.out2, d.ldo2 = _p(out2, torch.bfloat16), out2.stride(-2)
    return d


def launch(h, desc, stream=None, keep=None):
    if h.recorder is not None and stream is None:
        h.recorder.add(("gemm", desc, keep))      # (keep: the tensors the descriptor points into)
        return
    s = torch.cuda.current_stream().cuda_stream if stream is None else stream
    rc = h.lib.rg_gemm(h._h, ctypes.byref(desc), ctypes.c_void_p(s))
    if rc != 0:
        raise capi.RgError("rg_gemm failed (%d): %s" % (rc, h.lib.rg_last_error(h._h).decode()))


def gemm(h, stream=None, **kw):
    launch(h, make_desc(**kw), stream, keep=kw)


def stylize(h, segs, seg_len, M, out, stream=None, m_cond=None, unc_nseg=0, unc_tab=None, qmask=None, groups=None):
    """rg_stylize: materialise the fp32->bf16 A prologue of `segs` once into out [M, nseg*seg_len] bf16.
    groups = (scale_shift_b per segment, T, nseq, split): rows are [2][nseq][T]; sequences >= split take scale_shift_b
    (two diffusion steps in one batch, rg_stylize_groups)."""
    arr = (ASegment * MAX_SEG)()
    for i, sg in enumerate(segs):
        e = arr[i]
        e.src = _p(sg.src, torch.float32) + 4 * sg.col_offset
        e.ld, e.mode = sg.ld, sg.mode
        if sg.mode != A_IDENT:
            e.stats, e.nparts = _p(sg.stats, torch.float32), sg.stats.shape[-2]
            e.gamma, e.beta = _p(sg.gamma, torch.float32), _p(sg.beta, torch.float32)
        if sg.mode == A_STYL:
            e.scale_shift = _p(sg.ss, torch.float32)
    s = torch.cuda.current_stream().cuda_stream if stream is None else stream
    fixed = (h._h, arr, len(segs), seg_len, M, ctypes.c_void_p(_p(out, torch.bfloat16)), out.stride(-2),
             M if m_cond is None else m_cond, unc_nseg,
             ctypes.c_void_p(_p(unc_tab, torch.bfloat16) if unc_tab is not None else None),
             ctypes.c_void_p(_p(qmask, torch.float32) if qmask is not None else None))
    if groups is None:
        rc = h.lib.rg_stylize(*fixed, ctypes.c_void_p(s))
    else:
        ss_b, T, nseq, split = groups
        ptrs = (ctypes.c_void_p * len(segs))(*[_p(t, torch.float32) for t in ss_b])
        rc = h.lib.rg_stylize_groups(*fixed, ptrs, T, nseq, split, ctypes.c_void_p(s))
    if rc != 0:
        raise capi.RgError("rg_stylize failed (%d): %s" % (rc, h.lib.rg_last_error(h._h).decode()))
