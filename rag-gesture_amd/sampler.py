"""DDIM sampling / inversion / insertion-guided loops on top of DenoiserSession.

Mirrors the reference's samplers for the inference configuration (START_X prediction, eta = 0,
no clipping): mogen/models/utils/gaussian_diffusion.py
  :1042-1135 ddim_sample_loop(_progressive)   -> ddim_sample_loop
  :1137-1230 ddim_reverse_sample_loop          -> ddim_reverse_sample_loop
  :1233-1395 ddim_guided_sample_loop           -> ddim_guided_sample_loop
  :910-1001  ddim_sample (in_seq replacement + DDIM update)
Every step is a fixed sequence of kernel launches with no host read-back (CFG weights and
DDIM coefficients come from 50-entry host tables), so a whole loop can be captured into one
graph (`GraphedLoop`).  Noise is explicit: the caller supplies the tensors the reference would
have drawn from torch's global generator.
"""
import ctypes

import torch

from . import capi

_vp, _f, _i = ctypes.c_void_p, ctypes.c_float, ctypes.c_int


SPLICE_MAX = 64


class SpliceTable(ctypes.Structure):
    """include/rg_gesture.h: rg_splice_table (rg_splice_many)."""
    _fields_ = [("n", _i)] + [(k, _i * SPLICE_MAX) for k in ("e", "b", "r0", "q0", "nrows")]


from .seqfwd import GlueArgs      # (rg_glue_args: also the tail of rg_seq_args)


TAIL_GLUE = True     # the loops below end every forward with the step's update (rg_seq_args.glue_ctr, csrc/rg_tail.h) where they can


def _tail_ok(sess, x):
    """The loop step's update can ride at the end of the forward: sequence-stationary engine, x updated in place."""
    return (TAIL_GLUE and getattr(sess, "tail_glue", True) and getattr(sess, "sq", None) is not None and sess.h.recorder is None and x.is_cuda and x.is_contiguous()
            and x.dtype == torch.float32 and x.shape[0] == sess.B)


def _p(t):
    if t is None:
        return None
    if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
        raise capi.RgError("rg_cobatch_glue: contiguous fp32 device tensors expected")
    return t.data_ptr()


def _glue_sampling(sess, x, i, nxt, noise_nxt, g_next, lr):
    """rg_glue_args of a sampling step at respaced index i over ALL clips of the session (its group a): this step's CFG + DDIM
    update of x, then the NEXT step's guidance update (g_next iterations) and in-sequence replacement on the rows nxt marks."""
    sch, w, B, T = sess.w.schedule, sess.w, sess.B, sess.w.T
    a = GlueArgs()
    a.out_c_a, a.out_u_a, a.x_a, a.js = _p(sess.head), _p(sess.head[B * T:]), _p(x), _p(w.js)
    a.n_a, a.n_b, a.T, a.D = B, 0, T, w.D
    a.wc_a, a.wu_a = sch.cfg_weights(w.cfg["scale_func_cfg"], i)
    a.c_recip_a, a.c_recipm1_a, a.ca_a, a.cb_a = float(sch.c_recip[i]), float(sch.c_recipm1[i]), float(sch.c_prev_a[i]), float(sch.c_prev_b[i])
    a.in_seq_next, a.noise_next = _p(nxt), (None if nxt is None else _p(noise_nxt))
    a.g_iter_next, a.lr = (int(g_next) if nxt is not None else 0), float(lr)
    if i > 0:
        a.s_ab_next, a.s_1mab_next = float(sch.s_ab[i - 1]), float(sch.s_1mab[i - 1])
    return a


def _step(sess, x, i, in_seq=None, noise=None):
    """One ddim_sample call at respaced index i, in place on x [B,T,D]."""
    sch, w, h = sess.w.schedule, sess.w, sess.h
    if in_seq is not None:
        h.call("inseq_replace", x, in_seq, noise, sess.B * w.T, w.D, float(sch.s_ab[i]), float(sch.s_1mab[i]))
    sess.forward(x, i)
    sess.cfg_ddim(x, x, i, sch.c_prev_a[i], sch.c_prev_b[i])


def ddim_sample_loop(sess, x, in_seq=None, inseq_noise=None):
    """x: start noise [B,T,D] (updated in place and returned).  in_seq [B,T,D] or None;
    inseq_noise [S,B,T,D] = the randn_like(in_seq) draws, indexed by step."""
    sch, w = sess.w.schedule, sess.w
    S = sch.num_timesteps
    if _tail_ok(sess, x):        # one launch per step: the first step's insertion in front, every later one in the tail before it
        if in_seq is not None:
            sess.h.call("inseq_replace", x, in_seq, inseq_noise[S - 1], sess.B * w.T, w.D, float(sch.s_ab[S - 1]), float(sch.s_1mab[S - 1]))
        for i in range(S - 1, -1, -1):
            nxt = in_seq if i > 0 else None
            sess.forward(x, i, glue=_glue_sampling(sess, x, i, nxt, None if nxt is None else inseq_noise[i - 1], 0, 0.0))
        sess.chain_end()
        return x
    for i in range(S - 1, -1, -1):
        _step(sess, x, i, in_seq, None if in_seq is None else inseq_noise[i])
    sess.chain_end()
    return x


def p_sample_loop(sess, x, noise):
    """Ancestral sampling (gaussian_diffusion.py:805-905 `p_sample_loop`, fixed_large variance): x = start noise
    [B,T,D] (updated in place and returned), noise [S,B,T,D] = the randn_like(x) draw of every step."""
    S = sess.w.schedule.num_timesteps
    for i in range(S - 1, -1, -1):
        sess.forward(x, i)
        sess.cfg_ddpm(x, x, i, noise[i])
    sess.chain_end()
    return x


def ddim_reverse_sample_loop(sess, x, out):
    """DDIM inversion of x [B,T,D] (clean -> noise); out [S,B,T,D] receives every level
    (out[k] = latent at alphas_cumprod_next[k]), x is updated in place to out[S-1]."""
    sch, w = sess.w.schedule, sess.w
    if _tail_ok(sess, x) and out.is_contiguous() and out.dtype == torch.float32:
        B, T = sess.B, w.T           # one launch per step: x advances in place, the forward's tail also writes the level kept
        for i in range(sch.num_timesteps):
            a = GlueArgs()
            a.out_c_b, a.out_u_b, a.x_b, a.x_b_copy, a.js = _p(sess.head), _p(sess.head[B * T:]), _p(x), _p(out[i]), _p(w.js)
            a.n_a, a.n_b, a.T, a.D = 0, B, T, w.D
            a.wc_b, a.wu_b = sch.cfg_weights(w.cfg["scale_func_cfg"], i)
            a.c_recip_b, a.c_recipm1_b, a.ca_b, a.cb_b = float(sch.c_recip[i]), float(sch.c_recipm1[i]), float(sch.c_next_a[i]), float(sch.c_next_b[i])
            sess.forward(x, i, glue=a)
        sess.chain_end()
        return out
    cur = x
    for i in range(sch.num_timesteps):
        sess.forward(cur, i)
        sess.cfg_ddim(cur, out[i], i, sch.c_next_a[i], sch.c_next_b[i])   # the update writes level i in place of a copy
        cur = out[i]
    sess.chain_end()
    x.copy_(cur)
    return out


def ddim_guided_sample_loop(sess, x, inverted, guidance_iters, guidance_lr, inseq_noise, in_seq=None):
    """Insertion-guided sampling.  inverted [S,B,T,D] (zero rows = not guided); on every step but
    the first the reference sets in_seq = inverted[i], runs g_iter gradient steps on
    mse(x*mask, in_seq) and then ddim_sample re-inserts q_sample(in_seq) on the masked rows."""
    sch, w, h = sess.w.schedule, sess.w, sess.h
    S = sch.num_timesteps
    capi.require(len(guidance_iters) == S == inverted.shape[0],
            "unsupported argument: requires len(guidance_iters) == S == inverted.shape[0]")
    if _tail_ok(sess, x):        # one launch per step (as ddim_sample_loop; the next step's guidance update rides along)
        if in_seq is not None:
            h.call("inseq_replace", x, in_seq, inseq_noise[S - 1], sess.B * w.T, w.D, float(sch.s_ab[S - 1]), float(sch.s_1mab[S - 1]))
        for i in range(S - 1, -1, -1):
            nxt = inverted[i - 1] if i > 0 else None
            sess.forward(x, i, glue=_glue_sampling(sess, x, i, nxt, None if nxt is None else inseq_noise[i - 1],
                                                   guidance_iters[i - 1] if i > 0 else 0, guidance_lr))
        sess.chain_end()
        return x
    for i in range(S - 1, -1, -1):
        if i != S - 1:
            in_seq = inverted[i]
            h.call("guidance_update", x, in_seq, sess.B * w.T, w.D, int(guidance_iters[i]), float(guidance_lr))
        _step(sess, x, i, in_seq, None if in_seq is None else inseq_noise[i])
    sess.chain_end()
    return x


def cobatched_loop(sess, x_all, n_a, out_b, inverted_a=None, guidance_iters=None, guidance_lr=0.1, inseq_noise_a=None,
                   in_seq_a=None, fused_glue=True, tail_glue=True):
    """Two loops advancing in the same launches, one denoiser forward per step for both:
      clips [0, n_a) of the session: the (insertion-guided) DDIM sampling loop of one batch, exactly
        ddim_guided_sample_loop / ddim_sample_loop (inverted_a None) on x_all[:n_a], in place;
      clips [n_a, B): the DDIM inversion of the NEXT batch's exemplars, exactly ddim_reverse_sample_loop on x_all[n_a:]:
        out_b [S, B - n_a, T, D] receives every level.
    Step k runs the sampling at respaced index S-1-k and the inversion at index k.  Rows of a batch never mix in any
    kernel, so each group gets what its own loop would give; the point is the launch count and size: a forward over
    M = 2 (8 + 24) 43 rows costs ~1.1x the forward over the 24 exemplars alone (NOTEBOOK 6b; with the seq engine: one workgroup per sequence, the launch time does not depend on the count up to 256)."""
    sch, w, h = sess.w.schedule, sess.w, sess.h
    S, B, T, D = sch.num_timesteps, sess.B, w.T, w.D
    n_b = B - n_a
    xa, xb = x_all[:n_a], x_all[n_a:]
    in_seq = in_seq_a
    if fused_glue and h.recorder is None:
        # what lies between two forwards as ONE launch (rg_cobatch_glue: this step's two updates + the next step's guidance and
        # insertion on the sampling rows) instead of four; the first step's insertion in front of the loop as before
        if in_seq is not None:
            h.call("inseq_replace", xa, in_seq, inseq_noise_a[S - 1], n_a * T, D, float(sch.s_ab[S - 1]), float(sch.s_1mab[S - 1]))
        head = sess.head

        def p(t):
            if t is None:
                return None
            if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                raise capi.RgError("rg_cobatch_glue: contiguous fp32 device tensors expected")
            return t.data_ptr()
        cfgw = lambda step: sch.cfg_weights(w.cfg["scale_func_cfg"], step)
        # tail_glue: the forward's own workgroups do that launch's work as they finish (the one that ends a clip's second
        # sequence updates the clip: rg_seq_args.glue_ctr, csrc/rg_tail.h) -- a loop step is ONE launch
        tail = bool(tail_glue) and _tail_ok(sess, x_all) and 0 < n_a < B
        for k in range(S):
            i = S - 1 - k
            if not tail:
                sess.forward(x_all, i, step_b=k, split=n_a)
            a = GlueArgs()
            a.out_c_a, a.out_u_a, a.x_a = p(head), p(head[B * T:]), p(xa)
            a.out_c_b, a.out_u_b, a.x_b, a.x_b_copy = p(head[n_a * T:]), p(head[(B + n_a) * T:]), p(xb), p(out_b[k])
            nxt = None
            if i > 0:
                nxt = inverted_a[i - 1] if inverted_a is not None else in_seq_a
            a.in_seq_next, a.noise_next = p(nxt), (None if nxt is None else p(inseq_noise_a[i - 1]))
            a.js = p(w.js)
            a.n_a, a.n_b, a.T, a.D = n_a, n_b, T, D
            a.g_iter_next = int(guidance_iters[i - 1]) if (nxt is not None and inverted_a is not None) else 0
            a.wc_a, a.wu_a = cfgw(i)
            a.c_recip_a, a.c_recipm1_a, a.ca_a, a.cb_a = float(sch.c_recip[i]), float(sch.c_recipm1[i]), float(sch.c_prev_a[i]), float(sch.c_prev_b[i])
            a.wc_b, a.wu_b = cfgw(k)
            a.c_recip_b, a.c_recipm1_b, a.ca_b, a.cb_b = float(sch.c_recip[k]), float(sch.c_recipm1[k]), float(sch.c_next_a[k]), float(sch.c_next_b[k])
            a.lr = float(guidance_lr)
            if i > 0:
                a.s_ab_next, a.s_1mab_next = float(sch.s_ab[i - 1]), float(sch.s_1mab[i - 1])
            if tail:
                sess.forward(x_all, i, step_b=k, split=n_a, glue=a)
                continue
            rc = h.lib.rg_cobatch_glue(h._h, ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            if rc != 0:
                raise capi.RgError("rg_cobatch_glue failed (%d): %s" % (rc, h.lib.rg_last_error(h._h).decode()))
        sess.chain_end()
        return x_all, out_b
    for k in range(S):
        i = S - 1 - k
        if inverted_a is not None and i != S - 1:
            in_seq = inverted_a[i]
            h.call("guidance_update", xa, in_seq, n_a * T, D, int(guidance_iters[i]), float(guidance_lr))
        if in_seq is not None:
            h.call("inseq_replace", xa, in_seq, inseq_noise_a[i], n_a * T, D, float(sch.s_ab[i]), float(sch.s_1mab[i]))
        sess.forward(x_all, i, step_b=k, split=n_a)
        sess.cfg_ddim_rows(0, n_a, xa, xa, i, sch.c_prev_a[i], sch.c_prev_b[i])
        sess.cfg_ddim_rows(n_a, n_b, xb, xb, k, sch.c_next_a[k], sch.c_next_b[k], x_out2=out_b[k])
    sess.chain_end()
    return x_all, out_b


class GraphedLoop:
    """Capture fn() (a fixed launch sequence over static buffers) into a HIP graph and replay it.
    torch.cuda.CUDAGraph is used purely as the capture/replay plumbing."""

    def __init__(self, fn, warmup=True):
        if warmup:  # first call outside capture: lazy module loads, allocator warm-up
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fn()
            torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with capi.capture(self.graph):
            fn()

    def replay(self):
        self.graph.replay()
