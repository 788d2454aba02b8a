"""Host side of the sequence-stationary denoiser forward (include/rg_gesture.h: rg_seq_forward; kernel:
csrc/rg_seq.hip): the weight / parameter / table streams of a model and the per-clip-batch cross-attention
fragments, packed in the order and MFMA-fragment layout the kernel's waves consume them.

reference: raggesture.py:1041-1085 (`forward_test` up to the CFG mix), diffusion_transformer.py:620-668,
:105-127 (`DecoderLayer`), :74-87 (`FFN`), efficient_attention.py:23-45, 62-102, stylization_block.py:29-40.

Unit GEMMs (512 x 512) per layer, in stream order:
    KV (double unit: key | value rows, LayerNorm gain folded), Q, SA_OUT, MIX_X (ca_mix columns of the un-normalised x),
    (Q3_c, MIX_c) for c = text, audio, speaker, FF1_0, FF2_0, FF1_1, FF2_1, FFN_OUT
with the joint embedding in front and the output head behind: NU = 16 L + 2 unit slots.
"""
import ctypes

import torch

from . import capi

UPL = 16
(U_KV, U_KV2, U_Q, U_SAO, U_MIXX, U_Q3_0, U_MIX_0, U_Q3_1, U_MIX_1, U_Q3_2, U_MIX_2, U_FF1_0, U_FF2_0, U_FF1_1, U_FF2_1,
 U_FFO) = range(16)
CONDS = ("xf_text", "xf_audio", "xf_spk")
_vp = ctypes.c_void_p
LANE_STRIDE = 32          # include/rg_gesture.h: RG_LANE_STRIDE (ints per lane record of the launch-form arbitration state)


class GlueArgs(ctypes.Structure):
    """include/rg_gesture.h: rg_glue_args (rg_cobatch_glue)."""
    _fields_ = [("out_c_a", _vp), ("out_u_a", _vp), ("x_a", _vp), ("out_c_b", _vp), ("out_u_b", _vp), ("x_b", _vp), ("x_b_copy", _vp),
                ("in_seq_next", _vp), ("noise_next", _vp), ("js", _vp),
                ("n_a", ctypes.c_int), ("n_b", ctypes.c_int), ("T", ctypes.c_int), ("D", ctypes.c_int), ("g_iter_next", ctypes.c_int),
                ("wc_a", ctypes.c_float), ("wu_a", ctypes.c_float), ("c_recip_a", ctypes.c_float), ("c_recipm1_a", ctypes.c_float), ("ca_a", ctypes.c_float), ("cb_a", ctypes.c_float),
                ("wc_b", ctypes.c_float), ("wu_b", ctypes.c_float), ("c_recip_b", ctypes.c_float), ("c_recipm1_b", ctypes.c_float), ("ca_b", ctypes.c_float), ("cb_b", ctypes.c_float),
                ("lr", ctypes.c_float), ("s_ab_next", ctypes.c_float), ("s_1mab_next", ctypes.c_float)]


class SeqArgs(ctypes.Structure):
    _fields_ = [("wstream", _vp), ("pstream", _vp), ("ustream", _vp), ("afrag", _vp), ("x", _vp), ("tbias", _vp),
                ("src_mask", _vp), ("qmask", _vp), ("head", _vp), ("dump", _vp), ("xbuf", _vp), ("gbuf", _vp), ("form", _vp),
                ("L", ctypes.c_int), ("B", ctypes.c_int), ("T", ctypes.c_int), ("S", ctypes.c_int),
                ("step", ctypes.c_int), ("step_b", ctypes.c_int), ("split", ctypes.c_int),
                ("dump_stage", ctypes.c_int), ("dump_layer", ctypes.c_int), ("pairs", ctypes.c_int),
                ("glue_ctr", _vp), ("glue", GlueArgs)]


def supported(cfg, T, precision):
    """Shapes rg_seq_forward is specialised for (everything the reference configuration uses)."""
    return (precision == "bf16" and cfg["latent_dim"] == 512 and cfg["num_heads"] == 16 and cfg["ff_size"] == 1024
            and T <= 48 and 1 <= cfg["num_layers"] <= 8)


def pack_unit(W):
    """fp32 [512 n, 512 k] -> bf16 [8 waves][64 fragments][64 lanes][8]: fragment (s, j) of wave w, lane (g, nl), element e =
    W[64 w + 16 j + nl][32 s + 8 g + e] (fragment index 4 s + j, lane 16 g + nl)."""
    capi.require(W.shape == (512, 512), "unsupported argument: requires W.shape == (512, 512)")
    v = W.to(torch.bfloat16).view(8, 4, 16, 16, 4, 8)          # w, j, nl, s, g, e
    return v.permute(0, 3, 1, 4, 2, 5).contiguous().view(8, 64, 64, 8)


def pack_kv(Wk, Wv):
    """Key / value weights of the self attention as ONE double unit [8 waves][128 fragments][64][8]: per wave the fragments of
    head 2 w (key: 16 steps x 2 blocks, then value), then of head 2 w + 1."""
    kv = torch.stack([Wk, Wv]).to(torch.bfloat16).view(2, 8, 2, 2, 16, 16, 4, 8)   # kv, w, h, j2, nl, s, g, e
    return kv.permute(1, 2, 0, 5, 3, 6, 4, 7).contiguous().view(8, 128, 64, 8)


def a_fragments(a_pre):
    """fp32 A [..., H = 16, 32 i, 32 j] (softmax_N(K)^T V per head) -> bf16 MFMA A-operand fragments
    [..., 8 waves, 2 heads, 2 column blocks, 64 lanes, 8]: element e of lane (jj, g) = A[i][16 jb + jj],
    i = 4 g + e for e < 4, 16 + 4 g + e - 4 otherwise (the order in which the kernel's query accumulators enumerate i)."""
    dev = a_pre.device
    g = torch.arange(4, device=dev).view(4, 1)
    e = torch.arange(8, device=dev).view(1, 8)
    idx = torch.where(e < 4, 4 * g + e, 16 + 4 * g + e - 4).reshape(-1)            # [32] = (g, e)
    lead = a_pre.shape[:-3]
    A = a_pre.index_select(-2, idx).view(*lead, 16, 4, 8, 2, 16)                    # ..., h, g, e, jb, jj
    n = len(lead)
    A = A.permute(*range(n), n, n + 3, n + 1, n + 4, n + 2).contiguous()             # ..., h, jb, g, jj, e
    # (until late in round 5 a low-order bf16 half rode along for an hi + lo product that no build used: a quarter of the
    #  fragments a conditional sequence took off its ring per cross-attention were discarded)
    return A.to(torch.bfloat16).view(*lead, 8, 2, 2, 64, 8).contiguous()


class SeqStreams:
    """Device-resident streams of one model: wstream bf16 [NU][8][64][512], pstream fp32 [S][NU][8][4][64],
    ustream fp32 [S][L][8][8][64]."""

    def __init__(self, g, ss, layers_extra, cfg, S, dev):
        """g(name) -> fp32 CPU tensor of the reference state dict; ss [S, L, 5, 2 D] AdaLN (scale | shift) table;
        layers_extra[l] = dict(w_mix fp32 [512, 2048] fused, b_mix, unc_tab bf16 [S, 2, 1536])."""
        L, D = cfg["num_layers"], cfg["latent_dim"]
        self.L, self.S, self.NU = L, S, UPL * L + 2
        NU = self.NU
        f = lambda t: t.to(dev, torch.float32)
        W = torch.zeros(NU, 8, 64, 64, 8, device=dev, dtype=torch.bfloat16)
        P = torch.zeros(S, NU, 4, D, device=dev, dtype=torch.float32)
        U = torch.zeros(S, L, 8, D, device=dev, dtype=torch.float32)
        W[0] = pack_unit(f(g("joint_embed.weight")))
        P[:, 0, 0] = f(g("joint_embed.bias"))
        W[NU - 1] = pack_unit(f(g("out.weight")))
        P[:, NU - 1, 0] = f(g("out.bias"))
        ssd = ss.to(dev, torch.float32)

        def fold(wn, bn, gn, ben):     # LN(x) W^T + b = xhat (W diag(gamma))^T + (b + W beta)
            w, b, ga, be = (f(g(n)).double() for n in (wn, bn, gn, ben))
            return (w * ga[None, :]).float(), (b + w @ be).float()

        def styl(l, bi, gn, bn):       # gain = gamma (1 + scale), offset = beta (1 + scale) + shift, per step
            sc1 = 1.0 + ssd[:, l, bi, :D]
            return f(g(gn))[None] * sc1, f(g(bn))[None] * sc1 + ssd[:, l, bi, D:]

        for l in range(L):
            p = "temporal_decoder_blocks.%d." % l
            u0 = 1 + UPL * l
            sa = p + "sa_block."
            wq, bq = fold(sa + "query.weight", sa + "query.bias", sa + "norm.weight", sa + "norm.bias")
            wk, bk = fold(sa + "key.weight", sa + "key.bias", sa + "norm.weight", sa + "norm.bias")
            wv, bv = fold(sa + "value.weight", sa + "value.bias", sa + "norm.weight", sa + "norm.bias")
            W[u0 + U_KV:u0 + U_KV + 2] = pack_kv(wk, wv).view(2, 8, 64, 64, 8)      # [8 waves][128] over two slots
            # (the view above re-slices [8][128] as [2][8][64]: the kernel addresses the double unit as 8 x 128 fragments
            #  from the first slot's base, so the flat byte order is what counts)
            P[:, u0 + U_KV, 0], P[:, u0 + U_KV, 1] = bk, bv
            W[u0 + U_Q] = pack_unit(wq)
            P[:, u0 + U_Q, 0] = bq
            W[u0 + U_SAO] = pack_unit(f(g(sa + "proj_out.out_layers.2.weight")))
            P[:, u0 + U_SAO, 0] = f(g(sa + "proj_out.out_layers.2.bias"))
            P[:, u0 + U_SAO, 1], P[:, u0 + U_SAO, 2] = styl(l, 0, sa + "proj_out.norm.weight", sa + "proj_out.norm.bias")
            ex = layers_extra[l]
            wm = f(ex["w_mix"])                                                      # [512, 2048]
            wx = wm[:, 3 * D:]
            W[u0 + U_MIXX] = pack_unit(wx)
            P[:, u0 + U_MIXX, 0] = f(ex["b_mix"])
            P[:, u0 + U_MIXX, 1] = wx.to(torch.bfloat16).double().sum(1).float()
            tab = ex["unc_tab"].to(dev).double()                                     # [S, 2, 3 D]
            for c, cn in enumerate(CONDS):
                ca = p + "ca_blocks.%s." % cn
                w3, b3 = fold(ca + "query.weight", ca + "query.bias", ca + "norm.weight", ca + "norm.bias")
                W[u0 + U_Q3_0 + 2 * c] = pack_unit(w3)
                P[:, u0 + U_Q3_0 + 2 * c, 0] = b3
                wc = wm[:, c * D:(c + 1) * D]
                W[u0 + U_MIX_0 + 2 * c] = pack_unit(wc)
                P[:, u0 + U_MIX_0 + 2 * c, 1], P[:, u0 + U_MIX_0 + 2 * c, 2] = styl(
                    l, 1 + c, ca + "proj_out.norm.weight", ca + "proj_out.norm.bias")
                # classifier-free rows: W_c h_c with h_c one of the two tabulated rows
                wcb = wc.to(torch.bfloat16).double()
                U[:, l, 2 * c:2 * c + 2] = (tab[:, :, c * D:(c + 1) * D] @ wcb.T).float()
            w1, b1 = f(g(p + "ffn.linear1.weight")), f(g(p + "ffn.linear1.bias"))
            w2 = f(g(p + "ffn.linear2.weight"))
            for j in range(2):
                W[u0 + U_FF1_0 + 2 * j] = pack_unit(w1[j * D:(j + 1) * D])
                P[:, u0 + U_FF1_0 + 2 * j, 0] = b1[j * D:(j + 1) * D]
                W[u0 + U_FF2_0 + 2 * j] = pack_unit(w2[:, j * D:(j + 1) * D].contiguous())
            P[:, u0 + U_FF2_0, 0] = f(g(p + "ffn.linear2.bias"))
            W[u0 + U_FFO] = pack_unit(f(g(p + "ffn.proj_out.out_layers.2.weight")))
            P[:, u0 + U_FFO, 0] = f(g(p + "ffn.proj_out.out_layers.2.bias"))
            P[:, u0 + U_FFO, 1], P[:, u0 + U_FFO, 2] = styl(l, 4, p + "ffn.proj_out.norm.weight", p + "ffn.proj_out.norm.bias")
        self.wstream = W
        # per wave: [4 vectors][64 features]
        self.pstream = P.view(S, NU, 4, 8, 64).permute(0, 1, 3, 2, 4).contiguous()
        self.ustream = U.view(S, L, 8, 8, 64).permute(0, 1, 3, 2, 4).contiguous()   # [S][L][wave][8 slots][64]


class SeqForward:
    """Buffers of one DenoiserSession for rg_seq_forward."""

    def __init__(self, sess, pairs=False, duo=True, lane_dyn=None):
        """duo: two sequences of the same kind per workgroup (rg_seq2_forward: every streamed weight fragment feeds both; the
        fp32 residual stream and two bf16 panel images take round trips through scratch buffers in L2) instead of one
        (rg_seq_forward) -- same bits;
        pairs: the classifier-free sequences run behind the conditional ones in the SAME workgroups (half as many workgroups,
        ~1.6x as long) instead of in workgroups of their own;
        lane_dyn = (state int32 [n, LANE_STRIDE] on the device, lane, n, budget): the session belongs to lane `lane` of a pipeline whose
        lanes share `state` (include/rg_gesture.h: rg_lane_form).  Every forward first publishes the workgroups it will hold;
        a `duo` session then launches rg_seqx_forward, which runs this narrow form or -- when the other lanes leave room for
        2 B workgroups -- one workgroup per sequence (0.6 of the time per launch: what counts while the pipeline fills or
        drains), decided on the device when the launch starts.  `chain_end()` marks the lane idle."""
        w = sess.w
        self.sess, self.h, self.st = sess, sess.h, w.seq_streams
        B, dev = sess.B, w.dev
        self.duo = bool(duo)
        self.xbuf = self.gbuf = None
        if self.duo:
            nwg = B + 2          # at most ceil(split / 2) + ceil((B - split) / 2) pairs per kind, two kinds
            self.xbuf = torch.empty(nwg * 2 * 8 * 12 * 64 * 4, device=dev, dtype=torch.float32)
            self.gbuf = torch.empty(nwg * 8 * 48 * 1024, device=dev, dtype=torch.uint8)
        self.afrag = torch.zeros(w.L, 3, B, 8, 2, 2, 64, 8, device=dev, dtype=torch.bfloat16)
        self.glue_ctr = torch.zeros(B, device=dev, dtype=torch.int32)       # arrival counters of the forwards' tails (run(glue=...))
        a = self.args = SeqArgs()
        p = lambda t: t.data_ptr()
        a.wstream, a.pstream, a.ustream, a.afrag = p(self.st.wstream), p(self.st.pstream), p(self.st.ustream), p(self.afrag)
        a.tbias, a.src_mask, a.qmask, a.head = p(w.tbias), p(sess.src_mask), p(sess.qmask), p(sess.head)
        a.L, a.B, a.T, a.S = w.L, B, w.T, self.st.S
        a.pairs = int(bool(pairs))
        a.dump, a.dump_stage, a.dump_layer = None, 0, 0
        a.xbuf = p(self.xbuf) if self.xbuf is not None else None
        a.gbuf = p(self.gbuf) if self.gbuf is not None else None
        self._fn = self.h.lib.rg_seq2_forward if self.duo else self.h.lib.rg_seq_forward
        self.lane_dyn = None
        if lane_dyn is not None:
            state, lane, n, budget = lane_dyn
            if not (state.is_cuda and state.dtype == torch.int32 and state.is_contiguous() and state.numel() >= LANE_STRIDE * n and 0 <= lane < n):
                raise capi.RgError("lane_dyn: state must be a contiguous device int32 tensor [n, %d], 0 <= lane < n" % LANE_STRIDE)
            self.lane_dyn = (state, int(lane), int(n), int(budget))
            a.form = state.data_ptr() + 4 * (LANE_STRIDE * lane + 1)
            if self.duo:
                self._fn = self.h.lib.rg_seqx_forward

    def set_a(self, a_pre, o0, o1):
        """a_pre fp32 [L, 3, n, H, 32, 32] of the clips [o0, o1) of the session."""
        self.afrag[:, :, o0:o1] = a_fragments(a_pre)

    def run(self, x, step, step_b=None, split=None, dump=None, dump_stage=0, dump_layer=0, glue=None):
        """glue: a GlueArgs (n_a + n_b == B clips; n_b may be 0) -- the forward ends with the loop step's update of x that
        rg_cobatch_glue would do in a launch of its own (include/rg_gesture.h: rg_seq_args.glue_ctr; csrc/rg_tail.h)."""
        a = self.args
        if glue is not None:
            if glue.n_a + glue.n_b != self.sess.B or glue.T != self.sess.w.T or glue.D != self.sess.w.D or dump_stage:
                raise capi.RgError("rg_seq_forward: the tail's groups must cover the session's clips (n_a + n_b == B)")
            a.glue, a.glue_ctr = glue, self.glue_ctr.data_ptr()
        else:
            a.glue_ctr = None
        if not (x.is_contiguous() and x.dtype == torch.float32 and x.numel() == self.sess.B * self.sess.w.T * self.sess.w.D):
            raise capi.RgError("rg_seq_forward: x must be a contiguous fp32 [B, T, D] tensor")
        a.x, a.step = x.data_ptr(), int(step)
        a.step_b = int(step if step_b is None else step_b)
        a.split = int(self.sess.B if split is None else split)
        a.dump = dump.data_ptr() if dump is not None else None
        a.dump_stage, a.dump_layer = int(dump_stage), int(dump_layer)
        if self.lane_dyn is not None and not dump_stage:
            state, lane, n, budget = self.lane_dyn
            B = self.sess.B
            wide = 2 * B
            if self.duo:
                sp = max(0, min(B, a.split))
                npc = (sp + 1) // 2 + (B - sp + 1) // 2
                narrow = npc if a.pairs else 2 * npc
            else:
                narrow = wide = B if a.pairs else 2 * B
            self.h.call("lane_form", state, lane, n, narrow, wide, budget)
        s = torch.cuda.current_stream().cuda_stream
        fn = self._fn
        if dump_stage and self.lane_dyn is not None and self.duo:      # diagnostics: the fixed two-sequence form
            fn = self.h.lib.rg_seq2_forward
        rc = fn(self.h._h, ctypes.byref(a), ctypes.c_void_p(s))
        if rc != 0:
            raise capi.RgError("rg_seq_forward failed (%d): %s" % (rc, self.h.lib.rg_last_error(self.h._h).decode()))
        return self.sess.head

    def chain_end(self):
        """The lane's chain of forwards is over (sampler loops call this): it holds no workgroups."""
        if self.lane_dyn is not None:
            state, lane, n, budget = self.lane_dyn
            self.h.call("lane_form", state, lane, n, 0, 0, budget)
