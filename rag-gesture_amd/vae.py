"""Host side of the four body-part TransformerVAEs and the GestureRepEncoder wrapper.

Mirrors (same names, argument meaning):
  mogen/models/transformers/gesture_vae.py:25-239   TransformerVAE (encode / reparameterize / decode)
  mogen/models/utils/detr_utils.py:101-210, 335-480 Skip transformer stacks and layers
  mogen/models/transformers/diffusion_transformer.py:131-330 GestureRepEncoder (6D packing, 4 parts,
      separator tokens, output splitting)
All arithmetic runs in the HIP extension: linears through rg_gemm (bf16 MFMA, or bf16x3 in
precision="fp32"), attention/LayerNorm/rotations through the rg_vae kernels.  Token rows are
batch-major [B*S, D] (the reference is sequence-major; attention is per sequence either way).
The VAE hyper-parameters come from the YAML shipped with each checkpoint, so
nothing here is specialised to one width/depth/arch.
"""

import torch

from . import capi, gemm as G, vencfwd

PARTS = ("upper", "hands", "face", "lowertrans")  # reference encode / RNG order
ACT = {"relu": 2, "gelu": 1}


def _num_blocks(n):
    if n % 2 == 0:
        n += 1
    return (n - 1) // 2


class _Lin:
    def __init__(self, sd, name, dev, split, rows=None):
        w = sd[name + ".weight"] if rows is None else sd[name + "_weight"][rows]
        b = sd[name + ".bias"] if rows is None else sd[name + "_bias"][rows]
        self.n, self.k = w.shape
        self.w = G.pack_weight(w.detach().float(), dev, split=split)
        self.b = b.detach().float().to(dev).contiguous()


class _Block:
    def __init__(self, sd, name, dev, split, cross):
        D = sd[name + ".norm1.weight"].shape[0]
        f = lambda k: sd[name + k].detach().float().to(dev).contiguous()
        self.qkv = _Lin(sd, name + ".self_attn.in_proj", dev, split, rows=slice(0, 3 * D))
        self.qk = _Lin(sd, name + ".self_attn.in_proj", dev, split, rows=slice(0, 2 * D))
        self.v = _Lin(sd, name + ".self_attn.in_proj", dev, split, rows=slice(2 * D, 3 * D))
        self.out = _Lin(sd, name + ".self_attn.out_proj", dev, split)
        self.l1, self.l2 = _Lin(sd, name + ".linear1", dev, split), _Lin(sd, name + ".linear2", dev, split)
        self.n1 = (f(".norm1.weight"), f(".norm1.bias"))
        self.n2 = (f(".norm2.weight"), f(".norm2.bias"))
        if cross:
            self.cq = _Lin(sd, name + ".multihead_attn.in_proj", dev, split, rows=slice(0, D))
            self.ckv = _Lin(sd, name + ".multihead_attn.in_proj", dev, split, rows=slice(D, 3 * D))
            self.cout = _Lin(sd, name + ".multihead_attn.out_proj", dev, split)
            self.n3 = (f(".norm3.weight"), f(".norm3.bias"))


class _Stack:
    def __init__(self, sd, name, dev, split, num_layers, cross):
        nb = _num_blocks(num_layers)
        self.inp = [_Block(sd, "%s.input_blocks.%d" % (name, i), dev, split, cross) for i in range(nb)]
        self.mid = _Block(sd, name + ".middle_block", dev, split, cross)
        self.outb = [_Block(sd, "%s.output_blocks.%d" % (name, i), dev, split, cross) for i in range(nb)]
        self.lin = [_Lin(sd, "%s.linear_blocks.%d" % (name, i), dev, split) for i in range(nb)]
        self.norm = (sd[name + ".norm.weight"].float().to(dev).contiguous(), sd[name + ".norm.bias"].float().to(dev).contiguous())


class TransformerVAE:
    """One body part.  `vcfg` = the YAML dict (latent_dim, num_heads, ff_size, num_layers,
    decoder_arch, position_embedding, nfeats, num_frames, frame_chunk_size,
    transformer_activation, transformer_normalize_before, vae_dist)."""

    def __init__(self, state, vcfg, device="cuda", precision="bf16", chain=True, fused_encoder=True, fused_decoder=True):
        """chain: bf16 path only -- producers hand bf16 copies to the GEMMs that consume them (False: every GEMM converts
        its fp32 operand tiles itself; a measurement knob).
        fused_encoder: run the encoder stack as ONE launch (rg_venc_forward: two chunk sequences per workgroup, activations
        resident on chip, weights streamed) where the shape supports it (vencfwd.supported); False: the per-op launch chain.
        fused_decoder: the all_encoder decoder stack as one launch per block (rg_vdec_step: four 40-row tiles per 160-token
        sequence, keys / values exchanged through L2 between the launches) where vencfwd.decoder_supported; False: the chain."""
        capi.require(vcfg.get("vae_dist", "normal") == "normal", "only the Normal posterior is supported")
        self.cfg = vcfg
        self.dev = torch.device(device)
        self.h = capi.get_handle(self.dev.index if self.dev.index is not None else torch.cuda.current_device())
        self.precision = precision
        # bf16 path: every op that feeds a GEMM hands over a bf16 copy (LayerNorm, attention, FF1) so the GEMMs
        # read bf16 A operands instead of converting fp32 tiles in each of their column tiles
        self.chain = precision == "bf16" and bool(chain)
        self._copies = None
        split = precision == "fp32"
        D = self.D = vcfg["latent_dim"]
        self.nfeats, self.chunk, self.frames = vcfg["nfeats"], vcfg["frame_chunk_size"], vcfg["num_frames"]
        self.heads, self.pre = vcfg["num_heads"], bool(vcfg["transformer_normalize_before"])
        self.act = ACT[vcfg["transformer_activation"]]
        self.arch = vcfg["decoder_arch"]
        dev = self.dev
        f = lambda k: state[k].detach().float().to(dev).contiguous()
        self.embed, self.final = _Lin(state, "skel_embedding", dev, split), _Lin(state, "final_layer", dev, split)
        self.pe_enc, self.pe_dec, self.pe_mem = (f(n + ".pe")[:, 0].contiguous() for n in
                                                 ("query_pos_encoder", "query_pos_decoder", "mem_pos_decoder"))
        self.tok_pe = (f("global_motion_token") + self.pe_enc[:2]).contiguous()  # host-side constant fold
        self.encoder = _Stack(state, "encoder", dev, split, vcfg["num_layers"], cross=False)
        self.venc = None
        if fused_encoder and vencfwd.supported(vcfg, precision):
            self.venc = vencfwd.VencForward(self.h, vencfwd.VencStreams(state, "encoder", vcfg["num_layers"], self.heads, dev))
        self.vdec, self._fused_decoder = None, bool(fused_decoder)
        if self.arch == "all_encoder":
            self.decoder = _Stack(state, "decoder", dev, split, vcfg["num_layers"], cross=False)
            self.dec_heads = self.heads * 8
            if fused_decoder and vencfwd.decoder_supported(vcfg, precision, vcfg["num_frames"] // vcfg["frame_chunk_size"]):
                self.vdec = vencfwd.VdecForward(self.h, vencfwd.VencStreams(state, "decoder", vcfg["num_layers"], self.dec_heads, dev))
        elif self.arch == "encoder_decoder":
            self.decoder = _Stack(state, "decoder", dev, split, (vcfg["num_layers"] - 1) * 4 + 1, cross=True)
            self.dec_heads = self.heads * 4
        else:
            raise ValueError("Not support architecture!")

    # ---------------------------------------------------------------- building blocks
    def _lin(self, lin, x, M, out=None, residual=None, act=0, tbias=None, tb_period=0, segs=None, bf16_out=False):
        """x: fp32 [M,K] (converted tile by tile in the GEMM prologue) or bf16 [M,K] (used as is)."""
        if out is None:
            out = torch.empty(M, lin.n, device=self.dev, dtype=torch.bfloat16 if bf16_out else torch.float32)
        if x is not None and x.dtype == torch.bfloat16:
            G.gemm(self.h, M=M, N=lin.n, K=lin.k, W=lin.w, out=out, A=x, bias=lin.b, residual=residual, act=act,
                   tbias=tbias, tb_period=tb_period)
            return out
        segs = segs or [G.Seg(x)]
        seg_len = None if len(segs) == 1 else segs[0].src.shape[-1]
        G.gemm(self.h, M=M, N=lin.n, K=lin.k, W=lin.w, out=out, segs=segs, seg_len=seg_len, bias=lin.b,
               residual=residual, act=act, tbias=tbias, tb_period=tb_period)
        return out

    def _ln(self, x, gb, M):
        """fp32 LayerNorm; on the bf16 path the kernel also leaves a bf16 copy (the next GEMM's A operand),
        looked up by _b16()."""
        out = torch.empty_like(x)
        o16 = torch.empty(x.shape, device=self.dev, dtype=torch.bfloat16) if self.chain else None
        self.h.call("layernorm", x, gb[0], gb[1], out, M, self.D, o16)
        if o16 is not None:
            self._copies = (out, o16)
        return out

    def _b16(self, x):
        """the bf16 copy of x if the op that produced x left one (aligned rows only), else x itself"""
        c = self._copies
        return c[1] if (c is not None and c[0] is x and self.D % 8 == 0) else x

    def _self_attn(self, blk, x, B, S, heads, pos, M):
        D = self.D
        if pos is None:
            qkv = self._lin(blk.qkv, self._b16(x), M)
            q, k, v, ld = qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D
            ldv = 3 * D
        else:
            xp = torch.empty_like(x)
            self.h.call("add_rows", x, pos, xp, capi.I64(x.numel()), capi.I64(x.numel()))
            qk = self._lin(blk.qk, xp, M)
            v = self._lin(blk.v, self._b16(x), M)
            q, k, ld, ldv = qk, qk[:, D:], 2 * D, D
        o = torch.empty(M, D, device=self.dev, dtype=torch.bfloat16 if self._mha_fast(D // heads, S, ld, ld, ldv) else torch.float32)
        self._mha(q, ld, k, ld, v, ldv, o, B, heads, S, S)
        return o

    def _mha_fast(self, hd, Sk, ldq, ldk, ldv):
        return (self.precision == "bf16" and hd in (16, 32, 64, 128) and Sk <= 192 and ldq % 4 == 0 and ldk % 4 == 0
                and ldv % 4 == 0 and self.D % 8 == 0)

    def _mha(self, q, ldq, k, ldk, v, ldv, o, B, heads, Sq, Sk):
        import ctypes
        hd = self.D // heads
        if self.h.recorder is not None:
            fast = self._mha_fast(hd, Sk, ldq, ldk, ldv)
            capi.require(fast or o.dtype == torch.float32, "unsupported argument: requires o.dtype == torch.float32")
            tail = (self.D, 1 if o.dtype == torch.bfloat16 else 0, B, heads, Sq, Sk, hd) if fast else (self.D, B, heads, Sq, Sk, hd)
            self.h.recorder.add(("mha", fast, (q.data_ptr(), ldq, k.data_ptr(), ldk, v.data_ptr(), ldv, o.data_ptr()) + tail, (q, k, v, o)))
            return
        s = torch.cuda.current_stream().cuda_stream
        vp = ctypes.c_void_p
        # bf16 path: attention on the matrix cores; fp32 ("bf16x3") path: the exact fp32 VALU kernel
        if self._mha_fast(hd, Sk, ldq, ldk, ldv):
            rc = self.h.lib.rg_mha_bf16(self.h._h, vp(q.data_ptr()), ldq, vp(k.data_ptr()), ldk, vp(v.data_ptr()), ldv,
                                        vp(o.data_ptr()), self.D, 1 if o.dtype == torch.bfloat16 else 0, B, heads, Sq, Sk, hd,
                                        vp(s))
        else:
            capi.require(o.dtype == torch.float32, "unsupported argument: requires o.dtype == torch.float32")
            rc = self.h.lib.rg_mha(self.h._h, vp(q.data_ptr()), ldq, vp(k.data_ptr()), ldk, vp(v.data_ptr()), ldv,
                                   vp(o.data_ptr()), self.D, B, heads, Sq, Sk, hd, vp(s))
        if rc != 0:
            raise capi.RgError("rg_mha failed: %s" % self.h.lib.rg_last_error(self.h._h).decode())

    def _enc_layer(self, blk, x, B, S, heads, pos=None):
        """detr_utils.py:335-393 TransformerEncoderLayer (forward_post / forward_pre)."""
        M = B * S
        if not self.pre:
            a = self._self_attn(blk, x, B, S, heads, pos, M)
            x = self._ln(self._lin(blk.out, a, M, residual=x), blk.n1, M)
            hmid = self._lin(blk.l1, self._b16(x), M, act=self.act, bf16_out=self.chain)
            return self._ln(self._lin(blk.l2, hmid, M, residual=x), blk.n2, M)
        x2 = self._ln(x, blk.n1, M)
        a = self._self_attn(blk, x2, B, S, heads, pos, M)
        x = self._lin(blk.out, a, M, residual=x)
        hmid = self._lin(blk.l1, self._ln(x, blk.n2, M), M, act=self.act)
        return self._lin(blk.l2, hmid, M, residual=x)

    def _cross_attn(self, blk, t, mem, B, S, Sk, heads):
        D = self.D
        q = self._lin(blk.cq, t, B * S)
        kv = self._lin(blk.ckv, mem, B * Sk)
        o = torch.empty(B * S, D, device=self.dev)
        self._mha(q, D, kv, 2 * D, kv[:, D:], 2 * D, o, B, heads, S, Sk)
        return o

    def _dec_layer(self, blk, t, mem, B, S, Sk, heads):
        """detr_utils.py:396-480 TransformerDecoderLayer (pos = query_pos = None)."""
        M = B * S
        if not self.pre:
            a = self._self_attn(blk, t, B, S, heads, None, M)
            t = self._ln(self._lin(blk.out, a, M, residual=t), blk.n1, M)
            c = self._cross_attn(blk, t, mem, B, S, Sk, heads)
            t = self._ln(self._lin(blk.cout, c, M, residual=t), blk.n2, M)
            hmid = self._lin(blk.l1, t, M, act=self.act)
            return self._ln(self._lin(blk.l2, hmid, M, residual=t), blk.n3, M)
        t2 = self._ln(t, blk.n1, M)
        t = self._lin(blk.out, self._self_attn(blk, t2, B, S, heads, None, M), M, residual=t)
        t2 = self._ln(t, blk.n2, M)
        t = self._lin(blk.cout, self._cross_attn(blk, t2, mem, B, S, Sk, heads), M, residual=t)
        hmid = self._lin(blk.l1, self._ln(t, blk.n3, M), M, act=self.act)
        return self._lin(blk.l2, hmid, M, residual=t)

    def _skip_stack(self, st, x, layer_fn):
        """detr_utils.py:101-210 SkipTransformerEncoder/Decoder: U-Net style skips, concat + Linear(2D->D)."""
        M = x.shape[0]
        xs = []
        for blk in st.inp:
            x = layer_fn(blk, x)
            xs.append(x)
        x = layer_fn(st.mid, x)
        for blk, lin in zip(st.outb, st.lin):
            x = self._lin(lin, None, M, segs=[G.Seg(x), G.Seg(xs.pop())])
            x = layer_fn(blk, x)
        return self._ln(x, st.norm, M)

    # ---------------------------------------------------------------- public API
    def encode_to_latent(self, features, eps, latent, row_off):
        """gesture_vae.py:111-193 `encode_to_dist` with the rsample noise made explicit.
        features [B, nframes, nfeats] fp32 (device), eps [B*n_chunks, 1, D]; writes
        z into latent[:, row_off:row_off+n_chunks, :] ([B,T,D])."""
        B, nframes, nf = features.shape
        n_chunks, S = nframes // self.chunk, self.chunk + 2
        Bn, D = B * n_chunks, self.D
        x = self._lin(self.embed, features.reshape(B * nframes, nf), B * nframes, tbias=self.pe_enc[2:S].contiguous(),
                      tb_period=self.chunk)
        xseq = torch.empty(Bn * S, D, device=self.dev)
        self.h.call("copy_rows", self.tok_pe, xseq, Bn, 2, D, 0, 0, S, 0)
        self.h.call("copy_rows", x, xseq, Bn, self.chunk, D, self.chunk, 0, S, 2)
        if self.venc is not None:
            enc = self.venc.run(xseq, Bn, S)
        else:
            enc = self._skip_stack(self.encoder, xseq, lambda blk, t: self._enc_layer(blk, t, Bn, S, self.heads))
        self.h.call("vae_reparam", enc, S, eps.contiguous(), latent, B, n_chunks, D, latent.shape[1], row_off)

    def decode_latent(self, latent, row_off, n_chunks):
        """gesture_vae.py:195-239 `decode` for z = latent[:, row_off:row_off+n_chunks] -> [B*num_frames, nfeats]."""
        B, T, D = latent.shape
        F_ = self.frames
        if self.arch == "all_encoder":
            S = n_chunks + F_
            xseq = torch.zeros(B * S, D, device=self.dev)
            self.h.call("copy_rows", latent, xseq, B, n_chunks, D, T, row_off, S, 0)
            pos = torch.empty_like(xseq)
            self.h.call("add_rows", xseq, self.pe_dec[:S].contiguous(), pos, capi.I64(xseq.numel()), capi.I64(S * D))
            if self.vdec is not None and S == 160:
                out = self.vdec.run(xseq, pos, B)
            else:
                out = self._skip_stack(self.decoder, xseq,
                                       lambda blk, t: self._enc_layer(blk, t, B, S, self.dec_heads, pos=pos))
            fr = torch.empty(B * F_, D, device=self.dev)
            self.h.call("copy_rows", out, fr, B, F_, D, S, n_chunks, F_, 0)
        else:
            q = torch.empty(B * F_, D, device=self.dev)
            self.h.call("copy_rows", self.pe_dec[:F_].contiguous(), q, B, F_, D, 0, 0, F_, 0)
            z = torch.empty(B * n_chunks, D, device=self.dev)
            self.h.call("copy_rows", latent, z, B, n_chunks, D, T, row_off, n_chunks, 0)
            mem = torch.empty_like(z)
            self.h.call("add_rows", z, self.pe_mem[:n_chunks].contiguous(), mem, capi.I64(z.numel()), capi.I64(n_chunks * D))
            fr = self._skip_stack(self.decoder, q,
                                  lambda blk, t: self._dec_layer(blk, t, mem, B, F_, n_chunks, self.dec_heads))
        return self._lin(self.final, fr, B * F_)


class GestureRepEncoder:
    """diffusion_transformer.py:131-330: four VAEs, 6D rotation packing, separator tokens."""

    def __init__(self, state, vae_cfgs, device="cuda", precision="bf16", prefix="gesture_rep_encoder.", part_streams=True,
                 chain=True, grouped=True, fused_encoder=True, fused_decoder=True):
        """part_streams: run the four body-part VAEs as concurrent launch chains (False: one chain); chain: see
        TransformerVAE; grouped: where the parts run as ONE chain (asynchronous submission), layer i of all four parts goes out
        as one grouped launch (capi.OpRecorder): a quarter of the dependent launches, same bits."""
        self.grouped = bool(grouped)
        self.dev = torch.device(device)
        self.h = capi.get_handle(self.dev.index if self.dev.index is not None else torch.cuda.current_device())
        self.vaes = {}
        for part in PARTS:
            p = "%s%s_vae." % (prefix, part)
            sd = {k[len(p):]: v for k, v in state.items() if k.startswith(p)}
            self.vaes[part] = TransformerVAE(sd, vae_cfgs[part], device, precision, chain=chain, fused_encoder=fused_encoder, fused_decoder=fused_decoder)
        self.vae_latent_dim = vae_cfgs["upper"]["latent_dim"]
        self.frame_chunk_size = vae_cfgs["upper"]["frame_chunk_size"]
        self.uj = self.lj = self.fj = self.hj = self.tj = None
        # the four body-part VAEs are independent launch chains of small kernels: each runs on a stream of its own
        # (forked from / joined into the caller's stream, also inside a graph capture); part_streams=False: one chain
        self.part_streams = [torch.cuda.Stream(device=self.dev) for _ in PARTS] if part_streams else None
        self._part_streams_cfg = self.part_streams

    def concurrent_parts(self, on):
        """Switch the fan-out over part streams on (if the encoder was built with it) or off.  Off is what a caller needs
        who queues these graphs BEHIND long-running work: a captured fork / join replays on internal streams, whose
        waits occupy the runtime's (4) hardware queues from the moment the graph is queued -- every other stream of the
        process then stalls until that earlier work is done (measured: all 16 probed streams, profiles/dbg/host_block4.py)."""
        self.part_streams = self._part_streams_cfg if on else None

    def _fan_out(self, jobs):
        """Run the per-part jobs (callables) concurrently: job i on part stream i, all ordered after the work already
        queued on the current stream, which in turn waits for all of them.  Tensors that cross the fork or the join
        are allocated on the current stream by the caller; everything a job allocates stays on its stream."""
        if self.part_streams is None and self.grouped and len(jobs) <= 4:
            # record the parts' launch sequences, then issue them zipped: layer i of all four parts in one launch
            rec = self.h.recorder = capi.OpRecorder()
            try:
                for job in jobs:
                    rec.begin_job()
                    job()
            finally:
                self.h.recorder = None
            rec.issue(self.h)
            return
        if self.part_streams is None:
            for job in jobs:
                job()
            return
        cur = torch.cuda.current_stream()
        for st, job in zip(self.part_streams, jobs):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                job()
        for st in self.part_streams[:len(jobs)]:
            cur.wait_stream(st)

    def _aa6d(self, aa, out, col_off, joints):
        B, n, c = aa.shape
        self.h.call("aa_to_6d", aa.contiguous(), c, out, out.shape[-1], col_off, B * n, joints)

    def encode(self, motion_upper, motion_lower, motion_face, motion_hands, motion_transl, motion_facial,
               motion_contact, motion_mask, eps_list):
        """Same argument order as the reference + eps_list (4 tensors [B*10,1,D], order upper, hands,
        face, lowertrans = the reference's rsample order).  Returns (latent [B,43,D], mask [B,43]).
        Like the reference it re-zeroes x/z of `motion_transl` IN PLACE (:231-232)."""
        f = lambda t: t.to(self.dev).float().contiguous()
        latent, tr_rel = self.encode_device_graphed(f(motion_upper), f(motion_lower), f(motion_face), f(motion_hands),
                                                    f(motion_transl), f(motion_facial), f(motion_contact),
                                                    [f(e) for e in eps_list])
        motion_transl.copy_(tr_rel.to(motion_transl.device))  # the reference's in-place mutation
        return latent, self.latent_mask(motion_mask)

    graph_runner = None  # callable(key, inputs, fn) -> outputs; set by MotionDiffusion (HIP-graph cache)
    debug_poison = False
    debug_keep = None    # diagnostics only: a list that receives (part, decoder output [B * frames, nfeats]) of every decode

    def encode_device_graphed(self, up, lo, fa, ha, tr, fac, con, eps_list):
        """encode_device through the owner's graph cache (one captured launch sequence per batch size)."""
        if self.graph_runner is None:
            return self.encode_device(up, lo, fa, ha, tr, fac, con, eps_list)
        ins = dict(up=up, lo=lo, fa=fa, ha=ha, tr=tr, fac=fac, con=con, e0=eps_list[0], e1=eps_list[1],
                   e2=eps_list[2], e3=eps_list[3])
        return self.graph_runner(("enc", up.shape[0], self.part_streams is None), ins, lambda s: self.encode_device(
            s["up"], s["lo"], s["fa"], s["ha"], s["tr"], s["fac"], s["con"], [s["e0"], s["e1"], s["e2"], s["e3"]]))

    def latent_mask(self, motion_mask):
        mm = motion_mask.to(self.dev).float()[:, ::self.frame_chunk_size]
        sep = torch.zeros_like(mm[:, :1])
        return torch.cat([mm, sep, mm, sep, mm, sep, mm], dim=1)

    def encode_device(self, up, lo, fa, ha, tr, fac, con, eps_list):
        """Device-only part of encode (fixed launch sequence, graph-capturable): all arguments are
        contiguous fp32 device tensors.  Returns (latent [B,T,D], trans with x/z made relative)."""
        dev = self.dev
        B, n, _ = up.shape
        self.uj, self.lj, self.fj, self.hj = up.shape[-1] // 3, lo.shape[-1] // 3, fa.shape[-1] // 3, ha.shape[-1] // 3
        self.tj = tr.shape[-1]
        rows = B * n
        in_up = torch.empty(B, n, self.uj * 6, device=dev)
        self._aa6d(up, in_up, 0, self.uj)
        in_ha = torch.empty(B, n, self.hj * 6, device=dev)
        self._aa6d(ha, in_ha, 0, self.hj)
        in_fa = torch.empty(B, n, self.fj * 6 + fac.shape[-1], device=dev)
        self._aa6d(fa, in_fa, 0, self.fj)
        self.h.call("copy_cols", fac, fac.shape[-1], 0, in_fa, in_fa.shape[-1], self.fj * 6, rows, fac.shape[-1], 0, 0)
        wlt = self.lj * 6 + self.tj + con.shape[-1]
        in_lt = torch.empty(B, n, wlt, device=dev)
        self._aa6d(lo, in_lt, 0, self.lj)
        self.h.call("copy_cols", tr, self.tj, 0, in_lt, wlt, self.lj * 6, rows, self.tj, n, 0b101)
        self.h.call("copy_cols", con, con.shape[-1], 0, in_lt, wlt, self.lj * 6 + self.tj, rows, con.shape[-1], 0, 0)
        tr_rel = torch.empty_like(tr)
        self.h.call("copy_cols", in_lt, wlt, self.lj * 6, tr_rel, self.tj, 0, rows, self.tj, 0, 0)
        n_lat = n // self.frame_chunk_size
        T, D = 4 * n_lat + 3, self.vae_latent_dim
        latent = torch.zeros(B, T, D, device=dev)  # separator rows stay zero
        self._fan_out([lambda i=i, part=part, feats=feats: self.vaes[part].encode_to_latent(
            feats, eps_list[i], latent, i * (n_lat + 1))
            for i, (part, feats) in enumerate((("upper", in_up), ("hands", in_ha), ("face", in_fa), ("lowertrans", in_lt)))])
        return latent, tr_rel

    def decode(self, z_output):
        """Returns (upper, lower, facepose, hands, transl, exps, contact) like the reference (:270-330)."""
        B, T, D = z_output.shape
        n_lat = (T - 3) // 4
        z = z_output.contiguous()
        F_ = self.vaes["upper"].frames
        rows = B * F_
        new = lambda width: torch.empty(B, F_, width, device=self.dev)   # outputs: allocated before the fork
        upper, hands, face, lower = new(self.uj * 3), new(self.hj * 3), new(self.fj * 3), new(self.lj * 3)
        n_exp = self.vaes["face"].nfeats - self.fj * 6
        n_con = self.vaes["lowertrans"].nfeats - self.lj * 6 - self.tj
        exps, transl, contact = new(n_exp), new(self.tj), new(n_con)

        def aa(src, out, joints):
            self.h.call("6d_to_aa", src, src.shape[-1], 0, out, joints * 3, rows, joints)

        def cols(src, c0, out):
            self.h.call("copy_cols", src, src.shape[-1], c0, out, out.shape[-1], 0, rows, out.shape[-1], 0, 0)

        def job(i, part):
            d = self.vaes[part].decode_latent(z, i * (n_lat + 1), n_lat)
            if self.debug_keep is not None:      # diagnostics: the decoder output in front of the rotation conversion
                self.debug_keep.append((part, d))
                if self.debug_poison and self.h.recorder is not None:
                    d.fill_(float("nan"))        # runs now, i.e. BEFORE the recorded GEMM that writes d is issued
            if part == "upper":
                aa(d, upper, self.uj)
            elif part == "hands":
                aa(d, hands, self.hj)
            elif part == "face":
                aa(d, face, self.fj)
                cols(d, self.fj * 6, exps)
            else:
                aa(d, lower, self.lj)
                cols(d, self.lj * 6, transl)
                cols(d, self.lj * 6 + self.tj, contact)

        self._fan_out([lambda i=i, part=part: job(i, part) for i, part in enumerate(PARTS)])
        return upper, lower, face, hands, transl, exps, contact
