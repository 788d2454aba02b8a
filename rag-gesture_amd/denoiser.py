"""Host side of the MI355X denoiser: weight packing, once-per-clip conditioning precompute and
the per-step launch sequence of `ReGestureTransformer.forward` at inference
(reference: mogen/models/transformers/diffusion_transformer.py:620-668 `forward`,
raggesture.py:1041-1113 `forward_test`, diffusion_transformer.py:105-127 `DecoderLayer`).

Everything numeric runs in the C-ABI HIP extension (capi / gemm); torch only owns the device
buffers.  What is hoisted out of the 50-step loop, all exact algebra:
  * the 40 StylizationBlock `emb_layers` and the time_embed MLP depend only on the timestep:
    one [steps, layers, 5, 2*D] table per model, built at load time in fp32;
  * the cross-attention K/V projections and A = softmax_N(K)^T V depend only on the
    conditioning: built once per clip (`set_conditions`);
  * the classifier-free rows' A is the value bias broadcast (cond_type 0 zeroes the value input
    and softmax(K) sums to 1), built at load time;
  * `ca_mix(cat_c(x + out_c(h_c)))` = [h_text|h_audio|h_spk|x] @ [Wm_c Wo_c ... | sum_c Wm_c]^T + b:
    one K=2048 GEMM instead of three 512x512 GEMMs plus the 1536->512 mix.
"""
import math

import numpy as np
import torch

from . import capi, gemm as G, seqfwd as SQ

CONDS = ("xf_text", "xf_audio", "xf_spk")
BLOCKS = ("sa_block", "ca_blocks.xf_text", "ca_blocks.xf_audio", "ca_blocks.xf_spk", "ffn")


def _timestep_embedding(timesteps, dim, max_period=10000):
    """reference: diffusion_transformer.py:27-46 (host, fp32; 50 rows at load time)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class DenoiserWeights:
    KV_GROUP = 2      # layers per condition K / V projection GEMM (output [rows, KV_GROUP * 2 D] fp32: 260 MB for 64 x 499 audio tokens)

    @staticmethod
    def _fold_ln(W, b, gamma, beta, pw, f32):
        """LN(x) W^T + b = rstd (x W'^T - mean c1) + c2 with W' = W diag(gamma), c1 = rowsum(bf16(W')) (what the
        matrix cores actually multiply), c2 = b + W beta.  Returns (W' [packed if pw], c1, c2)."""
        Wp = (W.double() * gamma.double()[None, :]).float()
        c1 = Wp.to(torch.bfloat16).double().sum(1).float()
        c2 = (b.double() + W.double() @ beta.double()).float()
        if pw is None:
            return Wp, c1, c2
        return pw(Wp), f32(c1), f32(c2)

    """Device-resident packed weights of one ReGestureTransformer (bf16 GEMM operands [N,K]
    zero-padded to 64, fp32 biases / LayerNorm parameters / tables)."""

    def __init__(self, state, cfg, schedule, device="cuda", prefix="", precision="bf16"):
        """precision: "bf16" (MFMA operands rounded to bf16, the production path) or "fp32"
        (bf16x3 split operands: ~fp32 products at 3 MFMAs per tile, used for parity checks)."""
        capi.require(precision in ("bf16", "fp32"), "unsupported argument: requires precision in (\"bf16\", \"fp32\")")
        self.precision = precision
        self.cfg = cfg
        self.schedule = schedule
        self.dev = torch.device(device)
        self.h = capi.get_handle(self.dev.index if self.dev.index is not None else torch.cuda.current_device())
        D, L = cfg["latent_dim"], cfg["num_layers"]
        self.D, self.L, self.H = D, L, cfg["num_heads"]
        self.TE, self.FF = cfg["time_embed_dim"], cfg["ff_size"]
        capi.require(D % 128 == 0 and D // self.H == 32, "kernels are specialised for head_dim 32")
        n_lat = cfg["max_seq_len"] // cfg["frame_chunk_size"]
        self.n_lat, self.T = n_lat, 4 * n_lat + 3
        sd = {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)} if prefix else state
        f32 = lambda t: t.detach().to(torch.float32).to(self.dev).contiguous()
        pw = lambda t: G.pack_weight(t.detach().to(torch.float32), self.dev, split=(precision == "fp32"))
        g = lambda name: sd[name].detach().to(torch.float32)

        # --- embedding / head / conditioning projections
        self.w_embed, self.b_embed = pw(g("joint_embed.weight")), f32(g("joint_embed.bias"))
        pos = g("sequence_embedding.pe").permute(1, 0, 2)[0, :n_lat]
        sep = torch.zeros(1, D)
        pos_cat = torch.cat([pos, sep, pos, sep, pos, sep, pos], dim=0)
        self.tbias = f32(pos_cat + g("global_positional_embedding.pe")[:self.T, 0])
        self.w_out, self.b_out = pw(g("out.weight")), f32(g("out.bias"))
        self.w_text, self.b_text = pw(g("text_pre_proj.weight")), f32(g("text_pre_proj.bias"))
        self.w_audio, self.b_audio = pw(g("audio_pre_proj.weight")), f32(g("audio_pre_proj.bias"))
        self.spk_table = f32(g("speaker_embedding.weight"))
        self.num_speakers = self.spk_table.shape[0]

        # --- per layer
        self.layers = []
        for l in range(L):
            p = "temporal_decoder_blocks.%d." % l
            lw = {}
            lw["sa_g"], lw["sa_b"] = f32(g(p + "sa_block.norm.weight")), f32(g(p + "sa_block.norm.bias"))
            lw["w_qkv"] = pw(torch.cat([g(p + "sa_block.%s.weight" % n) for n in ("query", "key", "value")], 0))
            lw["b_qkv"] = f32(torch.cat([g(p + "sa_block.%s.bias" % n) for n in ("query", "key", "value")], 0))
            if precision == "bf16":  # LayerNorm folded into the GEMM epilogue (include/rg_gesture.h: ln_stats)
                lw["w_qkv_ln"], lw["c1_qkv"], lw["c2_qkv"] = self._fold_ln(
                    torch.cat([g(p + "sa_block.%s.weight" % n) for n in ("query", "key", "value")], 0),
                    torch.cat([g(p + "sa_block.%s.bias" % n) for n in ("query", "key", "value")], 0),
                    g(p + "sa_block.norm.weight"), g(p + "sa_block.norm.bias"), pw, f32)
            lw["sa_sg"], lw["sa_sb"] = f32(g(p + "sa_block.proj_out.norm.weight")), f32(g(p + "sa_block.proj_out.norm.bias"))
            lw["w_sao"], lw["b_sao"] = pw(g(p + "sa_block.proj_out.out_layers.2.weight")), f32(g(p + "sa_block.proj_out.out_layers.2.bias"))
            cq = [p + "ca_blocks.%s." % c for c in CONDS]
            lw["ca_g"] = f32(torch.stack([g(q + "norm.weight") for q in cq]))
            lw["ca_b"] = f32(torch.stack([g(q + "norm.bias") for q in cq]))
            lw["w_q3"] = pw(torch.cat([g(q + "query.weight") for q in cq], 0))
            lw["b_q3"] = f32(torch.cat([g(q + "query.bias") for q in cq], 0))
            if precision == "bf16":
                parts = [self._fold_ln(g(q + "query.weight"), g(q + "query.bias"), g(q + "norm.weight"), g(q + "norm.bias"),
                                       None, None) for q in cq]
                lw["w_q3_ln"] = pw(torch.cat([pt[0] for pt in parts], 0))
                lw["c1_q3"] = f32(torch.cat([pt[1] for pt in parts], 0))
                lw["c2_q3"] = f32(torch.cat([pt[2] for pt in parts], 0))
            lw["tn_g"] = [f32(g(q + "text_norm.weight")) for q in cq]
            lw["tn_b"] = [f32(g(q + "text_norm.bias")) for q in cq]
            lw["w_kv"] = [pw(torch.cat([g(q + "key.weight"), g(q + "value.weight")], 0)) for q in cq]
            lw["b_kv"] = [f32(torch.cat([g(q + "key.bias"), g(q + "value.bias")], 0)) for q in cq]
            lw["ca_sg"] = [f32(g(q + "proj_out.norm.weight")) for q in cq]
            lw["ca_sb"] = [f32(g(q + "proj_out.norm.bias")) for q in cq]
            lw["ca_sgs"], lw["ca_sbs"] = torch.stack(lw["ca_sg"]).contiguous(), torch.stack(lw["ca_sb"]).contiguous()
            # classifier-free rows: A[c][h][d][l] = b_v[h*32 + l]
            bv = torch.stack([g(q + "value.bias") for q in cq])  # [3, D]
            lw["a_unc"] = f32(bv.view(3, self.H, 1, 32).expand(3, self.H, 32, 32))
            # fused ca_mix o proj_out.out_layers: [D, 4D]
            wm = g(p + "ca_mix.weight").double()
            fused, bias = [], g(p + "ca_mix.bias").double()
            for ci, q in enumerate(cq):
                wmc = wm[:, ci * D:(ci + 1) * D]
                fused.append(wmc @ g(q + "proj_out.out_layers.2.weight").double())
                bias = bias + wmc @ g(q + "proj_out.out_layers.2.bias").double()
            fused.append(wm[:, :D] + wm[:, D:2 * D] + wm[:, 2 * D:])
            lw["w_mix_f32"] = torch.cat(fused, 1).float()
            lw["w_mix"], lw["b_mix"] = pw(lw["w_mix_f32"]), f32(bias.float())
            lw["w_ff1"], lw["b_ff1"] = pw(g(p + "ffn.linear1.weight")), f32(g(p + "ffn.linear1.bias"))
            lw["w_ff2"], lw["b_ff2"] = pw(g(p + "ffn.linear2.weight")), f32(g(p + "ffn.linear2.bias"))
            lw["ff_sg"], lw["ff_sb"] = f32(g(p + "ffn.proj_out.norm.weight")), f32(g(p + "ffn.proj_out.norm.bias"))
            lw["w_ffo"], lw["b_ffo"] = pw(g(p + "ffn.proj_out.out_layers.2.weight")), f32(g(p + "ffn.proj_out.out_layers.2.bias"))
            self.layers.append(lw)

        # --- condition K / V projections of SEVERAL layers as one GEMM per condition (bf16 path): the conditions' LayerNorm
        # differs between layers only in its affine (efficient_attention.py:74-80: `text_norm`), so the normalised rows
        # xhat = (xf - mean) * rstd are materialised ONCE as bf16 and every layer's gain / offset is folded into its weights,
        #   LN_l(xf) W_l^T + b_l = xhat (W_l diag(gamma_l))^T + (b_l + W_l beta_l),
        # and the [key | value] weights of KV_GROUP consecutive layers are stacked along N: 3 x L / KV_GROUP launches of
        # N = KV_GROUP x 1024 with a bf16 A operand instead of 3 x L launches of N = 1024 that each convert fp32 A tiles and
        # redo the LayerNorm in every column tile (NOTEBOOK section 9; was gemm_dma_kernel<false, ...>, 4 % of the step)
        self.kv_group = None
        if precision == "bf16":
            KVG = self.KV_GROUP if L % self.KV_GROUP == 0 else 1
            self.kv_group = KVG
            self.w_kv_grp, self.b_kv_grp = [], []      # [group][condition]
            for g0 in range(0, L, KVG):
                ws, bs = [], []
                for ci in range(3):
                    wl, bl = [], []
                    for l in range(g0, g0 + KVG):
                        q = "temporal_decoder_blocks.%d.ca_blocks.%s." % (l, CONDS[ci])
                        wkv = torch.cat([g(q + "key.weight"), g(q + "value.weight")], 0).double()
                        bkv = torch.cat([g(q + "key.bias"), g(q + "value.bias")], 0).double()
                        ga, be = g(q + "text_norm.weight").double(), g(q + "text_norm.bias").double()
                        wl.append((wkv * ga[None, :]).float())
                        bl.append((bkv + wkv @ be).float())
                    ws.append(pw(torch.cat(wl, 0)))
                    bs.append(f32(torch.cat(bl, 0)))
                self.w_kv_grp.append(ws)
                self.b_kv_grp.append(bs)
            # ... and all L layers of a condition as ONE plain bf16 matrix [L x 1024, 512] for the fused projection + reduction
            # (rg_cond_kv: one launch per condition, K | V never leave the registers)
            self.w_kv_all, self.b_kv_all = [], []
            if D == 512 and self.H == 16:
                for ci in range(3):
                    wl, bl = [], []
                    for l in range(L):
                        q = "temporal_decoder_blocks.%d.ca_blocks.%s." % (l, CONDS[ci])
                        wkv = torch.cat([g(q + "key.weight"), g(q + "value.weight")], 0).double()
                        bkv = torch.cat([g(q + "key.bias"), g(q + "value.bias")], 0).double()
                        ga, be = g(q + "text_norm.weight").double(), g(q + "text_norm.bias").double()
                        wl.append((wkv * ga[None, :]).float())
                        bl.append((bkv + wkv @ be).float())
                    self.w_kv_all.append(torch.cat(wl, 0).to(self.dev).to(torch.bfloat16).contiguous())
                    self.b_kv_all.append(f32(torch.cat(bl, 0)))
            self.ln_ones, self.ln_zeros = torch.ones(D, device=self.dev), torch.zeros(D, device=self.dev)

        # --- timestep tables: ss[step][layer][block] = emb_layers(SiLU(time_embed(t_step)))  (fp32, exact)
        S = schedule.num_timesteps
        self.S = S
        temb = f32(_timestep_embedding(torch.tensor(schedule.timestep_map, dtype=torch.long), D))
        e1 = torch.empty(S, self.TE, device=self.dev)
        emb = torch.empty(S, self.TE, device=self.dev)
        self.h.call("linear_f32", temb, f32(g("time_embed.0.weight")), f32(g("time_embed.0.bias")), e1, S, self.TE, D, 0, 1)
        self.h.call("linear_f32", e1, f32(g("time_embed.2.weight")), f32(g("time_embed.2.bias")), emb, S, self.TE, self.TE, 0, 0)
        self.ss = torch.empty(S, L, 5, 2 * D, device=self.dev)
        tmp = torch.empty(S, 2 * D, device=self.dev)
        for l in range(L):
            for bi, blk in enumerate(BLOCKS):
                q = "temporal_decoder_blocks.%d.%s.proj_out.emb_layers.1." % (l, blk)
                self.h.call("linear_f32", emb, f32(g(q + "weight")), f32(g(q + "bias")), tmp, S, 2 * D, self.TE, 1, 0)
                self.ss[:, l, bi].copy_(tmp)
        # --- stylization folded per (step, layer) for the GEMMs that stylize their bf16 A rows in LDS (rg_gemm_desc.seg):
        # gain = gamma (1 + scale), offset = beta (1 + scale) + shift; block 0 = self attention, 1 = FFN
        if precision == "bf16":
            self.styl_gain = torch.empty(S, L, 2, D, device=self.dev)
            self.styl_off = torch.empty(S, L, 2, D, device=self.dev)
            for l, lw in enumerate(self.layers):
                for j, (bi, gk, bk) in enumerate(((0, "sa_sg", "sa_sb"), (4, "ff_sg", "ff_sb"))):
                    sc1 = 1.0 + self.ss[:, l, bi, :D]
                    self.styl_gain[:, l, j] = lw[gk] * sc1
                    self.styl_off[:, l, j] = lw[bk] * sc1 + self.ss[:, l, bi, D:]
        # --- classifier-free rows (bf16 path): cross-attention output == value bias for every token
        # (cond_type 0), so its stylized bf16 form is a (step, layer, cond, masked?) table
        if precision == "bf16":
            for l, lw in enumerate(self.layers):
                cq = ["temporal_decoder_blocks.%d.ca_blocks.%s." % (l, c) for c in CONDS]
                bv = torch.stack([g(q + "value.bias") for q in cq])                      # [3, D]
                bq = (bv + torch.tensor(-1000000.0)) + torch.tensor(1000000.0)           # fp32 rounding of y - 1e6
                yu = f32(torch.stack([bv.reshape(-1), bq.reshape(-1)]))                  # [2, 3D]
                st = torch.empty(2 * 3, D // 64, 2, device=self.dev)
                self.h.call("row_stats", yu.view(6, D), st, 6, D)
                st = st.view(2, 3, D // 64, 2)
                tab = torch.empty(S, 2, 3 * D, device=self.dev, dtype=torch.bfloat16)
                stc = [st[:, c].contiguous() for c in range(3)]
                for si in range(S):
                    segs = [G.Seg(yu, ld=3 * D, mode=G.A_STYL, stats=stc[c], gamma=lw["ca_sg"][c], beta=lw["ca_sb"][c],
                                  scale_shift=self.ss[si, l, 1 + c], col_offset=c * D) for c in range(3)]
                    G.stylize(self.h, segs, D, 2, tab[si])
                lw["unc_tab"] = tab
        # --- streams of the sequence-stationary forward (rg_seq_forward): weights in MFMA-fragment order, per-step parameters
        self.seq_streams = None
        if SQ.supported(cfg, self.T, precision):
            extras = [dict(w_mix=lw.pop("w_mix_f32"), b_mix=lw["b_mix"], unc_tab=lw["unc_tab"]) for lw in self.layers]
            self.seq_streams = SQ.SeqStreams(g, self.ss, extras, cfg, S, self.dev)
        for lw in self.layers:
            lw.pop("w_mix_f32", None)
        torch.cuda.synchronize(self.dev)
        # per-joint CFG scale (raggesture.py:909-922), default all ones
        pjs = cfg.get("per_joint_scale") or dict(upper=1.0, hands=1.0, face=1.0, lowertransl=1.0)
        js = torch.ones(self.T)
        n = n_lat
        js[0:n], js[n + 1:2 * n + 1] = pjs["upper"], pjs["hands"]
        js[2 * n + 2:3 * n + 2], js[3 * n + 3:] = pjs["face"], pjs["lowertransl"]
        self.js = f32(js)


def xcd_affine_order(n_groups, items_per_group, T, tile_rows=64, n_xcd=8):
    """Launch order for per-row-group kernels (attention): workgroups are dealt round-robin to the 8
    XCDs (block b -> XCD b % 8) and the GEMMs put M-tile t (rows [64t, 64t+64)) on XCD t % 8, so
    row group g (rows [g*T, (g+1)*T)) is listed in the slots of the XCD that owns its middle row:
    its activations are then read from the L2 they were written to instead of across the fabric.
    Returns an int32 array: launch slot -> work item (g * items_per_group + i), -1 = idle slot."""
    lists = [[] for _ in range(n_xcd)]
    for g in range(n_groups):
        x = ((g * T + T // 2) // tile_rows) % n_xcd
        lists[x].extend(g * items_per_group + i for i in range(items_per_group))
    depth = max(len(l) for l in lists)
    out = np.full(depth * n_xcd, -1, dtype=np.int32)
    for x, l in enumerate(lists):
        out[x:x + n_xcd * len(l):n_xcd] = l
    return out


class DenoiserSession:
    """Buffers + conditioning state for B clips (R = 2B rows: conditional rows first, then the
    classifier-free rows).  Not re-entrant; one per (model, batch size, stream)."""
    DEFAULT_ENGINE = "seq"

    def __init__(self, weights, B, ln_mode="auto", styl_prepass=True, xcd_affine=True, engine=None,
                 kv_grouped=True, kv_fused=True, seq_pairs=False, seq_duo=None, lane_dyn=None, tail_glue=True):
        """engine: "seq" = the whole forward as ONE launch, one workgroup per sequence, activations resident in registers /
        LDS, weights streamed (rg_seq_forward, csrc/rg_seq.hip; bf16 production path, D = 512, FF = 1024, T <= 48); "chain" =
        one launch per op (~90 per forward: rg_gemm + attention + stylization kernels).  None = "seq" where the shape is
        supported, else "chain" (precision="fp32" always runs the chain).
        seq_pairs (engine "seq"): one workgroup per CLIP runs the conditional sequence and then its classifier-free twin (B
        workgroups for ~1.7x the time instead of 2 B of which the classifier-free half idles the last 0.3): less CU time per
        forward for pipelines that run enough narrow chains side by side to fill the chip.  Same bits.
        The remaining options belong to the launch chain:
        ln_mode: "folded" = LayerNorm folded into the consuming GEMM's epilogue (two passes per
        layer fewer; its bf16 operand is the UN-normalised row, so the error grows with |row mean| / std),
        "prologue" = LayerNorm in a pre-pass (exact for any offset), "auto" = folded unless the session's first
        forward finds rows more than LN_GUARD_SIGMAS standard deviations off centre (one read-back, once per session).
        (The "seq" engine evaluates every LayerNorm in fp32 from the fp32 rows.)
        styl_prepass / xcd_affine: measurement knobs of the launch chain (NOTEBOOK section 6).  (Round 5 removed three more
        that no default used and every measurement had gone against: sa_fused, styl_in_gemm, tile64; and seq_launches, the
        forward cut into several launches by layer ranges: no gain on the step, NOTEBOOK 8.4.)"""
        # tail_glue: the DDIM loops over this session end every forward with the step's update (one launch per step; sampler.py,
        # csrc/rg_tail.h) -- False keeps the update in launches of its own (same bits)
        self.tail_glue = bool(tail_glue)
        # kv_grouped (bf16): the conditions' K / V projections of DenoiserWeights.KV_GROUP layers per GEMM on a bf16 normalised
        # operand (set_conditions); False = one fp32-A GEMM with a LayerNorm prologue per layer and condition (round 1-3)
        self.kv_grouped = bool(kv_grouped)
        # kv_fused (with kv_grouped, bf16, D = 512, 16 heads, <= 512 tokens per clip and condition): projection AND reduction of a
        # condition in one launch (rg_cond_kv) instead of L / 2 GEMMs + L reductions through fp32 K | V in memory
        self.kv_fused = bool(kv_fused)
        if ln_mode not in ("auto", "folded", "prologue"):
            raise capi.RgError("ln_mode must be 'auto', 'folded' or 'prologue'")
        if engine not in (None, "seq", "chain"):
            raise capi.RgError("engine must be 'seq' or 'chain'")
        if engine == "seq" and weights.seq_streams is None:
            raise capi.RgError("engine='seq': unsupported shape / precision (bf16, D = 512, 16 heads, FF = 1024, T <= 48, L <= 8)")
        if engine is None:
            engine = self.DEFAULT_ENGINE if weights.seq_streams is not None else "chain"
        self.engine = engine
        w = self.w = weights
        self.h = w.h
        self.B, self.R = B, 2 * B
        D, T, dev = w.D, w.T, w.dev
        self.M = M = self.R * T
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        self.head = f(M, D)
        self.qmask_c = torch.ones(3, B, T, device=dev)
        self.a_pre = f(w.L, 3, B, w.H, 32, 32)
        self.src_mask = torch.ones(self.R, T, device=dev)
        self.qmask = torch.ones(3, self.R, T, device=dev)
        self.abf = self.xa_bf = self.hcat = None
        self.ln_ratio = None     # max mean^2 / var seen by the guard (chain, ln_mode "auto", after the first forward)
        self.sq = None
        if self.engine == "seq":           # activations never leave the CU: no per-op buffers
            self.ln_mode = "exact"
            # seq_duo: two same-kind sequences per workgroup (rg_seq2_forward: half the weight bytes per token row, 0.8 of the CU
            # time per forward with seq_pairs) -- for the wide launches of a pipeline that fills the chip with them; None: with
            # seq_pairs.  Narrow launches are faster (in latency) with one workgroup per sequence.  Same bits either way.
            duo = bool(seq_pairs) if seq_duo is None else bool(seq_duo)
            # lane_dyn: the session runs on one lane of a pipeline that arbitrates launch forms on the device (seqfwd.SeqForward)
            self.sq = SQ.SeqForward(self, pairs=seq_pairs, duo=duo, lane_dyn=lane_dyn)
            return
        self.xa, self.xb, self.xc = f(M, D), f(M, D), f(M, D)
        # partial LayerNorm statistics: one (sum, sumsq) pair per row and producer column tile (128 wide)
        self.st_a, self.st_b, self.st_c = (f(M, D // G.STATS_COLS, 2) for _ in range(3))
        self.qkv, self.q3 = f(M, 3 * D), f(M, 3 * D)
        self.y_sa, self.st_sa = f(M, D), f(M, D // 128, 2)
        self.y3, self.st3 = f(M, 3 * D), f(3, M, D // 128, 2)
        self.g = torch.empty(M, w.FF, device=dev, dtype=torch.bfloat16 if w.precision == "bf16" else torch.float32)
        self.yf, self.st_f = f(M, D), f(M, D // G.STATS_COLS, 2)
        self.hcat = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16) if w.precision == "bf16" else None
        self.ln_mode = ln_mode if (w.precision == "bf16" and styl_prepass) else "prologue"
        if w.precision == "bf16" and styl_prepass:
            self.abf = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
            self.abf3 = torch.empty(B * T, 3 * D, device=dev, dtype=torch.bfloat16)
            if self.ln_mode != "prologue":
                self._xa_bf_buf = self.xa_bf = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
        self.st3c = f(3, B * T, D // 128, 2)           # cross-attention stats of the conditional rows only
        self.a_pre_t = torch.empty(w.L, 3, B, w.H, 2, 32, 32, device=dev, dtype=torch.bfloat16)  # A^T as bf16 hi/lo
        ng = D // 128
        order = xcd_affine_order if xcd_affine else \
            (lambda n, ipg, T_: np.arange(n * ipg, dtype=np.int32))
        dv = lambda a: torch.from_numpy(a).to(dev)
        self.perm_sa = dv(order(self.R, ng, T))
        self.perm_ca = dv(order(self.R, 3 * ng, T))
        self.perm_cac = dv(order(B, 3 * ng, T))

    # ------------------------------------------------------------------ once per clip
    def set_conditions(self, word, audio, speaker_ids, motion_mask, query_masks=None, offset=0, finalize=True):
        """word [B,Nt,768], audio [B,Na,768], speaker_ids [B,Ns] int64, motion_mask [B,T];
        query_masks: dict cond -> [B,T] or None (no query masking).
        offset: the n = word.shape[0] <= B clips given are the session's clips [offset, offset + n) (a session that holds
        two batches side by side is filled in two calls; finalize=False on all but the last).
        reference: raggesture.py:957-1013 (conditions), diffusion_architecture.py:146-166 (masks)."""
        w, h, D = self.w, self.h, self.w.D
        dev = w.dev
        Bs, B = self.B, word.shape[0]          # session clips, clips of this call
        o0, o1 = offset, offset + B
        capi.require(0 <= o0 and o1 <= Bs, "unsupported argument: requires 0 <= o0 and o1 <= Bs")
        mm = motion_mask.to(dev).float()
        self.src_mask[o0:o1].copy_(mm)
        self.src_mask[Bs + o0:Bs + o1].copy_(mm)
        for ci, c in enumerate(CONDS):
            if query_masks is None:
                self.qmask[ci, o0:o1].fill_(1.0)
                self.qmask[ci, Bs + o0:Bs + o1].fill_(1.0)
            else:
                qm = query_masks[c].to(dev).float()
                self.qmask[ci, o0:o1].copy_(qm)
                self.qmask[ci, Bs + o0:Bs + o1].copy_(qm)
        self.qmask_c[:, o0:o1].copy_(self.qmask[:, o0:o1])
        srcs = []
        for name, x, wt, bt in (("xf_text", word, w.w_text, w.b_text), ("xf_audio", audio, w.w_audio, w.b_audio)):
            x = x.to(dev).float().contiguous()
            n_tok, kin = x.shape[1], x.shape[2]
            xf = torch.empty(B * n_tok, D, device=dev)
            st = torch.empty(B * n_tok, D // G.STATS_COLS, 2, device=dev)
            G.gemm(h, M=B * n_tok, N=D, K=kin, W=wt, out=xf, segs=[G.Seg(x.view(B * n_tok, kin))], seg_len=None,
                   bias=bt, stats_out=st)
            srcs.append((xf, st, n_tok))
        ids = speaker_ids.to(dev).long().contiguous()
        n_tok = ids.shape[1]
        if w.num_speakers == 1:
            # reference quirk: zeros of shape [B, B, D] (diffusion_transformer.py:545-546)
            n_tok = B
            xf = torch.zeros(B * n_tok, D, device=dev)
        else:
            xf = torch.empty(B * n_tok, D, device=dev)
            h.call("gather_rows", w.spk_table, ids.view(-1), xf, B * n_tok, D)
        # LayerNorm statistics of rows that no GEMM produced
        st = torch.empty(B * n_tok, D // 64, 2, device=dev)
        h.call("row_stats", xf, st, B * n_tok, D)
        srcs.append((xf, st, n_tok))
        kv_max = max(s[2] for s in srcs)
        if w.kv_group is not None and self.kv_grouped:
            KVG = w.kv_group
            ldk = KVG * 2 * D
            # the session's own scratch (one set per call shape, made on first use): inside a captured set_conditions a
            # torch.empty would pin ~260 MB per graph for the graph's lifetime
            kv = self._scratch("kv", (B * kv_max, ldk), torch.float32)
            scratch = self._scratch("ln", (B * kv_max, D), torch.float32)
            xhat_all = self._scratch("xhat", (B * kv_max, D), torch.bfloat16)
            fused_all = True
            for ci in range(3):
                xf, _, n_tok = srcs[ci]
                M = B * n_tok
                xhat = xhat_all[:M]
                h.call("layernorm", xf, w.ln_ones, w.ln_zeros, scratch, M, D, xhat)      # exact two-pass statistics, no affine
                if self.kv_fused and w.w_kv_all and n_tok <= 512:
                    af = self.sq.afrag if self.sq is not None else None      # (the fragments of the sequence-stationary forward at once)
                    h.call("cond_kv", xhat, w.w_kv_all[ci], w.b_kv_all[ci], self.a_pre[0, ci, o0:o1], self.a_pre.stride(0),
                           None if af is None else af[0, ci, o0:o1], 0 if af is None else af.stride(0), B, n_tok, w.L)
                    fused_all = fused_all and af is not None
                    continue
                fused_all = False
                for gi in range(w.L // KVG):
                    G.gemm(h, M=M, N=ldk, K=D, W=w.w_kv_grp[gi][ci], out=kv, A=xhat, bias=w.b_kv_grp[gi][ci], ldo=ldk)
                    for j in range(KVG):
                        h.call("kv_reduce", kv.data_ptr() + 4 * j * 2 * D, ldk, self.a_pre[gi * KVG + j, ci, o0:o1], B, n_tok, D)
        else:
            kv = torch.empty(B * kv_max, 2 * D, device=dev)
            for l, lw in enumerate(w.layers):
                for ci in range(3):
                    xf, st, n_tok = srcs[ci]
                    G.gemm(h, M=B * n_tok, N=2 * D, K=D, W=lw["w_kv"][ci], out=kv,
                           segs=[G.Seg(xf, mode=G.A_LN, stats=st, gamma=lw["tn_g"][ci], beta=lw["tn_b"][ci])],
                           seg_len=D, bias=lw["b_kv"][ci], ldo=2 * D)
                    h.call("kv_reduce", kv, 2 * D, self.a_pre[l, ci, o0:o1], B, n_tok, D)
        if self.sq is not None and not (w.kv_group is not None and self.kv_grouped and fused_all):
            self.sq.set_a(self.a_pre[:, :, o0:o1], o0, o1)
        elif self.abf is not None and finalize:
            h.call("split_transpose_bf16", self.a_pre, self.a_pre_t, w.L * 3 * Bs * w.H)
        self._keep = (getattr(self, "_keep", []) if offset else []) + srcs

    def _scratch(self, name, shape, dtype):
        key = (name, tuple(shape), dtype)
        pool = self.__dict__.setdefault("_scratch_pool", {})
        if key not in pool:
            pool[key] = torch.empty(*shape, device=self.w.dev, dtype=dtype)
        return pool[key]

    def chain_end(self):
        """End of a loop of forwards on this session (the samplers call it): a lane-arbitrated session marks its lane idle."""
        if self.sq is not None:
            self.sq.chain_end()

    # ------------------------------------------------------------------ per step
    def forward(self, x, step, step_b=None, split=None, glue=None):
        """x [B,T,D] fp32 (device) at respaced step index `step`; returns the head output
        [2B,T,D] (rows [0,B) conditional, [B,2B) classifier-free) in self.head.
        step_b / split: the clips [split, B) are at step index step_b instead (two diffusion loops advancing in the same
        launches: the sampling of one batch and the inversion of the next batch's exemplars, sampler.cobatched_loop).  Only
        the timestep-dependent stylization differs between the groups; launch chain, bf16 path."""
        w, h, B, R, M, D, T = self.w, self.h, self.B, self.R, self.M, self.w.D, self.w.T
        if split is not None and not (0 < split < B):
            step, step_b, split = (step if split >= B else step_b), None, None
        if self.sq is not None:
            if glue is not None and not x.is_contiguous():
                raise capi.RgError("DenoiserSession.forward(glue=...): x is updated in place and must be contiguous")
            return self.sq.run(x.contiguous(), step, step_b, split, glue=glue)
        if glue is not None:
            raise capi.RgError("DenoiserSession.forward(glue=...): the sequence-stationary engine only")
        if self.ln_mode == "auto":
            out = self._forward_guarded(x, step)     # settles the mode (one read-back; never inside a capture)
            if split is None:
                return out
        return self._forward_chain(x, step, step_b, split)

    LN_GUARD_SIGMAS = 3.0   # |row mean| beyond this many standard deviations: the folded LayerNorm is not used

    def _forward_guarded(self, x, step):
        """First forward of an ln_mode="auto" session: run the folded path with rg_ln_guard behind every GEMM that
        leaves row statistics, read the figure back once, settle the mode (and redo the step in the LayerNorm pre-pass
        form if the rows are off centre).  Never inside a graph capture: callers warm up before they capture."""
        if torch.cuda.is_current_stream_capturing():
            raise capi.RgError("DenoiserSession(ln_mode='auto'): run one forward before capturing a graph")
        self._guard = torch.zeros(1, device=self.w.dev)
        self.ln_mode = "folded"
        out = self._forward_chain(x, step)
        self.ln_ratio = float(self._guard.item())
        self._guard = None
        if self.ln_ratio > self.LN_GUARD_SIGMAS ** 2:
            self.ln_mode, self.xa_bf = "prologue", None
            out = self._forward_chain(x, step)
        return out

    def _guard_stats(self, stats):
        if getattr(self, "_guard", None) is not None:
            self.h.call("ln_guard", stats, self.M, stats.shape[1], self.w.D, self._guard)

    def _forward_chain(self, x, step, step_b=None, split=None):
        w, h, B, R, M, D, T = self.w, self.h, self.B, self.R, self.M, self.w.D, self.w.T
        two = split is not None
        if two and (self.abf is None or self.hcat is None):
            raise capi.RgError("two step groups: bf16 launch chain with the stylization pre-pass only")
        xa, xb, xc = self.xa, self.xb, self.xc
        sa_, sb_, sc_ = self.st_a, self.st_b, self.st_c     # sa_: written by the embed GEMM and by every FFN-out GEMM (128-wide tiles)
        sa_w = self.st_a
        # h = joint_embed(x) + positional tables, duplicated for the two CFG branches
        G.gemm(h, M=M, N=D, K=D, W=w.w_embed, out=xa, segs=[G.Seg(x.view(B * T, D))], seg_len=D, a_row_mod=B * T,
               bias=w.b_embed, tbias=w.tbias, tb_period=T, stats_out=sa_, out2=self.xa_bf)
        self._guard_stats(sa_)
        for l, lw in enumerate(w.layers):
            ss = w.ss[step, l]
            ssb = w.ss[step_b, l] if two else None
            grp = (lambda bi: ([ssb[bi]], T, B, split)) if two else (lambda bi: None)
            # --- self attention
            qkv_seg = G.Seg(xa, mode=G.A_LN, stats=sa_, gamma=lw["sa_g"], beta=lw["sa_b"])
            if self.xa_bf is not None:
                # A = bf16 copy of xa written by the producing GEMM; the LayerNorm is folded into this GEMM's
                # epilogue (rstd * (acc - mean * c1) + c2): no normalisation pass, no extra launch
                G.gemm(h, M=M, N=3 * D, K=D, W=lw["w_qkv_ln"], out=self.qkv, A=self.xa_bf, bias=lw["c2_qkv"],
                       ln_stats=sa_, ln_c1=lw["c1_qkv"], softmax_cols=D)
            elif self.abf is not None:
                # bf16 A operands: LayerNorm once per element in a pre-pass (the GEMM's 12 column tiles would
                # each redo it and fetch fp32 rows: measured 35.3 -> 4.1 + 22.7 us at M = 4128)
                G.stylize(h, [qkv_seg], D, M, self.abf)
                G.gemm(h, M=M, N=3 * D, K=D, W=lw["w_qkv"], out=self.qkv, A=self.abf, bias=lw["b_qkv"], softmax_cols=D)
            else:
                G.gemm(h, M=M, N=3 * D, K=D, W=lw["w_qkv"], out=self.qkv, segs=[qkv_seg], seg_len=D,
                       bias=lw["b_qkv"], softmax_cols=D)
            h.call("sa_attention", self.qkv, 3 * D, self.src_mask, self.y_sa, D, self.st_sa, R, T, D,
                   self.perm_sa, self.perm_sa.numel(), 1 if w.precision == "bf16" else 0)
            sa_seg = G.Seg(self.y_sa, mode=G.A_STYL, stats=self.st_sa, gamma=lw["sa_sg"], beta=lw["sa_sb"], scale_shift=ss[0])
            if self.abf is not None:
                # stylization (LN, scale/shift, SiLU: 2 transcendentals per element) once per element in a
                # pre-pass instead of once per column tile and wave pair inside the GEMM's A prologue
                G.stylize(h, [sa_seg], D, M, self.abf, groups=grp(0))
                # out2: bf16 copy of xb = 4th K-segment of the ca_mix GEMM's A operand
                G.gemm(h, M=M, N=D, K=D, W=lw["w_sao"], out=xb, A=self.abf, bias=lw["b_sao"], residual=xa, stats_out=sb_,
                       out2=self.hcat[:, 3 * D:])
            else:
                G.gemm(h, M=M, N=D, K=D, W=lw["w_sao"], out=xb, segs=[sa_seg], seg_len=D, bias=lw["b_sao"], residual=xa,
                       stats_out=sb_)
            self._guard_stats(sb_)
            # --- three parallel cross attentions on the same input
            if self.hcat is not None:
                # production path: query projection + cross attention on the conditional rows only; the
                # classifier-free rows take their (constant) stylized cross-attention rows from the table
                Mc = B * T
                if self.xa_bf is not None:
                    # A = the bf16 copy of xb the SA-out epilogue left in hcat[:, 3D:]; per-condition LayerNorms
                    # folded into the epilogue (gamma in the weights, beta in the bias)
                    G.gemm(h, M=Mc, N=3 * D, K=D, W=lw["w_q3_ln"], out=self.q3, A=self.hcat[:, 3 * D:], bias=lw["c2_q3"],
                           ln_stats=sb_, ln_c1=lw["c1_q3"], softmax_cols=3 * D)
                elif self.abf is not None:
                    # the three query projections normalise the same rows with their own gamma/beta:
                    # pre-pass writes [Mc, 3*D] bf16 (one LN segment per condition), then one GEMM per condition
                    q3_segs = [G.Seg(xb, mode=G.A_LN, stats=sb_, gamma=lw["ca_g"][c], beta=lw["ca_b"][c]) for c in range(3)]
                    G.stylize(h, q3_segs, D, Mc, self.abf3)
                    G.gemm(h, M=Mc, N=3 * D, K=D, W=lw["w_q3"], out=self.q3, A=self.abf3, bias=lw["b_q3"],
                           softmax_cols=3 * D, gb_group=D, gb_stride=D)
                else:
                    G.gemm(h, M=Mc, N=3 * D, K=D, W=lw["w_q3"], out=self.q3,
                           segs=[G.Seg(xb, mode=G.A_LN, stats=sb_, gamma=lw["ca_g"], beta=lw["ca_b"])], seg_len=D,
                           bias=lw["b_q3"], softmax_cols=3 * D, gb_group=D, gb_stride=D)
                if self.abf is not None:
                    # cross attention + LN/stylization/SiLU of its three outputs in one launch, straight into
                    # the bf16 A operand of the ca_mix GEMM (classifier-free rows: the (step, layer) table)
                    if two:
                        h.call("ca_stylize_groups", self.q3, self.a_pre_t[l], self.qmask, lw["ca_sgs"], lw["ca_sbs"], ss[1:4],
                               lw["unc_tab"][step], self.hcat, 4 * D, B, B, T, D, 3, ssb[1:4], lw["unc_tab"][step_b], split)
                    else:
                        h.call("ca_stylize", self.q3, self.a_pre_t[l], self.qmask, lw["ca_sgs"], lw["ca_sbs"], ss[1:4],
                               lw["unc_tab"][step], self.hcat, 4 * D, B, B, T, D, 3)
                else:
                    h.call("ca_attention", self.q3, self.a_pre[l], None, self.qmask_c, self.y3, self.st3c, B, B, T, D, 3,
                           self.perm_cac, self.perm_cac.numel())
                    segs = [G.Seg(self.y3, ld=3 * D, mode=G.A_STYL, stats=self.st3c[c], gamma=lw["ca_sg"][c],
                                  beta=lw["ca_sb"][c], scale_shift=ss[1 + c], col_offset=c * D) for c in range(3)]
                    segs.append(G.Seg(xb))
                    # the SiLU prologue of the K = 2048 GEMM once per element instead of once per column tile
                    G.stylize(h, segs, D, M, self.hcat, m_cond=Mc, unc_nseg=3, unc_tab=lw["unc_tab"][step], qmask=self.qmask)
                G.gemm(h, M=M, N=D, K=4 * D, W=lw["w_mix"], out=xc, A=self.hcat, bias=lw["b_mix"], out2=self.abf)
            else:
                G.gemm(h, M=M, N=3 * D, K=D, W=lw["w_q3"], out=self.q3,
                       segs=[G.Seg(xb, mode=G.A_LN, stats=sb_, gamma=lw["ca_g"], beta=lw["ca_b"])], seg_len=D,
                       bias=lw["b_q3"], softmax_cols=3 * D, gb_group=D, gb_stride=D)
                h.call("ca_attention", self.q3, self.a_pre[l], lw["a_unc"], self.qmask, self.y3, self.st3, R, B, T, D, 3,
                       self.perm_ca, self.perm_ca.numel())
                segs = [G.Seg(self.y3, ld=3 * D, mode=G.A_STYL, stats=self.st3[c], gamma=lw["ca_sg"][c],
                              beta=lw["ca_sb"][c], scale_shift=ss[1 + c], col_offset=c * D) for c in range(3)]
                segs.append(G.Seg(xb))
                G.gemm(h, M=M, N=D, K=4 * D, W=lw["w_mix"], out=xc, segs=segs, seg_len=D, bias=lw["b_mix"])
            # --- FFN
            if self.abf is not None and self.hcat is not None:   # A = bf16 copy of xc written by the ca_mix epilogue
                G.gemm(h, M=M, N=w.FF, K=D, W=lw["w_ff1"], out=self.g, A=self.abf, bias=lw["b_ff1"], act=1)
            else:
                G.gemm(h, M=M, N=w.FF, K=D, W=lw["w_ff1"], out=self.g, segs=[G.Seg(xc)], seg_len=D, bias=lw["b_ff1"], act=1)
            if w.precision == "bf16":
                G.gemm(h, M=M, N=D, K=w.FF, W=lw["w_ff2"], out=self.yf, A=self.g, bias=lw["b_ff2"], stats_out=self.st_f)
            else:
                G.gemm(h, M=M, N=D, K=w.FF, W=lw["w_ff2"], out=self.yf, segs=[G.Seg(self.g)], seg_len=w.FF,
                       bias=lw["b_ff2"], stats_out=self.st_f)
            ff_seg = G.Seg(self.yf, mode=G.A_STYL, stats=self.st_f, gamma=lw["ff_sg"], beta=lw["ff_sb"], scale_shift=ss[4])
            if self.abf is not None:
                G.stylize(h, [ff_seg], D, M, self.abf, groups=grp(4))
                G.gemm(h, M=M, N=D, K=D, W=lw["w_ffo"], out=xa, A=self.abf, bias=lw["b_ffo"], residual=xc, stats_out=sa_w,
                       out2=self.xa_bf)
                sa_ = sa_w   # the next layer's QKV reads the statistics in this tile split
                self._guard_stats(sa_)
            else:
                G.gemm(h, M=M, N=D, K=D, W=lw["w_ffo"], out=xa, segs=[ff_seg], seg_len=D, bias=lw["b_ffo"], residual=xc,
                       stats_out=sa_)
        if self.xa_bf is not None:   # the last FFN-out epilogue left the bf16 copy of xa
            G.gemm(h, M=M, N=D, K=D, W=w.w_out, out=self.head, A=self.xa_bf, bias=w.b_out)
        else:
            G.gemm(h, M=M, N=D, K=D, W=w.w_out, out=self.head, segs=[G.Seg(xa)], seg_len=D, bias=w.b_out)
        return self.head

    def cfg_ddim(self, x, x_out, step, c_a, c_b, x0_out=None):
        """CFG mix of self.head + DDIM (or inversion) update of x -> x_out."""
        w, sch = self.w, self.w.schedule
        wc, wu = sch.cfg_weights(w.cfg["scale_func_cfg"], step)
        self.h.call("cfg_ddim_update", self.head, x, x_out, x0_out, w.js, self.B, w.T, w.D, wc, wu,
                    float(sch.c_recip[step]), float(sch.c_recipm1[step]), float(c_a), float(c_b))

    def cfg_ddim_rows(self, b0, nb, x, x_out, step, c_a, c_b, x_out2=None):
        """cfg_ddim for the clips [b0, b0 + nb) of the batch: x / x_out [nb,T,D] (x_out may alias x), x_out2 an optional
        second copy of the result."""
        w, sch, T = self.w, self.w.schedule, self.w.T
        wc, wu = sch.cfg_weights(w.cfg["scale_func_cfg"], step)
        self.h.call("cfg_ddim_update_rows", self.head[b0 * T:], self.head[(self.B + b0) * T:], x, x_out, x_out2, w.js, nb, T, w.D,
                    wc, wu, float(sch.c_recip[step]), float(sch.c_recipm1[step]), float(c_a), float(c_b))

    def cfg_ddpm(self, x, x_out, step, noise):
        """CFG mix of self.head + one ancestral step (inference_type="ddpm"; gaussian_diffusion.py:795-803)."""
        w, sch = self.w, self.w.schedule
        wc, wu = sch.cfg_weights(w.cfg["scale_func_cfg"], step)
        self.h.call("cfg_ddpm_update", self.head, x, noise, x_out, w.js, self.B, w.T, w.D, wc, wu,
                    float(sch.post_c1[step]), float(sch.post_c2[step]), float(sch.ddpm_sigma[step]))
