"""Per-window conditioning features: the two pretrained encoders the reference runs in front of the
hot path, on the HIP extension.

    tools/longform_synthesis.py:64-82   get_text_feature: BERT-base-cased, hidden_states of `tokenizer.encode_plus(sentence)`,
                                        sum of the last four layers -> [L_tokens, 768]
    tools/longform_synthesis.py:84-94   get_wav2vec2_feature: Wav2Vec2Processor (zero-mean / unit-variance waveform) +
                                        Wav2Vec2Model("facebook/wav2vec2-base-960h").last_hidden_state -> [499, 768] per 10 s
    mogen/datasets/beatx_dataset.py:498-505, 823-832, 1171-1179   the same two calls when the dataset caches are built

Weights are the Hugging Face state dicts (`BertModel.state_dict()`, `Wav2Vec2Model.state_dict()`: the released
checkpoints cannot be fetched here, tests use random-initialised models of the same configuration as the oracle).
Tokenisation (WordPiece vocabulary file) stays with the caller: `BertFeatures` takes token ids.

Everything numeric is a launch of the C-ABI extension: rg_gemm (bf16 MFMA operands, fp32 accumulate; bias / GELU /
residual epilogues; the strided convolutions of the wav2vec2 feature extractor are GEMMs over OVERLAPPING rows of the
previous layer's [T, 512] output: row t = elements [t * stride * 512, + kernel * 512), no im2col), rg_mha_bf16 (softmax
attention on the matrix cores, 499 keys resident in LDS), rg_layernorm_res, and the small kernels of csrc/rg_features.hip.
precision="fp32": bf16x3 split operands in the GEMMs (parity checks).
"""
import torch

from . import capi, gemm as G


def _f32(t, dev):
    return t.detach().to(torch.float32).to(dev).contiguous()


class _Linear:
    def __init__(self, w, b, dev, split):
        self.w = G.pack_weight(w.detach().float(), dev, split=split)
        self.b = _f32(b, dev) if b is not None else None
        self.n, self.k = w.shape


class _Encoder:
    """Post-norm transformer encoder stack shared by the two models (BertLayer / Wav2Vec2EncoderLayer):
    x -> LN1(x + O(MHA(x))) -> LN2(. + FF2(GELU(FF1(.))))."""

    def __init__(self, layers, heads, eps, dev, precision):
        self.layers, self.heads, self.eps, self.dev, self.precision = layers, heads, eps, dev, precision
        self.h = capi.get_handle(dev.index if dev.index is not None else torch.cuda.current_device())

    def lin(self, lw, x32, xbf, out, act=0, residual=None):
        """out = act(x W^T + b) (+ residual); bf16 mode multiplies the bf16 copy, fp32 mode the fp32 rows (bf16x3)."""
        M = x32.shape[0] if x32 is not None else xbf.shape[0]
        if self.precision == "bf16":
            a = xbf if xbf is not None else None
            if a is not None:
                G.gemm(self.h, M=M, N=lw.n, K=lw.k, W=lw.w, out=out, A=a, bias=lw.b, act=act, residual=residual)
                return
        G.gemm(self.h, M=M, N=lw.n, K=lw.k, W=lw.w, out=out, segs=[G.Seg(x32)], bias=lw.b, act=act, residual=residual)

    def layer_norm(self, x, res, g, b, want_bf16=True):
        out = torch.empty_like(x)
        obf = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16) if (want_bf16 and self.precision == "bf16") else None
        self.h.call("layernorm_res", x, res, g, b, out, x.shape[0], x.shape[1], float(self.eps), obf)
        return out, obf

    def attention(self, qkv, L, D):
        hd = D // self.heads
        q, k, v = qkv.data_ptr(), qkv.data_ptr() + 4 * D, qkv.data_ptr() + 8 * D
        if self.precision == "fp32" and L <= 192:
            o = torch.empty(L, D, device=self.dev)
            self.h.call("mha", q, 3 * D, k, 3 * D, v, 3 * D, o, D, 1, self.heads, L, L, hd)
            return o, None
        o = torch.empty(L, D, device=self.dev, dtype=torch.bfloat16 if self.precision == "bf16" else torch.float32)
        self.h.call("mha_bf16", q, 3 * D, k, 3 * D, v, 3 * D, o, D, int(self.precision == "bf16"), 1, self.heads, L, L, hd)
        return (None, o) if self.precision == "bf16" else (o, None)

    def run(self, x, xbf, collect=None):
        L, D = x.shape
        bf = self.precision == "bf16"
        for lw in self.layers:
            qkv = torch.empty(L, 3 * D, device=self.dev)
            self.lin(lw["qkv"], x, xbf, qkv)
            a32, abf = self.attention(qkv, L, D)
            t = torch.empty(L, D, device=self.dev)
            self.lin(lw["o"], a32, abf, t, residual=x)
            x1, x1bf = self.layer_norm(t, None, lw["ln1_g"], lw["ln1_b"])
            f = torch.empty(L, lw["ff1"].n, device=self.dev, dtype=torch.bfloat16 if bf else torch.float32)
            self.lin(lw["ff1"], x1, x1bf, f, act=1)
            t2 = torch.empty(L, D, device=self.dev)
            self.lin(lw["ff2"], None if bf else f, f if bf else None, t2, residual=x1)
            x, xbf = self.layer_norm(t2, None, lw["ln2_g"], lw["ln2_b"])
            if collect is not None:
                collect.append(x)
        return x


class BertFeatures:
    """BertModel (bert-base-cased shapes) hidden states.  state: `BertModel.state_dict()` (keys may carry a `bert.`
    prefix).  __call__(input_ids [L]) -> [L, 768] = sum of the last four hidden states (longform_synthesis.py:72-80)."""

    def __init__(self, state, num_heads=12, device="cuda", precision="bf16", eps=1e-12):
        dev = self.dev = torch.device(device)
        sd = {(k[5:] if k.startswith("bert.") else k): v for k, v in state.items()}
        split = precision == "fp32"
        f = lambda k: _f32(sd[k], dev)
        self.word, self.pos = f("embeddings.word_embeddings.weight"), f("embeddings.position_embeddings.weight")
        self.type0 = f("embeddings.token_type_embeddings.weight")[0].contiguous()
        self.eg, self.eb = f("embeddings.LayerNorm.weight"), f("embeddings.LayerNorm.bias")
        layers, i = [], 0
        while "encoder.layer.%d.attention.self.query.weight" % i in sd:
            p = "encoder.layer.%d." % i
            qkv_w = torch.cat([sd[p + "attention.self.%s.weight" % n] for n in ("query", "key", "value")], 0)
            qkv_b = torch.cat([sd[p + "attention.self.%s.bias" % n] for n in ("query", "key", "value")], 0)
            layers.append(dict(
                qkv=_Linear(qkv_w, qkv_b, dev, split),
                o=_Linear(sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"], dev, split),
                ln1_g=f(p + "attention.output.LayerNorm.weight"), ln1_b=f(p + "attention.output.LayerNorm.bias"),
                ff1=_Linear(sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"], dev, split),
                ff2=_Linear(sd[p + "output.dense.weight"], sd[p + "output.dense.bias"], dev, split),
                ln2_g=f(p + "output.LayerNorm.weight"), ln2_b=f(p + "output.LayerNorm.bias")))
            i += 1
        self.enc = _Encoder(layers, num_heads, eps, dev, precision)
        self.h = self.enc.h

    def hidden_states(self, input_ids):
        ids = input_ids.to(self.dev).long().view(-1).contiguous()
        L, D = ids.shape[0], self.word.shape[1]
        if L > self.pos.shape[0]:
            raise capi.RgError("BertFeatures: %d tokens exceed the %d position embeddings" % (L, self.pos.shape[0]))
        e = torch.empty(L, D, device=self.dev)
        self.h.call("embed_sum3", ids, self.word, self.pos, self.type0, e, L, D)
        x, xbf = self.enc.layer_norm(e, None, self.eg, self.eb)
        states = [x]
        self.enc.run(x, xbf, collect=states)
        return states

    def __call__(self, input_ids, layers=(-4, -3, -2, -1)):
        st = self.hidden_states(input_ids)
        out = st[layers[0]].clone()
        for i in layers[1:]:
            out += st[i]
        return out


class Wav2Vec2Features:
    """Wav2Vec2Model (wav2vec2-base shapes: 7 conv layers, feat_extract_norm "group", 12 post-norm encoder layers).
    state: `Wav2Vec2Model.state_dict()` (keys may carry a `wav2vec2.` prefix).
    __call__(waveform [n] at 16 kHz) -> last_hidden_state [T, 768] (T = 499 for 10 s), input normalised like
    Wav2Vec2FeatureExtractor(do_normalize=True) unless normalize=False."""

    def __init__(self, state, num_heads=12, device="cuda", precision="bf16", eps=1e-5, conv_stride=(5, 2, 2, 2, 2, 2, 2),
                 pos_groups=16):
        dev = self.dev = torch.device(device)
        sd = {(k[9:] if k.startswith("wav2vec2.") else k): v for k, v in state.items()}
        split = precision == "fp32"
        f = lambda k: _f32(sd[k], dev)
        self.precision, self.eps, self.stride = precision, eps, tuple(conv_stride)
        self.convs = []
        for i in range(len(self.stride)):
            w = sd["feature_extractor.conv_layers.%d.conv.weight" % i].detach().float()      # [C_out, C_in, k]
            b = sd.get("feature_extractor.conv_layers.%d.conv.bias" % i)
            self.convs.append(dict(lin=_Linear(w.permute(0, 2, 1).reshape(w.shape[0], -1), b, dev, split), k=w.shape[2], cin=w.shape[1]))
        self.gn_g, self.gn_b = f("feature_extractor.conv_layers.0.layer_norm.weight"), f("feature_extractor.conv_layers.0.layer_norm.bias")
        self.fp_g, self.fp_b = f("feature_projection.layer_norm.weight"), f("feature_projection.layer_norm.bias")
        self.fp = _Linear(sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"], dev, split)
        # positional convolution: weight_norm(dim=2) stored as (g, v) -- either parametrization naming
        pk = "encoder.pos_conv_embed.conv."
        if pk + "weight" in sd:
            w = sd[pk + "weight"].detach().float()
        else:
            g = sd.get(pk + "weight_g", sd.get(pk + "parametrizations.weight.original0")).detach().float()
            v = sd.get(pk + "weight_v", sd.get(pk + "parametrizations.weight.original1")).detach().float()
            w = v * (g / v.norm(p=2, dim=(0, 1), keepdim=True))
        C, Cg, K = w.shape                                                                       # [768, 48, 128]
        self.pos_groups, self.pos_k = pos_groups, K
        capi.require(C // pos_groups == Cg, "unsupported argument: requires C // pos_groups == Cg")
        pb = sd[pk + "bias"]
        self.pos = [_Linear(w[g * Cg:(g + 1) * Cg].permute(0, 2, 1).reshape(Cg, K * Cg), pb[g * Cg:(g + 1) * Cg], dev, split)
                    for g in range(pos_groups)]
        self.enc_g, self.enc_b = f("encoder.layer_norm.weight"), f("encoder.layer_norm.bias")
        layers, i = [], 0
        while "encoder.layers.%d.attention.q_proj.weight" % i in sd:
            p = "encoder.layers.%d." % i
            qkv_w = torch.cat([sd[p + "attention.%s_proj.weight" % n] for n in ("q", "k", "v")], 0)
            qkv_b = torch.cat([sd[p + "attention.%s_proj.bias" % n] for n in ("q", "k", "v")], 0)
            layers.append(dict(
                qkv=_Linear(qkv_w, qkv_b, dev, split),
                o=_Linear(sd[p + "attention.out_proj.weight"], sd[p + "attention.out_proj.bias"], dev, split),
                ln1_g=f(p + "layer_norm.weight"), ln1_b=f(p + "layer_norm.bias"),
                ff1=_Linear(sd[p + "feed_forward.intermediate_dense.weight"], sd[p + "feed_forward.intermediate_dense.bias"], dev, split),
                ff2=_Linear(sd[p + "feed_forward.output_dense.weight"], sd[p + "feed_forward.output_dense.bias"], dev, split),
                ln2_g=f(p + "final_layer_norm.weight"), ln2_b=f(p + "final_layer_norm.bias")))
            i += 1
        self.enc = _Encoder(layers, num_heads, eps, dev, precision)
        self.h = self.enc.h

    def conv_features(self, wave):
        """Feature extractor: [n] -> [T, 512] fp32 (Wav2Vec2FeatureEncoder; transposed to time-major)."""
        h, dev, bf = self.h, self.dev, self.precision == "bf16"
        n = wave.numel()
        x = torch.zeros(n + 64, device=dev)      # tail padding: the GEMM's 64-wide K tiles may look past the last sample
        x[:n] = wave.to(dev).float().view(-1)
        c0 = self.convs[0]
        T = (n - c0["k"]) // self.stride[0] + 1
        y = torch.empty(T, c0["lin"].n, device=dev)
        # layer 0: rows of 10 samples at a hop of 5 (fp32 source, K = 10)
        G.gemm(h, M=T, N=c0["lin"].n, K=c0["lin"].k, W=c0["lin"].w, out=y, segs=[G.Seg(x, ld=self.stride[0])], bias=c0["lin"].b)
        C = y.shape[1]
        cur = torch.empty(T, C, device=dev, dtype=torch.bfloat16 if bf else torch.float32)
        ws = torch.empty(2 * C, device=dev)
        h.call("time_groupnorm_gelu", y, self.gn_g, self.gn_b, cur if bf else None, None if bf else cur, T, C, float(self.eps), ws)
        out = None
        for i in range(1, len(self.convs)):
            cv, s = self.convs[i], self.stride[i]
            Tn = (T - cv["k"]) // s + 1
            last = i == len(self.convs) - 1
            if bf:
                out = torch.empty(Tn, cv["lin"].n, device=dev, dtype=torch.float32 if last else torch.bfloat16)
                G.gemm(h, M=Tn, N=cv["lin"].n, K=cv["lin"].k, W=cv["lin"].w, out=out, A=cur.view(-1), lda=s * C, bias=cv["lin"].b, act=1)
            else:
                out = torch.empty(Tn, cv["lin"].n, device=dev)
                G.gemm(h, M=Tn, N=cv["lin"].n, K=cv["lin"].k, W=cv["lin"].w, out=out, segs=[G.Seg(cur.view(-1), ld=s * C)],
                       bias=cv["lin"].b, act=1)
            cur, T = out, Tn
        return cur.float() if cur.dtype != torch.float32 else cur

    def __call__(self, wave, normalize=True):
        h, dev = self.h, self.dev
        x = wave.to(dev).float().view(-1)
        if normalize:   # Wav2Vec2FeatureExtractor.zero_mean_unit_var_norm
            x = (x - x.mean()) / torch.sqrt(x.var(unbiased=False) + 1e-7)
        feats = self.conv_features(x)                                   # [T, 512]
        T = feats.shape[0]
        fn, fnbf = self.enc.layer_norm(feats, None, self.fp_g, self.fp_b)
        hid = torch.empty(T, self.fp.n, device=dev)
        self.enc.lin(self.fp, fn, fnbf, hid)
        D = hid.shape[1]
        # positional convolution: one patch matrix + GEMM per group, GELU, added to the hidden states, LayerNorm
        Cg, K = D // self.pos_groups, self.pos_k
        cols = torch.empty(self.pos_groups, T, K * Cg, device=dev, dtype=torch.bfloat16)
        h.call("im2col_grouped", hid, cols, T, D, self.pos_groups, K, K // 2)
        pos = torch.empty(T, D, device=dev)
        for g in range(self.pos_groups):
            lw = self.pos[g]
            o = pos[:, g * Cg:(g + 1) * Cg]
            if self.precision == "bf16":
                G.gemm(h, M=T, N=Cg, K=K * Cg, W=lw.w, out=o, A=cols[g], bias=lw.b, act=1)
            else:
                G.gemm(h, M=T, N=Cg, K=K * Cg, W=lw.w, out=o, segs=[G.Seg(cols[g].float())], bias=lw.b, act=1)
        x0, x0bf = self.enc.layer_norm(hid, pos, self.enc_g, self.enc_b)
        return self.enc.run(x0, x0bf)


def merge_disco_textsegs(textsegs):
    """mogen/datasets/beatx_dataset.py:1099-1113: consecutive segments with identical (start, end) are one segment whose
    words are concatenated."""
    merged = []
    for i, seg in enumerate(textsegs):
        seg = [list(seg[0]), seg[1]]
        if i > 0 and seg[0] == list(textsegs[i - 1][0]):
            merged[-1][1] += seg[1]
        else:
            merged.append(seg)
    return merged


class WindowFeatures:
    """The per-window conditioning of tools/longform_synthesis.py:320-343 as the `features` callback of
    longform.LongformSynthesizer: wav2vec2 on the window's 16 kHz samples (`raw_audio` of the sample, [1, n]) and BERT
    (sum of the last four layers) on the window's transcript.  `tokenize(sentence) -> token ids` is the caller's
    WordPiece tokenizer (`tokenizer.encode_plus(sentence)["input_ids"]`); the models are BertFeatures / Wav2Vec2Features."""

    def __init__(self, bert, wav2vec2, tokenize, sample_rate=16000):
        self.bert, self.w2v, self.tokenize, self.sr = bert, wav2vec2, tokenize, sample_rate

    def window(self, raw_audio, t0, t1, text_segments):
        a0, a1 = int(t0 * self.sr), int(t1 * self.sr)
        wave = raw_audio.view(-1)[a0:a1]
        if wave.numel() < a1 - a0:    # the padded tail of the last window
            wave = torch.cat([wave, wave.new_zeros(a1 - a0 - wave.numel())])
        audio = self.w2v(wave).unsqueeze(0)
        sentence = " ".join(seg[1] for seg in merge_disco_textsegs(text_segments))
        ids = torch.as_tensor(self.tokenize(sentence), dtype=torch.long)
        return dict(audio=audio, raw_word=[sentence], text_features=[self.bert(ids)])

    def for_sample(self, raw_audio):
        """-> features(cidx, t0, t1, annotations) for LongformSynthesizer.run on that sample."""
        return lambda cidx, t0, t1, ann: self.window(raw_audio, t0, t1, ann["text_segments"][0])
