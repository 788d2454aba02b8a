"""Synthetic (random-init) weights and inputs at the reference's shapes.

There is no network on the build or GPU boxes, so the released checkpoint and the
BEAT-X data cannot be fetched; bench.py, smoke(), the tests and the golden-vector
generator all use the deterministic tensors made here.  Every tensor is drawn
from its own numpy Philox stream keyed by (seed, crc32(name)), so values do not
depend on generation order, on torch, or on the platform.

State-dict key names follow the reference module tree (SURVEY.md Appendix A;
reference: mogen/models/transformers/diffusion_transformer.py:335-420,
gesture_vae.py:25-98, utils/detr_utils.py:101-210,335-480), so the same dict
loads into the reference classes with strict=True (done by
tests/golden/make_goldens.py) and into this package's weight packer.
"""
import math
import zlib

import numpy as np
import torch

NFEATS = {"upper": 78, "hands": 180, "face": 106, "lowertrans": 61}
PARTS = ("upper", "hands", "face", "lowertrans")  # reference RNG/encode order


def default_model_cfg(num_layers=8):
    """Shapes of configs/raggesture_beatx/basegesture_len150_beat.py:32-160."""
    return dict(
        latent_dim=512, time_embed_dim=2048, num_heads=16, ff_size=1024,
        num_layers=num_layers, max_seq_len=150, frame_chunk_size=15,
        text_latent_dim=768, num_speakers=25,
        scale_func_cfg=dict(coarse_scale=6.5, both_coef=0.52351, text_coef=-0.28419,
                            retr_coef=2.39872),
        per_joint_scale=dict(upper=1.0, hands=1.0, face=1.0, lowertransl=1.0),
    )


def default_vae_cfg(part, decoder_arch="all_encoder", latent_dim=512, num_layers=8,
                    num_heads=4, ff_size=1024, position_embedding="learned",
                    normalize_before=False, activation="gelu"):
    """The VAE YAMLs ship with the weights, not with the repo; these are
    the survey's probe hyper-parameters."""
    return dict(
        latent_dim=latent_dim, num_heads=num_heads, ff_size=ff_size, num_layers=num_layers,
        decoder_arch=decoder_arch, position_embedding=position_embedding,
        nfeats=NFEATS[part], vae_dist="normal", num_frames=150, frame_chunk_size=15,
        transformer_activation=activation, transformer_normalize_before=normalize_before,
        dropout=0.0, test_ckpt="synthetic.bin",
    )


def _rng(seed, name):
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, zlib.crc32(name.encode())]))


def _uniform(seed, name, shape, bound):
    a = _rng(seed, name).uniform(-bound, bound, size=shape).astype(np.float32)
    return torch.from_numpy(a)


def _normal(seed, name, shape, std=1.0, mean=0.0):
    a = (_rng(seed, name).standard_normal(size=shape) * std + mean).astype(np.float32)
    return torch.from_numpy(a)


def _linear(sd, seed, name, out_f, in_f):
    b = 1.0 / math.sqrt(in_f)
    sd[name + ".weight"] = _uniform(seed, name + ".weight", (out_f, in_f), b)
    sd[name + ".bias"] = _uniform(seed, name + ".bias", (out_f,), b)


def _layernorm(sd, seed, name, d):
    sd[name + ".weight"] = _normal(seed, name + ".weight", (d,), 0.1, 1.0)
    sd[name + ".bias"] = _normal(seed, name + ".bias", (d,), 0.1)


def sine_pe(max_len, d_model):
    """PositionEmbeddingSine1D table (reference utils/detr_utils.py:33-41), [max_len,1,d]."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-np.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0).transpose(0, 1).contiguous()


def _stylization(sd, seed, name, d, te):
    _linear(sd, seed, name + ".emb_layers.1", 2 * d, te)
    _layernorm(sd, seed, name + ".norm", d)
    _linear(sd, seed, name + ".out_layers.2", d, d)  # zero_module in the reference; re-randomised


def synth_denoiser_state(seed=0, cfg=None, prefix=""):
    """ReGestureTransformer state dict (without the VAEs)."""
    cfg = cfg or default_model_cfg()
    d, te, ff = cfg["latent_dim"], cfg["time_embed_dim"], cfg["ff_size"]
    n_lat = cfg["max_seq_len"] // cfg["frame_chunk_size"]
    t_tok = 4 * n_lat + 3
    sd = {}
    bound = math.sqrt(6.0 / (t_tok + d))
    sd["global_positional_embedding.pe"] = _uniform(seed, "global_positional_embedding.pe",
                                                    (t_tok, 1, d), bound)
    sd["sequence_embedding.pe"] = sine_pe(n_lat, d)
    _linear(sd, seed, "text_pre_proj", d, cfg["text_latent_dim"])
    _linear(sd, seed, "audio_pre_proj", d, cfg["text_latent_dim"])
    sd["speaker_embedding.weight"] = _normal(seed, "speaker_embedding.weight",
                                             (cfg["num_speakers"], d), 1.0 / d)
    _linear(sd, seed, "joint_embed", d, d)
    _linear(sd, seed, "time_embed.0", te, d)
    _linear(sd, seed, "time_embed.2", te, te)
    _linear(sd, seed, "out", d, d)
    for l in range(cfg["num_layers"]):
        p = "temporal_decoder_blocks.%d." % l
        _layernorm(sd, seed, p + "sa_block.norm", d)
        for n in ("query", "key", "value"):
            _linear(sd, seed, p + "sa_block." + n, d, d)
        _stylization(sd, seed, p + "sa_block.proj_out", d, te)
        for c in ("xf_text", "xf_audio", "xf_spk"):
            q = p + "ca_blocks.%s." % c
            _layernorm(sd, seed, q + "norm", d)
            _layernorm(sd, seed, q + "text_norm", d)
            for n in ("query", "key", "value"):
                _linear(sd, seed, q + n, d, d)
            _stylization(sd, seed, q + "proj_out", d, te)
        _linear(sd, seed, p + "ca_mix", d, 3 * d)
        _linear(sd, seed, p + "ffn.linear1", ff, d)
        _linear(sd, seed, p + "ffn.linear2", d, ff)
        _stylization(sd, seed, p + "ffn.proj_out", d, te)
    return {prefix + k: v for k, v in sd.items()}


def _mha_block(sd, seed, name, d, ff, cross=False):
    xb = math.sqrt(6.0 / (4 * d))
    sd[name + ".self_attn.in_proj_weight"] = _uniform(seed, name + ".self_attn.in_proj_weight", (3 * d, d), xb)
    sd[name + ".self_attn.in_proj_bias"] = _normal(seed, name + ".self_attn.in_proj_bias", (3 * d,), 0.02)
    _linear(sd, seed, name + ".self_attn.out_proj", d, d)
    if cross:
        sd[name + ".multihead_attn.in_proj_weight"] = _uniform(seed, name + ".multihead_attn.in_proj_weight", (3 * d, d), xb)
        sd[name + ".multihead_attn.in_proj_bias"] = _normal(seed, name + ".multihead_attn.in_proj_bias", (3 * d,), 0.02)
        _linear(sd, seed, name + ".multihead_attn.out_proj", d, d)
    _linear(sd, seed, name + ".linear1", ff, d)
    _linear(sd, seed, name + ".linear2", d, ff)
    _layernorm(sd, seed, name + ".norm1", d)
    _layernorm(sd, seed, name + ".norm2", d)
    if cross:
        _layernorm(sd, seed, name + ".norm3", d)


def _skip_stack(sd, seed, name, d, ff, num_layers, cross=False):
    if num_layers % 2 == 0:
        num_layers += 1
    nb = (num_layers - 1) // 2
    for i in range(nb):
        _mha_block(sd, seed, "%s.input_blocks.%d" % (name, i), d, ff, cross)
    _mha_block(sd, seed, name + ".middle_block", d, ff, cross)
    for i in range(nb):
        _mha_block(sd, seed, "%s.output_blocks.%d" % (name, i), d, ff, cross)
        _linear(sd, seed, "%s.linear_blocks.%d" % (name, i), d, 2 * d)
    _layernorm(sd, seed, name + ".norm", d)


def synth_vae_state(seed, vcfg, prefix=""):
    """TransformerVAE state dict for one body part (reference gesture_vae.py:25-98)."""
    d, ff = vcfg["latent_dim"], vcfg["ff_size"]
    sd = {}
    sd["global_motion_token"] = _normal(seed, "global_motion_token", (2, d), 1.0)
    for n in ("query_pos_encoder", "query_pos_decoder", "mem_pos_decoder"):
        if vcfg["position_embedding"] == "learned":
            sd[n + ".pe"] = _uniform(seed, n + ".pe", (1024, 1, d), math.sqrt(6.0 / (1024 + d)))
        else:
            sd[n + ".pe"] = sine_pe(1024, d)
    _skip_stack(sd, seed, "encoder", d, ff, vcfg["num_layers"], cross=False)
    if vcfg["decoder_arch"] == "all_encoder":
        _skip_stack(sd, seed, "decoder", d, ff, vcfg["num_layers"], cross=False)
    else:
        _skip_stack(sd, seed, "decoder", d, ff, (vcfg["num_layers"] - 1) * 4 + 1, cross=True)
    _linear(sd, seed, "skel_embedding", d, vcfg["nfeats"])
    _linear(sd, seed, "final_layer", vcfg["nfeats"], d)
    return {prefix + k: v for k, v in sd.items()}


def synth_vae_cfgs(decoder_arch="all_encoder", **kw):
    return {p: default_vae_cfg(p, decoder_arch=decoder_arch, **kw) for p in PARTS}


def synth_full_state(seed=0, cfg=None, vae_cfgs=None):
    """Denoiser + 4 VAEs under the reference's sub-module names (no `model.` prefix)."""
    cfg = cfg or default_model_cfg()
    vae_cfgs = vae_cfgs or synth_vae_cfgs()
    sd = synth_denoiser_state(seed, cfg)
    for i, part in enumerate(PARTS):
        sd.update(synth_vae_state(seed + 101 + i, vae_cfgs[part],
                                  prefix="gesture_rep_encoder.%s_vae." % part))
    return sd


class NoiseTape:
    """Explicit noise source shared by the reference (patched), the oracle and the HIP path.

    The reference draws from torch's global generator in the order of SURVEY.md
    Appendix D; "same seed" therefore means "same noise tensors".  draw(shape) returns
    the next standard-normal fp32 tensor of a numpy PCG64 stream.
    """

    def __init__(self, seed):
        self._g = np.random.Generator(np.random.PCG64(seed))
        self.count = 0

    def draw(self, shape, device=None):
        self.count += 1
        t = torch.from_numpy(self._g.standard_normal(size=tuple(shape), dtype=np.float32))
        return t.to(device) if device is not None else t


class ClipTapes:
    """One NoiseTape per clip behind the batch interface: draw((n * k, ...)) for n clips = the clips' own draws of
    (k, ...) stacked in clip order.  A clip's noise then does not depend on which other clips share its batch: the
    batched long-form driver (longform.run_many) and a per-clip run consume identical numbers."""

    def __init__(self, seeds, clips=None):
        self.tapes = {c: NoiseTape(s) for c, s in (seeds.items() if isinstance(seeds, dict) else enumerate(seeds))}
        self.clips = list(self.tapes) if clips is None else list(clips)

    def for_clips(self, clips):
        out = ClipTapes({}, clips)
        out.tapes = self.tapes
        return out

    def draw(self, shape, device=None):
        n = len(self.clips)
        assert shape[0] % n == 0, (shape, n)
        per = (shape[0] // n,) + tuple(shape[1:])
        t = torch.cat([self.tapes[c].draw(per) for c in self.clips], dim=0)
        return t.to(device) if device is not None else t


def synth_batch(batch, seed=1234, device="cpu"):
    """Synthetic collated batch with the schema of mogen/datasets/builder.py:55-92 and the
    value distributions of SURVEY.md section 8(d)."""
    g = np.random.Generator(np.random.PCG64(seed))

    def u(shape, a):
        return torch.from_numpy(g.uniform(-a, a, size=shape).astype(np.float32))

    def n(shape, s):
        return torch.from_numpy((g.standard_normal(size=shape) * s).astype(np.float32))

    B = batch
    d = dict(
        motion_upper=u((B, 150, 39), 0.3), motion_lower=u((B, 150, 27), 0.3),
        motion_face=u((B, 150, 3), 0.3), motion_hands=u((B, 150, 90), 0.3),
        facial=n((B, 150, 100), 0.3), trans=n((B, 150, 3), 0.1),
        contact=torch.from_numpy((g.uniform(size=(B, 150, 4)) < 0.5).astype(np.float32)),
        motion_mask=torch.ones(B, 150),
        audio=n((B, 499, 768), 1.0), word=n((B, 150, 768), 1.0),
        speaker_ids=torch.zeros(B, 150, dtype=torch.int64),
    )
    d["motion"] = torch.zeros(B, 150, 165)
    d = {k: v.to(device) for k, v in d.items()}
    d.update(
        motion_length=[150] * B, raw_word=[[""] * 150 for _ in range(B)], raw_audio=None,
        text_features=[n((int(g.integers(12, 49)), 768), 1.0) for _ in range(B)],
        text_segments=[[] for _ in range(B)], gesture_labels=[[] for _ in range(B)],
        discourse=[[] for _ in range(B)], prominence=[[] for _ in range(B)],
        sample_idx=list(range(B)), sample_name=["synthetic_%04d" % i for i in range(B)],
    )
    return d


def reference_style_model_cfg(cfg=None, vae_cfgs=None, inference_type="ddim", with_retrieval=False):
    """`cfg.model` in the layout of configs/raggesture_beatx/basegesture_len150_beat.py:45-160, with the
    VAE YAML paths replaced by the hyper-parameter dicts themselves and `per_joint_scale` supplied
    (SURVEY F6)."""
    cfg = cfg or default_model_cfg()
    vae_cfgs = vae_cfgs or synth_vae_cfgs()
    d, te, H = cfg["latent_dim"], cfg["time_embed_dim"], cfg["num_heads"]
    return dict(
        type="MotionDiffusion",
        model=dict(
            type="ReGestureTransformer", input_feats=189, max_seq_len=cfg["max_seq_len"],
            frame_chunk_size=cfg["frame_chunk_size"], latent_dim=d, time_embed_dim=te,
            num_layers=cfg["num_layers"], body_part_cat_axis="time",
            sa_block_cfg=dict(type="EfficientSelfAttention", latent_dim=d, num_heads=H, dropout=0, time_embed_dim=te),
            ca_block_cfg=dict(type="EfficientCrossAttention", latent_dim=d, text_latent_dim=d, num_heads=H,
                              dropout=0, time_embed_dim=te),
            ffn_cfg=dict(latent_dim=d, ffn_dim=cfg["ff_size"], dropout=0, time_embed_dim=te),
            vae_cfg=dict(upper_cfg=vae_cfgs["upper"], lowertrans_cfg=vae_cfgs["lowertrans"], face_cfg=vae_cfgs["face"],
                         hands_cfg=vae_cfgs["hands"], latent_dim=d, frame_chunk_size=cfg["frame_chunk_size"]),
            text_encoder=dict(pretrained_model=None, latent_dim=cfg["text_latent_dim"], num_layers=0,
                              ff_size=2048, dropout=0, use_text_proj=False),
            audio_encoder=dict(pretrained_model=None, latent_dim=cfg["text_latent_dim"], num_layers=0, dropout=0.1),
            speaker_embedding=dict(num_speakers=cfg["num_speakers"]),
            retrieval_train=False,
            retrieval_cfg=(dict(motion_feat_dim=189, num_retrieval=1, topk=2, latent_dim=d, text_latent_dim=cfg["text_latent_dim"],
                                max_seq_len=cfg["max_seq_len"], motion_fps=15, motion_framechunksize=cfg["frame_chunk_size"],
                                lmdb_paths="experiments/retrieval_cache_stratified/", new_lmdb_cache=False)
                           if with_retrieval else None),
            use_retrieval_for_test=with_retrieval,
            scale_func_cfg=dict(cfg["scale_func_cfg"]), per_joint_scale=dict(cfg["per_joint_scale"]),
        ),
        loss_recon=dict(type="MSELoss", loss_weight=1, reduction="none"),
        body_part_lossweights=dict(upper=1.0, hands=1.0, face=1.0, lowertransl=1.0),
        diffusion_train=dict(beta_scheduler="scaled_linear", diffusion_steps=1000, model_mean_type="start_x",
                             model_var_type="fixed_large"),
        diffusion_test=dict(beta_scheduler="scaled_linear", diffusion_steps=1000, model_mean_type="start_x",
                            model_var_type="fixed_large", respace="15,15,8,6,6", num_inference_timesteps=50,
                            classifier_free_guidance_scale=0),
        inference_type=inference_type,
    )


# ---------------------------------------------------------------------------------------------
# Synthetic retrieval database (discourse relations, prominence, BERT-like token features)
SENSES = ("Contingency.Cause", "Comparison.Contrast", "Expansion.Conjunction", "Temporal.Synchronous")
_BASE_CONNS = ("and", "but", "because", "so", "when", "while", "then", "also", "however", "although",
               "as well", "in fact", "for example", "after", "before", "since", "if", "or", "yet", "still")


def connective_vocab(n=50):
    out = list(_BASE_CONNS)
    i = 0
    while len(out) < n:
        out.append("conn%02d" % i)
        i += 1
    return out[:n]


GESTURE_TYPES = ("beat", "iconic", "metaphoric", "deictic")
GESTURE_WORDS = ("round", "big", "this", "that one", "over there", "grow", "tiny", "spiral staircase", "you", "up", "down",
                 "together", "apart", "huge", "little bit")


def synth_word_similarity(a, b):
    """Deterministic stand-in for the reference's get_word_similarity_score (a fasttext / word2vec model,
    rag/utils.py:239-272): a symmetric pseudo-similarity in [0, 1) from the crc32 of the sorted word pair."""
    x, y = sorted((a, b))
    return (zlib.crc32((x + "|" + y).encode()) % 10007) / 10007.0


def synth_gesture_query(seed, n_labels=2):
    g = np.random.Generator(np.random.PCG64(seed + 31337))
    out, t = [], 0.4
    for k in range(n_labels):
        st = t + float(g.uniform(0.2, 2.5))
        en = st + float(g.uniform(0.3, 1.0))
        t = en
        name = GESTURE_TYPES[1 + int(g.integers(0, 3))] if k else "iconic"
        word = GESTURE_WORDS[int(g.integers(0, len(GESTURE_WORDS)))] if k else "round"
        if k == 2 or (k and seed % 2 == 0):     # words absent from the DB: the word-similarity branch decides
            word = ("gigantic", "very small", "Round")[(seed + k) % 3]
        out.append(dict(name=name, word=word, start=st, end=en))
    out.insert(1, dict(name="beat", word="so", start=0.1, end=0.3))   # beat labels are ignored by the method
    return out


LLM_FILLERS = ("well", "i", "think", "the", "was", "really", "and", "then", "we", "went", "it", "is", "like", "a", "of")


def synth_llm_query(seed):
    """One clip for the llm retrieval method: transcript words with timings ((start, end), word) as
    beatx_dataset's text_times, word prominence, and a canned LLM answer naming two of its gesture words."""
    g = np.random.Generator(np.random.PCG64(seed + 4242))
    picks = [GESTURE_WORDS[int(k)] for k in g.choice(len(GESTURE_WORDS), size=3, replace=False)]
    if seed % 2:
        picks[0] = "round"
    words = []
    for p in picks:
        words += [LLM_FILLERS[int(g.integers(0, len(LLM_FILLERS)))] for _ in range(int(g.integers(1, 4)))]
        words += p.split()
    words += [LLM_FILLERS[int(g.integers(0, len(LLM_FILLERS)))], picks[0].split()[0]]     # a repeated word
    times, prom, t = [], [], 0.2
    for w in words:
        st = t + float(g.uniform(0.0, 0.2))
        en = st + float(g.uniform(0.15, 0.5))
        t = en
        shown = w.capitalize() + "," if g.uniform() < 0.2 else w           # punctuation / case are stripped by the method
        times.append(((st, en), shown))
        if g.uniform() < 0.8:
            prom.append((w, st, en, float(g.uniform(0, 3))))
    types = [GESTURE_TYPES[1 + int(g.integers(0, 3))] for _ in picks]
    fmt = seed % 3
    if fmt == 0:
        ans = "[(\"%s\", \"%s\"), (\"%s\", \"%s\")]" % (picks[0], types[0], picks[1].title(), types[1])
    elif fmt == 1:
        ans = "1. '%s', %s\n2. '%s', %s\n3. 'unicorn', metaphoric" % (picks[1], types[1], picks[0], types[0])
    else:
        ans = "(%s, %s), (%s, beat), (%s, %s)" % (picks[2], types[2], words[0], picks[0], types[0])
    q = synth_query(seed)
    return dict(text=" ".join(w for _, w in times), text_times=times, prominence=prom, llm_output=ans,
                speaker_id=q["speaker_id"], text_features=q["text_features"])


def synth_llm_answer(text):
    """Stand-in for the GPT call of the llm retrieval method (rag/llm_retrieval.py:69-96): names the gesture words it
    finds in the window's transcript, in the list-of-tuples answer format.  Deterministic, so an LLMResponseCache filled
    with it during warm-up serves the timed / checked run from cache (BASELINE config 5: cached LLM calls)."""
    words = [w for w in GESTURE_WORDS if (" " + w + " ") in (" " + text.lower().replace(",", "") + " ")][:2]
    return "[" + ", ".join('("%s", "%s")' % (w, GESTURE_TYPES[1 + len(w) % 3]) for w in words) + "]"


def synth_longform_clip(seed, windows=3, hop_s=9.0, device=None):
    """One long-form sample of `windows` overlapping 150-frame windows (hop 135 frames = 9 s at 15 fps): the motion-side
    tensors of synth_batch chained along time plus the transcript annotations the llm method reads (text_segments with
    word timings, prominence), one synth_llm_query per window shifted to its start."""
    import torch
    parts = [synth_batch(1, seed=seed + w, device=device) for w in range(windows)]
    n = 135 * windows       # sample lengths in (135 (w - 1), 135 w] give w windows (longform_synthesis.py:262-265)
    clip = {k: torch.cat([p[k] for p in parts], dim=1)[:, :n] for k in parts[0]
            if torch.is_tensor(parts[0][k]) and parts[0][k].dim() >= 2 and parts[0][k].shape[1] == 150}
    segs, prom = [], []
    for w in range(windows):
        q = synth_llm_query(seed + 10 * w + 1)
        sh = hop_s * w + 0.3
        segs += [[[t[0][0] + sh, t[0][1] + sh], t[1]] for t in q["text_times"] if t[0][1] + sh < hop_s * (w + 1)]
        prom += [(p[0], p[1] + sh, p[2] + sh, p[3]) for p in q["prominence"] if p[2] + sh < hop_s * (w + 1)]
    clip["text_segments"], clip["prominence"] = [segs], [prom]
    clip["discourse"], clip["gesture_labels"] = [[]], [[]]
    clip["sample_name"] = ["9_longform_%d_0/0" % seed]
    return clip


def synth_retrieval_samples(n_entries, seed=2025, n_speakers=25, feat_dim=768, tie_groups=True, feat_device=None):
    """Raw per-sample records with the fields the reference's DB builder reads
    (raggesture.py:244-293): sample_name, speaker_id, discourse (8-tuples
    (conn, sense, arg1, arg2, start, end, conn_start, conn_end), beatx_dataset.py:1082-1093),
    prominence ((word, start, end, value)), text_feature [L, feat_dim].  Returned sorted by
    sample_name (the LMDB cursor order the reference iterates in)."""
    g = np.random.Generator(np.random.PCG64(seed))
    conns = connective_vocab()
    recs = []
    for i in range(n_entries):
        spk = int(g.integers(0, n_speakers))
        name = "%d_spk%02d_%d_%d/%d" % (spk, spk, int(g.integers(0, 3)), i, int(g.integers(0, 40)) * 15)
        nrel = int(g.integers(0, 5))
        disc, prom = [], []
        if tie_groups and i % 5 == 0:           # forced ties: identical categorical signature, no prominence
            nrel, spk = 1, 3
        t = 0.2
        for r in range(nrel):
            conn = conns[int(g.integers(0, len(conns)))] if not (tie_groups and i % 5 == 0) else "because"
            sense = SENSES[int(g.integers(0, len(SENSES)))] if not (tie_groups and i % 5 == 0) else SENSES[0]
            cs = t + float(g.uniform(0.0, 1.5))
            ce = cs + float(g.uniform(0.15, 0.6))
            t = ce
            disc.append((conn, sense, "arg1 text", "arg2 text", max(0.0, cs - 1.0), ce + 1.0, cs, ce))
            if not (tie_groups and i % 5 == 0) and g.uniform() < 0.75:   # prominence known for most connectives
                words = conn.split()
                dur = (ce - cs) / len(words)
                for wi, w in enumerate(words):
                    prom.append((w, cs + wi * dur, cs + (wi + 1) * dur, float(g.uniform(0, 3))))
        # some unrelated prominence words in between (at least one: the reference's
        # map_conns_to_prominence traps on an empty prominence list)
        for _ in range(int(g.integers(1, 3))):
            prom.append(("filler%d" % int(g.integers(0, 9)), float(g.uniform(0, 9)), float(g.uniform(0, 9)), float(g.uniform(0, 3))))
        prom.sort(key=lambda p: p[1])
        L = int(g.integers(8, 49))
        feat = L if feat_device is not None else torch.from_numpy(g.standard_normal((L, feat_dim)).astype(np.float32))
        recs.append(dict(sample_name=name, speaker_id=spk, discourse=disc, prominence=prom, text_feature=feat))
    # semantic gesture labels (beatx_dataset gesture_labels: name / word / start / end), from a generator of their own
    # so that the fields above (and the goldens pinned on them) do not move
    gl = np.random.Generator(np.random.PCG64(seed + 777))
    gp = np.random.Generator(np.random.PCG64(seed + 778))
    for i, r in enumerate(recs):
        labels, t = [], 0.1
        for _ in range(int(gl.integers(0, 4))):
            st = t + float(gl.uniform(0.0, 2.0))
            en = st + float(gl.uniform(0.2, 1.2))
            t = en
            labels.append(dict(name=GESTURE_TYPES[int(gl.integers(0, len(GESTURE_TYPES)))],
                               word=GESTURE_WORDS[int(gl.integers(0, len(GESTURE_WORDS)))], start=st, end=en))
        if tie_groups and i % 7 == 0:   # forced ties: same type, same word, same speaker as many other entries
            labels = [dict(name="iconic", word="round", start=1.0, end=1.6)]
        r["gesture_labels"] = labels
        # prominence of most gesture words (idx_2_gestprom, raggesture.py:270-272); appended behind the connectives'
        # records: no gesture word is a word of a connective, so idx_2_prominence does not move
        if not (tie_groups and i % 7 == 0):
            for lab in labels:
                if gp.uniform() < 0.7:
                    words = lab["word"].split()
                    dur = (lab["end"] - lab["start"]) / len(words)
                    for wi, w in enumerate(words):
                        r["prominence"].append((w, lab["start"] + wi * dur, lab["start"] + (wi + 1) * dur,
                                                float(gp.uniform(0, 3))))
    if feat_device is not None:
        # benchmark-sized DBs: draw all token features with one device-side generator call
        tot = sum(r["text_feature"] for r in recs)
        gen = torch.Generator(device=feat_device).manual_seed(seed)
        big = torch.randn(tot, feat_dim, device=feat_device, generator=gen)
        o = 0
        for r in recs:
            L = r["text_feature"]
            r["text_feature"] = big[o:o + L]
            o += L
    recs.sort(key=lambda r: r["sample_name"])
    return recs


class SyntheticDataset:
    """Stand-in for the BEAT-X train set handed to `build_architecture(cfg.model, database=...)`:
    `dataset[name]` returns the per-sample dict RetrievalDatabase.forward reads (raggesture.py:558-570),
    `retrieval_samples` are the raw DB records.  Samples are generated from crc32(name) and cached."""

    def __init__(self, n_entries, seed=2025, device="cpu", feat_device=None):
        self.device = device
        self.retrieval_samples = synth_retrieval_samples(n_entries, seed, feat_device=feat_device)
        self.names = [r["sample_name"] for r in self.retrieval_samples]
        self._cache = {}

    def __len__(self):
        return len(self.names)

    def __getitem__(self, key):
        name = self.names[0] if isinstance(key, int) else key
        if name not in self._cache:
            d = synth_batch(1, seed=zlib.crc32(name.encode()) & 0x7FFFFFFF, device=self.device)
            out = {k: d[k][0] for k in ("motion", "motion_upper", "motion_lower", "motion_face", "motion_hands", "facial",
                                        "trans", "contact", "motion_mask", "word", "audio")}
            out["speaker_id"] = d["speaker_ids"][0]
            out["sample_name"] = name
            self._cache[name] = out
        return self._cache[name]


def synth_query(seed, n_rel=3, n_speakers=25, feat_dim=768):
    g = np.random.Generator(np.random.PCG64(seed))
    conns = connective_vocab()
    disc, prom = [], []
    t = 0.3
    for r in range(n_rel):
        conn = conns[int(g.integers(0, len(conns)))] if r else "because"
        if r == 2 and conn == disc[1][0]:
            # relation 1 carries no prominence entry (the None path): the same connective again would hand relation 2's entry
            # to relation 1 and leave the map short, where the reference drops into a debugger (rag/utils.py:171-228)
            conn = conns[(conns.index(conn) + 1) % len(conns)]
        sense = SENSES[int(g.integers(0, len(SENSES)))] if r else SENSES[0]
        cs = t + float(g.uniform(0.2, 2.0))
        ce = cs + float(g.uniform(0.15, 0.6))
        t = ce
        disc.append((conn, sense, "a1", "a2", max(0.0, cs - 1.0), ce + 1.0, cs, ce))
        if r != 1:
            words = conn.split()
            dur = (ce - cs) / len(words)
            for wi, w in enumerate(words):
                prom.append((w, cs + wi * dur, cs + (wi + 1) * dur, float(g.uniform(0, 3))))
    L = int(g.integers(12, 49))
    return dict(discourse=disc, prominence=prom, speaker_id=3 if seed % 2 == 0 else int(g.integers(0, n_speakers)),
                text_features=torch.from_numpy(g.standard_normal((L, feat_dim)).astype(np.float32)))
