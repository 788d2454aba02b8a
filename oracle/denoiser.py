"""Oracle: ReGestureTransformer denoiser forward (test infrastructure, see oracle/__init__.py).

Functional restatement over a state dict `P` whose keys are the reference's
(SURVEY.md Appendix A).  torch fp32 on CPU; faithful to the reference's op order,
including the additive -1e6 masks and the per-step recomputation of the conditioning
K/V projections.
"""
import math

import torch
import torch.nn.functional as F

T_TOKENS = 43


# Checker-side switches (defaults = the reference's arithmetic):
#   bf16      : round both GEMM operands to bfloat16 (fp32 accumulate), i.e. what the HIP path's
#               MFMA GEMMs consume, so structural parity can be checked at a tight tolerance.
#   masked_ln : "torch" = F.layer_norm in fp32 on the -1e6-offset rows (implementation-defined
#               rounding of a mean near -1e6, see DESIGN.md "masked query rows");
#               "exact" = the same LayerNorm evaluated in fp64 on the fp32-quantised row.
# "hoist":      False = reference-faithful (the cross-attention K/V side is recomputed on every denoiser call, as the
#               reference does: SURVEY F7); True = it is computed once per conditioning tensor and reused (the
#               loop-invariant part hoisted) -- only bench.py's cpu_baseline "hoisted" mode sets it; hoist_clear()
#               drops the memo between clips.
OPTS = {"bf16": False, "masked_ln": "torch", "hoist": False}
_HOIST = {}


def hoist_clear():
    _HOIST.clear()


def linear(P, name, x):
    w = P[name + ".weight"]
    if OPTS["bf16"]:
        return F.linear(x.bfloat16().float(), w.bfloat16().float(), P[name + ".bias"])
    return F.linear(x, w, P[name + ".bias"])


def layer_norm(P, name, x):
    y = F.layer_norm(x, (x.shape[-1],), P[name + ".weight"], P[name + ".bias"], 1e-5)
    if OPTS["masked_ln"] == "exact":
        big = x.abs().amax(dim=-1) > 1e5
        if bool(big.any()):
            y64 = F.layer_norm(x.double(), (x.shape[-1],), P[name + ".weight"].double(),
                               P[name + ".bias"].double(), 1e-5).float()
            y = torch.where(big.unsqueeze(-1), y64, y)
    return y


def timestep_embedding(timesteps, dim, max_period=10000):
    """reference: mogen/models/transformers/diffusion_transformer.py:27-46"""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def stylization_block(P, name, h, emb):
    """reference: mogen/models/utils/stylization_block.py:29-40"""
    emb_out = linear(P, name + ".emb_layers.1", F.silu(emb)).unsqueeze(1)
    scale, shift = torch.chunk(emb_out, 2, dim=2)
    h = layer_norm(P, name + ".norm", h) * (1 + scale) + shift
    return linear(P, name + ".out_layers.2", F.silu(h))


def efficient_self_attention(P, name, x, src_mask, emb, num_heads):
    """reference: mogen/models/attentions/efficient_attention.py:23-45"""
    B, T, D = x.shape
    H = num_heads
    xn = layer_norm(P, name + ".norm", x)
    query = linear(P, name + ".query", xn)
    key = linear(P, name + ".key", xn) + (1 - src_mask) * -1000000
    query = F.softmax(query.view(B, T, H, -1), dim=-1)
    key = F.softmax(key.view(B, T, H, -1), dim=1)
    value = (linear(P, name + ".value", xn) * src_mask).view(B, T, H, -1)
    attention = torch.einsum("bnhd,bnhl->bhdl", key, value)
    y = torch.einsum("bnhd,bhdl->bnhl", query, attention).reshape(B, T, D)
    return x + stylization_block(P, name + ".proj_out", y, emb)


def efficient_cross_attention(P, name, x, xf, emb, query_mask, cond_type, num_heads):
    """reference: mogen/models/attentions/efficient_attention.py:62-102"""
    B, T, D = x.shape
    N = xf.shape[1]
    H = num_heads
    query = linear(P, name + ".query", layer_norm(P, name + ".norm", x))
    ck = None
    if OPTS.get("hoist"):   # cpu_baseline "hoisted" mode only: the K/V side of the block depends on the conditioning alone
        # (keyed on the tensor's content, not its address: exemplar conditionings are freed and reallocated)
        ck = (name, tuple(xf.shape), float(xf.sum()), float(xf.view(-1)[::997].abs().sum()),
              None if cond_type is None else tuple(cond_type.view(-1).tolist()))
    hit = ck is not None and ck in _HOIST
    if not hit:
        xfn = layer_norm(P, name + ".text_norm", xf)
        key = linear(P, name + ".key", xfn)
    query = F.softmax(query.view(B, T, H, -1), dim=-1)
    if hit:
        attention = _HOIST[ck]
    else:
        if cond_type is None:
            key = F.softmax(key.view(B, N, H, -1), dim=1)
            value = linear(P, name + ".value", xfn).view(B, N, H, -1)
        else:
            tct = ((cond_type % 10) > 0).float().view(B, 1, 1).repeat(1, N, 1)
            key = key + (1 - tct) * -1000000
            key = F.softmax(key.view(B, N, H, -1), dim=1)
            value = linear(P, name + ".value", xfn * tct).view(B, N, H, -1)
        attention = torch.einsum("bnhd,bnhl->bhdl", key, value)
        if ck is not None:
            _HOIST[ck] = attention
    y = torch.einsum("bnhd,bhdl->bnhl", query, attention)
    if query_mask is not None:
        y = y + (1 - query_mask).view(B, T, 1, 1) * -1000000
    y = y.reshape(B, T, D)
    return x + stylization_block(P, name + ".proj_out", y, emb)


COND_ORDER = ("xf_text", "xf_audio", "xf_spk")


def decoder_layer(P, name, x, xf, emb, src_mask, query_mask, cond_type, num_heads):
    """reference: mogen/models/transformers/diffusion_transformer.py:74-87, 105-127"""
    x = efficient_self_attention(P, name + ".sa_block", x, src_mask, emb, num_heads)
    outs = []
    for cond in xf.keys():  # dict order of get_precompute_condition: text, audio, spk
        qm = query_mask[cond] if query_mask is not None else None
        outs.append(efficient_cross_attention(P, name + ".ca_blocks." + cond, x, xf[cond], emb,
                                              qm, cond_type, num_heads))
    x = linear(P, name + ".ca_mix", torch.cat(outs, dim=-1))
    y = linear(P, name + ".ffn.linear2", F.gelu(linear(P, name + ".ffn.linear1", x)))
    return x + stylization_block(P, name + ".ffn.proj_out", y, emb)


def encode_conditions(P, text, audio, speaker_ids, num_speakers=25):
    """reference: raggesture.py:957-1013 + diffusion_transformer.py:544-606 (pre_proj only:
    pretrained_model=None, num_layers=0, use_text_proj=False in the shipped config)."""
    if num_speakers == 1:
        spk = torch.zeros((speaker_ids.shape[0], speaker_ids.shape[0], P["joint_embed.weight"].shape[0]))
    else:
        spk = F.embedding(speaker_ids, P["speaker_embedding.weight"])
    return {
        "xf_text": linear(P, "text_pre_proj", text),
        "xf_audio": linear(P, "audio_pre_proj", audio),
        "xf_spk": spk,
    }


def joint_scale_mask(per_joint_scale, T=T_TOKENS):
    """reference: raggesture.py:909-922"""
    n = (T - 3) // 4
    m = torch.ones(T)
    m[0:n] = per_joint_scale["upper"]
    m[n + 1:2 * n + 1] = per_joint_scale["hands"]
    m[2 * n + 2:3 * n + 2] = per_joint_scale["face"]
    m[3 * n + 3:T] = per_joint_scale["lowertransl"]
    return m


def scale_func_retr(scale_func_cfg, timestep):
    """reference: raggesture.py:925-954.  The random.randint branch picks between two
    coefficient sets that give the same mix; the first is returned."""
    w = (1 - (1000 - timestep) / 1000) * scale_func_cfg["coarse_scale"] + 1
    if timestep > 100:
        return dict(both_coef=w, text_coef=0, retr_coef=1 - w, none_coef=0)
    both, text, retr = (scale_func_cfg[k] for k in ("both_coef", "text_coef", "retr_coef"))
    return dict(both_coef=both, text_coef=text, retr_coef=retr, none_coef=1 - both - text - retr)


def embed_input(P, motion, n_lat):
    """reference: diffusion_transformer.py:646-659 (joint_embed + per-part sine PE + learned PE)."""
    h = linear(P, "joint_embed", motion)
    pos = P["sequence_embedding.pe"].permute(1, 0, 2)[:, :n_lat, :]
    sep = torch.zeros_like(pos[:, :1, :])
    h = h + torch.cat([pos, sep, pos, sep, pos, sep, pos], dim=1)
    return h + P["global_positional_embedding.pe"].permute(1, 0, 2)[:, :h.shape[1], :]


def denoiser_forward(P, cfg, motion, timesteps, motion_mask, xf_out, query_mask):
    """One model call at inference = DiffusionTransformer.forward (diffusion_transformer.py:620-668)
    + ReGestureTransformer.forward_test (raggesture.py:1041-1113).

    motion [B,T,D], timesteps [B] (ORIGINAL 0..999 indices), motion_mask [B,T],
    xf_out dict cond -> [B,N,D], query_mask dict cond -> [B,T].  Returns the CFG-mixed
    x0 prediction [B,T,D].
    """
    B, T, _ = motion.shape
    H, L = cfg["num_heads"], cfg["num_layers"]
    src_mask = motion_mask.clone().unsqueeze(-1)
    emb = linear(P, "time_embed.2", F.silu(linear(P, "time_embed.0",
                                                  timestep_embedding(timesteps, cfg["latent_dim"]))))
    h = embed_input(P, motion, (T - 3) // 4)

    # forward_test: CFG doubles the batch, rows [cond ; uncond]
    cond_type = torch.cat([torch.ones(B, 1, 1), torch.zeros(B, 1, 1)], dim=0)
    h = h.repeat(2, 1, 1)
    xf = {k: v.repeat(2, 1, 1) for k, v in xf_out.items()}
    emb = emb.repeat(2, 1)
    src_mask = src_mask.repeat(2, 1, 1)
    qm = {k: v.repeat(2, 1) for k, v in query_mask.items()} if query_mask is not None else None
    for l in range(L):
        h = decoder_layer(P, "temporal_decoder_blocks.%d" % l, h, xf, emb, src_mask, qm, cond_type, H)
    out = linear(P, "out", h).view(2 * B, T, -1)

    coef = scale_func_retr(cfg["scale_func_cfg"], int(timesteps[0]))
    out_text, out_none = out[:B], out[B:2 * B]
    js = joint_scale_mask(cfg["per_joint_scale"], T).unsqueeze(0).unsqueeze(-1).expand(B, -1, out.shape[-1])
    return (out_text * coef["both_coef"] * js + out_text * coef["text_coef"] * js
            + out_none * coef["retr_coef"] * (1 / js) + out_none * coef["none_coef"] * (1 / js))


def make_query_masks(motion_mask):
    """reference: diffusion_architecture.py:146-166 (separator rows masked as CA queries)."""
    T = motion_mask.shape[1]
    n = (T - 3) // 4
    m = torch.ones_like(motion_mask)
    m[:, [n, 2 * n, 3 * n]] = 0
    return {"xf_text": m, "xf_audio": m, "xf_spk": m.clone()}
