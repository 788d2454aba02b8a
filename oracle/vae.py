"""Oracle: body-part TransformerVAE encode/decode and the 4-part GestureRepEncoder
(test infrastructure, see oracle/__init__.py).

Restates mogen/models/transformers/gesture_vae.py, mogen/models/utils/detr_utils.py and
GestureRepEncoder (diffusion_transformer.py:131-330) functionally over a state dict.
torch.nn.MultiheadAttention is restated explicitly (in_proj split, 1/sqrt(hd) scaling,
softmax over keys with -inf on padded keys, out_proj).  Sequence-first layout [S,B,D]
as in the reference.
"""
import math

import torch
import torch.nn.functional as F

from . import rotation as rot

PARTS = ("upper", "hands", "face", "lowertrans")


def _lin(P, name, x):
    return F.linear(x, P[name + ".weight"], P[name + ".bias"])


def _ln(P, name, x):
    return F.layer_norm(x, (x.shape[-1],), P[name + ".weight"], P[name + ".bias"], 1e-5)


def _act(name):
    return {"relu": F.relu, "gelu": F.gelu}[name]


def multihead_attention(P, name, q_in, k_in, v_in, nhead, key_padding_mask=None):
    """torch.nn.MultiheadAttention forward (batch_first=False, no attn_mask, eval)."""
    W, b = P[name + ".in_proj_weight"], P[name + ".in_proj_bias"]
    D = W.shape[1]
    Sq, B, _ = q_in.shape
    Sk = k_in.shape[0]
    hd = D // nhead
    q = F.linear(q_in, W[:D], b[:D]).reshape(Sq, B * nhead, hd).transpose(0, 1)
    k = F.linear(k_in, W[D:2 * D], b[D:2 * D]).reshape(Sk, B * nhead, hd).transpose(0, 1)
    v = F.linear(v_in, W[2 * D:], b[2 * D:]).reshape(Sk, B * nhead, hd).transpose(0, 1)
    s = torch.bmm(q, k.transpose(1, 2)) / math.sqrt(hd)
    if key_padding_mask is not None:
        m = key_padding_mask.view(B, 1, 1, Sk).expand(-1, nhead, -1, -1).reshape(B * nhead, 1, Sk)
        s = s.masked_fill(m, float("-inf"))
    o = torch.bmm(F.softmax(s, dim=-1), v).transpose(0, 1).reshape(Sq, B, D)
    return _lin(P, name + ".out_proj", o)


def encoder_layer(P, name, src, nhead, act, pre_norm, kpm=None, pos=None):
    """reference: detr_utils.py:335-393 (TransformerEncoderLayer forward_post / forward_pre)."""
    wp = (lambda t: t) if pos is None else (lambda t: t + pos)
    if not pre_norm:
        q = k = wp(src)
        src = _ln(P, name + ".norm1", src + multihead_attention(P, name + ".self_attn", q, k, src, nhead, kpm))
        src2 = _lin(P, name + ".linear2", act(_lin(P, name + ".linear1", src)))
        return _ln(P, name + ".norm2", src + src2)
    src2 = _ln(P, name + ".norm1", src)
    q = k = wp(src2)
    src = src + multihead_attention(P, name + ".self_attn", q, k, src2, nhead, kpm)
    src2 = _ln(P, name + ".norm2", src)
    return src + _lin(P, name + ".linear2", act(_lin(P, name + ".linear1", src2)))


def decoder_layer(P, name, tgt, memory, nhead, act, pre_norm, tgt_kpm=None):
    """reference: detr_utils.py:396-480 (TransformerDecoderLayer, pos/query_pos = None)."""
    if not pre_norm:
        tgt = _ln(P, name + ".norm1", tgt + multihead_attention(P, name + ".self_attn", tgt, tgt, tgt, nhead, tgt_kpm))
        tgt = _ln(P, name + ".norm2", tgt + multihead_attention(P, name + ".multihead_attn", tgt, memory, memory, nhead))
        tgt2 = _lin(P, name + ".linear2", act(_lin(P, name + ".linear1", tgt)))
        return _ln(P, name + ".norm3", tgt + tgt2)
    t2 = _ln(P, name + ".norm1", tgt)
    tgt = tgt + multihead_attention(P, name + ".self_attn", t2, t2, t2, nhead, tgt_kpm)
    t2 = _ln(P, name + ".norm2", tgt)
    tgt = tgt + multihead_attention(P, name + ".multihead_attn", t2, memory, memory, nhead)
    t2 = _ln(P, name + ".norm3", tgt)
    return tgt + _lin(P, name + ".linear2", act(_lin(P, name + ".linear1", t2)))


def _num_blocks(num_layers):
    if num_layers % 2 == 0:
        num_layers += 1
    return (num_layers - 1) // 2


def skip_encoder(P, name, x, num_layers, nhead, act, pre_norm, kpm=None, pos=None):
    """reference: detr_utils.py:101-152 (SkipTransformerEncoder)."""
    nb = _num_blocks(num_layers)
    xs = []
    for i in range(nb):
        x = encoder_layer(P, "%s.input_blocks.%d" % (name, i), x, nhead, act, pre_norm, kpm, pos)
        xs.append(x)
    x = encoder_layer(P, name + ".middle_block", x, nhead, act, pre_norm, kpm, pos)
    for i in range(nb):
        x = _lin(P, "%s.linear_blocks.%d" % (name, i), torch.cat([x, xs.pop()], dim=-1))
        x = encoder_layer(P, "%s.output_blocks.%d" % (name, i), x, nhead, act, pre_norm, kpm, pos)
    return _ln(P, name + ".norm", x)


def skip_decoder(P, name, x, memory, num_layers, nhead, act, pre_norm, tgt_kpm=None):
    """reference: detr_utils.py:154-210 (SkipTransformerDecoder)."""
    nb = _num_blocks(num_layers)
    xs = []
    for i in range(nb):
        x = decoder_layer(P, "%s.input_blocks.%d" % (name, i), x, memory, nhead, act, pre_norm, tgt_kpm)
        xs.append(x)
    x = decoder_layer(P, name + ".middle_block", x, memory, nhead, act, pre_norm, tgt_kpm)
    for i in range(nb):
        x = _lin(P, "%s.linear_blocks.%d" % (name, i), torch.cat([x, xs.pop()], dim=-1))
        x = decoder_layer(P, "%s.output_blocks.%d" % (name, i), x, memory, nhead, act, pre_norm, tgt_kpm)
    return _ln(P, name + ".norm", x)


def vae_encode(P, vcfg, features, eps):
    """reference: gesture_vae.py:111-193 (encode_to_dist; lengths=None => no padding).
    features [B,150,nfeats]; eps [B*n_chunks,1,D] standard normal (the rsample draw).
    Returns z [B,n_chunks,D], mu, logvar."""
    bs, nframes, _ = features.shape
    chunk = vcfg["frame_chunk_size"]
    n_chunks = nframes // chunk
    D = vcfg["latent_dim"]
    x = features.reshape(bs * n_chunks, chunk, -1).permute(1, 0, 2)
    x = _lin(P, "skel_embedding", x)
    nb = bs * n_chunks
    dist = P["global_motion_token"][:, None, :].expand(-1, nb, -1)
    xseq = torch.cat((dist, x), dim=0)
    xseq = xseq + P["query_pos_encoder.pe"][:xseq.shape[0]]
    latent = skip_encoder(P, "encoder", xseq, vcfg["num_layers"], vcfg["num_heads"],
                          _act(vcfg["transformer_activation"]), vcfg["transformer_normalize_before"])[:2]
    latent = latent.permute(1, 0, 2)
    mu, logvar = latent[:, 0:1, :], latent[:, 1:, :]
    std = logvar.exp().pow(0.5)
    z = mu + std * eps  # Normal(mu, std).rsample()
    return z.reshape(bs, n_chunks, D), mu, logvar


def vae_decode(P, vcfg, z):
    """reference: gesture_vae.py:195-239 (lengths=None => num_frames each, no padding)."""
    bs, n_chunks, D = z.shape
    nframes = vcfg["num_frames"]
    act = _act(vcfg["transformer_activation"])
    pre = vcfg["transformer_normalize_before"]
    queries = torch.zeros(nframes, bs, D)
    zt = z.permute(1, 0, 2)
    if vcfg["decoder_arch"] == "all_encoder":
        xseq = torch.cat((zt, queries), dim=0)
        query_pos = xseq + P["query_pos_decoder.pe"][:xseq.shape[0]]
        out = skip_encoder(P, "decoder", xseq, vcfg["num_layers"], vcfg["num_heads"] * 8, act, pre,
                           pos=query_pos)[n_chunks:]
    else:
        queries = queries + P["query_pos_decoder.pe"][:nframes]
        mem = zt + P["mem_pos_decoder.pe"][:n_chunks]
        out = skip_decoder(P, "decoder", queries, mem, (vcfg["num_layers"] - 1) * 4 + 1,
                           vcfg["num_heads"] * 4, act, pre)
    out = _lin(P, "final_layer", out)
    return out.permute(1, 0, 2)


def sub_state(P, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in P.items() if k.startswith(prefix)}


def gesture_encode(P, vae_cfgs, data, eps_list):
    """reference: GestureRepEncoder.encode (diffusion_transformer.py:190-268).
    `data` holds motion_upper/lower/face/hands, trans, facial, contact, motion_mask;
    eps_list = 4 tensors [B*10,1,D] in the order upper, hands, face, lowertrans.
    NOTE: mutates data["trans"] in place exactly as the reference does (:231-232).
    Returns latent [B,43,D], mask [B,43]."""
    up6 = rot.aa_to_6d(data["motion_upper"], data["motion_upper"].shape[-1] // 3)
    lo6 = rot.aa_to_6d(data["motion_lower"], data["motion_lower"].shape[-1] // 3)
    ha6 = rot.aa_to_6d(data["motion_hands"], data["motion_hands"].shape[-1] // 3)
    fa6 = rot.aa_to_6d(data["motion_face"], data["motion_face"].shape[-1] // 3)
    tr = data["trans"]
    tr[:, :, 0] = tr[:, :, 0] - tr[:, 0:1, 0]
    tr[:, :, 2] = tr[:, :, 2] - tr[:, 0:1, 2]
    inputs = {
        "upper": up6, "hands": ha6, "face": torch.cat([fa6, data["facial"]], dim=-1),
        "lowertrans": torch.cat([lo6, tr, data["contact"]], dim=-1),
    }
    zs = {}
    for part, eps in zip(PARTS, eps_list):
        sp = sub_state(P, "gesture_rep_encoder.%s_vae." % part)
        zs[part] = vae_encode(sp, vae_cfgs[part], inputs[part], eps)[0]
    sep = torch.zeros_like(zs["upper"][:, :1, :])
    motion = torch.cat([zs["upper"], sep, zs["hands"], sep, zs["face"], sep, zs["lowertrans"]], dim=1)
    mm = data["motion_mask"][:, ::vae_cfgs["upper"]["frame_chunk_size"]]
    ms = torch.zeros_like(mm[:, :1])
    return motion, torch.cat([mm, ms, mm, ms, mm, ms, mm], dim=1)


def gesture_decode(P, vae_cfgs, z, joints=(13, 9, 1, 30)):
    """reference: GestureRepEncoder.decode (diffusion_transformer.py:270-330).
    joints = (upper, lower, face, hands) joint counts.  Returns the 7-tuple
    (upper, lower, facepose, hands, transl, exps, contact)."""
    uj, lj, fj, hj = joints
    n = (z.shape[1] - 3) // 4
    zp = {"upper": z[:, :n], "hands": z[:, n + 1:2 * n + 1], "face": z[:, 2 * n + 2:3 * n + 2],
          "lowertrans": z[:, 3 * n + 3:]}
    dec = {p: vae_decode(sub_state(P, "gesture_rep_encoder.%s_vae." % p), vae_cfgs[p], zp[p]) for p in PARTS}
    upper = rot.sixd_to_aa(dec["upper"], uj)
    hands = rot.sixd_to_aa(dec["hands"], hj)
    face = rot.sixd_to_aa(dec["face"][:, :, :fj * 6], fj)
    exps = dec["face"][:, :, fj * 6:]
    lt = dec["lowertrans"]
    lower = rot.sixd_to_aa(lt[:, :, :lj * 6], lj)
    transl = lt[:, :, lj * 6:lj * 6 + 3]
    contact = lt[:, :, lj * 6 + 3:]
    return upper, lower, face, hands, transl, exps, contact
