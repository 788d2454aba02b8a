"""Oracle: MotionDiffusion.forward, eval branch (test infrastructure, see oracle/__init__.py).

Restates mogen/models/architectures/diffusion_architecture.py:117-176, 213-582 with the
noise made explicit: `tape.draw(shape)` is called in exactly the order the reference
consumes torch's global generator.
"""
import torch

from . import denoiser as od
from . import diffusion as odf
from . import vae as ovae


def part_indices(T):
    n = (T - 3) // 4
    return (list(range(0, n)), list(range(n + 1, 2 * n + 1)), list(range(2 * n + 2, 3 * n + 2)),
            list(range(3 * n + 3, T)))


def motion_diffusion_forward(P, cfg, vae_cfgs, sch, data, tape, re_dict=None, use_inversion=False,
                             insertion_guidance=False, guidance_iters=None, guidance_lr=0.1,
                             use_prev_latent=False, prev_latent=None, outpaint=False,
                             inversion_start_time=-1, trace=None, inference_type="ddim", visualize_inversion=False):
    """Returns dict(pred_upper, pred_lower, pred_facepose, pred_hands, pred_transl, pred_exps,
    prev_latentout, latent_in, start_noise, inverted)."""
    if use_prev_latent:
        assert not outpaint
    if outpaint:
        assert not use_inversion and not insertion_guidance
    if insertion_guidance:
        assert use_inversion
    B = data["motion_upper"].shape[0]
    D = cfg["latent_dim"]
    eps_list = [tape.draw((B * 10, 1, D)) for _ in range(4)]
    motion, motion_mask = ovae.gesture_encode(P, vae_cfgs, data, eps_list)
    T = motion.shape[1]
    up_i, ha_i, fa_i, lt_i = part_indices(T)
    qm = od.make_query_masks(motion_mask)
    xf = od.encode_conditions(P, data["word"], data["audio"], data["speaker_ids"], cfg["num_speakers"])
    if callable(re_dict):  # RetrievalDatabase.forward runs here in the reference (its rsample draws come now)
        re_dict = re_dict(tape)

    def model_fn(xf_, qm_, mm_):
        return lambda x, t: od.denoiser_forward(P, cfg, x, t, mm_, xf_, qm_)

    if outpaint:
        retrieval_motion_latents = re_dict["raw_motion_latents"].squeeze(1)
    if use_prev_latent and prev_latent is not None:
        masked = torch.zeros_like(prev_latent)
        for idx in (up_i, ha_i, fa_i, lt_i):
            masked[:, idx[0]] = prev_latent[:, idx[-1]]
        prev_latent = masked

    start_noise = None
    invl = None
    inverted_all, recon_pairs = [], []
    if use_inversion:
        start_noise = tape.draw((B, T, D))
        n_lat = (T - 3) // 4
        per_batch = []
        for b in range(B):
            retr_se = re_dict["retr_startends"][b]
            query_se = re_dict["query_startends"][b]
            lats = re_dict["retr_uncropped_latents"][b]
            zero_inv = [torch.zeros((1, T, D)) for _ in range(sch.num_timesteps)]
            for q_idx in lats.keys():
                ex = lats[q_idx]
                xf_r = od.encode_conditions(P, ex["retr_text"], ex["retr_audio"], ex["retr_spkid"], cfg["num_speakers"])
                qm_r = {k: v[b:b + 1] for k, v in qm.items()}
                inv = odf.ddim_reverse_sample_loop(sch, model_fn(xf_r, qm_r, ex["retr_motion_mask"]),
                                                   ex["retr_motion_latent"])
                inverted_all.append(inv)
                if visualize_inversion:   # diffusion_architecture.py:357-382: DDIM reconstruction from the last level
                    recon_pairs.append((ex["retr_motion_latent"], odf.ddim_sample_loop(
                        sch, model_fn(xf_r, qm_r, ex["retr_motion_mask"]), inv[-1].clone(), tape.draw)))
                start_lat = inv[inversion_start_time]
                r0, r1 = retr_se[q_idx]
                q0, q1 = query_se[q_idx]
                assert r1 - r0 == q1 - q0
                start_noise[b:b + 1, q0:q1] = start_lat[:, r0:r1]
                start_noise[b:b + 1, n_lat + 1 + q0:n_lat + 1 + q1] = start_lat[:, n_lat + 1 + r0:n_lat + 1 + r1]
                if insertion_guidance:
                    for k, lat in enumerate(inv):
                        zero_inv[k][:, q0:q1] = lat[:, r0:r1]
                        zero_inv[k][:, n_lat + 1 + q0:n_lat + 1 + q1] = lat[:, n_lat + 1 + r0:n_lat + 1 + r1]
            if insertion_guidance:
                per_batch.append(torch.cat(zero_inv, dim=0))
        if insertion_guidance:
            invl = torch.stack(per_batch, dim=0).permute(1, 0, 2, 3).contiguous()
            if use_prev_latent and prev_latent is not None:
                for idx in (up_i, ha_i, fa_i, lt_i):
                    invl[:, :, idx[0], :] = 0

    model = model_fn(xf, qm, motion_mask)
    img = start_noise if use_inversion else tape.draw((B, T, D))
    if inference_type == "ddpm":   # diffusion_architecture.py:424-432: fresh start noise, ancestral sampling
        out = odf.p_sample_loop(sch, model, tape.draw((B, T, D)) if use_inversion else img, tape.draw)
    elif insertion_guidance:
        out = odf.ddim_guided_sample_loop(sch, model, img, tape.draw, guidance_iters, invl, guidance_lr,
                                          in_seq=prev_latent if use_prev_latent else None, trace=trace)
    elif use_prev_latent:
        out = odf.ddim_sample_loop(sch, model, img, tape.draw, in_seq=prev_latent, trace=trace)
    else:
        out = odf.ddim_sample_loop(sch, model, img, tape.draw,
                                   in_seq=retrieval_motion_latents if outpaint else None, trace=trace)
    up, lo, fa, ha, tr, ex, co = ovae.gesture_decode(P, vae_cfgs, out)
    extra = {}
    if use_inversion and visualize_inversion and inverted_all:
        S = sch.num_timesteps
        inv_lat = torch.stack([torch.cat(list(inv), dim=0) for inv in inverted_all], dim=0).reshape(-1, T, D)
        pair_lat = torch.stack([torch.cat(list(pr), dim=0) for pr in recon_pairs], dim=0).reshape(-1, T, D)
        for name, lat, k in (("inverted_output", inv_lat, S), ("reconspair_output", pair_lat, 2)):
            dec = ovae.gesture_decode(P, vae_cfgs, lat)
            for kname, t in zip(("upper", "lower", "facepose", "hands", "transl", "exps"), dec[:6]):
                extra["%s_%s" % (name, kname)] = t.reshape(len(inverted_all), k, t.shape[1], -1)
    return dict(extra, pred_upper=up, pred_lower=lo, pred_facepose=fa, pred_hands=ha, pred_transl=tr,
                pred_exps=ex, pred_contact=co, prev_latentout=out, latent_in=motion,
                latent_mask=motion_mask, start_noise=start_noise, inverted=inverted_all,
                inverted_latent_list=invl)


def synthetic_re_dict(B, seed, exemplars=((2, 5, 1, 4), (6, 8, 7, 9)), D=512, T=43):
    """A retrieval dict with the schema RetrievalDatabase.forward returns (raggesture.py:860-884):
    per clip, one exemplar per (retr_start, retr_end, query_start, query_end) latent span."""
    import numpy as np
    g = np.random.Generator(np.random.PCG64(seed))

    def n(shape, s=1.0):
        return torch.from_numpy((g.standard_normal(size=shape) * s).astype(np.float32))

    mask = torch.ones(1, T)
    mask[:, [10, 21, 32]] = 0
    retr_se, query_se, lats = [], [], []
    for b in range(B):
        rs, qs, ls = {}, {}, {}
        for q_idx, (r0, r1, q0, q1) in enumerate(exemplars):
            rs[q_idx] = (r0, r1)
            qs[q_idx] = (q0, q1)
            lat = n((1, T, D))
            lat[:, [10, 21, 32]] = 0
            ls[q_idx] = dict(retr_motion_latent=lat, retr_text=n((1, 150, 768)), retr_audio=n((1, 499, 768)),
                             retr_spkid=torch.full((1, 150), int(g.integers(0, 25)), dtype=torch.int64),
                             retr_motion_mask=mask.clone())
        retr_se.append(rs)
        query_se.append(qs)
        lats.append(ls)
    return dict(retr_startends=retr_se, query_startends=query_se, retr_uncropped_latents=lats)
