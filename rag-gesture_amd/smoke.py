"""__graft_entry__.smoke(): one small hot-path invocation on cuda:0, checked against the oracle."""
import numpy as np
import torch

from . import capi, pipeline, synth


def run():
    if not torch.cuda.is_available():
        raise capi.RgError("smoke() needs a GPU: the HIP path has no CPU fallback")
    from oracle import diffusion as odf, pipeline as opipe  # the checker (allowed here only)

    cfg = synth.default_model_cfg(num_layers=2)
    vae_cfgs = synth.synth_vae_cfgs(decoder_arch="all_encoder", num_layers=2)
    P = synth.synth_full_state(0, cfg, vae_cfgs)
    gi = [0] * 25 + list(range(25))
    keep = [r for r in range(43) if r not in (10, 20, 30)]
    rows = []
    for precision, tol in (("fp32", 2e-3), ("bf16", 1e-2)):
        model = pipeline.build_architecture(synth.reference_style_model_cfg(cfg, vae_cfgs), database=None,
                                            device="cuda:0", precision=precision)
        model.load_state_dict(P)
        model.eval()
        for tag, ikw, need_re in (("base", {}, False),
                                  ("guided", dict(use_inversion=True, insertion_guidance=True, guidance_iters=gi,
                                                  guidance_lr=0.1), True)):
            re_dict = opipe.synthetic_re_dict(1, seed=7) if need_re else None
            data = synth.synth_batch(1, seed=11)
            if need_re:
                data["re_dict"] = re_dict
            out = model(**dict(data, retrieval_method="discourse",
                               inference_kwargs=dict(ikw, noise_tape=synth.NoiseTape(3))))
            torch.cuda.synchronize()
            with torch.no_grad():
                ref = opipe.motion_diffusion_forward(P, cfg, vae_cfgs, odf.SpacedSchedule(), synth.synth_batch(1, seed=11),
                                                     synth.NoiseTape(3), re_dict=re_dict, **ikw)
            a, b = out["prev_latentout"].cpu()[:, keep], ref["prev_latentout"][:, keep]
            err = ((a - b).norm() / b.norm()).item()
            et = ((out["pred_transl"].cpu() - ref["pred_transl"]).norm() / ref["pred_transl"].norm()).item()
            worst = ((a - b).norm(dim=-1) / b.norm(dim=-1)).max().item()
            engine = ",".join(sorted({s.engine for s in model._sessions.values()})) or "-"
            rows.append((precision, tag, engine, err, worst, et))
            print("smoke %s %s (denoiser engine %s): final latent rel err %.3e (bound %.0e), worst token row %.3e, pred_transl %.3e"
                  % (precision, tag, engine, err, tol, worst, et))
            if not (err <= tol and worst <= 5 * tol and et <= 3 * tol):
                raise AssertionError("smoke: HIP path disagrees with the oracle (%s %s: %g, %g, %g)" % (precision, tag, err, worst, et))
    # the wide-launch form of the denoiser forward (rg_seq2_forward: two sequences of a kind per workgroup, what the pipelined
    # path launches) must give the bits of the one-workgroup-per-sequence form that the forwards above ran
    from . import denoiser, schedule
    from oracle import denoiser as od
    W = denoiser.DenoiserWeights(synth.synth_denoiser_state(0, cfg), cfg, schedule.Schedule(), "cuda:0")
    d3 = synth.synth_batch(3, seed=12)
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(4)).standard_normal((3, 43, 512)).astype(np.float32)).cuda()
    mm = torch.ones(3, 43)
    mm[:, [10, 21, 32]] = 0
    heads = []
    for kw in (dict(seq_duo=False), dict(seq_duo=True, seq_pairs=True)):
        sess = denoiser.DenoiserSession(W, 3, engine="seq", **kw)
        sess.set_conditions(d3["word"], d3["audio"], d3["speaker_ids"], mm, od.make_query_masks(mm))
        heads.append(sess.forward(x, 31, 7, 1).clone())
    torch.cuda.synchronize()
    if not (torch.isfinite(heads[1]).all() and torch.equal(heads[0], heads[1])):
        raise AssertionError("smoke: rg_seq2_forward differs from rg_seq_forward")
    print("smoke: rg_seq2_forward (two sequences per workgroup) == rg_seq_forward, bit for bit")
    print("smoke ok: " + "; ".join("%s/%s %.1e" % (p, t, e) for p, t, _, e, _, _ in rows))
