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
    for precision, tol in (("fp32", 1e-2), ("bf16", 3e-2)):
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
            print("smoke %s %s: latent rel err %.3e, pred_transl rel err %.3e" % (precision, tag, err, et))
            if not (err <= tol and et <= 5 * tol):
                raise AssertionError("smoke: HIP path disagrees with the oracle (%s %s: %g, %g)" % (precision, tag, err, et))
    print("smoke ok")
