"""Generates tests/golden/e2e_ddpm_L2.npz from the REAL reference (imported from /root/reference through
tests/golden/_ref_import.py, build container only): `MotionDiffusion.forward` with
  * inference_type="ddpm"  (diffusion_architecture.py:424-432 -> gaussian_diffusion.py:741-905 p_sample_loop), base run;
  * inference_kwargs["visualize_inversion"] = True with use_inversion (diffusion_architecture.py:357-382, 488-571): the
    decoded inversion levels and (exemplar, DDIM reconstruction) pairs.
Same synthetic weights / inputs / noise tape as make_goldens.py's L2_allenc end-to-end runs (seeds only: the fixture
stores outputs).  Also checks the oracle restatement against the reference before writing."""
import contextlib
import os
import sys
sys.dont_write_bytecode = True   # /root/reference is read-only: no __pycache__ there
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402
import make_goldens as mg  # noqa: E402
from oracle import diffusion as odf, pipeline as opipe  # noqa: E402

synth = mg.synth


def main():
    ns = _ref_import.load_reference()
    torch.set_num_threads(8)
    cfg = synth.default_model_cfg(num_layers=2)
    vae_cfgs = synth.synth_vae_cfgs(decoder_arch="all_encoder")
    sch = odf.SpacedSchedule()
    res = {}
    B = 2
    with tempfile.TemporaryDirectory() as tmp:
        # ---- ddpm, base
        model, full = mg.build_reference_model(ns, cfg, vae_cfgs, 0, tmp, inference_type="ddpm")
        data = synth.synth_batch(B, seed=4321)
        with torch.no_grad(), mg.taped_noise(synth.NoiseTape(2024)), contextlib.redirect_stdout(open(os.devnull, "w")):
            ref = model(**dict(data, retrieval_method="discourse", inference_kwargs=dict()))
        mine = opipe.motion_diffusion_forward(full, cfg, vae_cfgs, sch, synth.synth_batch(B, seed=4321), synth.NoiseTape(2024),
                                              inference_type="ddpm")
        for k in ("prev_latentout", "pred_upper", "pred_hands", "pred_transl", "pred_exps"):
            print("ddpm", k, "oracle-vs-ref max abs %.3e |ref| %.3f" % ((ref[k] - mine[k]).abs().max().item(), ref[k].abs().max().item()))
            res["ddpm_%s" % k] = mg.t2n(ref[k])
        # ---- visualize_inversion (ddim model)
        model, full = mg.build_reference_model(ns, cfg, vae_cfgs, 0, tmp, inference_type="ddim")
        net = model.model
        re_dict = opipe.synthetic_re_dict(B, seed=77)
        net.database = lambda *a, **k: re_dict
        data = synth.synth_batch(B, seed=4321)
        with torch.no_grad(), mg.taped_noise(synth.NoiseTape(2024)), contextlib.redirect_stdout(open(os.devnull, "w")):
            ref = model(**dict(data, retrieval_method="discourse", inference_kwargs=dict(use_inversion=True, visualize_inversion=True)))
        mine = opipe.motion_diffusion_forward(full, cfg, vae_cfgs, sch, synth.synth_batch(B, seed=4321), synth.NoiseTape(2024),
                                              re_dict=re_dict, use_inversion=True, visualize_inversion=True)
        for k in ("prev_latentout", "inverted_output_upper", "inverted_output_transl", "reconspair_output_upper",
                  "reconspair_output_hands", "reconspair_output_exps"):
            print("visinv", k, tuple(ref[k].shape), "oracle-vs-ref max abs %.3e |ref| %.3f"
                  % ((ref[k] - mine[k]).abs().max().item(), ref[k].abs().max().item()))
        res["visinv_prev_latentout"] = mg.t2n(ref["prev_latentout"])
        # the decoded levels are large ([4, 50, 150, *]): keep levels 0 / 24 / 49 of the upper body and the pairs
        res["visinv_inverted_output_upper_lv"] = mg.t2n(ref["inverted_output_upper"][:, [0, 24, 49]])
        res["visinv_inverted_output_transl_lv"] = mg.t2n(ref["inverted_output_transl"][:, [0, 24, 49]])
        for k in ("reconspair_output_upper", "reconspair_output_transl", "reconspair_output_exps"):
            res["visinv_%s" % k] = mg.t2n(ref[k])
    np.savez_compressed(os.path.join(HERE, "e2e_ddpm_L2.npz"), **res)
    print("wrote e2e_ddpm_L2.npz:", {k: v.shape for k, v in res.items()})


if __name__ == "__main__":
    main()
