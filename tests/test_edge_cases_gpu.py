"""GPU edge cases of the hot path against the oracle (batch of one, ragged clips, clips without
retrieved exemplars, empty discourse annotation)."""
import numpy as np
import pytest
import torch

from oracle import diffusion as odf, pipeline as opipe, retrieval as oret

pytestmark = pytest.mark.gpu
KEEP = [r for r in range(43) if r not in (10, 20, 30)]
GI = [0] * 25 + list(range(25))


def relerr(a, b):
    return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module")
def setup(rg):
    cfg = rg.synth.default_model_cfg(num_layers=2)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder", num_layers=2)
    P = rg.synth.synth_full_state(0, cfg, vae_cfgs)
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs), database=None, precision="fp32")
    model.load_state_dict(P)
    return cfg, vae_cfgs, P, model


def test_batch_of_one_base(rg, setup):
    """BASELINE config 0: tools/visualize.py base diffusion, batch = 1."""
    cfg, vae_cfgs, P, model = setup
    out = model(**dict(rg.synth.synth_batch(1, seed=5), retrieval_method="discourse",
                       inference_kwargs=dict(noise_tape=rg.synth.NoiseTape(1))))
    with torch.no_grad():
        ref = opipe.motion_diffusion_forward(P, cfg, vae_cfgs, odf.SpacedSchedule(), rg.synth.synth_batch(1, seed=5),
                                             rg.synth.NoiseTape(1))
    assert out["pred_upper"].shape == (1, 150, 39) and out["prev_latentout"].shape == (1, 43, 512)
    assert relerr(out["prev_latentout"].cpu()[:, KEEP], ref["prev_latentout"][:, KEEP]) <= 1e-2
    assert relerr(out["pred_transl"].cpu(), ref["pred_transl"]) <= 2e-2


def test_ragged_clip_lengths(rg, setup):
    """motion_mask with padded tails (clips shorter than 150 frames): masked latent tokens drop out of the
    self-attention keys/values (efficient_attention.py:33-36) and the exemplar inversion sees the same masks."""
    cfg, vae_cfgs, P, model = setup
    B = 3

    def make():
        d = rg.synth.synth_batch(B, seed=77)
        d["motion_mask"][1, 90:] = 0      # 6 of 10 latent tokens per part valid
        d["motion_mask"][2, 15:] = 0      # a single valid latent token per part
        d["motion_length"] = [150, 90, 15]
        d["re_dict"] = opipe.synthetic_re_dict(B, seed=3)
        return d

    ikw = dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1)
    out = model(**dict(make(), retrieval_method="discourse", inference_kwargs=dict(ikw, noise_tape=rg.synth.NoiseTape(2))))
    with torch.no_grad():
        ref = opipe.motion_diffusion_forward(P, cfg, vae_cfgs, odf.SpacedSchedule(), make(), rg.synth.NoiseTape(2),
                                             re_dict=opipe.synthetic_re_dict(B, seed=3), **ikw)
    lat, r = out["prev_latentout"].cpu(), ref["prev_latentout"]
    assert torch.isfinite(lat).all()
    for b in range(B):
        e = relerr(lat[b:b + 1, KEEP], r[b:b + 1, KEEP])
        print("ragged clip %d: final latent rel err %.3e" % (b, e))
        assert e <= 1e-2


def test_clips_without_exemplars_and_empty_discourse(rg):
    """A guided batch in which one clip has no discourse relation at all: it gets no inversion rows (start
    noise untouched, zero guidance canvas) while the others are guided; retrieval results and the final
    latents match the oracle chain."""
    cfg = rg.synth.default_model_cfg(num_layers=2)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder", num_layers=2)
    ds = rg.synth.SyntheticDataset(200, seed=5)
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs, with_retrieval=True), database=ds,
                                  precision="fp32")
    P = rg.synth.synth_full_state(0, cfg, vae_cfgs)
    model.load_state_dict(P)
    B = 3
    qs = [rg.synth.synth_query(7), rg.synth.synth_query(8), rg.synth.synth_query(9)]

    def make():
        d = rg.synth.synth_batch(B, seed=21)
        d["discourse"] = [qs[0]["discourse"], [], qs[2]["discourse"]]          # clip 1: nothing to retrieve
        d["prominence"] = [qs[0]["prominence"], [], qs[2]["prominence"]]
        d["text_features"] = [q["text_features"] for q in qs]
        d["speaker_ids"] = torch.tensor([[q["speaker_id"]] * 150 for q in qs])
        d["sample_name"] = ["q_a", "q_b", "q_c"]
        return d

    ikw = dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1)
    out = model(**dict(make(), retrieval_method="discourse", inference_kwargs=dict(ikw, noise_tape=rg.synth.NoiseTape(9))))
    rd = out["retrieval_dict"]
    assert len(rd["retr_startends"][1]) == 0 and len(rd["retr_uncropped_latents"][1]) == 0
    assert len(rd["retr_startends"][0]) > 0
    db = oret.build_db_dicts(ds.retrieval_samples)
    d2 = make()
    cond = dict(text_features=d2["text_features"], discourse=d2["discourse"], prominence=d2["prominence"],
                speaker_ids=d2["speaker_ids"])
    with torch.no_grad():
        ref = opipe.motion_diffusion_forward(
            P, cfg, vae_cfgs, odf.SpacedSchedule(), d2, rg.synth.NoiseTape(9),
            re_dict=lambda tp: oret.database_forward(P, vae_cfgs, db, ds, cond, d2["sample_name"], tp), **ikw)
    e = relerr(out["prev_latentout"].cpu()[:, KEEP], ref["prev_latentout"][:, KEEP])
    print("mixed batch (clip without exemplars): final latent rel err %.3e" % e)
    assert e <= 1e-2
