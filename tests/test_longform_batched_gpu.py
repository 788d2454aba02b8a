"""GPU: the batched long-form driver (longform.run_many: window k of ALL clips in one forward, per-clip prev-latent
chains; BASELINE config 5's 10 clips x N windows) against the per-clip loop of tools/longform_synthesis.py:256-403
(`run`, batch 1 per clip): clips never interact inside the model, so each clip's result must not depend on its
batch mates.  Clips of different lengths: the batch shrinks as clips run out of windows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _clip(rg, seeds):
    parts = [rg.synth.synth_batch(1, seed=s) for s in seeds]
    return {k: torch.cat([p[k] for p in parts], dim=1) for k in rg.longform.MOTION_KEYS + rg.longform.REPEAT_KEYS
            if k in parts[0] and torch.is_tensor(parts[0][k]) and parts[0][k].dim() >= 2 and parts[0][k].shape[1] == 150}


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_run_many_equals_per_clip_runs(rg, precision):
    cfg = rg.synth.default_model_cfg(num_layers=2)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder", num_layers=2)
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs), database=None, precision=precision)
    model.load_state_dict(rg.synth.synth_full_state(0, cfg, vae_cfgs))
    model.eval()
    clips = [_clip(rg, (11, 12)), _clip(rg, (13,)), _clip(rg, (14, 15, 16))]          # 300, 150, 430 frames: 3, 2, 4 windows (hop 135)
    clips[2] = {k: v[:, :430] for k, v in clips[2].items()}
    audio = lambda ci, cidx: rg.synth.synth_batch(1, seed=1000 + 10 * ci + cidx)["audio"]
    synth = rg.longform.LongformSynthesizer(model, overlap=15)
    copy = lambda d: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items()}
    many = synth.run_many([copy(c) for c in clips], lambda ci, cidx, t0, t1, ann: dict(audio=audio(ci, cidx), text_features=None),
                          noise_tape=rg.synth.ClipTapes([71, 72, 73]))
    assert sorted(many) == [0, 1, 2]
    assert [len(many[c]["windows"]) for c in range(3)] == [3, 2, 4]
    for ci, clip in enumerate(clips):
        one = synth.run(copy(clip), lambda cidx, t0, t1, ann, ci=ci: dict(audio=audio(ci, cidx), text_features=None),
                        noise_tape=rg.synth.NoiseTape(71 + ci))
        assert one["windows"] == many[ci]["windows"] and one["poses"].shape == many[ci]["poses"].shape
        for lat_a, lat_b in zip(one["latents"], many[ci]["latents"]):
            e = ((lat_a - lat_b).norm() / lat_b.norm()).item()
            assert e <= 1e-6, (ci, e)    # same kernels on the same rows; only the GEMM tile a row falls into changes
        for k in ("expressions", "trans"):
            d = np.abs(one[k] - many[ci][k]).max() / max(1e-9, np.abs(one[k]).max())
            assert d <= 1e-5, (ci, k, d)
        assert np.isfinite(many[ci]["poses"]).all()
        assert np.abs(one["poses"] - many[ci]["poses"]).max() <= 1e-4
