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


@pytest.mark.parametrize("guided,batch_lanes,n_windows", [(True, 2, (5, 2, 5, 5)), (True, 4, (5, 2, 5, 5)), (False, 4, (5, 2, 5, 5)),
                                                          (True, 4, (4, 2, 1, 1)), (True, 2, (4, 2, 1, 1))])
def test_pipelined_windows_equal_the_sequential_loop(rg, guided, batch_lanes, n_windows, tmp_path):
    """run_many(pipelined=True): the windows go through submit() / flush() with the previous window's latent still pending
    (pipeline.PendingLatent) -- retrieval + exemplar inversion of window k + 1 beside the sampling loop of window k.  Same
    noise tape, clips of different lengths (the batch shrinks: the pending latent is row-selected), BASELINE config 5's
    flags (llm retrieval on cached answers, inversion + insertion guidance + prev-latent): every latent and every output
    must equal the sequential loop's, bit for bit.  Window batches rotate over `batch_lanes` lanes: with two, window k + 2 shares
    its launches with window k (co-batched chains); with four (the default) the inversions of the next windows run as chains
    of their own on the other lanes.  Window counts (4, 2, 1, 1): batches of 4, 2, 1, 1 clips -- several batches of fewer clips
    than lanes in a row (the lane of a batch must not depend on its size: ADVICE round 4)."""
    dev = torch.device("cuda", 0)
    cfg = rg.synth.default_model_cfg(num_layers=2)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder", num_layers=2)
    ds = rg.synth.SyntheticDataset(300, seed=31, device=dev, feat_device=dev)
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs, with_retrieval=True), database=ds, device=dev,
                                  batch_lanes=batch_lanes)
    model.load_state_dict(rg.synth.synth_full_state(0, cfg, vae_cfgs))
    model.eval()
    cache = rg.retrieval.LLMResponseCache(str(tmp_path / "llm_cache.json"), call=rg.synth.synth_llm_answer)
    model.model.database.llm_output = cache.get
    n_windows = list(n_windows)       # (5, 2, 5, 5): windows 0-1: 4 clips, 2-4: 3 clips (window 4 shares its launches with window 2)
    clips = [rg.synth.synth_longform_clip(40 + 100 * ci, windows=w, device=dev) for ci, w in enumerate(n_windows)]
    feats = {(ci, w): rg.synth.synth_query(300 + 10 * ci + w)["text_features"].to(dev) for ci, n in enumerate(n_windows) for w in range(n)}
    audio = {(ci, w): rg.synth.synth_batch(1, seed=1000 + 10 * ci + w, device=dev)["audio"] for ci, n in enumerate(n_windows) for w in range(n)}

    def features(ci, cidx, t0, t1, ann):
        text = " ".join(seg[1] for seg in ann["text_segments"][0])
        return dict(audio=audio[(ci, cidx)], raw_word=[text], text_features=[feats[(ci, cidx)]])

    synth = rg.longform.LongformSynthesizer(model, overlap=15)
    copy = lambda d: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items()}
    flags = dict(use_inversion=True, insertion_guidance=True, guidance_iters=[2] * 25 + [0] * 25, guidance_lr=0.1) if guided else {}
    seq = synth.run_many([copy(c) for c in clips], features, noise_tape=rg.synth.NoiseTape(71), retrieval_method="llm",
                         pipelined=False, **flags)
    model.async_results = True
    pip = synth.run_many([copy(c) for c in clips], features, noise_tape=rg.synth.NoiseTape(71), retrieval_method="llm", **flags)
    torch.cuda.synchronize()
    if guided and batch_lanes == 2 and max(n_windows) > 4:
        assert any(k[0] == "cobatch" for k in model._graphs), "the pipelined run should have gone through co-batched chains"
    assert not model._pend and not model._ready
    for ci in range(len(clips)):
        assert [len(seq[ci]["windows"]), len(pip[ci]["windows"])] == [n_windows[ci]] * 2
        for w, (a, b) in enumerate(zip(seq[ci]["latents"], pip[ci]["latents"])):
            assert torch.equal(a, b), (ci, w, (a - b).abs().max().item())
        for k in ("poses", "expressions", "trans"):
            assert np.array_equal(seq[ci][k], pip[ci][k]), (ci, k)
