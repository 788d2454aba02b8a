"""GPU: BASELINE config 5 in miniature -- tools/longform_synthesis.py's loop (overlapping 150-frame windows, prev-latent
chaining, blend, 30 fps) with the llm retrieval method on cached LLM answers, inversion + insertion guidance.
Per window the retrieval result must equal the oracle's llm_retrieval on the same window annotations."""
import importlib

import numpy as np
import pytest
import torch

from oracle import retrieval as oret

pytestmark = pytest.mark.gpu


def _answer(rg):
    def call(text):   # stands for the GPT call: names the gesture words of the window
        words = [w for w in rg.synth.GESTURE_WORDS if (" " + w + " ") in (" " + text.lower().replace(",", "") + " ")][:2]
        return "[" + ", ".join('("%s", "%s")' % (w, rg.synth.GESTURE_TYPES[1 + len(w) % 3]) for w in words) + "]"
    return call


def test_longform_llm_guidance_windows(rg, tmp_path):
    cfg = rg.synth.default_model_cfg(num_layers=2)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder", num_layers=2)
    ds = rg.synth.SyntheticDataset(300, seed=31)
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs, with_retrieval=True), database=ds,
                                  precision="bf16")
    model.load_state_dict(rg.synth.synth_full_state(0, cfg, vae_cfgs))
    model.eval()
    rdb = model.model.database
    cache = rg.retrieval.LLMResponseCache(str(tmp_path / "llm_cache.json"), call=_answer(rg))
    rdb.word_similarity, rdb.llm_output = rg.synth.synth_word_similarity, cache.get
    assert rdb.gesture_index is not None and rdb.gesture_index.lab_prom is not None

    a, b = rg.synth.synth_batch(1, seed=11), rg.synth.synth_batch(1, seed=12)
    data = {k: torch.cat([a[k], b[k]], dim=1) for k in rg.longform.MOTION_KEYS + rg.longform.REPEAT_KEYS
            if k in a and torch.is_tensor(a[k]) and a[k].dim() >= 2 and a[k].shape[1] == 150}
    q0, q1 = rg.synth.synth_llm_query(51), rg.synth.synth_llm_query(52)
    shift = 10.2
    data["text_segments"] = [[[[t[0][0], t[0][1]], t[1]] for t in q0["text_times"]] +
                             [[[t[0][0] + shift, t[0][1] + shift], t[1]] for t in q1["text_times"]]]
    data["prominence"] = [list(q0["prominence"]) + [(p[0], p[1] + shift, p[2] + shift, p[3]) for p in q1["prominence"]]]
    data["discourse"], data["gesture_labels"] = [[]], [[]]
    data["sample_name"] = ["9_longform_0_0/0"]
    feats, seen = [rg.synth.synth_query(200 + i) for i in range(3)], {}

    def features(cidx, t0, t1, ann):
        text = " ".join(s[1] for s in ann["text_segments"][0])
        seen[cidx] = dict(text=text, ann=ann)
        return dict(audio=rg.synth.synth_batch(1, seed=100 + cidx)["audio"], raw_word=[text],
                    text_features=[feats[cidx]["text_features"]])

    synth = rg.longform.LongformSynthesizer(model, overlap=15)
    got = synth.run({k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()}, features, use_inversion=True,
                    insertion_guidance=True, guidance_iters=[2] * 25 + [0] * 25, guidance_lr=0.1, retrieval_method="llm",
                    noise_tape=rg.synth.NoiseTape(5))
    assert got["windows"] == [(0, 150), (135, 285), (270, 420)] and got["poses"].shape == (600, 165)
    for k in ("poses", "expressions", "trans"):
        assert np.isfinite(got[k]).all()
    # windows 0 and 1 carry text (one LLM call each, persisted); window 2 is padding: empty text, no call
    assert seen[2]["text"] == "" and cache.misses == 2 and cache.hits == 0
    assert len(rg.retrieval.LLMResponseCache(str(tmp_path / "llm_cache.json")).data) == 2
    odb = oret.build_db_dicts(ds.retrieval_samples)
    spk = int(data["speaker_ids"][0, 0])
    n_exemplars = 0
    for cidx in range(3):
        name = "9_longform_0_0/%d" % cidx
        ann = seen[cidx]["ann"]
        want = oret.llm_retrieval(seen[cidx]["text"], ann["text_segments"][0], spk, ann["prominence"][0],
                                  odb["idx_2_gesture_labels"], odb["idx_2_gestprom"], feats[cidx]["text_features"],
                                  odb["idx_2_text"], rg.synth.synth_word_similarity, cache.get)
        assert rdb.test_indexes[name]["llm"] == want[0], "window %d: llm retrieval differs from the oracle" % cidx
        assert rdb.test_dbounds[name]["llm"] == want[1] and rdb.test_qbounds[name]["llm"] == want[2]
        n_exemplars += len(want[0])
    assert n_exemplars >= 2
