"""GPU: BASELINE config 5 in miniature -- tools/longform_synthesis.py's loop (overlapping 150-frame windows, prev-latent
chaining, blend, 30 fps) with the llm retrieval method on cached LLM answers, inversion + insertion guidance.
Per window the retrieval result must equal the oracle's llm_retrieval on the same window annotations."""
import importlib

import numpy as np
import pytest
import torch

from oracle import retrieval as oret

pytestmark = pytest.mark.gpu
KEEP = [r for r in range(43) if r not in (10, 20, 30)]


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


def test_config5_one_clip_three_windows_l8_vs_oracle(rg, tmp_path, parity):
    """BASELINE config 5 at full depth (8 denoiser layers, bf16), one clip x 3 windows, exactly the flags of
    tools/longform_synthesis.py:389-403 -- retrieval_method="llm" on cached LLM answers (default word similarity = the
    reference's effective fuzz.partial_ratio / 100), use_inversion + insertion_guidance + use_prev_latent per window,
    prev-latent chaining, overlap blend, 30 fps -- against the oracle pipeline driven by the same loop: retrieval results
    exact per window, latents / poses within the bf16 bars."""
    from oracle import diffusion as odf, fuzzy as ofz, packing as opk, pipeline as opipe, rotation as orot
    GI = [0] * 25 + list(range(25))
    cfg = rg.synth.default_model_cfg(num_layers=8)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
    ds = rg.synth.SyntheticDataset(300, seed=31)
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs, with_retrieval=True), database=ds, precision="bf16")
    P = rg.synth.synth_full_state(0, cfg, vae_cfgs)
    model.load_state_dict(P)
    model.eval()
    rdb = model.model.database
    cache = rg.retrieval.LLMResponseCache(str(tmp_path / "llm_cache.json"), call=rg.synth.synth_llm_answer)
    rdb.llm_output = cache.get
    clip = rg.synth.synth_longform_clip(40, windows=3)
    feats = [rg.synth.synth_query(200 + i) for i in range(3)]
    audio = [rg.synth.synth_batch(1, seed=100 + i)["audio"] for i in range(3)]
    seen = {}

    def features(cidx, t0, t1, ann):
        text = " ".join(s[1] for s in ann["text_segments"][0])
        seen[cidx] = dict(text=text, ann=ann)
        return dict(audio=audio[cidx], raw_word=[text], text_features=[feats[cidx]["text_features"]])

    synth = rg.longform.LongformSynthesizer(model, overlap=15)
    flags = dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1)
    seq = synth.run({k: (v.clone() if torch.is_tensor(v) else v) for k, v in clip.items()}, features, retrieval_method="llm",
                    noise_tape=rg.synth.NoiseTape(5), **flags)
    assert sum(bool(seen[c]["text"]) for c in range(3)) == 3 and cache.misses == 3
    # the throughput path: the same loop with the windows pipelined through submit() / flush() (window k + 1's retrieval and
    # inversion beside window k's sampling, its prev_latent bound late); this is what is compared with the oracle below
    model.async_results = True
    rdb.test_indexes.clear(); rdb.test_dbounds.clear(); rdb.test_qbounds.clear()
    got = synth.run({k: (v.clone() if torch.is_tensor(v) else v) for k, v in clip.items()}, features, retrieval_method="llm",
                    noise_tape=rg.synth.NoiseTape(5), **flags)
    assert any(k[0] == "cobatch" for k in model._graphs)
    for a, b in zip(seq["latents"], got["latents"]):
        assert torch.equal(a, b)
    assert np.array_equal(seq["poses"], got["poses"])
    assert got["windows"] == [(0, 150), (135, 285), (270, 420)] and got["poses"].shape == (810, 165)
    assert cache.misses == 3

    # ---- oracle loop (same noise tape order, same cached answers)
    odb = oret.build_db_dicts(ds.retrieval_samples)
    sim = lambda a, b: ofz.partial_ratio(a, b) / 100
    otape, sch = rg.synth.NoiseTape(5), odf.SpacedSchedule()
    starts, ends, rem = opk.window_bounds(405)
    od = rg.longform.pad_tail({k: (v.clone() if torch.is_tensor(v) else v) for k, v in clip.items()}, rem)
    spk = int(clip["speaker_ids"][0, 0])
    prev, so_far, n_ex = None, None, 0
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    for cidx, (c0, c1) in enumerate(zip(starts, ends)):
        chunk = {k: od[k][:, c0:c1] for k in od if torch.is_tensor(od[k])}
        chunk["audio"] = audio[cidx]
        ann, name = seen[cidx]["ann"], "9_longform_40_0/%d" % cidx
        want = oret.llm_retrieval(seen[cidx]["text"], ann["text_segments"][0], spk, ann["prominence"][0], odb["idx_2_gesture_labels"],
                                  odb["idx_2_gestprom"], feats[cidx]["text_features"], odb["idx_2_text"], sim, cache.get)
        assert rdb.test_indexes[name]["llm"] == want[0], "window %d: llm retrieval differs from the oracle" % cidx
        assert rdb.test_dbounds[name]["llm"] == want[1] and rdb.test_qbounds[name]["llm"] == want[2]
        n_ex += len(want[0])
        cond = dict(text_features=[feats[cidx]["text_features"]], speaker_ids=chunk["speaker_ids"])
        with torch.no_grad():
            o = opipe.motion_diffusion_forward(
                P, cfg, vae_cfgs, sch, chunk, otape, use_prev_latent=True, prev_latent=prev,
                re_dict=lambda tp: oret.database_forward(P, vae_cfgs, odb, ds, cond, [name], tp, retrieval_method="llm",
                                                         retrieve=lambda b: want), **flags)
        lat = got["latents"][cidx].cpu()
        parity.check("config 5 (llm, L8, bf16) window %d: final latent vs oracle" % cidx,
                     ((lat - o["prev_latentout"])[:, KEEP].norm() / o["prev_latentout"][:, KEEP].norm()).item(), 1.5e-2)
        prev = o["prev_latentout"]
        cur = (opk.scatter_parts(o["pred_upper"], o["pred_lower"], o["pred_hands"], o["pred_facepose"]), o["pred_exps"], o["pred_transl"])
        so_far = cur if cidx == 0 else opk.blend_window(*so_far, *cur, 15)
    assert cache.misses == 3 and cache.hits >= 3, "the oracle loop was served from the response cache"
    assert n_ex >= 2, "the windows should retrieve exemplars"
    want_m = opk.interp_motion(so_far[0], 2)[0, :810]
    want_f, want_t = opk.interp_features(so_far[1], 2)[0, :810], opk.interp_features(so_far[2], 2)[0, :810]
    rel = lambda x, y: ((x - y).norm() / y.norm()).item()
    ma = orot.axis_angle_to_matrix(torch.from_numpy(got["poses"]).reshape(-1, 3))
    parity.check("config 5 (llm, L8, bf16): 30-fps poses vs oracle loop (rotation matrices)", rel(ma, orot.axis_angle_to_matrix(want_m.reshape(-1, 3))), 3e-2)
    parity.check("config 5 (llm, L8, bf16): expressions", rel(torch.from_numpy(got["expressions"]), want_f), 3e-2)
    parity.check("config 5 (llm, L8, bf16): trans", rel(torch.from_numpy(got["trans"]), want_t), 3e-2)


def test_config5_two_clips_run_many_pipelined_l8_vs_oracle(rg, tmp_path, parity):
    """The batched, PIPELINED long-form driver (longform.run_many: window k of all clips in one forward, submitted through
    submit() / flush() with the previous window's latent still pending -- pipeline.PendingLatent) at full depth against
    the oracle driven by the loop of tools/longform_synthesis.py:256-403 over the same window batches (the oracle pipeline is
    batch-capable and consumes one noise tape in the reference's order, exactly like the product): two clips of different
    lengths (3 and 2 windows: the batch SHRINKS at window 2, the pending latent is row-selected, :389-403), the flags of
    BASELINE config 5 (llm retrieval on cached answers, inversion + insertion guidance + prev-latent).  Per clip and window:
    retrieval results exact, latents and the blended 30-fps outputs within the bf16 bars."""
    from oracle import diffusion as odf, fuzzy as ofz, packing as opk, pipeline as opipe, rotation as orot
    GI = [0] * 25 + list(range(25))
    cfg = rg.synth.default_model_cfg(num_layers=8)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
    ds = rg.synth.SyntheticDataset(300, seed=31)
    # (cobatch_lanes="split": every window batch is cut over the two lanes, so consecutive windows share their launches even with
    #  two clips; with whole batches alternating between the lanes a lane sees every other window only)
    model = rg.build_architecture(rg.synth.reference_style_model_cfg(cfg, vae_cfgs, with_retrieval=True), database=ds, precision="bf16",
                                  async_results=True, cobatch_lanes="split")
    P = rg.synth.synth_full_state(0, cfg, vae_cfgs)
    model.load_state_dict(P)
    model.eval()
    rdb = model.model.database
    cache = rg.retrieval.LLMResponseCache(str(tmp_path / "llm_cache.json"), call=rg.synth.synth_llm_answer)
    rdb.llm_output = cache.get
    n_windows, seeds = [3, 2], [40, 140]
    clips = [rg.synth.synth_longform_clip(s, windows=w) for s, w in zip(seeds, n_windows)]
    feats = {(ci, w): rg.synth.synth_query(200 + 10 * ci + w) for ci in range(2) for w in range(3)}
    audio = {(ci, w): rg.synth.synth_batch(1, seed=100 + 10 * ci + w)["audio"] for ci in range(2) for w in range(3)}
    seen = {}

    def features(ci, cidx, t0, t1, ann):
        text = " ".join(s[1] for s in ann["text_segments"][0])
        seen[(ci, cidx)] = dict(text=text, ann=ann)
        return dict(audio=audio[(ci, cidx)], raw_word=[text], text_features=[feats[(ci, cidx)]["text_features"]])

    synth = rg.longform.LongformSynthesizer(model, overlap=15)
    flags = dict(use_inversion=True, insertion_guidance=True, guidance_iters=GI, guidance_lr=0.1)
    copy = lambda d: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items()}
    got = synth.run_many([copy(c) for c in clips], features, retrieval_method="llm", noise_tape=rg.synth.NoiseTape(5), **flags)
    torch.cuda.synchronize()
    assert not model._pend and not model._ready
    assert [len(got[c]["windows"]) for c in range(2)] == n_windows
    assert any(k[0] == "cobatch" for k in model._graphs), "the pipelined run should have gone through co-batched chains"

    # ---- the oracle: window k of the clips that still have one as ONE oracle forward (same tape, same cached answers)
    odb = oret.build_db_dicts(ds.retrieval_samples)
    sim = lambda a, b: ofz.partial_ratio(a, b) / 100
    sch, otape = odf.SpacedSchedule(), rg.synth.NoiseTape(5)
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    rel = lambda x, y: ((x - y).norm() / y.norm()).item()
    st = []
    for clip in clips:
        sample_len = clip["motion"].shape[1]
        starts, ends, rem = opk.window_bounds(sample_len)
        st.append(dict(od=rg.longform.pad_tail(copy(clip), rem), starts=starts, ends=ends, n=sample_len, prev=None, so_far=None,
                       spk=int(clip["speaker_ids"][0, 0])))
    n_ex = 0
    for cidx in range(max(n_windows)):
        act = [ci for ci in range(2) if cidx < n_windows[ci]]
        chunks, wants, names = [], [], []
        for ci in act:
            c0, c1 = st[ci]["starts"][cidx], st[ci]["ends"][cidx]
            ch = {k: st[ci]["od"][k][:, c0:c1] for k in st[ci]["od"] if torch.is_tensor(st[ci]["od"][k])}
            ch["audio"] = audio[(ci, cidx)]
            chunks.append(ch)
            ann, name = seen[(ci, cidx)]["ann"], "9_longform_%d_0/%d" % (seeds[ci], cidx)
            f = feats[(ci, cidx)]["text_features"]
            want = oret.llm_retrieval(seen[(ci, cidx)]["text"], ann["text_segments"][0], st[ci]["spk"], ann["prominence"][0],
                                      odb["idx_2_gesture_labels"], odb["idx_2_gestprom"], f, odb["idx_2_text"], sim, cache.get)
            assert rdb.test_indexes[name]["llm"] == want[0], "clip %d window %d: llm retrieval differs from the oracle" % (ci, cidx)
            assert rdb.test_dbounds[name]["llm"] == want[1] and rdb.test_qbounds[name]["llm"] == want[2]
            n_ex += len(want[0])
            wants.append(want)
            names.append(name)
        batch = {k: torch.cat([c[k] for c in chunks], dim=0) for k in chunks[0]}
        cond = dict(text_features=[feats[(ci, cidx)]["text_features"] for ci in act], speaker_ids=batch["speaker_ids"])
        prev = None if cidx == 0 else torch.cat([st[ci]["prev"] for ci in act], dim=0)
        with torch.no_grad():
            o = opipe.motion_diffusion_forward(
                P, cfg, vae_cfgs, sch, batch, otape, use_prev_latent=True, prev_latent=prev,
                re_dict=lambda tp: oret.database_forward(P, vae_cfgs, odb, ds, cond, names, tp, retrieval_method="llm",
                                                         retrieve=lambda b: wants[b]), **flags)
        for j, ci in enumerate(act):
            lat, ref = got[ci]["latents"][cidx].cpu(), o["prev_latentout"][j:j + 1]
            parity.check("config 5 run_many pipelined (llm, L8, bf16) clip %d window %d: final latent vs oracle" % (ci, cidx),
                         ((lat - ref)[:, KEEP].norm() / ref[:, KEEP].norm()).item(), 1.5e-2)
            st[ci]["prev"] = ref
            cur = (opk.scatter_parts(o["pred_upper"][j:j + 1], o["pred_lower"][j:j + 1], o["pred_hands"][j:j + 1], o["pred_facepose"][j:j + 1]),
                   o["pred_exps"][j:j + 1], o["pred_transl"][j:j + 1])
            st[ci]["so_far"] = cur if cidx == 0 else opk.blend_window(*st[ci]["so_far"], *cur, 15)
    for ci in range(2):
        n_out = 2 * st[ci]["n"]
        so_far = st[ci]["so_far"]
        want_m = opk.interp_motion(so_far[0], 2)[0, :n_out]
        want_f, want_t = opk.interp_features(so_far[1], 2)[0, :n_out], opk.interp_features(so_far[2], 2)[0, :n_out]
        assert got[ci]["poses"].shape == (n_out, 165)
        ma = orot.axis_angle_to_matrix(torch.from_numpy(got[ci]["poses"]).reshape(-1, 3))
        parity.check("config 5 run_many pipelined clip %d: 30-fps poses vs oracle loop (rotation matrices)" % ci,
                     rel(ma, orot.axis_angle_to_matrix(want_m.reshape(-1, 3))), 3e-2)
        parity.check("config 5 run_many pipelined clip %d: expressions" % ci, rel(torch.from_numpy(got[ci]["expressions"]), want_f), 3e-2)
        parity.check("config 5 run_many pipelined clip %d: trans" % ci, rel(torch.from_numpy(got[ci]["trans"]), want_t), 3e-2)
    assert n_ex >= 3, "the windows should retrieve exemplars"
