"""Retrieval: the oracle against the real reference's golden results (CPU), and the HIP sweep +
host tier walk against both (GPU).  Retrieval indices must match bit-exactly."""
import json
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import retrieval as oret


def _golden(golden_dir):
    with open(os.path.join(golden_dir, "retrieval_L2_allenc.json")) as f:
        return json.load(f)


def _db(rg, n=200):
    return rg.synth.synth_retrieval_samples(n, seed=2025)


def _unpack(q):
    si = {int(k): v for k, v in q["sample_indexes"].items()}
    db = {int(k): {n: tuple(b) for n, b in v.items()} for k, v in q["d_bounds"].items()}
    qb = {int(k): tuple(v) for k, v in q["query_bounds"].items()}
    return si, db, qb


def test_oracle_discourse_retrieval_matches_reference(rg, golden_dir):
    g = _golden(golden_dir)
    db = oret.build_db_dicts(_db(rg))
    for q in g["queries"]:
        qq = rg.synth.synth_query(q["seed"])
        si, dbb, qb = oret.discourse_retrieval(qq["discourse"], qq["prominence"], qq["speaker_id"], db, qq["text_features"])
        gsi, gdb, gqb = _unpack(q)
        assert si == gsi and dbb == gdb and qb == gqb
    assert any(len(set(v)) == 10 for v in _unpack(g["queries"][0])[0].values())


def test_product_host_logic_matches_oracle(rg):
    """map_conns_to_prominence / build_db_dicts / place_exemplars of the product == the oracle's."""
    smp = _db(rg, 120)
    a, b = oret.build_db_dicts(smp), rg.retrieval.build_db_dicts(smp)
    assert a["idx_2_prominence"] == b["idx_2_prominence"] and a["idx_2_sense"] == b["idx_2_sense"]
    assert a["idx_2_discbounds"] == b["idx_2_discbounds"]
    # stratified DB creation (raggesture.py:250-255): windows whose in-sequence index is a multiple of the interval
    c = rg.retrieval.build_db_dicts(smp, stratified_db_creation=True, stratification_interval=30)
    keep = [x["sample_name"] for x in smp if int(x["sample_name"].split("/")[1]) % 30 == 0]
    assert 0 < len(keep) < len(smp) and list(c["idx_2_sense"]) == keep
    assert all(c["idx_2_prominence"][n] == a["idx_2_prominence"][n] for n in keep)
    g = np.random.Generator(np.random.PCG64(5))
    for _ in range(300):
        ri, rb, qb = {}, {}, {}
        for qp in range(int(g.integers(1, 5))):
            qs = float(g.uniform(0.3, 9.9))
            qe = qs + float(g.uniform(-0.2, 2.0))
            rs = float(g.uniform(0, 10))
            re = rs + float(g.uniform(0.0, 3.0))
            ri[qp] = ["s%d" % qp] if g.uniform() > 0.1 else []
            qb[qp] = ("w", "t", qs, qe)
            rb[qp] = {"s%d" % qp: ("w", "t", round(rs, 3), round(min(re, 10.0), 3))}
        for method in ("discourse", "llm"):
            want = oret.place_exemplars(ri, rb, qb, method)
            got = {qp: (n, p[0], p[1]) for qp, n, p in rg.retrieval.place_exemplars(ri, rb, qb, method) if p is not None}
            assert want == got


def test_oracle_placement_matches_reference_forward(rg, golden_dir):
    g = _golden(golden_dir)
    for b, ent in enumerate(g["forward"]):
        own = ent["own"]
        t = g["test_indexes"][own]["discourse"]
        qq = rg.synth.synth_query(21 + b)
        db = oret.build_db_dicts(_db(rg))
        si, dbb, qb = oret.discourse_retrieval(qq["discourse"], qq["prominence"], qq["speaker_id"], db, qq["text_features"])
        assert {str(k): v for k, v in si.items()} == t
        sel = oret.select_retrieved(si, own, 1)
        placed = oret.place_exemplars(sel, dbb, qb, "discourse")
        assert {str(k): list(v[1]) for k, v in placed.items()} == ent["retr_startends"]
        assert {str(k): list(v[2]) for k, v in placed.items()} == ent["query_startends"]


class _FakeDataset:
    def __init__(self, rg, samples):
        self.rg, self.names, self.retrieval_samples = rg, [s["sample_name"] for s in samples], samples

    def __getitem__(self, key):
        name = self.names[0] if isinstance(key, int) else key
        d = self.rg.synth.synth_batch(1, seed=zlib.crc32(name.encode()) & 0x7FFFFFFF)
        out = {k: d[k][0] for k in ("motion", "motion_upper", "motion_lower", "motion_face", "motion_hands", "facial",
                                    "trans", "contact", "motion_mask", "word", "audio")}
        out["speaker_id"] = d["speaker_ids"][0]
        out["sample_name"] = name
        return out


@pytest.mark.gpu
def test_hip_sweep_matches_reference_golden(rg, golden_dir):
    g = _golden(golden_dir)
    index = rg.retrieval.DiscourseIndex(rg.retrieval.build_db_dicts(_db(rg)), "cuda")
    for q in g["queries"]:
        qq = rg.synth.synth_query(q["seed"])
        si, dbb, qb = rg.retrieval.discourse_retrieval(index, qq["discourse"], qq["prominence"], qq["speaker_id"],
                                                       qq["text_features"])
        gsi, gdb, gqb = _unpack(q)
        assert si == gsi, "retrieval indices differ from the reference"
        assert dbb == gdb and qb == gqb


@pytest.mark.gpu
def test_hip_sweep_matches_oracle_large_db(rg):
    """4096-entry DB, 12 queries: scores are bit-identical float64, indices exact."""
    smp = rg.synth.synth_retrieval_samples(4096, seed=7)
    db = oret.build_db_dicts(smp)
    index = rg.retrieval.DiscourseIndex(rg.retrieval.build_db_dicts(smp), "cuda")
    for seed in range(100, 112):
        qq = rg.synth.synth_query(seed)
        want = oret.discourse_retrieval(qq["discourse"], qq["prominence"], qq["speaker_id"], db, qq["text_features"])
        got = rg.retrieval.discourse_retrieval(index, qq["discourse"], qq["prominence"], qq["speaker_id"], qq["text_features"])
        assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2]


@pytest.mark.gpu
@pytest.mark.parametrize("n_entries,dense", [(50, False), (4096, False), (32768, False), (4096, True)])
def test_fused_sweep_selects_exactly_what_the_two_step_form_selects(rg, n_entries, dense):  # (dense: three times the relations per entry)
    """rg_discourse_select_fused (scores in registers, two launches) against rg_discourse_scores_batched +
    rg_select_top_scores_batched (scores through memory, four launches) and against the full score rows: the same survivors
    with bit-identical float64 scores and the same relation indices for every query relation of a batch (BASELINE DB size
    included; 50 entries: the `fewer than 64 entries: keep every positive score` branch; dense: every entry's relations three
    times over)."""
    import numpy as np
    smp = rg.synth.synth_retrieval_samples(n_entries, seed=2025)
    if dense:
        smp = [dict(s, discourse=list(s["discourse"]) * 3) for s in smp]
    index = rg.retrieval.DiscourseIndex(rg.retrieval.build_db_dicts(smp), "cuda")
    queries = []
    for i in range(16):
        q = rg.synth.synth_query(1000 + i)
        queries += rg.retrieval.discourse_queries(q["discourse"], q["prominence"], q["speaker_id"])
    assert len(queries) >= 16
    index.fused_sweep = True
    fused = index.collect(index.sweep_async(queries))
    index.fused_sweep = False
    plain = index.collect(index.sweep_async(queries))
    n_surv = 0
    for qi, ((fi, fs, ft), (pi, ps, pt)) in enumerate(zip(fused, plain)):
        assert np.array_equal(fi, pi) and np.array_equal(fs, ps) and np.array_equal(ft, pt), qi
        # ... and against the whole score row of the single-query sweep
        sc, tp = index.scores(*queries[qi])
        pos = np.sort(sc[sc > 0])[::-1]
        thr = pos[9] if (len(pos) >= 10 and n_entries > 64) else 0.0
        keep = np.nonzero((sc >= thr) & (sc > 0))[0]
        assert np.array_equal(fi, keep) and np.array_equal(fs, sc[keep]) and np.array_equal(ft, tp[keep]), qi
        n_surv += len(keep)
    assert n_surv > 0


@pytest.mark.gpu
def test_retrieval_database_forward_vs_reference(rg, golden_dir):
    g = _golden(golden_dir)
    lat = np.load(os.path.join(golden_dir, "retrieval_L2_allenc.npz"))
    samples = _db(rg)
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
    P = {}
    for i, part in enumerate(rg.synth.PARTS):
        P.update(rg.synth.synth_vae_state(101 + i, vae_cfgs[part], prefix="gesture_rep_encoder.%s_vae." % part))
    gre = rg.vae.GestureRepEncoder(P, vae_cfgs, "cuda", "fp32")
    rdb = rg.retrieval.RetrievalDatabase(num_retrieval=1, dataset=_FakeDataset(rg, samples), device="cuda")
    qs = [rg.synth.synth_query(21), rg.synth.synth_query(22)]
    own = [e["own"] for e in g["forward"]]
    cond = dict(text_features=[q["text_features"] for q in qs], discourse=[q["discourse"] for q in qs],
                prominence=[q["prominence"] for q in qs],
                speaker_ids=torch.tensor([[q["speaker_id"]] * 150 for q in qs]))
    re = rdb(cond, [150, 150], "cuda", idx=own, retrieval_method="discourse", gesture_rep_encoder=gre,
             noise=rg.synth.NoiseTape(4242))
    for b, ent in enumerate(g["forward"]):
        assert {str(k): list(v) for k, v in re["retr_startends"][b].items()} == ent["retr_startends"]
        assert {str(k): list(v) for k, v in re["query_startends"][b].items()} == ent["query_startends"]
        assert re["raw_sample_names"][b] == ent["names"]
        for k, v in re["retr_uncropped_latents"][b].items():
            ref = torch.from_numpy(lat["lat_%d_%d" % (b, k)])
            err = ((v["retr_motion_latent"].cpu() - ref).norm() / ref.norm()).item()
            assert err <= 1e-3, "exemplar latent (VAE encode with taped noise) differs: %g" % err
            assert torch.equal(v["retr_spkid"].cpu(), torch.from_numpy(lat["spk_%d_%d" % (b, k)]))
    # the raw_motion_latents template carries upper+hands rows only (raggesture.py:856-857)
    rml = re["raw_motion_latents"]
    assert rml.shape == (2, 1, 43, 512) and rml[:, :, 22:].abs().max() == 0


def test_llm_output_parser_and_response_cache(rg, golden_dir, tmp_path):
    """rag/llm_retrieval.py:131-165 (parser) against answers parsed by the real reference function; the response
    cache replays answers without calling out."""
    import json
    from oracle import retrieval as oret
    cases = json.load(open(os.path.join(golden_dir, "llm_parser.json")))
    assert len(cases) >= 8
    for c in cases:
        assert oret.parse_gesture_labels_from_llm_output(c["llm_output"]) == c["labels"]
        assert rg.retrieval.parse_gesture_labels_from_llm_output(c["llm_output"]) == c["labels"]
    calls = []

    def fake_llm(text):
        calls.append(text)
        return "[('%s', 'iconic'), ('so', 'beat')]" % text.split()[0]

    path = str(tmp_path / "llm_cache.json")
    cache = rg.retrieval.LLMResponseCache(path, call=fake_llm)
    assert cache.labels("round table talk") == [{"word": "round", "name": "iconic"}]
    assert cache.labels("round table talk") == [{"word": "round", "name": "iconic"}] and calls == ["round table talk"]
    assert cache.labels("   ") == [] and (cache.hits, cache.misses) == (1, 1)
    replay = rg.retrieval.LLMResponseCache(path)          # a new process: cached answers only
    assert replay.labels("round table talk") == [{"word": "round", "name": "iconic"}]
    with pytest.raises(KeyError):
        replay.get("never asked")


# ------------------------------------------------------------------ gesture_type retrieval
def _gesture_golden(golden_dir):
    with open(os.path.join(golden_dir, "gesture_type.json")) as f:
        return json.load(f)["queries"]


def _f32_sim(rg):
    return lambda a, b: np.float32(rg.synth.synth_word_similarity(a, b))


def test_oracle_gesture_type_retrieval_matches_reference(rg, golden_dir):
    """oracle/retrieval.py::gesture_type_retrieval against the real reference's outputs (make_goldens.py), with the
    word-similarity model replaced by the same deterministic stand-in on both sides (python float and float32)."""
    smp = _db(rg)
    db = oret.build_db_dicts(smp)
    assert "idx_2_gesture_labels" in db and len(db["idx_2_gesture_labels"]) == len(smp)
    for q in _gesture_golden(golden_dir):
        sim = rg.synth.synth_word_similarity if q["sim"] == "f64" else _f32_sim(rg)
        qq = rg.synth.synth_query(q["seed"])
        si, dbb, qb = oret.gesture_type_retrieval(q["labels"], qq["speaker_id"], db["idx_2_gesture_labels"],
                                                  qq["text_features"], db["idx_2_text"], sim)
        gsi, gdb, gqb = _unpack(q)
        assert si == gsi and dbb == gdb and qb == gqb
    # only beat labels / no labels: three empty dicts (gesture_type_retrieval.py:26-27)
    assert oret.gesture_type_retrieval([], 1, db["idx_2_gesture_labels"], None, db["idx_2_text"], None) == ({}, {}, {})


@pytest.mark.gpu
def test_hip_gesture_type_matches_reference_golden(rg, golden_dir):
    meta = rg.retrieval.build_db_dicts(_db(rg))
    gindex = rg.retrieval.GestureTypeIndex(meta, rg.retrieval.DiscourseIndex(meta, "cuda"))
    for q in _gesture_golden(golden_dir):
        sim = rg.synth.synth_word_similarity if q["sim"] == "f64" else _f32_sim(rg)
        qq = rg.synth.synth_query(q["seed"])
        si, dbb, qb = rg.retrieval.gesture_type_retrieval(gindex, q["labels"], qq["speaker_id"], qq["text_features"], sim)
        gsi, gdb, gqb = _unpack(q)
        assert si == gsi, "gesture_type retrieval indices differ from the reference"
        assert dbb == gdb and qb == gqb


@pytest.mark.gpu
def test_hip_gesture_type_matches_oracle_large_db(rg):
    """4096-entry DB: device scores (float64 and the float32 variant) give the oracle's ranking exactly."""
    smp = rg.synth.synth_retrieval_samples(4096, seed=7)
    db = oret.build_db_dicts(smp)
    meta = rg.retrieval.build_db_dicts(smp)
    gindex = rg.retrieval.GestureTypeIndex(meta, rg.retrieval.DiscourseIndex(meta, "cuda"))
    for seed in range(100, 110):
        qq = rg.synth.synth_query(seed)
        labels = rg.synth.synth_gesture_query(seed, n_labels=3)
        for sim in (rg.synth.synth_word_similarity, _f32_sim(rg)):
            want = oret.gesture_type_retrieval(labels, qq["speaker_id"], db["idx_2_gesture_labels"], qq["text_features"],
                                               db["idx_2_text"], sim)
            got = rg.retrieval.gesture_type_retrieval(gindex, labels, qq["speaker_id"], qq["text_features"], sim)
            assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2]


@pytest.mark.gpu
def test_retrieval_database_gesture_type_and_llm_methods(rg):
    """RetrievalDatabase.forward(retrieval_method="gesture_type"): exemplar choice and placement equal the oracle's
    retrieval + placement arithmetic (reduced padding for labels longer than 0.9 s, raggesture.py:628-636)."""
    smp = _db(rg)
    odb = oret.build_db_dicts(smp)
    names = [x["sample_name"] for x in smp]
    vae_cfgs = rg.synth.synth_vae_cfgs(decoder_arch="all_encoder")
    P = {}
    for i, part in enumerate(rg.synth.PARTS):
        P.update(rg.synth.synth_vae_state(101 + i, vae_cfgs[part], prefix="gesture_rep_encoder.%s_vae." % part))
    vae = rg.vae.GestureRepEncoder(P, vae_cfgs, "cuda", "fp32")
    rdb = rg.retrieval.RetrievalDatabase(num_retrieval=1, dataset=_FakeDataset(rg, smp), device="cuda",
                                         word_similarity=rg.synth.synth_word_similarity)
    B = 2
    qs = [rg.synth.synth_query(21), rg.synth.synth_query(22)]
    labels = [rg.synth.synth_gesture_query(21, 3), rg.synth.synth_gesture_query(22, 2)]
    cond = dict(text_features=[q["text_features"] for q in qs], discourse=[q["discourse"] for q in qs],
                prominence=[q["prominence"] for q in qs], gesture_labels=labels,
                speaker_ids=torch.tensor([[q["speaker_id"]] * 150 for q in qs]))
    re = rdb(cond, [150] * B, "cuda", idx=["clip_a", names[7]], retrieval_method="gesture_type", gesture_rep_encoder=vae)
    for b in range(B):
        si, dbb, qb = oret.gesture_type_retrieval(labels[b], qs[b]["speaker_id"], odb["idx_2_gesture_labels"],
                                                  qs[b]["text_features"], odb["idx_2_text"], rg.synth.synth_word_similarity)
        own = ["clip_a", names[7]][b]
        ri = {k: [s for s in v if s != own][:1] for k, v in si.items()}
        plan = oret.place_exemplars(ri, dbb, qb, "gesture_type")
        assert {k: tuple(v[1]) for k, v in plan.items()} == {k: tuple(v) for k, v in re["retr_startends"][b].items()}
        assert {k: tuple(v[2]) for k, v in plan.items()} == {k: tuple(v) for k, v in re["query_startends"][b].items()}
    # llm method through the same front door: cached answers, text / text_times from the conditions
    lq = [rg.synth.synth_llm_query(31), rg.synth.synth_llm_query(32)]
    cache = rg.retrieval.LLMResponseCache()
    for q in lq:
        cache.data[q["text"]] = q["llm_output"]
    rdb = rg.retrieval.RetrievalDatabase(num_retrieval=1, dataset=_FakeDataset(rg, smp), device="cuda",
                                         word_similarity=rg.synth.synth_word_similarity, llm_output=cache)
    cond = dict(text_features=[q["text_features"] for q in lq], discourse=[[], []], prominence=[q["prominence"] for q in lq],
                text=[q["text"] for q in lq], text_times=[q["text_times"] for q in lq], gesture_labels=[[], []],
                speaker_ids=torch.tensor([[q["speaker_id"]] * 150 for q in lq]))
    re = rdb(cond, [150] * B, "cuda", idx=["clip_a", "clip_b"], retrieval_method="llm", gesture_rep_encoder=vae)
    assert cache.hits == 2 and cache.misses == 0
    for b in range(B):
        q = lq[b]
        si, dbb, qb = oret.llm_retrieval(q["text"], q["text_times"], q["speaker_id"], q["prominence"],
                                         odb["idx_2_gesture_labels"], odb["idx_2_gestprom"], q["text_features"],
                                         odb["idx_2_text"], rg.synth.synth_word_similarity, cache.get)
        plan = oret.place_exemplars({k: v[:1] for k, v in si.items()}, dbb, qb, "llm")
        assert len(plan) > 0
        assert {k: tuple(v[1]) for k, v in plan.items()} == {k: tuple(v) for k, v in re["retr_startends"][b].items()}
        assert {k: tuple(v[2]) for k, v in plan.items()} == {k: tuple(v) for k, v in re["query_startends"][b].items()}


# ------------------------------------------------------------------ llm retrieval
def _llm_golden(golden_dir):
    with open(os.path.join(golden_dir, "llm_retrieval.json")) as f:
        return json.load(f)["queries"]


def _llm_query(rg, q):
    """The clip a golden entry was produced from: the stored fields (hand-made edge cases overwrite the synthetic
    ones), token features from the seed."""
    qq = rg.synth.synth_llm_query(q["seed"])
    qq.update(llm_output=q["llm_output"], text_times=[(tuple(t[0]), t[1]) for t in q["text_times"]],
              prominence=[tuple(p) for p in q["prominence"]], speaker_id=q["speaker_id"])
    return qq


def test_oracle_llm_retrieval_matches_reference(rg, golden_dir):
    """oracle/retrieval.py::llm_retrieval against the real reference function (get_llm_output replaced by the clip's
    canned answer, get_word_similarity_score by the deterministic stand-in; see make_goldens.py)."""
    smp = _db(rg)
    db = oret.build_db_dicts(smp)
    for q in _llm_golden(golden_dir):
        sim = rg.synth.synth_word_similarity if q["sim"] == "f64" else _f32_sim(rg)
        qq = _llm_query(rg, q)
        si, dbb, qb = oret.llm_retrieval(q["text"], qq["text_times"], qq["speaker_id"], qq["prominence"],
                                         db["idx_2_gesture_labels"], db["idx_2_gestprom"], qq["text_features"],
                                         db["idx_2_text"], sim, lambda t: qq["llm_output"])
        gsi, gdb, gqb = _unpack(q)
        assert si == gsi and dbb == gdb and qb == gqb
    # product host logic: label alignment == oracle's (synthetic clips and the hand-made golden cases)
    clips = [rg.synth.synth_llm_query(seed) for seed in range(1, 40)] + [_llm_query(rg, q) for q in _llm_golden(golden_dir)]
    for qq in clips:
        labs = oret.parse_gesture_labels_from_llm_output(qq["llm_output"])
        assert rg.retrieval.parse_gesture_labels_from_llm_output(qq["llm_output"]) == labs
        assert rg.retrieval.llm_query_bounds(labs, qq["text_times"]) == oret.llm_query_bounds(labs, qq["text_times"])[0]
    assert rg.retrieval.build_db_dicts(smp)["idx_2_gestprom"] == db["idx_2_gestprom"]


@pytest.mark.gpu
def test_hip_llm_retrieval_matches_reference_golden_and_oracle(rg, golden_dir):
    meta = rg.retrieval.build_db_dicts(_db(rg))
    gindex = rg.retrieval.GestureTypeIndex(meta, rg.retrieval.DiscourseIndex(meta, "cuda"))
    for q in _llm_golden(golden_dir):
        sim = rg.synth.synth_word_similarity if q["sim"] == "f64" else _f32_sim(rg)
        qq = _llm_query(rg, q)
        si, dbb, qb = rg.retrieval.llm_retrieval(gindex, q["text"], qq["text_times"], qq["speaker_id"], qq["prominence"],
                                                 qq["text_features"], sim, lambda t: qq["llm_output"])
        gsi, gdb, gqb = _unpack(q)
        assert si == gsi, "llm retrieval indices differ from the reference"
        assert dbb == gdb and qb == gqb
    # larger DB against the oracle, answers served from a response cache
    smp = rg.synth.synth_retrieval_samples(4096, seed=7)
    db, meta = oret.build_db_dicts(smp), rg.retrieval.build_db_dicts(smp)
    gindex = rg.retrieval.GestureTypeIndex(meta, rg.retrieval.DiscourseIndex(meta, "cuda"))
    cache = rg.retrieval.LLMResponseCache(call=None)
    for seed in range(100, 110):
        qq = rg.synth.synth_llm_query(seed)
        cache.data[qq["text"]] = qq["llm_output"]
        for sim in (rg.synth.synth_word_similarity, _f32_sim(rg)):
            want = oret.llm_retrieval(qq["text"], qq["text_times"], qq["speaker_id"], qq["prominence"],
                                      db["idx_2_gesture_labels"], db["idx_2_gestprom"], qq["text_features"],
                                      db["idx_2_text"], sim, cache.get)
            got = rg.retrieval.llm_retrieval(gindex, qq["text"], qq["text_times"], qq["speaker_id"], qq["prominence"],
                                             qq["text_features"], sim, cache.get)
            assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2]


def test_host_ranking_walk_matches_oracle_without_gpu(rg):
    """The host half of the product's discourse retrieval (survivor ordering, tier cut, tie-break ordering, the
    10-entry walk, bounds) against the oracle, on CPU: the device results it consumes (survivors of the top-score
    selection, tie-break similarities) are produced here from the oracle's own scores."""
    smp = _db(rg)
    db = oret.build_db_dicts(smp)
    names = list(db["idx_2_sense"].keys())

    class HostIndex:                      # what retrieval.py reads from a DiscourseIndex
        dev = torch.device("cpu")

        def __init__(self):
            self.names, self.db = names, db

        def sims_async(self, q, cand):   # rag/utils.py:109-121 in float64, like rg_text_diag_sim
            out = []
            for e in cand:
                f = db["idx_2_text"][names[e]][0].double()
                n = min(q.shape[0], f.shape[0])
                out.append((q[:n].double() * f[:n]).sum() / n if n else torch.tensor(0.0, dtype=torch.float64))
            return torch.stack(out)

    index = HostIndex()
    for seed in (11, 12, 13, 14, 21, 22, 100, 101, 102):
        q = rg.synth.synth_query(seed)
        trace = []
        want = oret.discourse_retrieval(q["discourse"], q["prominence"], q["speaker_id"], db, q["text_features"], trace)
        survivors = []
        for t in trace:
            sc = np.array([float(t["score"][n]) for n in names])
            pos = np.sort(sc[sc > 0])[::-1]
            thr = pos[9] if len(pos) >= 10 else 0.0
            keep = np.nonzero((sc >= thr) & (sc > 0))[0]
            survivors.append((keep, sc[keep], np.array([t["top"].get(names[e], -1) for e in keep])))
        got = rg.retrieval.discourse_retrieval(index, q["discourse"], q["prominence"], q["speaker_id"], q["text_features"],
                                               survivors=survivors)
        assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2]
