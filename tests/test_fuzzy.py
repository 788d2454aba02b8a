"""The reference's effective word similarity (rag/utils.py:239-272 -> fuzzywuzzy 0.18 fuzz.partial_ratio / 100):
oracle restatement, product host form, and the device form rg_partial_ratio.  Parity is pinned against the standard
library's difflib and hand-checkable pairs (fuzzywuzzy itself is not installed: unpinned against the package)."""
import json
import os
import random

import pytest
import torch

from oracle import fuzzy, retrieval as oret


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "fuzzy.json")) as f:
        return json.load(f)


def _random_pairs(n, seed, max_len=20, alpha="abcdeab "):
    rnd = random.Random(seed)
    return [("".join(rnd.choice(alpha) for _ in range(rnd.randint(0, max_len))),
             "".join(rnd.choice(alpha) for _ in range(rnd.randint(0, max_len)))) for _ in range(n)]


def test_oracle_on_goldens_and_restated_matcher(golden):
    for a, b, want in golden["hand"] + golden["pairs"]:
        assert fuzzy.partial_ratio(a, b) == want
        assert fuzzy.partial_ratio_restated(a, b) == want
        assert fuzzy.get_word_similarity_score(a, b) == want / 100
    # difflib-backed form == the matcher written out (what the kernel implements), argument order included
    for a, b in _random_pairs(3000, 5):
        assert fuzzy.partial_ratio(a, b) == fuzzy.partial_ratio_restated(a, b)
        assert fuzzy.partial_ratio(b, a) == fuzzy.partial_ratio_restated(b, a)


def test_product_host_form_equals_oracle(rg, golden):
    for a, b, want in golden["hand"] + golden["pairs"]:
        assert rg.retrieval.partial_ratio_host(a, b) == want
        assert rg.retrieval.fuzzy_word_similarity(a, b) == want / 100
    for a, b in _random_pairs(1000, 6):
        assert rg.retrieval.partial_ratio_host(a, b) == fuzzy.partial_ratio(a, b)


def _gindex(rg, words):
    """A GestureTypeIndex whose vocabulary is exactly `words` (one single-label entry per word)."""
    smp = rg.synth.synth_retrieval_samples(len(words), seed=3)
    for s, w in zip(smp, words):
        s["gesture_labels"] = [dict(name="iconic", word=w, start=0.5, end=1.0)]
    meta = rg.retrieval.build_db_dicts(smp)
    return rg.retrieval.GestureTypeIndex(meta, rg.retrieval.DiscourseIndex(meta, "cuda"))


@pytest.mark.gpu
def test_device_partial_ratio_matches_oracle(rg, golden):
    vocab = sorted({a for a, _, _ in golden["hand"] + golden["pairs"] if a}
                   | {a for a, _ in _random_pairs(400, 9, max_len=48) if a}
                   | {"x" * 48, "ab" * 24, "z" * 60, "long " * 12, "naïve", "日本語のテキスト", "tie", "eit"})
    gi = _gindex(rg, vocab)
    assert gi.vocab and set(gi.vocab) == set(vocab)
    queries = ["big", "this way", "abcab", "", "x" * 48, "y" * 55, "naïve", "日本語", "tie", "a b", "bcab abca"] + \
              [b for _, b in _random_pairs(40, 10, max_len=48)]
    sims = gi.fuzzy_similarities(queries).cpu()
    torch.cuda.synchronize()
    for qi, q in enumerate(queries):
        for v, w in enumerate(gi.vocab):
            want = fuzzy.get_word_similarity_score(w, q)      # (DB word, query word): the reference's argument order
            assert sims[qi, v].item() == want, (w, q, sims[qi, v].item(), want)


@pytest.mark.gpu
def test_llm_and_gesture_type_retrieval_with_the_default_similarity(rg):
    """RetrievalDatabase without a word_similarity callable: the reference's effective behaviour is the default, and
    `retrieval_method="llm"` needs nothing from the caller but the (cached) LLM answers."""
    smp = rg.synth.synth_retrieval_samples(2048, seed=7)
    db, meta = oret.build_db_dicts(smp), rg.retrieval.build_db_dicts(smp)
    cache = rg.retrieval.LLMResponseCache(call=None)
    rdb = rg.retrieval.RetrievalDatabase(metadata=meta, device="cuda", llm_output=cache)
    assert rdb.word_similarity is rg.retrieval.fuzzy_word_similarity
    for seed in range(200, 212):
        qq = rg.synth.synth_llm_query(seed)
        cache.data[qq["text"]] = qq["llm_output"]
        want = oret.llm_retrieval(qq["text"], qq["text_times"], qq["speaker_id"], qq["prominence"],
                                  db["idx_2_gesture_labels"], db["idx_2_gestprom"], qq["text_features"], db["idx_2_text"],
                                  fuzzy.get_word_similarity_score, cache.get)
        got = rdb.retrieve("llm", qq["text_features"], None, qq["prominence"], qq["speaker_id"], idx=None,
                           text=qq["text"], text_times=qq["text_times"])
        assert got[0] == {q: idxs[:1] for q, idxs in want[0].items()} and got[1] == want[1] and got[2] == want[2]
        labels = oret.parse_gesture_labels_from_llm_output(qq["llm_output"])
        if labels:
            gl = [dict(name=g["name"], word=g["word"], start=0.5 * i, end=0.5 * i + 0.4) for i, g in enumerate(labels)]
            want_g = oret.gesture_type_retrieval(gl, qq["speaker_id"], db["idx_2_gesture_labels"], qq["text_features"],
                                                 db["idx_2_text"], fuzzy.get_word_similarity_score)
            got_g = rdb.retrieve("gesture_type", qq["text_features"], None, None, qq["speaker_id"], gesture_labels=gl)
            assert got_g[1] == want_g[1] and got_g[2] == want_g[2]
            assert got_g[0] == {q: idxs[:1] for q, idxs in want_g[0].items()}
