"""Oracle: discourse-based exemplar retrieval and exemplar placement (test infrastructure, see
oracle/__init__.py).

Restates mogen/models/transformers/rag/discourse_retrieval.py:8-316,
rag/utils.py:86-132 (`sort_sidx_by_textsimilarity`), :171-228 (`map_conns_to_prominence`),
the DB dict construction raggesture.py:244-293, `RetrievalDatabase.retrieve` :470-477 and the
placement arithmetic of `RetrievalDatabase.forward` :542-760.
All score arithmetic is Python float64 in the reference's order; tie-break similarities are fp32
`torch.mm` diagonals like the reference.
"""
import copy

import torch


def _clean(s):
    return "".join([c for c in str(s) if c.isalnum() or c.isspace()])


def map_conns_to_prominence(conn_list, prominence_list):
    """reference: rag/utils.py:171-228"""
    relevant = {}
    residual = copy.deepcopy(conn_list)
    for dp in prominence_list:
        dp_word = _clean(dp[0])
        for si, sc in enumerate(conn_list):
            if si not in relevant:
                relevant[si] = []
            if residual[si] is None:
                continue
            sc = _clean(sc)
            if dp_word == sc or dp_word in sc.split():
                relevant[si].append((sc, dp[3]))
                if dp_word == sc or dp_word == sc.split()[-1]:
                    residual[si] = None
                break
    for si, dps in relevant.items():
        if len(dps) > 1:
            assert dps[0][0] == _clean(conn_list[si])
            relevant[si] = (conn_list[si], sum([d[1] for d in dps]) / len(dps))
        else:
            relevant[si] = dps[0] if len(dps) > 0 else None
    # the reference asserts len(relevant) == len(conn_list): an empty prominence list with a
    # non-empty connective list trips it; mirror the dict the loop would have produced otherwise
    if len(prominence_list) == 0:
        relevant = {si: None for si in range(len(conn_list))}
    assert len(relevant) == len(conn_list)
    return relevant


def build_db_dicts(samples):
    """reference: raggesture.py:255-276 (one record per DB sample, in iteration order)."""
    idx_2_text, idx_2_sense, idx_2_discbounds, idx_2_prominence = {}, {}, {}, {}
    for smp in samples:
        n, spk = smp["sample_name"], int(smp["speaker_id"])
        idx_2_text[n] = (smp["text_feature"], spk)
        idx_2_sense[n] = [spk] + [(d[1], d[0]) for d in smp["discourse"]]
        idx_2_discbounds[n] = [(d[1], d[0], d[4], d[5], d[6], d[7]) for d in smp["discourse"]]
        idx_2_prominence[n] = map_conns_to_prominence([d[0] for d in smp["discourse"]], smp["prominence"])
    out = dict(idx_2_text=idx_2_text, idx_2_sense=idx_2_sense, idx_2_discbounds=idx_2_discbounds,
               idx_2_prominence=idx_2_prominence)
    if all("gesture_labels" in smp for smp in samples):   # raggesture.py:262
        out["idx_2_gesture_labels"] = {smp["sample_name"]: [int(smp["speaker_id"])] + list(smp["gesture_labels"])
                                       for smp in samples}
        out["idx_2_gestprom"] = {smp["sample_name"]: map_conns_to_prominence([g["word"] for g in smp["gesture_labels"]],
                                                                             smp["prominence"]) for smp in samples}  # :270-272
    return out


def sort_sidx_by_textsimilarity(smp_indexes, encoded_text, feature_cache):
    """reference: rag/utils.py:86-132"""
    if len(smp_indexes) == 0:
        return smp_indexes
    sims = {}
    for s in smp_indexes:
        f = feature_cache[s][0]
        sims[s] = torch.diagonal(torch.mm(encoded_text, f.T)).mean()
    return sorted(sims, key=sims.get, reverse=True)


def discourse_retrieval(discourse, prominence, speaker_id, db, encoded_text, trace=None):
    """reference: rag/discourse_retrieval.py:8-316.  Returns (sample_indexes, d_bounds, query_bounds).
    trace (test aid): a list that receives, per query relation, {"score": name -> score, "top": name -> index of the
    relation whose bounds are reported} -- the state before the ranking walk."""
    d_bounds, sample_indexes, query_bounds = {}, {}, {}
    if len(discourse) == 0:
        return sample_indexes, d_bounds, query_bounds
    senses = [d[1] for d in discourse]
    conns = [d[0] for d in discourse]
    query_bounds = {i: (d[0].lower(), d[1], d[6], d[7]) for i, d in enumerate(discourse)}
    q_prom = map_conns_to_prominence(conns, prominence)
    for i, cv in q_prom.items():
        if cv is not None:
            q_prom[i] = (senses[i], cv[1])
    for qi, (q_sense, q_conn) in enumerate(zip(senses, conns)):
        score, rel_bounds, tops = {}, {}, {}
        for name, rec in db["idx_2_sense"].items():
            score[name] = 0
            spk, rels = rec[0], rec[1:]
            if len(rels) == 0:
                continue
            s_senses = [d[0] for d in rels]
            s_conns = [d[1] for d in rels]
            s_prom = {k: (None if v is None else (s_senses[k], v[1])) for k, v in db["idx_2_prominence"][name].items()}
            if q_sense in s_senses:
                score[name] += 2
                rel = [k for k, s in enumerate(s_senses) if s == q_sense]
                top, chosen = rel[0], False
                rel_conns = [s_conns[k] for k in rel]
                if q_conn in rel_conns:
                    score[name] += 4
                    top, chosen = rel[rel_conns.index(q_conn)], True
                if spk == speaker_id:
                    score[name] += 3
                ssum, cnt, diffs = 0, 0, {}
                for k in rel:
                    if s_prom[k] is None or q_prom[qi] is None:
                        continue
                    diff = abs(s_prom[k][1] - q_prom[qi][1])
                    diffs[k] = diff
                    ssum += 4 / (1 + 2 * diff)
                    cnt += 1
                if cnt > 0:
                    score[name] += ssum / cnt
                    best = sorted(diffs, key=diffs.get)
                    if top != best[0] and not chosen:
                        top = best[0]
                rel_bounds[name] = db["idx_2_discbounds"][name][top]
                tops[name] = top
        if trace is not None:
            trace.append(dict(score=dict(score), top=dict(tops)))
        order = sorted(score, key=score.get, reverse=True)
        tiers = {}
        for name in order:
            tiers.setdefault(score[name], [])
            if score[name] > 0:
                tiers[score[name]].append(name)
        ranked = []
        for sc in sorted(tiers.keys(), reverse=True):
            tier = tiers[sc]
            if len(tier) > 1:
                tier = sort_sidx_by_textsimilarity(tier, encoded_text, db["idx_2_text"])
            ranked += tier
            if len(ranked) >= 10:
                break
        sample_indexes[qi] = ranked[:10]
        d_bounds[qi] = {}
        for name in ranked[:10]:
            b = rel_bounds[name]
            d_bounds[qi][name] = (b[1], b[0], round(b[4], 3), round(b[5], 3))
    return sample_indexes, d_bounds, query_bounds


def select_retrieved(sample_indexes, own_name, num_retrieval=1):
    """reference: raggesture.py:470-477"""
    return {q: [s for s in idxs if s != own_name][:num_retrieval] for q, idxs in sample_indexes.items()}


def place_exemplars(retr_indexes, retr_bounds, query_bounds, retrieval_method="discourse", fps=15, chunk=15,
                    motion_len=150):
    """reference: raggesture.py:542-760 (time -> latent-index arithmetic only).
    Returns {query_point: (sample_name, (retr_lat_start, retr_lat_end), (start_lat, end_lat))} in
    placement order."""
    latent_len = motion_len // chunk
    prev_end = -1
    out = {}
    for qp, smp_idxs in retr_indexes.items():
        if len(smp_idxs) == 0 or qp not in query_bounds:
            continue
        _, _, q_start, q_end = query_bounds[qp]
        if q_start > q_end:
            continue
        for smp in smp_idxs:
            _, _, r_start, r_end = retr_bounds[qp][smp]
            q_start = max(0, q_start)
            q_end = min(motion_len / fps, q_end)
            q_start, q_end = int(q_start * fps), int(q_end * fps)
            q_lat_start, q_lat_end = q_start // chunk, q_end // chunk + 1
            assert q_lat_start < q_lat_end
            if retrieval_method in ("gesture_type", "llm") and (r_end - r_start) > 0.9:
                r_start, r_end = max(0, r_start - 0.2), min(motion_len / fps, r_end + 0.1)
            else:
                r_start, r_end = max(0, r_start - 0.666), min(motion_len / fps, r_end + 0.333)
            r_start, r_end = int(r_start * fps), int(r_end * fps)
            if r_start == r_end:
                continue
            if r_end == motion_len:
                r_end = motion_len - 1
                r_start = max(0, r_start - 1)
            r_lat_start, r_lat_end = r_start // chunk, r_end // chunk + 1
            mid_lat = ((q_start + q_end) // 2) // chunk
            n = r_lat_end - r_lat_start
            assert n > 0
            side = n // 2
            if n == 1:
                s, e = mid_lat - side, mid_lat + side + 1
            elif n == 2:
                s, e = mid_lat, mid_lat + side + 1
            elif n % 2 == 1:
                s, e = mid_lat - side - 1, mid_lat + side
            else:
                s, e = mid_lat - side, mid_lat + side
            if s < 0:
                s, e = 0, n
            if e > latent_len:
                s -= e - latent_len
                e = latent_len
            if s < prev_end:
                s = prev_end
                e = s + n
                if e > latent_len:
                    e = latent_len
                    n = e - s
                    if n <= 0:
                        continue
                    r_lat_end = r_lat_start + n
            prev_end = e
            out[qp] = (smp, (r_lat_start, r_lat_end), (s, e))
    return out


def database_forward(P, vae_cfgs, db, dataset, conditions, own_names, tape, retrieval_method="discourse", retrieve=None):
    """reference: RetrievalDatabase.forward (raggesture.py:479-884), the parts the sampler consumes:
    per clip retrieve -> select -> fetch + VAE-encode each visited exemplar (4 rsample draws each, in
    visiting order, even when the exemplar is skipped afterwards) -> placement -> re_dict.
    retrieve(b) -> (sample_indexes, db_bounds, query_bounds): the retrieval method of clip b when it is not the discourse
    one (raggesture.py:313-477 dispatches on retrieval_method; llm / gesture_type take other annotations)."""
    from . import vae as ovae
    B = len(conditions["text_features"])
    retr_se, query_se, lats = [], [], []
    for b in range(B):
        spk = int(conditions["speaker_ids"][b, 0].item())
        if retrieve is not None:
            si, dbb, qb = retrieve(b)
        else:
            si, dbb, qb = discourse_retrieval(conditions["discourse"][b], conditions["prominence"][b], spk, db,
                                              conditions["text_features"][b])
        sel = select_retrieved(si, own_names[b], 1)
        rs, qs, ls = {}, {}, {}
        # visiting order and skip rules as in place_exemplars, but the encode happens before the skips
        visited = []
        for qp, smp_idxs in sel.items():
            if len(smp_idxs) == 0 or qp not in qb or qb[qp][2] > qb[qp][3]:
                continue
            visited.append((qp, smp_idxs[0]))
        placed = place_exemplars(sel, dbb, qb, retrieval_method)
        for qp, name in visited:
            rec = dataset[name]
            data = {k: rec[k].unsqueeze(0).clone() for k in ("motion_upper", "motion_lower", "motion_face", "motion_hands",
                                                             "facial", "trans", "contact", "motion_mask")}
            eps = [tape.draw((10, 1, 512)) for _ in range(4)]
            lat, mask = ovae.gesture_encode(P, vae_cfgs, data, eps)
            if qp in placed:
                _, r, q = placed[qp]
                rs[qp], qs[qp] = r, q
                ls[qp] = dict(retr_motion_latent=lat, retr_text=rec["word"].unsqueeze(0), retr_audio=rec["audio"].unsqueeze(0),
                              retr_spkid=rec["speaker_id"].unsqueeze(0), retr_motion_mask=mask)
        retr_se.append(rs), query_se.append(qs), lats.append(ls)
    return dict(retr_startends=retr_se, query_startends=query_se, retr_uncropped_latents=lats)


def parse_gesture_labels_from_llm_output(llm_output):
    """reference: rag/llm_retrieval.py:131-165 (regex over the LLM's "(word, type)" list; beat labels and
    duplicates dropped)."""
    import re
    labels = []
    rx = r"[\"\']*([\w \-\']+\w)[\"\']*\,\s*[\"\']*(?P<gesttype>b*eat|m*etaphoric|iconic|deictic)"
    for m in re.finditer(rx, llm_output, re.MULTILINE):
        t = m.group("gesttype")
        name = "metaphoric" if "etaphoric" in t else "beat" if "eat" in t else "iconic" if "iconic" in t else "deictic"
        labels.append({"word": m.group(1).strip(), "name": name})
    labels = [g for g in labels if g["name"] != "beat"]
    out = []
    for g in labels:
        if g not in out:
            out.append(g)
    return out


def gesture_type_retrieval(gesture_labels, speaker_id, db_labels, encoded_text, db_text, word_similarity):
    """reference: rag/gesture_type_retrieval.py:8-176.  db_labels: name -> [speaker_id, {name, word, start, end}, ...]
    (raggesture.py:262), db_text: idx_2_text, word_similarity(a, b): the reference's get_word_similarity_score
    (rag/utils.py:239-272, a fasttext / word2vec model with a fuzzy-ratio fallback -- injected here).
    Returns (sample_indexes, d_bounds, query_gest_bounds)."""
    gesture_labels = [g for g in gesture_labels if g["name"] != "beat"]
    d_bounds, sample_indexes, query_bounds = {}, {}, {}
    if len(gesture_labels) == 0:
        return sample_indexes, d_bounds, query_bounds
    query_bounds = {i: (g["word"].lower(), g["name"], g["start"], g["end"]) for i, g in enumerate(gesture_labels)}
    for qi, g in enumerate(gesture_labels):
        q_type, q_word = g["name"], g["word"]
        score, rel_bounds = {}, {}
        for name, rec in db_labels.items():
            score[name] = 0
            spk = rec[0]
            labels = [x for x in rec[1:] if x["name"] != "beat"]
            types = [x["name"] for x in labels]
            words = [x["word"] for x in labels]
            if q_type in types:
                score[name] += 2
                rel = [k for k, t in enumerate(types) if t == q_type]
                rel_words = [words[k] for k in rel]
                if spk == speaker_id:
                    score[name] += 2
                if q_word in rel_words:
                    score[name] += 5
                    top = rel[rel_words.index(q_word)]
                else:
                    sims = [word_similarity(w, q_word) for w in rel_words]
                    best = max(range(len(sims)), key=lambda k: (sims[k], -k))   # np.argmax: first maximum
                    top = rel[best]
                    score[name] += 3 / (1 + 2 * sims[best])
                rel_bounds[name] = labels[top]
        order = sorted(score, key=score.get, reverse=True)
        tiers = {}
        for name in order:
            tiers.setdefault(score[name], [])
            if score[name] > 0:
                tiers[score[name]].append(name)
        ranked = []
        for sc in sorted(tiers.keys(), reverse=True):
            tier = tiers[sc]
            if len(tier) > 1:
                tier = sort_sidx_by_textsimilarity(tier, encoded_text, db_text)
            ranked += tier
            if len(ranked) >= 10:
                break
        sample_indexes[qi] = ranked[:10]
        d_bounds[qi] = {}
        for name in ranked[:10]:
            b = rel_bounds[name]
            d_bounds[qi][name] = (b["word"], b["name"], round(b["start"], 3), round(b["end"], 3))
    return sample_indexes, d_bounds, query_bounds


def llm_query_bounds(gesture_labels, text_times):
    """reference: rag/llm_retrieval.py:191-262 -- align the LLM's (word, type) labels with the word timings of the
    clip.  text_times: [((start, end), word), ...].  Returns ({k: (word, type, start, end)} keyed 0.. in order of
    first occurrence in the text, types, words)."""
    q_types = [g["name"] for g in gesture_labels]
    q_words = [_clean_str(g["word"].lower()) for g in gesture_labels]
    bounds = {}
    residual = copy.deepcopy(q_words)
    for t_time in text_times:
        t_word = _clean_str(t_time[1].lower())
        t_start, t_end = t_time[0][0], t_time[0][1]
        for qi, q_word in enumerate(q_words):
            if residual[qi] is None:
                continue
            q_word = q_word.lower()
            if q_word == t_word or t_word in q_word.split():
                bounds.setdefault(qi, []).append((q_word, q_types[qi], t_start, t_end))
                if q_word == t_word or t_word == q_word.split()[-1]:
                    residual[qi] = None
                break
    for qi, bs in bounds.items():
        if len(bs) > 1:
            bounds[qi] = (bs[0][0], bs[0][1], min(b[2] for b in bs), max(b[3] for b in bs))
        else:
            bounds[qi] = bs[0]
    bounds = {k: v for k, v in enumerate(bounds.values())}
    return bounds, [bounds[i][1] for i in sorted(bounds)], [bounds[i][0] for i in sorted(bounds)]


def _clean_str(s):
    return "".join([c for c in s if c.isalnum() or c.isspace()])


def llm_retrieval(text, text_times, speaker_id, prominence, db_labels, db_gestprom, encoded_text, db_text,
                  word_similarity, llm_output):
    """reference: rag/llm_retrieval.py:166-466.  db_gestprom: name -> {label idx: (word, prominence) | None}
    (raggesture.py:270-272, over ALL labels of the sample, beat included); llm_output(text) -> str stands for
    get_llm_output (the GPT call, :69-96); word_similarity as in gesture_type_retrieval."""
    d_bounds, sample_indexes, query_bounds = {}, {}, {}
    if text.strip() == "":
        return sample_indexes, d_bounds, query_bounds
    gesture_labels = parse_gesture_labels_from_llm_output(llm_output(text))
    if len(gesture_labels) == 0:
        return sample_indexes, d_bounds, query_bounds
    query_bounds, q_types, q_words = llm_query_bounds(gesture_labels, text_times)
    if len(query_bounds) == 0:
        return sample_indexes, d_bounds, query_bounds
    q_prom = map_conns_to_prominence(q_words, prominence)
    for i in range(len(q_words)):
        q_prom[i] = None if q_prom[i] is None else (q_types[i], *q_prom[i])
    for qi, (q_type, q_word) in enumerate(zip(q_types, q_words)):
        score, rel_bounds = {}, {}
        for name, rec in db_labels.items():
            score[name] = 0
            spk, all_labels = rec[0], rec[1:]
            if len(all_labels) == 0:
                continue
            labels = [g for g in all_labels if g["name"] != "beat"]
            proms = [db_gestprom[name][gi] for gi, g in enumerate(all_labels) if g["name"] != "beat"]
            types = [x["name"] for x in labels]
            words = [x["word"] for x in labels]
            if len(types) == 0:
                continue
            if q_type in types:
                score[name] += 2
                rel = [k for k, t in enumerate(types) if t == q_type]
                rel_words = [words[k] for k in rel]
                if spk == speaker_id:
                    score[name] += 1
                if q_word in rel_words:
                    score[name] += 5
                    top = rel[rel_words.index(q_word)]
                else:
                    sims = [word_similarity(w, q_word) for w in rel_words]
                    best = max(range(len(sims)), key=lambda k: (sims[k], -k))   # np.argmax: first maximum
                    top = rel[best]
                    score[name] += 3 / (1 + 2 * sims[best])
                total, count, diffs = 0, 0, {}
                for k in rel:
                    if proms[k] is None or q_prom[qi] is None:
                        continue
                    diff = abs(proms[k][1] - q_prom[qi][-1])
                    diffs[k] = diff
                    total += 4 / (1 + 2 * diff)
                    count += 1
                if count > 0:
                    score[name] += total / count
                    top = sorted(diffs, key=diffs.get)[0]
                rel_bounds[name] = labels[top]
        order = sorted(score, key=score.get, reverse=True)
        tiers = {}
        for name in order:
            tiers.setdefault(score[name], [])
            if score[name] > 0:
                tiers[score[name]].append(name)
        ranked = []
        for sc in sorted(tiers.keys(), reverse=True):
            tier = tiers[sc]
            if len(tier) > 1:
                tier = sort_sidx_by_textsimilarity(tier, encoded_text, db_text)
            ranked += tier
            if len(ranked) >= 10:
                break
        sample_indexes[qi] = ranked[:10]
        d_bounds[qi] = {}
        for name in ranked[:10]:
            b = rel_bounds[name]
            d_bounds[qi][name] = (b["word"], b["name"], round(b["start"], 3), round(b["end"], 3))
    return sample_indexes, d_bounds, query_bounds
