"""Motion-exemplar retrieval: replicated DB on the GPU, discourse scoring sweep + text-similarity
tie-break as HIP reductions, exemplar placement and `re_dict` assembly.

Mirrors (names, argument meaning, result schema):
  mogen/models/transformers/raggesture.py:157-303 RetrievalDatabase.__init__ (DB dicts),
      :313-477 retrieve (method dispatch, self-exclusion, num_retrieval), :479-884 forward
      (exemplar fetch + VAE encode, placement, re_dict)
  mogen/models/transformers/rag/discourse_retrieval.py:8-316, rag/utils.py:86-132, 171-228
What runs where: string handling (connective cleaning, vocabulary coding, prominence matching) and
the final tier walk are host integer/string logic exactly as in the reference; the O(N_db) score
sweep (float64, reference operation order -> bit-identical scores) and the tie-break similarity
reduction run on the GPU over an integer-coded CSR copy of the DB (`DiscourseIndex`).
The LMDB-backed cache of the reference (6 LMDB dicts) is replaced by in-memory dicts handed in by
the caller (`metadata=` or `dataset.retrieval_samples`); `lmdb` is not available in this image.
"""
import copy
import ctypes
import json
import os
import re

import numpy as np
import torch

from . import capi


def _clean(s):
    return "".join([c for c in str(s) if c.isalnum() or c.isspace()])


def map_conns_to_prominence(conn_list, prominence_list):
    """rag/utils.py:171-228: attach prominence values to connectives (multi-word = average)."""
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
            if dps[0][0] != _clean(conn_list[si]):
                raise ValueError("prominence words do not spell the connective %r" % (conn_list[si],))
            relevant[si] = (conn_list[si], sum([d[1] for d in dps]) / len(dps))
        else:
            relevant[si] = dps[0] if len(dps) > 0 else None
    if len(relevant) != len(conn_list):  # the reference drops into a debugger here
        raise ValueError("prominence list does not cover the connective list (reference would trap)")
    return relevant


def build_db_dicts(samples, stratified_db_creation=False, stratification_interval=15):
    """raggesture.py:244-276: DB dicts from per-sample records (sample_name, speaker_id, discourse,
    prominence, text_feature[, gesture_labels]), in iteration order; with stratified_db_creation only the windows
    whose in-sequence index (the number behind "/" in sample_name) is a multiple of the interval are kept (:250-255)."""
    if stratified_db_creation:
        samples = [smp for smp in samples if int(smp["sample_name"].split("/")[1]) % stratification_interval == 0]
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
        out["idx_2_gestprom"] = {}                        # raggesture.py:270-272 (llm method)
        for smp in samples:
            words = [g["word"] for g in smp["gesture_labels"]]
            out["idx_2_gestprom"][smp["sample_name"]] = map_conns_to_prominence(words, smp["prominence"]) if words else {}
    return out


def to_device_async(host, dev):
    """Host tensor / array -> device without stalling the host: staged in page-locked memory (torch's caching host
    allocator keeps the block until the copy has run), copied asynchronously on the current stream.  A copy from pageable
    memory instead returns only after everything queued on that stream before it has run -- on a stream that holds a
    just-launched VAE graph that is milliseconds of host time, per copy."""
    t = torch.from_numpy(np.ascontiguousarray(host)) if isinstance(host, np.ndarray) else host
    if t.device.type != "cpu" or torch.device(dev).type == "cpu":
        return t.to(dev)
    return t.pin_memory().to(dev, non_blocking=True)

class DiscourseIndex:
    """Integer-coded, device-resident copy of the discourse DB (replicated per GPU)."""

    def __init__(self, db, device="cuda"):
        self.dev = torch.device(device)
        self.h = capi.get_handle(self.dev.index if self.dev.index is not None else torch.cuda.current_device())
        self.db = db
        self.names = list(db["idx_2_sense"].keys())
        self.name_to_idx = {n: i for i, n in enumerate(self.names)}
        self.sense_code, self.conn_code = {}, {}
        spk, off, rs, rc, rp = [], [0], [], [], []
        for n in self.names:
            rec = db["idx_2_sense"][n]
            spk.append(int(rec[0]))
            prom = db["idx_2_prominence"][n]
            for k, (sense, conn) in enumerate(rec[1:]):
                rs.append(self.sense_code.setdefault(sense, len(self.sense_code)))
                rc.append(self.conn_code.setdefault(conn, len(self.conn_code)))
                pv = prom.get(k)
                rp.append(float("nan") if pv is None else float(pv[1]))
            off.append(len(rs))
        i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=self.dev)
        self.spk, self.rel_off, self.rel_sense, self.rel_conn = i32(spk), i32(off), i32(rs or [0]), i32(rc or [0])
        self.rel_prom = torch.tensor(rp or [0.0], dtype=torch.float64, device=self.dev)
        feats = [db["idx_2_text"][n][0].float() for n in self.names]
        foff = np.zeros(len(feats) + 1, dtype=np.int64)
        foff[1:] = np.cumsum([f.shape[0] for f in feats])
        self.feat_off = torch.from_numpy(foff).to(self.dev)
        self.feats = torch.cat(feats, dim=0).to(self.dev).contiguous()
        self.dim = self.feats.shape[1]
        self.n = len(self.names)
        self._score = torch.empty(self.n, dtype=torch.float64, device=self.dev)
        self._top = torch.empty(self.n, dtype=torch.int32, device=self.dev)

    fused_sweep = True     # sweep_async: rg_discourse_select_fused (two launches, no score arrays); False: sweep + 3-launch selection

    def scores(self, sense, conn, speaker_id, q_prom):
        """HIP sweep for one query relation -> (scores float64 [N], top_rel int32 [N]) on the host."""
        lib, vp = self.h.lib, ctypes.c_void_p
        s = torch.cuda.current_stream().cuda_stream
        rc = lib.rg_discourse_scores(self.h._h, vp(self.spk.data_ptr()), vp(self.rel_off.data_ptr()),
                                     vp(self.rel_sense.data_ptr()), vp(self.rel_conn.data_ptr()),
                                     vp(self.rel_prom.data_ptr()), self.n, self.sense_code.get(sense, -2),
                                     self.conn_code.get(conn, -1), int(speaker_id),
                                     ctypes.c_double(float("nan") if q_prom is None else float(q_prom)),
                                     vp(self._score.data_ptr()), vp(self._top.data_ptr()), vp(s))
        if rc != 0:
            raise capi.RgError("rg_discourse_scores failed: %s" % lib.rg_last_error(self.h._h).decode())
        return self._score.cpu().numpy(), self._top.cpu().numpy()

    def sweep_buffers(self, queries):
        """Device buffers of one sweep over `queries` [(sense, conn, speaker_id, q_prom)]: the query parameters (uploaded
        asynchronously from pinned memory), the selection workspace and the survivor lists."""
        lib = self.h.lib
        Q, n = len(queries), self.n
        cap = n
        nan = float("nan")
        params = to_device_async(torch.tensor([[float(self.sense_code.get(sense, -2)), float(self.conn_code.get(conn, -1)),
                                                float(int(spk)), nan if q_prom is None else float(q_prom)]
                                               for sense, conn, spk, q_prom in queries], dtype=torch.float64), self.dev)
        e = lambda *shape, dt=torch.int32: torch.empty(*shape, dtype=dt, device=self.dev)
        bufs = dict(Q=Q, cap=cap, params=params, ws=e(Q, lib.rg_select_workspace_doubles(n), dt=torch.float64),
                    cursor=e(Q), idx=e(Q, cap), top=e(Q, cap), score=e(Q, cap, dt=torch.float64))
        if not self.fused_sweep:
            bufs.update(all_score=e(Q, n, dt=torch.float64), all_top=e(Q, n))
        return bufs

    def sweep_launch(self, bufs):
        """The launches of one sweep + device-side candidate selection on the current stream (no allocation, no copy, no
        synchronisation): fused = rg_discourse_select_fused (two launches, scores in registers), else the sweep + the
        three-launch selection over score arrays.  Returns the ticket for collect()."""
        lib, vp = self.h.lib, ctypes.c_void_p
        s = torch.cuda.current_stream().cuda_stream
        p = lambda k: vp(bufs[k].data_ptr())
        db = (vp(self.spk.data_ptr()), vp(self.rel_off.data_ptr()), vp(self.rel_sense.data_ptr()), vp(self.rel_conn.data_ptr()),
              vp(self.rel_prom.data_ptr()), self.n)
        if self.fused_sweep:
            rc = lib.rg_discourse_select_fused(self.h._h, *db, p("params"), bufs["Q"], p("ws"), p("cursor"), bufs["cap"], p("idx"),
                                               p("top"), p("score"), vp(s))
        else:
            rc = lib.rg_discourse_scores_batched(self.h._h, *db, p("params"), bufs["Q"], p("all_score"), p("all_top"), vp(s))
            if rc == 0:
                rc = lib.rg_select_top_scores_batched(self.h._h, p("all_score"), p("all_top"), self.n, bufs["Q"], p("ws"), p("cursor"),
                                                      bufs["cap"], p("idx"), p("top"), p("score"), vp(s))
        if rc != 0:
            raise capi.RgError("retrieval sweep failed: %s" % lib.rg_last_error(self.h._h).decode())
        return dict(cursor=bufs["cursor"], idx=bufs["idx"], top=bufs["top"], score=bufs["score"], keep=bufs)

    def sweep_async(self, queries):
        """Launch sweep + device-side candidate selection for a list of (sense, conn, speaker_id, q_prom)
        without any host synchronisation; returns a ticket for collect()."""
        return self.sweep_launch(self.sweep_buffers(queries))

    @staticmethod
    def collect(ticket):
        """One synchronisation for the whole batch: per query (entry idx ascending, score, top_rel_idx)."""
        counts = ticket["cursor"].cpu().numpy()
        m = int(counts.max()) if counts.size else 0
        idx = ticket["idx"][:, :m].cpu().numpy()
        top = ticket["top"][:, :m].cpu().numpy()
        score = ticket["score"][:, :m].cpu().numpy()
        out = []
        for q, c in enumerate(counts):
            o = np.argsort(idx[q, :c], kind="stable")
            out.append((idx[q, :c][o], score[q, :c][o], top[q, :c][o]))
        return out

    def sims(self, q_feat_dev, cand):
        """HIP tie-break reduction: mean diagonal similarity of the query to each candidate entry."""
        return self.sims_async(q_feat_dev, cand).cpu().numpy()

    def sims_async(self, q_feat_dev, cand):
        """Launch only: float64 device tensor [len(cand)]."""
        lib, vp = self.h.lib, ctypes.c_void_p
        c = to_device_async(torch.tensor(cand, dtype=torch.int32), self.dev)
        out = torch.empty(len(cand), dtype=torch.float64, device=self.dev)
        s = torch.cuda.current_stream().cuda_stream
        rc = lib.rg_text_diag_sim(self.h._h, vp(q_feat_dev.data_ptr()), q_feat_dev.shape[0], vp(self.feats.data_ptr()),
                                  vp(self.feat_off.data_ptr()), vp(c.data_ptr()), len(cand), self.dim,
                                  vp(out.data_ptr()), vp(s))
        if rc != 0:
            raise capi.RgError("rg_text_diag_sim failed: %s" % lib.rg_last_error(self.h._h).decode())
        return out


def discourse_queries(discourse, prominence, speaker_id):
    """The (sense, connective, speaker, query prominence) tuples discourse_retrieval sweeps the DB with."""
    conns = [d[0] for d in discourse]
    q_prom = map_conns_to_prominence(conns, prominence)
    return [(d[1], d[0], speaker_id, None if q_prom[i] is None else q_prom[i][1]) for i, d in enumerate(discourse)]


def _cut_tiers(index, q_dev, survivor, sims):
    """Only the top tiers are ever visited by the reference's ranking walk (it stops once it holds 10 entries): the
    device kept the entries whose score reaches the 10th largest one (all ties included); order them like
    sorted(..., reverse=True) (stable among equals), cut the leading score tiers and launch the tie-break
    similarity of every multi-member tier (appended to `sims`).  Returns (tiers [(entries, sims slot)], top)."""
    keep, kscore, ktop = survivor
    top = dict(zip(keep.tolist(), ktop.tolist()))
    o = np.argsort(-kscore, kind="stable")
    order, oscore = keep[o].tolist(), kscore[o].tolist()
    tiers, i, n = [], 0, 0
    while i < len(order) and n < 10:
        sc = oscore[i]
        if not sc > 0:
            break
        j = i
        while j < len(order) and oscore[j] == sc:
            j += 1
        tier = [int(e) for e in order[i:j]]
        slot = None
        if len(tier) > 1:
            slot = len(sims)
            sims.append(index.sims_async(q_dev, tier))
        tiers.append((tier, slot))
        n += len(tier)
        i = j
    return tiers, top


def _walk_tiers(tiers, sims_host):
    ranked = []
    for tier, slot in tiers:
        if slot is not None:
            tier = [tier[k] for k in np.argsort(-sims_host[slot], kind="stable")]
        ranked += tier
    return ranked[:10]


def discourse_retrieval_begin(index, discourse, prominence, speaker_id, encoded_text, survivors=None):
    """First half of discourse_retrieval: orders the survivors of every query relation, cuts the leading score
    tiers the reference's ranking walk can reach (it stops once it holds 10 entries) and LAUNCHES the
    tie-break similarity of every multi-member tier without reading anything back.  Returns a pending record
    for discourse_retrieval_finish; `pending["sims"]` are device tensors (one per tie tier)."""
    pend = dict(discourse=discourse, queries=[], sims=[])
    if len(discourse) == 0:
        return pend
    q_dev = encoded_text.to(index.dev).float().contiguous()
    pend["q_dev"] = q_dev   # keeps the features alive until the kernels have run
    if survivors is None:
        survivors = index.collect(index.sweep_async(discourse_queries(discourse, prominence, speaker_id)))
    for qi in range(len(discourse)):
        pend["queries"].append(_cut_tiers(index, q_dev, survivors[qi], pend["sims"]))
    return pend


def discourse_retrieval_finish(index, pend, sims_host):
    """Second half: sims_host[slot] = the tie-break similarities (numpy) of pending["sims"][slot]."""
    discourse = pend["discourse"]
    d_bounds, sample_indexes, query_bounds = {}, {}, {}
    if len(discourse) == 0:
        return sample_indexes, d_bounds, query_bounds
    query_bounds = {i: (d[0].lower(), d[1], d[6], d[7]) for i, d in enumerate(discourse)}
    for qi, (tiers, top) in enumerate(pend["queries"]):
        ranked = _walk_tiers(tiers, sims_host)
        sample_indexes[qi] = [index.names[e] for e in ranked]
        d_bounds[qi] = {}
        for e in ranked:
            b = index.db["idx_2_discbounds"][index.names[e]][int(top[e])]
            d_bounds[qi][index.names[e]] = (b[1], b[0], round(b[4], 3), round(b[5], 3))
    return sample_indexes, d_bounds, query_bounds


def fetch_sims(pendings):
    """One read-back for the tie-break similarities of any number of pending retrievals."""
    tensors = [t for p in pendings for t in p["sims"]]
    if not tensors:
        return [[] for _ in pendings]
    flat = torch.cat(tensors).cpu().numpy()
    out, pos = [], 0
    for p in pendings:
        cur = []
        for t in p["sims"]:
            cur.append(flat[pos:pos + t.numel()])
            pos += t.numel()
        out.append(cur)
    return out


def discourse_retrieval(index, discourse, prominence, speaker_id, encoded_text, survivors=None):
    """Same contract as rag/discourse_retrieval.py:8-316 (returns sample_indexes, d_bounds,
    query_bounds) with the DB sweep, the candidate selection and the tie-break similarity on the GPU.
    `survivors`: the collect()ed results of a sweep_async() over discourse_queries(...) launched earlier
    (batched over clips); None = sweep here."""
    pend = discourse_retrieval_begin(index, discourse, prominence, speaker_id, encoded_text, survivors)
    return discourse_retrieval_finish(index, pend, fetch_sims([pend])[0])


_NEP50 = int(np.__version__.split(".")[0]) >= 2


def partial_ratio_host(s1, s2):
    """fuzzywuzzy 0.18 `fuzz.partial_ratio`, pure-python flavour (the package calls difflib.SequenceMatcher when
    python-Levenshtein is absent, as in the reference's requirements.txt).  Host form of rg_partial_ratio: used for
    single pairs and for strings longer than the kernel handles."""
    from difflib import SequenceMatcher
    if s1 is None or s2 is None:
        return 0
    if s1 == s2:
        return 100
    if len(s1) == 0 or len(s2) == 0:
        return 0
    shorter, longer = (s1, s2) if len(s1) <= len(s2) else (s2, s1)
    scores = []
    for i, j, _ in SequenceMatcher(None, shorter, longer).get_matching_blocks():
        start = j - i if j - i > 0 else 0
        r = SequenceMatcher(None, shorter, longer[start:start + len(shorter)]).ratio()
        if r > .995:
            return 100
        scores.append(r)
    return int(round(100 * max(scores)))


def fuzzy_word_similarity(db_word, query_word):
    """The reference's get_word_similarity_score as it effectively behaves (rag/utils.py:239-272): its word2vec /
    fasttext models are never defined, every call raises inside the `try` and returns
    `fuzz.partial_ratio(word1, word2) / 100`.  This is the DEFAULT `word_similarity` of RetrievalDatabase; whole
    vocabularies go through the device form (rg_partial_ratio) in GestureTypeIndex.sweep."""
    return partial_ratio_host(db_word, query_word) / 100


class GestureTypeIndex:
    """Device copy of the DB's semantic gesture labels (raggesture.py:262 idx_2_gesture_labels, beat labels dropped
    like gesture_type_retrieval.py:57-59 does): CSR over entries with integer-coded types and words.  Shares the
    entry order, names and text-feature table (tie-break) with a DiscourseIndex."""

    def __init__(self, db, base):
        self.base, self.dev, self.h, self.n = base, base.dev, base.h, base.n
        self.type_code, self.word_code, self.labels = {}, {}, []
        spk, off, lt, lw, lp = [], [0], [], [], []
        gestprom = db.get("idx_2_gestprom")
        for n in base.names:
            rec = db["idx_2_gesture_labels"][n]
            spk.append(int(rec[0]))
            labs = [g for g in rec[1:] if g["name"] != "beat"]
            self.labels.append(labs)
            for g in labs:
                lt.append(self.type_code.setdefault(g["name"], len(self.type_code)))
                lw.append(self.word_code.setdefault(g["word"], len(self.word_code)))
            off.append(len(lt))
            if gestprom is not None:      # aligned with ALL labels of the sample (beat included), llm_retrieval.py:293-299
                for gi, g in enumerate(rec[1:]):
                    if g["name"] != "beat":
                        pv = gestprom[n][gi]
                        lp.append(float("nan") if pv is None else float(pv[1]))
        i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=self.dev)
        self.spk, self.lab_off, self.lab_type, self.lab_word = i32(spk), i32(off), i32(lt or [0]), i32(lw or [0])
        self.lab_prom = None if gestprom is None else torch.tensor(lp or [0.0], dtype=torch.float64, device=self.dev)
        self.vocab = list(self.word_code.keys())
        # the vocabulary as code-point rows for rg_partial_ratio (the default word similarity)
        self.pr_max = int(self.h.lib.rg_partial_ratio_max_len())
        codes = np.zeros((max(1, len(self.vocab)), self.pr_max), dtype=np.int32)
        lens = np.zeros(max(1, len(self.vocab)), dtype=np.int32)
        for v, w in enumerate(self.vocab):
            lens[v] = len(w)
            cp = [ord(ch) for ch in w[:self.pr_max]]
            codes[v, :len(cp)] = cp
        self.vocab_codes, self.vocab_len = torch.from_numpy(codes).to(self.dev), torch.from_numpy(lens).to(self.dev)

    def fuzzy_similarities(self, words):
        """[len(words), len(vocab)] float64 on the device: partial_ratio(vocab word, query word) / 100 for every query
        word, one rg_partial_ratio launch each; pairs with a string beyond the kernel's length are patched from the host."""
        V = max(1, len(self.vocab))
        out = torch.empty(len(words), V, dtype=torch.float64, device=self.dev)
        vp, s = ctypes.c_void_p, torch.cuda.current_stream().cuda_stream
        for q, word in enumerate(words):
            cp = (ctypes.c_int * max(1, len(word)))(*[ord(ch) for ch in word])
            rc = self.h.lib.rg_partial_ratio(self.h._h, vp(self.vocab_codes.data_ptr()), vp(self.vocab_len.data_ptr()),
                                             len(self.vocab) or 1, self.pr_max, cp, len(word), vp(out[q].data_ptr()), vp(s))
            if rc != 0:
                raise capi.RgError("rg_partial_ratio failed: %s" % self.h.lib.rg_last_error(self.h._h).decode())
        long_rows = [v for v, w in enumerate(self.vocab) if len(w) > self.pr_max]
        for q, word in enumerate(words):
            rows = range(len(self.vocab)) if len(word) > self.pr_max else long_rows
            for v in rows:
                out[q, v] = fuzzy_word_similarity(self.vocab[v], word)
        if not self.vocab:
            out.zero_()
        return out

    def sweep(self, queries, word_similarity, spk_bonus=2.0, proms=None):
        """[(type, word, speaker_id)] (+ per-query prominence, None = unknown: the llm method) -> per query (entry idx ascending, score, top label index) of the entries that
        can be visited; the word-similarity vector of each query word is computed here (host: the similarity model
        is the caller's), the scores and the selection on the device."""
        lib, vp = self.h.lib, ctypes.c_void_p
        Q, n = len(queries), self.n
        nws = lib.rg_select_workspace_doubles(n)
        score = torch.empty(Q, n, dtype=torch.float64, device=self.dev)
        top = torch.empty(Q, n, dtype=torch.int32, device=self.dev)
        ws = torch.empty(Q, nws, dtype=torch.float64, device=self.dev)
        cursor = torch.zeros(Q, dtype=torch.int32, device=self.dev)
        o_idx = torch.empty(Q, n, dtype=torch.int32, device=self.dev)
        o_top = torch.empty(Q, n, dtype=torch.int32, device=self.dev)
        o_score = torch.empty(Q, n, dtype=torch.float64, device=self.dev)
        if word_similarity is fuzzy_word_similarity:
            # the reference's effective similarity (python floats -> float64 scores): whole vocabulary on the device
            sims = self.fuzzy_similarities([q[1] for q in queries])
            f32 = [0] * Q
        else:
            raw = [[word_similarity(w, q[1]) for w in self.vocab] or [0.0] for q in queries]
            # float32 similarities make the reference's score float32 only under NumPy >= 2 promotion rules (NEP 50); the
            # reference pins numpy < 1.24, where scalar-scalar arithmetic promotes to float64: follow the installed numpy
            f32 = [int(_NEP50 and any(isinstance(v, np.float32) for v in row)) for row in raw]
            sims = torch.tensor([[float(v) for v in row] for row in raw], dtype=torch.float64, device=self.dev)
        if proms is not None and self.lab_prom is None:
            raise capi.RgError("llm retrieval needs idx_2_gestprom in the DB metadata (raggesture.py:270-272)")
        nan = float("nan")
        s = torch.cuda.current_stream().cuda_stream
        for q, (q_type, q_word, speaker_id) in enumerate(queries):
            rc = lib.rg_gesture_scores(self.h._h, vp(self.spk.data_ptr()), vp(self.lab_off.data_ptr()),
                                       vp(self.lab_type.data_ptr()), vp(self.lab_word.data_ptr()),
                                       vp(self.lab_prom.data_ptr()) if proms is not None else vp(None),
                                       vp(sims[q].data_ptr()), n, self.type_code.get(q_type, -2),
                                       self.word_code.get(q_word, -1), int(speaker_id), ctypes.c_double(spk_bonus),
                                       ctypes.c_double(nan if proms is None or proms[q] is None else float(proms[q])),
                                       f32[q], vp(score[q].data_ptr()), vp(top[q].data_ptr()), vp(s))
            if rc == 0:
                rc = lib.rg_select_top_scores(self.h._h, vp(score[q].data_ptr()), vp(top[q].data_ptr()), n,
                                              vp(ws[q].data_ptr()), vp(cursor[q:].data_ptr()), n, vp(o_idx[q].data_ptr()),
                                              vp(o_top[q].data_ptr()), vp(o_score[q].data_ptr()), vp(s))
            if rc != 0:
                raise capi.RgError("gesture-type sweep failed: %s" % lib.rg_last_error(self.h._h).decode())
        return DiscourseIndex.collect(dict(cursor=cursor, idx=o_idx, top=o_top, score=o_score))


def gesture_type_retrieval(gindex, gesture_labels, speaker_id, encoded_text, word_similarity):
    """Same contract as rag/gesture_type_retrieval.py:8-176 (sample_indexes, d_bounds, query_gest_bounds); the DB
    sweep, the candidate selection and the tie-break text similarity run on the GPU, `word_similarity(db_word,
    query_word)` is the reference's get_word_similarity_score (an external embedding model: supplied by the caller)."""
    gesture_labels = [g for g in gesture_labels if g["name"] != "beat"]
    d_bounds, sample_indexes, query_bounds = {}, {}, {}
    if len(gesture_labels) == 0:
        return sample_indexes, d_bounds, query_bounds
    query_bounds = {i: (g["word"].lower(), g["name"], g["start"], g["end"]) for i, g in enumerate(gesture_labels)}
    sample_indexes, d_bounds = _gesture_ranked(gindex, [(g["name"], g["word"], speaker_id) for g in gesture_labels],
                                               encoded_text, word_similarity, 2.0, None)
    return sample_indexes, d_bounds, query_bounds


def _clean_str(s):
    return "".join([c for c in s if c.isalnum() or c.isspace()])


def llm_query_bounds(gesture_labels, text_times):
    """rag/llm_retrieval.py:191-262: align the LLM's (word, type) labels with the clip's word timings
    [((start, end), word), ...]; multi-word labels span their words, a label is closed by its last word, the result is
    keyed 0.. in order of first occurrence in the text."""
    q_types = [g["name"] for g in gesture_labels]
    q_words = [_clean_str(g["word"].lower()) for g in gesture_labels]
    open_, spans = [True] * len(q_words), {}
    for (t_start, t_end), t_word in ((t[0], t[1]) for t in text_times):
        t_word = _clean_str(t_word.lower())
        for qi, q_word in enumerate(q_words):
            if not open_[qi]:
                continue
            parts = q_word.split()
            if q_word == t_word or t_word in parts:
                spans.setdefault(qi, []).append((t_start, t_end))
                if q_word == t_word or t_word == parts[-1]:
                    open_[qi] = False
                break
    return {k: (q_words[qi], q_types[qi], min(a for a, _ in sp), max(b for _, b in sp))
            for k, (qi, sp) in enumerate(spans.items())}


def llm_retrieval(gindex, text, text_times, speaker_id, prominence, encoded_text, word_similarity, llm_output):
    """Same contract as rag/llm_retrieval.py:166-466; `llm_output(text) -> str` replaces get_llm_output (the GPT
    call: use LLMResponseCache.get), the label alignment is host string work, the DB sweep (type / speaker / word /
    prominence score), selection and tie-break run on the GPU."""
    if text.strip() == "":
        return {}, {}, {}
    labels = parse_gesture_labels_from_llm_output(llm_output(text))
    if len(labels) == 0:
        return {}, {}, {}
    query_bounds = llm_query_bounds(labels, text_times)
    if len(query_bounds) == 0:
        return {}, {}, {}
    nq = len(query_bounds)
    q_types, q_words = [query_bounds[i][1] for i in range(nq)], [query_bounds[i][0] for i in range(nq)]
    q_prom = map_conns_to_prominence(q_words, prominence)
    proms = [None if q_prom[i] is None else q_prom[i][-1] for i in range(nq)]
    si, db_b = _gesture_ranked(gindex, [(t, w, speaker_id) for t, w in zip(q_types, q_words)], encoded_text,
                               word_similarity, 1.0, proms)
    return si, db_b, query_bounds


def _gesture_ranked(gindex, queries, encoded_text, word_similarity, spk_bonus, proms):
    base = gindex.base
    q_dev = encoded_text.to(base.dev).float().contiguous()
    survivors = gindex.sweep(queries, word_similarity, spk_bonus, proms)
    sims, cut = [], []
    for qi in range(len(queries)):
        cut.append(_cut_tiers(base, q_dev, survivors[qi], sims))
    sims_host = [t.cpu().numpy() for t in sims]
    sample_indexes, d_bounds = {}, {}
    for qi, (tiers, top) in enumerate(cut):
        ranked = _walk_tiers(tiers, sims_host)
        sample_indexes[qi] = [base.names[e] for e in ranked]
        d_bounds[qi] = {}
        for e in ranked:
            b = gindex.labels[e][int(top[e])]
            d_bounds[qi][base.names[e]] = (b["word"], b["name"], round(b["start"], 3), round(b["end"], 3))
    return sample_indexes, d_bounds


def place_exemplars(retr_indexes, retr_bounds, query_bounds, retrieval_method="discourse", fps=15, chunk=15,
                    motion_len=150):
    """Time -> latent-index placement of raggesture.py:542-760.  Returns a list of
    (query_point, sample_name, placed) in visiting order, where placed is None (the exemplar was
    fetched and VAE-encoded by the reference but then skipped) or
    ((retr_lat_start, retr_lat_end), (start_lat, end_lat))."""
    latent_len = motion_len // chunk
    prev_end = -1
    out = []
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
            if not q_start // chunk < q_end // chunk + 1:
                raise ValueError("empty query span")
            if retrieval_method in ("gesture_type", "llm") and (r_end - r_start) > 0.9:
                r_start, r_end = max(0, r_start - 0.2), min(motion_len / fps, r_end + 0.1)
            else:
                r_start, r_end = max(0, r_start - 0.666), min(motion_len / fps, r_end + 0.333)
            r_start, r_end = int(r_start * fps), int(r_end * fps)
            if r_start == r_end:
                out.append((qp, smp, None))
                continue
            if r_end == motion_len:
                r_end = motion_len - 1
                r_start = max(0, r_start - 1)
            r_lat_start, r_lat_end = r_start // chunk, r_end // chunk + 1
            mid_lat = ((q_start + q_end) // 2) // chunk
            n = r_lat_end - r_lat_start
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
                        out.append((qp, smp, None))
                        continue
                    r_lat_end = r_lat_start + n
            prev_end = e
            out.append((qp, smp, ((r_lat_start, r_lat_end), (s, e))))
    return out


LMDB_DICTS = ("idx_2_text", "idx_2_sense", "idx_2_discbounds", "idx_2_gesture_labels", "idx_2_prominence", "idx_2_gestprom")


def read_lmdb_dicts(lmdb_paths):
    """The reference's six retrieval caches (raggesture.py:90-155 `LMDBDict`, :219-224: one LMDB environment per dict
    under `lmdb_paths`, ascii keys in cursor order = sample-name order, values `pyarrow.serialize`d, tensors stored as
    numpy arrays and converted back for idx_2_text) -> the metadata dict RetrievalDatabase takes, or None when the
    caches cannot be read here (directory missing / empty, or `lmdb` / the legacy `pyarrow.deserialize` absent --
    pyarrow >= 2 removed that serializer; this image has neither lmdb nor it).  Raises only on a corrupt cache."""
    import os
    try:
        import lmdb
        import pyarrow
    except ImportError:
        return None
    if not hasattr(pyarrow, "deserialize") or lmdb_paths is None or not os.path.isdir(str(lmdb_paths)):
        return None
    out = {}
    for name in LMDB_DICTS:
        path = os.path.join(str(lmdb_paths), name)
        if not os.path.isdir(path):
            if name in ("idx_2_text", "idx_2_sense", "idx_2_discbounds", "idx_2_prominence"):
                return None
            continue
        env = lmdb.open(path, readonly=True, lock=False)
        d = {}
        with env.begin(write=False) as txn:
            for key, val in txn.cursor():
                v = pyarrow.deserialize(val)
                if name == "idx_2_text":   # torch_converter=True (:219)
                    v = torch.from_numpy(v) if isinstance(v, np.ndarray) else \
                        [torch.from_numpy(x) if isinstance(x, np.ndarray) else x for x in v] if isinstance(v, list) else v
                d[key.decode("ascii")] = v
        env.close()
        if name in ("idx_2_text", "idx_2_sense") and not d:
            return None      # fresh checkout: the reference rebuilds the caches from the dataset (:226-243)
        out[name] = d
    return out


class RetrievalDatabase:
    """Drop-in for raggesture.py:157-884 `RetrievalDatabase` (inference; `discourse` and `gesture_type` methods).

    `dataset[name]` must return the per-sample dict the reference reads (motion, motion_upper/lower/
    face/hands, facial, trans, contact, motion_mask, word, audio, speaker_id).  DB metadata comes
    from `metadata` (dict of idx_2_text / idx_2_sense / idx_2_discbounds / idx_2_prominence) or from
    `dataset.retrieval_samples` (raw records, see build_db_dicts)."""
    CACHE_CAP = 4096   # clips kept in the write-only test_indexes / test_dbounds / test_qbounds dicts
    # keys of the reference's constructor (raggesture.py:159-184) that only size its TRAINING-time retrieval encoders
    # (never built or used at inference): accepted and ignored, anything else is an error
    REFERENCE_TRAINING_KEYS = frozenset(("motion_feat_dim", "output_dim", "num_layers", "num_motion_layers", "kinematic_coef",
                                         "num_heads", "ff_size", "stride", "sa_block_cfg", "ffn_cfg", "dropout"))

    def __init__(self, num_retrieval=None, topk=None, latent_dim=512, text_latent_dim=768, max_seq_len=150,
                 motion_fps=15, motion_framechunksize=15, dataset=None, metadata=None, device="cuda", word_similarity=None,
                 llm_output=None, stratified_db_creation=False, stratification_interval=15, lmdb_paths=None,
                 new_lmdb_cache=False, **_cfg):
        """DB metadata, in this order: `metadata=` (the six dicts), the reference's LMDB caches under `lmdb_paths`
        (when readable here and new_lmdb_cache is off, raggesture.py:219-243), `dataset.retrieval_samples` (raw
        records -> build_db_dicts, what the reference does when its caches are empty)."""
        unknown = sorted(set(_cfg) - self.REFERENCE_TRAINING_KEYS)
        if unknown:
            raise capi.RgError("RetrievalDatabase: unknown configuration key(s) %s" % ", ".join(unknown))
        if metadata is None and not new_lmdb_cache:
            metadata = read_lmdb_dicts(lmdb_paths)
        if metadata is None:
            samples = getattr(dataset, "retrieval_samples", None)
            if samples is None:
                raise capi.RgError("RetrievalDatabase needs `metadata=`, readable LMDB caches under lmdb_paths=%r (lmdb and "
                                   "the legacy pyarrow.deserialize must be importable) or `dataset.retrieval_samples`"
                                   % (lmdb_paths,))
            metadata = build_db_dicts(samples, stratified_db_creation, stratification_interval)
        self.dataset = dataset
        self.num_retrieval, self.topk = num_retrieval or 1, topk
        self.max_seq_len, self.motion_fps, self.motion_framechunksize = max_seq_len, motion_fps, motion_framechunksize
        self.latent_dim, self.text_latent_dim = latent_dim, text_latent_dim
        self.index = DiscourseIndex(metadata, device)
        # gesture_type / llm methods: need idx_2_gesture_labels in the metadata.  word_similarity(db_word, query_word):
        # None = the reference's effective behaviour, fuzz.partial_ratio / 100 (rag/utils.py:239-272; device form
        # rg_partial_ratio); a callable = the embedding-model similarity the reference's comments intend
        self.gesture_index = GestureTypeIndex(metadata, self.index) if "idx_2_gesture_labels" in metadata else None
        self.word_similarity = fuzzy_word_similarity if word_similarity is None else word_similarity
        # llm method: `llm_output(text) -> str` stands for get_llm_output (rag/llm_retrieval.py:69-96, the GPT call);
        # an LLMResponseCache (cached answers, BASELINE config 5) or any callable
        self.llm_output = llm_output.get if isinstance(llm_output, LLMResponseCache) else llm_output
        self.test_indexes, self.test_dbounds, self.test_qbounds = {}, {}, {}
        self.phase_ms = None  # dict: MotionDiffusion's phase profiler also collects the sub-phases here

    def retrieve(self, retr_method, text_features, discourse, prominence, speaker_id, idx=None, ready=None,
                 gesture_labels=None, text=None, text_times=None):
        """raggesture.py:313-477, eval branch: the reference reads its per-idx cache only under `self.training` (:335);
        in eval every call recomputes and overwrites test_indexes[idx] -- so do we (the dicts are written for parity
        of that side effect only, bounded to the last CACHE_CAP clips)."""
        if retr_method not in ("discourse", "gesture_type", "llm"):
            raise NotImplementedError("retrieval method %r (the reference's `prosody` raises too, raggesture.py:431)" % retr_method)
        if True:
            if ready is not None:
                si, db_b, qb = ready
            elif retr_method in ("gesture_type", "llm"):
                if self.gesture_index is None:
                    raise capi.RgError("%s retrieval needs idx_2_gesture_labels in the DB metadata (raggesture.py:262)" % retr_method)
                if retr_method == "gesture_type":
                    si, db_b, qb = gesture_type_retrieval(self.gesture_index, gesture_labels, speaker_id, text_features,
                                                          self.word_similarity)
                else:
                    if self.llm_output is None:
                        raise capi.RgError("llm retrieval needs `llm_output=` (an LLMResponseCache or a callable text -> answer)")
                    si, db_b, qb = llm_retrieval(self.gesture_index, text, text_times, speaker_id, prominence, text_features,
                                                 self.word_similarity, self.llm_output)
            else:
                si, db_b, qb = discourse_retrieval(self.index, discourse, prominence, speaker_id, text_features)
            self.test_indexes.setdefault(idx, {})[retr_method] = si
            self.test_dbounds.setdefault(idx, {})[retr_method] = db_b
            self.test_qbounds.setdefault(idx, {})[retr_method] = qb
            while len(self.test_indexes) > self.CACHE_CAP:
                old = next(iter(self.test_indexes))
                for d in (self.test_indexes, self.test_dbounds, self.test_qbounds):
                    d.pop(old, None)
        data = {q: [s for s in idxs if s != idx][:self.num_retrieval] for q, idxs in si.items()}
        return data, db_b, qb

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    @staticmethod
    def _clip_text(conditions, b):
        """The transcript the llm method prompts with.  The reference hands llm_retrieval conditions["text"]
        (raggesture.py:501), which MotionDiffusion fills with the frame-aligned word EMBEDDINGS
        (diffusion_architecture.py:190) -- `text.strip()` cannot run on those; the transcript string is what
        the same dict carries as "raw_text" (:191), so a non-string "text" falls back to it."""
        for key in ("text", "raw_text"):
            v = conditions.get(key)
            if v is not None and not torch.is_tensor(v) and len(v) > b and isinstance(v[b], str):
                return v[b]
        return None

    def _tick(self, name):
        """Sub-phase wall times (with device syncs) when a profiler dict is attached; else a no-op."""
        if self.phase_ms is None:
            return
        import time
        torch.cuda.synchronize()
        now = time.perf_counter()
        if name is not None:
            self.phase_ms[name] = self.phase_ms.get(name, 0.0) + (now - self._t_last) * 1e3
        self._t_last = now

    def forward(self, conditions, lengths, device, idx=None, retrieval_method="gesture_type", gesture_rep_encoder=None,
                noise=None, on_exemplars=None, search_stream=None, inputs_ready=None):
        """conditions: the model's kwargs dict (text_features, discourse, prominence, speaker_ids, ...)."""
        gre = gesture_rep_encoder
        dev = torch.device(device)
        B = len(conditions["text_features"])
        chunk, L = self.motion_framechunksize, self.max_seq_len // self.motion_framechunksize
        T, D = 4 * L + 3, self.latent_dim
        tick = self._tick
        tick(None)
        # The DB sweeps, their read-back and the host-side ranking walk only need the query annotations: on a
        # stream of their own (search_stream, waiting for `inputs_ready` only) they do not queue behind the
        # caller's pending work on the current stream (the batch VAE encode), and the walk overlaps with it.
        import contextlib
        ctx = torch.cuda.stream(search_stream) if search_stream is not None else contextlib.nullcontext()
        if search_stream is not None and inputs_ready is not None:
            search_stream.wait_event(inputs_ready)
        with ctx:
            plans, ex = [], []
            spks = [int(v) for v in conditions["speaker_ids"][:, 0].tolist()]
            # every clip's DB sweeps are launched before the first host read-back: one synchronisation per batch
            pending, queries = {}, []
            if retrieval_method == "discourse":
                for b in range(B):
                    qs = discourse_queries(conditions["discourse"][b], conditions["prominence"][b], spks[b])
                    pending[b] = (len(queries), len(qs))
                    queries += qs
            swept = self.index.collect(self.index.sweep_async(queries)) if queries else []
            # tie-break similarities of all clips: launched back to back, one read-back
            order_b = sorted(pending)
            begun = [discourse_retrieval_begin(self.index, conditions["discourse"][b], conditions["prominence"][b], spks[b],
                                               conditions["text_features"][b],
                                               swept[pending[b][0]:pending[b][0] + pending[b][1]]) for b in order_b]
            sims = fetch_sims(begun)
            ready = {b: discourse_retrieval_finish(self.index, p, sm) for b, p, sm in zip(order_b, begun, sims)}
            for b in range(B):
                spk = spks[b]
                ri, rb, qb = self.retrieve(retrieval_method, conditions["text_features"][b], conditions["discourse"][b],
                                           conditions["prominence"][b], spk, idx=idx[b] if idx is not None else None,
                                           ready=ready.get(b),
                                           gesture_labels=(conditions.get("gesture_labels") or [None] * B)[b],
                                           text=self._clip_text(conditions, b),
                                           text_times=(conditions.get("text_times") or [None] * B)[b])
                plan = place_exemplars(ri, rb, qb, retrieval_method, self.motion_fps, chunk, self.max_seq_len)
                plans.append((plan, rb, qb))
                for qp, name, placed in plan:
                    ex.append((b, qp, name, placed))
        tick("retrieval.search")
        self.last_exemplars = [(b, qp, name, placed is not None) for b, qp, name, placed in ex]   # visiting order (tests)
        # ---- fetch + VAE-encode every visited exemplar in one batch (noise in the reference's order)
        recs = [self.dataset[name] for _, _, name, _ in ex]
        lat = None
        if ex:
            E = len(ex)
            stack = lambda k: torch.stack([r[k] for r in recs]).to(dev).float().contiguous()
            if noise is not None and not getattr(noise, "order_free", False):
                # explicit noise: the reference draws 4 x [L,1,D] per exemplar, exemplar-major (rsample order)
                eps = [[noise.draw((L, 1, D)) for _ in range(4)] for _ in range(E)]
                eps_list = [torch.cat([e[p].to(dev) for e in eps], dim=0) for p in range(4)]
            else:
                draw = noise.draw if noise is not None else (lambda shape: torch.randn(*shape, device=dev))
                eps_list = [draw((E * L, 1, D)) for _ in range(4)]  # generator noise: order is immaterial
            fork = torch.cuda.Event()     # work started by on_exemplars orders itself after this point, not after the encode
            fork.record()
            # batch of Ep = E rounded up to a multiple of 4 (padding = copies of exemplar 0 with zero noise, dropped below):
            # the captured encode graphs exist per Ep, not per exemplar count
            Ep = -(-E // 4) * 4
            padr = lambda t: t if t.shape[0] == Ep else torch.cat([t, t[:1].expand(Ep - E, *t.shape[1:])], dim=0).contiguous()
            eps_list = [torch.cat([e, e.new_zeros((Ep - E) * L, 1, D)], dim=0) if Ep != E else e for e in eps_list]
            lat, _ = gre.encode(padr(stack("motion_upper")), padr(stack("motion_lower")), padr(stack("motion_face")),
                                padr(stack("motion_hands")), padr(stack("trans")), padr(stack("facial")), padr(stack("contact")),
                                padr(stack("motion_mask")), eps_list)
            lat = lat[:E]
        if on_exemplars is not None and ex:
            # the caller may start work that needs the exemplars' conditioning only (their K/V projections) on other
            # streams while this stream VAE-encodes their motion; the encode (one graph launch) is queued first so
            # that the device is not left waiting for the host to issue the projections' launches
            on_exemplars(ex, recs, fork)
        tick("retrieval.exemplar_encode")
        retr_se, query_se, retr_lats, names_out, type2words = ([{} for _ in range(B)] for _ in range(5))
        zero_motion = torch.zeros(B, T, D, device=dev)
        if self.dataset is None:
            raise capi.RgError("RetrievalDatabase.forward needs `dataset=` (the exemplars' motion / audio / text records); "
                               "metadata alone only supports retrieve()")
        tmpl = self.dataset[0]
        raw_motion = torch.zeros(B, self.max_seq_len, tmpl["motion"].shape[-1], device=dev)
        raw_trans = torch.zeros(B, self.max_seq_len, tmpl["trans"].shape[-1], device=dev)
        raw_facial = torch.zeros(B, self.max_seq_len, tmpl["facial"].shape[-1], device=dev)
        # all row copies of the exemplar canvases are gathered into index lists and issued as one gather/scatter
        # per tensor (a guided batch has ~50 exemplars x 7 small copies otherwise); later spans overwrite earlier
        # ones exactly as the reference's sequential slice assignments do (index_copy_ with unique keys)
        lat_dst, lat_src, frame_dst, frame_src = [], [], [], []
        lmask = gre.latent_mask(torch.stack([r["motion_mask"] for r in recs])) if ex else None
        for e, (b, qp, name, placed) in enumerate(ex):
            if placed is None:
                continue
            (r0, r1), (s0, s1) = placed
            rec, (plan, rb, qb) = recs[e], plans[b]
            retr_se[b][qp], query_se[b][qp] = (r0, r1), (s0, s1)
            retr_lats[b][qp] = dict(retr_motion_latent=lat[e:e + 1], retr_text=rec["word"].unsqueeze(0).to(dev),
                                    retr_audio=rec["audio"].unsqueeze(0).to(dev),
                                    retr_spkid=rec["speaker_id"].unsqueeze(0).to(dev),
                                    retr_motion_mask=lmask[e:e + 1])
            span = np.arange(s1 - s0)
            for part in range(4):
                o = part * (L + 1)
                lat_dst.append(b * T + o + s0 + span)
                lat_src.append(e * T + o + r0 + span)
            frames = np.arange((s1 - s0) * chunk)
            frame_dst.append(b * self.max_seq_len + s0 * chunk + frames)
            frame_src.append(e * self.max_seq_len + r0 * chunk + frames)
            q_word, q_type = qb[qp][0], qb[qp][1]
            r_word, r_type = rb[qp][name][0], rb[qp][name][1]
            type2words[b][qp] = (q_word, q_type, r_word, r_type)
            names_out[b][q_word] = name
        if lat_dst:
            def idx(dst, src):
                # a row written twice keeps its LAST source (sequential slice assignment); index_copy_ needs unique keys
                dst, src = np.concatenate(dst), np.concatenate(src)
                _, last = np.unique(dst[::-1], return_index=True)
                keep = len(dst) - 1 - last
                return to_device_async(dst[keep], dev), to_device_async(src[keep], dev)
            dst, src = idx(lat_dst, lat_src)
            zero_motion.view(B * T, D).index_copy_(0, dst, lat.reshape(E * T, D).index_select(0, src))
            dst, src = idx(frame_dst, frame_src)
            for canvas, key in ((raw_motion, "motion"), (raw_trans, "trans"), (raw_facial, "facial")):
                allrec = torch.stack([r[key] for r in recs]).to(dev).float()
                canvas.view(B * self.max_seq_len, -1).index_copy_(0, dst, allrec.view(E * self.max_seq_len, -1).index_select(0, src))
        src_mask = (zero_motion != 0).any(dim=-1).to(torch.int)
        raw_latent_mask = src_mask.clone()
        raw_motion_latents = zero_motion.clone()
        # (face and lower-body rows; as two slices: a LIST index is uploaded from pageable memory, and that copy returns
        # only after everything queued on this stream -- the exemplar encode -- has run)
        for r0, r1 in ((2 * L + 2, 3 * L + 2), (3 * L + 3, T)):
            src_mask[:, r0:r1] = 0
            raw_motion_latents[:, r0:r1, :] = 0
        tick("retrieval.assemble")
        return dict(re_text=None, re_motion=None, re_mask=src_mask,
                    raw_motion_latents=raw_motion_latents.view(B, self.num_retrieval, T, D),
                    raw_motion=raw_motion.view(B, self.num_retrieval, self.max_seq_len, -1),
                    raw_trans=raw_trans.view(B, self.num_retrieval, self.max_seq_len, -1),
                    raw_facial=raw_facial.view(B, self.num_retrieval, self.max_seq_len, -1),
                    raw_sample_names=names_out, raw_type2words=type2words, raw_latent_mask=raw_latent_mask,
                    retr_startends=retr_se, query_startends=query_se, retr_uncropped_latents=retr_lats)


# ---------------------------------------------------------------------------------------------- LLM guidance (host side)
_LLM_LABEL_RX = re.compile(r"[\"\']*([\w \-\']+\w)[\"\']*\,\s*[\"\']*(?P<gesttype>b*eat|m*etaphoric|iconic|deictic)", re.MULTILINE)


def parse_gesture_labels_from_llm_output(llm_output):
    """rag/llm_retrieval.py:131-165: the "(word, type)" pairs of an LLM answer -> [{"word", "name"}], beat labels and
    duplicates (pairs repeated in the model's explanation) dropped.  Same regular expression as the reference."""
    labels = []
    for m in _LLM_LABEL_RX.finditer(llm_output):
        t = m.group("gesttype")
        if "etaphoric" in t:
            name = "metaphoric"
        elif "eat" in t:
            name = "beat"
        elif "iconic" in t:
            name = "iconic"
        else:
            name = "deictic"
        labels.append({"word": m.group(1).strip(), "name": name})
    out = []
    for g in labels:
        if g["name"] != "beat" and g not in out:
            out.append(g)
    return out


class LLMResponseCache:
    """Response cache for the LLM call of the llm retrieval method (rag/llm_retrieval.py:72-96 calls the API for
    every window of every clip; BASELINE config 5 runs on cached calls).  `call(text) -> str` is supplied by the
    user (there is no network code here); answers are keyed by the exact prompt text and persisted as JSON."""

    def __init__(self, path=None, call=None):
        self.path, self.call, self.hits, self.misses = path, call, 0, 0
        self.data = {}
        if path is not None and os.path.exists(path):
            with open(path, "r", encoding="utf-8") as f:
                self.data = json.load(f)

    def get(self, text):
        if text in self.data:
            self.hits += 1
            return self.data[text]
        if self.call is None:
            raise KeyError("no cached LLM answer for this text and no `call` to produce one")
        self.misses += 1
        self.data[text] = self.call(text)
        if self.path is not None:
            with open(self.path, "w", encoding="utf-8") as f:
                json.dump(self.data, f, ensure_ascii=False, indent=0)
        return self.data[text]

    def labels(self, text):
        """parse(get(text)); the empty text short-circuits like llm_retrieval (:180-181)."""
        return [] if text.strip() == "" else parse_gesture_labels_from_llm_output(self.get(text))
