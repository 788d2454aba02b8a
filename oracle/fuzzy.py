"""TEST INFRASTRUCTURE (CPU oracle) -- the word similarity the reference EFFECTIVELY uses for the gesture_type / llm
retrieval methods.

reference: mogen/models/transformers/rag/utils.py:231-272 `get_word_similarity_score`.  `word2vec_model` /
`fasttext_model` are never defined (their loads are commented out, :7-8), so every call raises NameError inside the
`try` and lands in `except Exception: return fuzz.partial_ratio(word1, word2) / 100` (:269-270) -- on the WHOLE
strings, whatever branch (single / multi word) raised.

Third-party algorithm restated: fuzzywuzzy==0.18.0 (requirements.txt:14) `fuzz.partial_ratio`, pure-python flavour
(python-Levenshtein is not in requirements.txt, so fuzzywuzzy falls back to `difflib.SequenceMatcher`).  Published
algorithm (fuzzywuzzy/fuzz.py):
    decorators: either argument None -> 0; s1 == s2 -> 100; either string empty -> 0
    shorter, longer = (s1, s2) if len(s1) <= len(s2) else (s2, s1)
    blocks = SequenceMatcher(None, shorter, longer).get_matching_blocks()        # includes the (la, lb, 0) sentinel
    for (i, j, n) in blocks:
        start = max(j - i, 0); sub = longer[start:start + len(shorter)]
        r = SequenceMatcher(None, shorter, sub).ratio();  if r > .995: return 100
    return int(round(100 * max(r)))
PARITY UNPINNED AGAINST THE PACKAGE: fuzzywuzzy is not installed in the build container (no network), so this
restatement is pinned against the standard library's difflib (the very matcher the package calls) and against
hand-checkable pairs (tests/golden/fuzzy.json), not against fuzzywuzzy's own output.

Two forms:
  partial_ratio(s1, s2)          on difflib.SequenceMatcher (what the package runs)
  partial_ratio_restated(s1, s2) difflib's Ratcliff-Obershelp matcher written out (find_longest_match scan order and
                                 tie-breaks, LIFO block queue), the form the HIP kernel rg_partial_ratio implements
"""
from difflib import SequenceMatcher


def partial_ratio(s1, s2):
    if s1 is None or s2 is None:
        return 0
    if s1 == s2:
        return 100
    if len(s1) == 0 or len(s2) == 0:
        return 0
    shorter, longer = (s1, s2) if len(s1) <= len(s2) else (s2, s1)
    blocks = SequenceMatcher(None, shorter, longer).get_matching_blocks()
    scores = []
    for block in blocks:
        long_start = block[1] - block[0] if (block[1] - block[0]) > 0 else 0
        long_substr = longer[long_start:long_start + len(shorter)]
        r = SequenceMatcher(None, shorter, long_substr).ratio()
        if r > .995:
            return 100
        scores.append(r)
    return int(round(100 * max(scores)))


def get_word_similarity_score(word1, word2):
    """rag/utils.py:239-272 as it actually behaves (see the module docstring)."""
    return partial_ratio(word1, word2) / 100


# ------------------------------------------------------------------ the matcher written out (no junk: len < 200)
def _find_longest_match(a, b, alo, ahi, blo, bhi):
    """difflib.SequenceMatcher.find_longest_match for isjunk=None and sequences below the autojunk length: the
    longest common substring of a[alo:ahi] and b[blo:bhi]; among equals the one starting earliest in a, then earliest
    in b (rows i ascending, columns j ascending, strict improvement)."""
    besti, bestj, bestsize = alo, blo, 0
    prev = {}
    for i in range(alo, ahi):
        cur = {}
        for j in range(blo, bhi):
            if b[j] == a[i]:
                k = cur[j] = prev.get(j - 1, 0) + 1
                if k > bestsize:
                    besti, bestj, bestsize = i - k + 1, j - k + 1, k
        prev = cur
    return besti, bestj, bestsize


def _matching_blocks(a, b):
    """Raw blocks of get_matching_blocks (before sorting / merging of adjacent blocks, which changes neither the
    number of matched characters nor the set of diagonals j - i)."""
    out, queue = [], [(0, len(a), 0, len(b))]
    while queue:
        alo, ahi, blo, bhi = queue.pop()
        i, j, k = _find_longest_match(a, b, alo, ahi, blo, bhi)
        if k:
            out.append((i, j, k))
            if alo < i and blo < j:
                queue.append((alo, i, blo, j))
            if i + k < ahi and j + k < bhi:
                queue.append((i + k, ahi, j + k, bhi))
    return out


def _ratio(a, b):
    m = sum(k for _, _, k in _matching_blocks(a, b))
    return 2.0 * m / (len(a) + len(b)) if (len(a) + len(b)) else 1.0


def partial_ratio_restated(s1, s2):
    if s1 is None or s2 is None:
        return 0
    if s1 == s2:
        return 100
    if len(s1) == 0 or len(s2) == 0:
        return 0
    shorter, longer = (s1, s2) if len(s1) <= len(s2) else (s2, s1)
    la, lb = len(shorter), len(longer)
    diagonals = {max(j - i, 0) for i, j, _ in _matching_blocks(shorter, longer)} | {lb - la}
    best = 0.0
    for d in sorted(diagonals):
        r = _ratio(shorter, longer[d:d + la])
        if r > .995:
            return 100
        best = max(best, r)
    return int(round(100 * best))
