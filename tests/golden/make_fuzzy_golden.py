"""Generates tests/golden/fuzzy.json: pairs for the effective word similarity of the gesture_type / llm retrieval
(rag/utils.py:239-272 -> fuzz.partial_ratio / 100).  fuzzywuzzy is NOT installed in the build container, so the expected
values come from the standard library's difflib through oracle/fuzzy.py::partial_ratio (the matcher fuzzywuzzy 0.18
itself calls without python-Levenshtein) -- PARITY UNPINNED AGAINST THE PACKAGE.  The first block is hand-checkable:
each expectation is derived in the comment next to it and asserted here before the file is written."""
import json
import os
import random
import sys
sys.dont_write_bytecode = True   # /root/reference is read-only: no __pycache__ there

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import fuzzy  # noqa: E402

HAND = [
    ("abcd", "XXXbcdeEEE", 75),      # fuzzywuzzy's own docstring example: best window "Xbcd": 3 of 4 match -> 2*3/8
    ("hello", "hello world", 100),   # the shorter string is a substring
    ("same", "same", 100),           # equal strings
    ("", "word", 0), ("word", "", 0),
    ("ab", "ba", 67),                # block "a" = (0, 1, 1): window longer[1:3] = "a" -> 2*1/3; the sentinel's window "ba" -> 2*1/4
    ("abc", "xyz", 0),
    ("big", "huge", 33),             # "g" only: windows "hug" (1 match: 2/6 = .333) and "uge" -> 33
    ("pointing", "point", 100),
    ("round", "around", 100),
    ("this way", "that way", 75),    # equal length 8: t, h, " way" match -> 2*6/16
    ("aaaa", "aa", 100),
    ("abcde", "abXde", 80),          # 4 of 5 -> 2*4/10
]


def main():
    for a, b, want in HAND:
        assert fuzzy.partial_ratio(a, b) == want == fuzzy.partial_ratio_restated(a, b), (a, b, fuzzy.partial_ratio(a, b))
    rnd = random.Random(20260)
    words = ["big", "huge", "small", "round", "circle", "up", "down", "over there", "this one", "that", "me", "you",
             "everything", "nothing", "rectangular", "square", "going", "gone", "left side", "right", "naïve", "café",
             "número uno", "x", "xx", "abracadabra", "cadabra", "the the the", "a b c d", "mississippi", "miss"]
    pairs = [(a, b) for a in words for b in words if a != b][::7]
    alpha = "abcab "
    for _ in range(150):
        pairs.append(("".join(rnd.choice(alpha) for _ in range(rnd.randint(1, 12))),
                      "".join(rnd.choice(alpha) for _ in range(rnd.randint(1, 16)))))
    out = dict(hand=[[a, b, w] for a, b, w in HAND],
               pairs=[[a, b, fuzzy.partial_ratio(a, b)] for a, b in pairs])
    with open(os.path.join(HERE, "fuzzy.json"), "w") as f:
        json.dump(out, f, ensure_ascii=True, indent=0)
    print("fuzzy.json:", len(out["hand"]), "hand-checked,", len(out["pairs"]), "difflib-derived pairs")


if __name__ == "__main__":
    main()
