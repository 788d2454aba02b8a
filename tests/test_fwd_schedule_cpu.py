"""CPU: the tile schedule of the persistent denoiser forward (rag-gesture_amd/fwd.py) is a valid dataflow
program: every tile waits for exactly the earlier tiles of its sequence, queues are topologically ordered,
and any number of in-order workers drains it without deadlock."""
import importlib

import numpy as np
import pytest

F = importlib.import_module("rag-gesture_amd.fwd")


def _tiles(sched):
    starts = sched[:F.N_SHARD + 1]
    t = sched[F.SCHED_HEADER:].reshape(-1, 4)
    return starts, t


@pytest.mark.parametrize("B,L", [(1, 2), (3, 8), (16, 8), (48, 8)])
def test_schedule_is_topological(B, L):
    starts, t = _tiles(F.build_schedule(B, L))
    assert starts[0] == 0 and starts[-1] == len(t) and np.all(np.diff(starts) >= 0)
    shard_of = {}
    for q in range(F.N_SHARD):
        seen = {}
        for typ_l, s, nt, target in t[starts[q]:starts[q + 1]]:
            assert shard_of.setdefault(int(s), q) == q, "a sequence lives in one queue"
            # completion count to wait for = tiles of this sequence in EARLIER stages (same-stage tiles are independent)
            typ, l = typ_l & 0xff, typ_l >> 8
            stage_key = (l if typ not in (F.EMBED, F.HEAD) else (-1 if typ == F.EMBED else L), typ)
            prev = seen.setdefault(int(s), [])
            assert target == sum(1 for k in prev if k != stage_key)
            assert all(k <= stage_key for k in prev), "stages of a sequence appear in chain order"
            prev.append(stage_key)
    for s in range(2 * B):
        n = int(np.sum(t[:, 1] == s))
        assert n == F.tiles_per_sequence(L, s < B)
    for b in range(B):   # the two branches of a clip share a queue (the CFG mix reads both)
        assert shard_of[b] == shard_of[B + b]
    # conditional-only stage
    q3 = t[(t[:, 0] & 0xff) == F.Q3_CA]
    assert np.all(q3[:, 1] < B) and len(q3) == B * L * 24


@pytest.mark.parametrize("workers", [1, 3, 32])
def test_in_order_workers_never_deadlock(workers):
    """Each worker holds one ticket and one prefetched ticket (as the kernel does) and runs them in order."""
    B, L = 5, 2
    starts, t = _tiles(F.build_schedule(B, L))
    heads = list(starts[:-1])
    done = np.zeros(2 * B, dtype=np.int64)
    rng = np.random.default_rng(0)

    def take(w):
        for k in range(F.N_SHARD):
            q = (w + k) % F.N_SHARD
            if heads[q] < starts[q + 1]:
                heads[q] += 1
                return heads[q] - 1
        return -1

    cur = [take(w) for w in range(workers)]
    nxt = [take(w) if cur[w] >= 0 else -1 for w in range(workers)]
    finished, idle_rounds = 0, 0
    while finished < len(t):
        progressed = False
        for w in rng.permutation(workers):
            if cur[w] < 0:
                continue
            _, s, _, target = t[cur[w]]
            if done[s] >= target:   # inputs exist: run the tile
                done[s] += 1
                finished += 1
                cur[w], nxt[w] = nxt[w], (take(w) if nxt[w] >= 0 else -1)
                progressed = True
        idle_rounds = 0 if progressed else idle_rounds + 1
        assert idle_rounds < 2, "deadlock"
    assert all(h == e for h, e in zip(heads, starts[1:]))


def test_ctypes_structs_mirror_the_header():
    """rg_fwd_layer / rg_fwd_args in include/rg_gesture.h and their ctypes mirrors list the same fields in the same
    order (every field is a pointer or an int: same layout)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = open(os.path.join(root, "include", "rg_gesture.h")).read()
    for cname, cls in (("rg_fwd_layer", F.FwdLayer), ("rg_fwd_args", F.FwdArgs)):
        i, j = h.index("typedef struct %s {" % cname), h.index("} %s;" % cname)
        body = re.sub(r"/\*.*?\*/", "", h[i:j], flags=re.S)
        names = re.findall(r"[\*\s](\w+)\s*[;,]", body.split("{", 1)[1])
        assert names == [f[0] for f in cls._fields_], cname
