"""CPU: host-side helpers of the product that need no device (launch-order permutation, schedule tables, guidance
presets live in test_packing.py, retrieval host logic in test_retrieval.py)."""
import numpy as np


def test_xcd_affine_order_is_a_permutation_on_the_owning_xcd(rg):
    """denoiser.xcd_affine_order: every work item appears exactly once, idle slots are -1, and block b (-> XCD b % 8)
    processes a row group whose middle row lies in an M-tile of the same XCD (tile t -> XCD t % 8)."""
    order = rg.denoiser.xcd_affine_order
    for n_groups, ipg, T in ((32, 4, 43), (16, 1, 43), (96, 4, 43), (5, 3, 64), (1, 1, 33)):
        o = order(n_groups, ipg, T)
        items = o[o >= 0]
        assert sorted(items.tolist()) == list(range(n_groups * ipg))
        assert len(o) % 8 == 0
        for slot, it in enumerate(o.tolist()):
            if it < 0:
                continue
            g = it // ipg
            assert ((g * T + T // 2) // 64) % 8 == slot % 8
        # items of one group keep their order
        for g in range(n_groups):
            pos = [int(np.nonzero(o == g * ipg + i)[0][0]) for i in range(ipg)]
            assert pos == sorted(pos)


def test_schedule_tables_match_the_oracle(rg):
    """schedule.Schedule (what the kernels are given per respaced step) against the oracle's SpacedSchedule (pinned to
    the reference's tables in test_oracle_golden.py)."""
    from oracle import diffusion as odf
    s, o = rg.schedule.Schedule(), odf.SpacedSchedule()
    assert list(s.timestep_map) == list(o.timestep_map) and s.num_timesteps == 50
    # gaussian_diffusion.py:934-947 / 981-1040 (eta = 0): x_prev = sqrt(ab_prev) x0 + sqrt(1 - ab_prev) eps,
    # x_next = sqrt(ab_next) x0 + sqrt(1 - ab_next) eps; eps from sqrt_recip / sqrt_recipm1
    f32 = np.float32
    abp, abn = np.asarray(o.alphas_cumprod_prev).astype(f32), np.asarray(o.alphas_cumprod_next).astype(f32)
    assert np.array_equal(np.asarray(s.c_prev_a), np.sqrt(abp)) and np.array_equal(np.asarray(s.c_prev_b), np.sqrt(f32(1) - abp))
    assert np.array_equal(np.asarray(s.c_next_a), np.sqrt(abn)) and np.array_equal(np.asarray(s.c_next_b), np.sqrt(f32(1) - abn))
    assert np.array_equal(np.asarray(s.c_recip), np.asarray(o.sqrt_recip_alphas_cumprod).astype(f32))
    assert np.array_equal(np.asarray(s.c_recipm1), np.asarray(o.sqrt_recipm1_alphas_cumprod).astype(f32))


def test_lmdb_cache_reader_with_stub_modules(rg, tmp_path, monkeypatch):
    """read_lmdb_dicts against the reference's cache layout (raggesture.py:90-155, 219-224) through in-memory stand-ins
    for `lmdb` and the legacy `pyarrow.serialize / deserialize` (neither exists in this image: the reader then reports
    "not readable" and RetrievalDatabase falls back to dataset.retrieval_samples)."""
    import pickle
    import sys
    import types
    import numpy as np
    import torch
    assert rg.retrieval.read_lmdb_dicts(str(tmp_path)) is None      # real environment: nothing to read with

    stores = {}

    class _Txn:
        def __init__(self, d):
            self.d = d

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def cursor(self):
            return iter(sorted(self.d.items()))

    class _Env:
        def __init__(self, path):
            self.d = stores[path]

        def begin(self, write=False):
            return _Txn(self.d)

        def close(self):
            pass

    lmdb = types.ModuleType("lmdb")
    lmdb.open = lambda path, **kw: _Env(path)
    pa = types.ModuleType("pyarrow")
    pa.deserialize = pickle.loads
    monkeypatch.setitem(sys.modules, "lmdb", lmdb)
    monkeypatch.setitem(sys.modules, "pyarrow", pa)

    smp = rg.synth.synth_retrieval_samples(12, seed=4)
    want = rg.retrieval.build_db_dicts(smp)
    for name in rg.retrieval.LMDB_DICTS:
        path = tmp_path / name
        path.mkdir()
        d = {}
        for k, v in want[name].items():
            if name == "idx_2_text":     # LMDBDict stores tensors as numpy (torch_converter=True)
                v = [x.numpy() if torch.is_tensor(x) else x for x in v] if isinstance(v, (list, tuple)) else v.numpy()
            d[k.encode("ascii")] = pickle.dumps(v)
        stores[str(path)] = d
    got = rg.retrieval.read_lmdb_dicts(str(tmp_path))
    assert sorted(got) == sorted(rg.retrieval.LMDB_DICTS)
    for name in rg.retrieval.LMDB_DICTS:
        assert list(got[name].keys()) == sorted(want[name].keys())       # cursor order = sample-name order
        if name != "idx_2_text":
            assert got[name] == {k: want[name][k] for k in got[name]}
    k0 = next(iter(got["idx_2_text"]))
    a, b = got["idx_2_text"][k0], want["idx_2_text"][k0]
    a0, b0 = (a[0], b[0]) if isinstance(a, (list, tuple)) else (a, b)
    assert torch.is_tensor(a0) and np.array_equal(a0.numpy(), b0.numpy())
    # an empty cache (fresh checkout) is "not readable": the caller rebuilds from the dataset, like the reference
    stores[str(tmp_path / "idx_2_sense")] = {}
    assert rg.retrieval.read_lmdb_dicts(str(tmp_path)) is None


def test_op_recorder_zips_identical_jobs_into_grouped_launches(rg):
    """capi.OpRecorder (the four body-part VAEs as grouped launches): jobs with the same launch sequence go out position by
    position, uniform positions as ONE grouped call with per-job pointer arrays, anything else one by one in job order;
    jobs of different lengths are not zipped at all.  (Host logic only: a fake handle records what would be launched.)"""
    import ctypes
    import types

    class FakeLib:
        def __init__(self, log):
            self.log = log

        def rg_gemm(self, h, d, s):
            self.log.append(("gemm", d._obj.M))
            return 0

        def rg_gemm_grouped(self, h, descs, n, s):
            self.log.append(("gemm_grouped", n, [descs[i].M for i in range(n)]))
            return 0

        def rg_last_error(self, h):
            return b""

    class FakeHandle:
        def __init__(self):
            self.log, self._h, self.recorder = [], None, None
            self.lib = FakeLib(self.log)

        def call(self, name, *args, stream=None):
            if self.recorder is not None and stream is None:
                return self.recorder.add(("call", name, args))
            self.log.append((name,) + tuple(list(a) if isinstance(a, ctypes.Array) else a for a in args))

    torch_stream = types.SimpleNamespace(cuda_stream=0)
    import torch
    real = torch.cuda.current_stream
    torch.cuda.current_stream = lambda *a, **k: torch_stream
    try:
        h = FakeHandle()
        G = rg.gemm.GemmDesc
        def job(rec, j, extra=False):
            rec.begin_job()
            d = G(); d.M = 100 + j
            rec.add(("gemm", d, None))
            rec.add(("call", "layernorm", (1000 + j, 2000 + j, 3000 + j, 4000 + j, 64, 512, None)))
            rec.add(("call", "copy_cols" if j != 2 else "add_rows", (1, 2, 3, 4, 5)))          # position 2 is not uniform
            rec.add(("call", "add_rows", (10 + j, 20 + j, 30 + j, 4096, 4096 if not extra or j else 2048)))
        rec = rg.capi.OpRecorder()
        for j in range(4):
            job(rec, j)
        rec.issue(h)
        assert h.log[0] == ("gemm_grouped", 4, [100, 101, 102, 103])
        assert h.log[1] == ("layernorm_grouped", 4, [1000, 1001, 1002, 1003], [2000, 2001, 2002, 2003], [3000, 3001, 3002, 3003],
                            [4000, 4001, 4002, 4003], 64, 512, [None] * 4)
        assert [e[0] for e in h.log[2:6]] == ["copy_cols", "copy_cols", "add_rows", "copy_cols"]      # singles, in job order
        assert h.log[6] == ("add_rows_grouped", 4, [10, 11, 12, 13], [20, 21, 22, 23], [30, 31, 32, 33], 4096, 4096)
        assert len(h.log) == 7 and rec.jobs == []
        # a shared (non-pointer) argument that differs between the jobs: not grouped
        h.log.clear()
        for j in range(4):
            job(rec, j, extra=True)
        rec.issue(h)
        assert [e[0] for e in h.log[-4:]] == ["add_rows"] * 4
        # jobs of different lengths: job by job
        h.log.clear()
        job(rec, 0)
        job(rec, 1)
        rec.add(("call", "vae_reparam", (1, 2)))
        rec.issue(h)
        assert [e[0] for e in h.log] == ["gemm", "layernorm", "copy_cols", "add_rows", "gemm", "layernorm", "copy_cols", "add_rows", "vae_reparam"]
    finally:
        torch.cuda.current_stream = real


def test_async_results_wait_before_every_way_of_handing_tensors_out(rg, monkeypatch):
    """pipeline.AsyncResults promises code written against the reference's synchronous results (tools/visualize.py:201-260)
    finished tensors: the wait (`_ready`) must be issued by every idiom that copies entries out, including the ones CPython
    implements in C for dict subclasses (dict(out), {**out}, f(**out), other.update(out), out.copy(), copy.copy(out))."""
    import copy
    import pickle
    AR = rg.pipeline.AsyncResults
    calls = []
    monkeypatch.setattr(AR, "_ready", lambda self: calls.append(1))
    out = AR(pred_upper=1, pred_lower=2, done_event="ev", done_stream="st")
    idioms = {
        "getitem": lambda: out["pred_upper"], "get": lambda: out.get("pred_upper"), "values": lambda: list(out.values()),
        "items": lambda: list(out.items()), "iter": lambda: list(out), "dict()": lambda: dict(out), "{**}": lambda: {**out},
        "f(**)": lambda: (lambda **kw: kw)(**out), "update": lambda: {}.update(out), "copy": lambda: out.copy(),
        "copy.copy": lambda: copy.copy(out), "deepcopy": lambda: copy.deepcopy(out), "pickle": lambda: pickle.dumps(out),
        "setdefault": lambda: out.setdefault("pred_upper", 0),
    }
    for name, f in idioms.items():
        del calls[:]
        f()
        assert calls, "%s hands tensors out without waiting for the batch" % name
    # bookkeeping entries and membership tests need no wait
    del calls[:]
    assert out["done_event"] == "ev" and out.get("done_stream") == "st" and "pred_upper" in out and len(out) == 4
    assert not calls
    assert type(out.copy()) is dict and type(copy.copy(out)) is dict
    del calls[:]
    assert out.pop("pred_lower") == 2 and calls


def test_slot_bookkeeping_is_per_lane(rg):
    """Sessions and graphs are keyed by (lane, slot): the events a new batch waits for must be found under the same key
    whichever pipeline (split batches: lanes 0..n-1 under one pipeline id; whole batches per lane: one id per lane) left them."""
    MD = rg.pipeline.MotionDiffusion
    plan = [(0, "s0", 0, 2), (1, "s1", 2, 4)]
    assert MD._lanes_of(plan, plan) == [0, 1] and MD._lanes_of([(1, "s1", 0, 4)]) == [1]

    class _Main:
        def __init__(self):
            self.waited = []

        def wait_event(self, ev):
            self.waited.append(ev)

    m = MD.__new__(MD)
    m.slots, m._slots, m._slot_done, m._slot = 2, {}, {(0, 1): ["a"], (1, 1): ["b"], (0, 0): ["c"]}, 0
    main = _Main()
    assert m._take_slot(None, main, [0, 1]) == 1 and main.waited == ["a", "b"]
    main = _Main()
    assert m._take_slot(0, main, [0]) == 1 and main.waited == ["a"]      # pipeline 0 meets what pipeline None left on lane 0
    main = _Main()
    assert m._take_slot(0, main, [0]) == 0 and main.waited == ["c"]


def test_batch_lane_rules_results_in_submission_order_and_paired_launches(rg):
    """Two host-side rules of the four-lane pipeline.  (i) Wherever a pending batch has to finish on its own, everything
    submitted before it finishes first: a batch without inversion picks its lane among `base_lanes`, so the pending batch it meets
    there need not be the oldest one (found on the GPU as results out of order, NOTEBOOK 9.8).  (ii) The narrower launch
    forms are chosen exactly where the batch lanes' launches would not fit the compute units side by side."""
    import types
    MD = rg.pipeline.MotionDiffusion
    m = MD.__new__(MD)
    m._pend = {3: types.SimpleNamespace(seq=3), 0: types.SimpleNamespace(seq=4), 1: types.SimpleNamespace(seq=5),
               2: types.SimpleNamespace(seq=6)}
    assert m._pending_upto(5) == [3, 0, 1]           # a base batch on lane 1 meets batch 5: 3 and 4 go first
    assert m._pending_upto(3) == [3] and m._pending_upto(2) == [] and m._pending_upto(99) == [3, 0, 1, 2]
    m.async_results, m.batch_lanes, m._cob = True, 4, dict(lane=3)
    F = lambda B: m._seq_form_auto(B, cus=256)                 # -> (seq_pairs, seq_duo)
    assert F(64) == (False, True)       # co-batched chains of the default four lanes: 64 workgroups of two sequences of a kind (rg_seq2)
    assert F(48) == (False, True)       # 48-exemplar inversions of a filling pipeline: 48 workgroups
    assert F(16) == (False, False) and F(32) == (False, False)      # a draining batch's clips: 2 x 32 x 4 = 256 fits, one workgroup per sequence
    m.batch_lanes = 8
    assert F(64) == (True, True)        # eight lanes: 32 workgroups, the classifier-free pair behind the conditional one
    assert F(48) == (True, True) and F(32) == (False, True) and F(16) == (False, False)      # 2 x 16 x 8 = 256 fits
    m.batch_lanes = 2
    assert F(64) == (False, False)      # two lanes of 128 workgroups fit
    m.batch_lanes = 8
    m._cur_rot = 3                      # base batches rotate over base_lanes
    assert F(32) == (False, False)      # 2 x 32 x 3 = 192 fits
    m._cur_rot = 2                      # long-form window batches alternate between two lanes
    assert F(96) == (False, True)       # 96 exemplars: 96 workgroups of two sequences, not 48 of two clips
    m._cur_rot = None
    m.batch_lanes, m._cob = 4, None
    assert F(64) == (False, False)      # synchronous forwards: always one workgroup per sequence
    m._cob, m.async_results = dict(lane=None), True
    assert F(64) == (False, False)      # lanes split a batch (cobatch_lanes="split"): one workgroup per sequence

