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
