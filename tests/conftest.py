import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Leaf to composite: kernels against the oracle / the reference's golden vectors first, then the end-to-end goldens, the
# full-size BASELINE configs, the boundary, and the HIP-vs-HIP self-comparisons of the scheduling layers (async pipeline,
# co-batching, long-form batching) last.  Nothing is skipped or hidden -- a failure anywhere still fails the run -- but under
# `-x` a scheduling problem can no longer keep the oracle-parity evidence of every kernel from being collected.
ORDER = ("test_oracle_golden", "test_fuzzy", "test_host_logic_cpu", "test_seq_pack_cpu", "test_capi_cpu", "test_reference_config_cpu",
         "test_dist_cpu",
         "test_sampler_gpu", "test_rotation_gpu", "test_gemm_gpu", "test_denoiser_gpu", "test_vae_oracle_gpu", "test_retrieval",
         "test_packing", "test_features_gpu", "test_ln_guard_gpu",
         "test_pipeline_gpu", "test_edge_cases_gpu", "test_fullsize_gpu", "test_longform_llm_gpu", "test_boundary_gpu",
         "test_ebucket_gpu", "test_cobatch_gpu", "test_longform_batched_gpu", "test_async_gpu")


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(ORDER)}
    stem = lambda it: os.path.splitext(os.path.basename(str(it.fspath)))[0]
    items.sort(key=lambda it: rank.get(stem(it), len(ORDER) - 4))      # (stable: the order inside a file is kept; unknown
    #                                                                     files run before the scheduling-layer files)


@pytest.fixture(scope="session")
def rg():
    """The product package (directory name has a hyphen, so import it by name)."""
    return importlib.import_module("rag-gesture_amd")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ---------------------------------------------------------------------------------------------------------------------------
# Parity bookkeeping: every tolerance-based comparison of the GPU suite goes through `parity.check(name, measured, bound)`.
# The measured values are asserted, kept, printed as one table at the end of the run (so they survive `pytest -q`) and
# written to gpurun_out/parity_gpu.json when that directory is writable.
class _Parity:
    def __init__(self):
        self.rows = []

    def check(self, name, measured, bound):
        measured = float(measured)
        self.rows.append((name, measured, float(bound)))
        assert measured <= bound, "%s: measured %.3e > bound %.3e" % (name, measured, bound)
        return measured


_PARITY = _Parity()


@pytest.fixture(scope="session")
def parity():
    return _PARITY


def relerr(a, b):
    """Frobenius-norm ratio ||a - b|| / ||b||."""
    return ((a - b).norm() / b.norm()).item()


def rowerr(a, b, dim=-1):
    """Worst row: max over rows of ||a_row - b_row|| / ||b_row|| (rows along `dim`), rows of b below 1e-3 of the mean row
    norm are priced against that floor instead of their own norm."""
    nb = b.norm(dim=dim)
    floor = 1e-3 * nb.mean()
    return ((a - b).norm(dim=dim) / nb.clamp_min(floor)).max().item()


def pytest_terminal_summary(terminalreporter):
    if not _PARITY.rows:
        return
    tr = terminalreporter
    tr.write_line("")
    tr.write_line("PARITY (measured <= bound) -- %d checks" % len(_PARITY.rows))
    for name, m, b in _PARITY.rows:
        tr.write_line("PARITY %-78s %.3e <= %.1e  (margin x%.1f)" % (name[:78], m, b, b / m if m > 0 else float("inf")))
    try:
        import json
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_gpu.json"), "w") as f:
            json.dump([dict(name=n, measured=m, bound=b) for n, m, b in _PARITY.rows], f, indent=1)
    except OSError:
        pass
