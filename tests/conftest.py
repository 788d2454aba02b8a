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
