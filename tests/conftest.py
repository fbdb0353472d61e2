import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible and -m gpu was not requested."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Achieved parity errors (tests.util.report) as the last lines of the run, one compact line per record, so that a
    `pytest -q` tail shows the numbers behind the green assertions (bounds are in the tests)."""
    try:
        from tests.util import BOUNDS, RECORDS
    except Exception:  # noqa: BLE001
        return
    tr = terminalreporter
    if BOUNDS:
        tr.write_sep("-", "BOUNDS (worst achieved value / asserted limit per check)")
        for tag, (v, lim) in sorted(BOUNDS.items()):
            tr.write_line(f"BOUND {tag}: {v:.2e} / {lim:.0e}")
    if not RECORDS:
        return
    tr.write_sep("-", "PARITY (achieved errors; stated tolerances: train_loss mean 1e-4, per sample 1e-3, x_hat 1e-2, gradients 1e-2)")

    def fmt(v):
        if isinstance(v, float):
            return f"{v:.2e}"
        if isinstance(v, list):
            return "[" + ",".join(fmt(x) for x in v[:4]) + (",..]" if len(v) > 4 else "]")
        if isinstance(v, dict):
            return "{" + ",".join(f"{k}={fmt(x)}" for k, x in list(v.items())[:4]) + "}"
        return str(v)

    for rec in RECORDS:
        body = " ".join(f"{k}={fmt(v)}" for k, v in rec.items() if k != "test")
        tr.write_line(f"PARITY {rec['test']}: {body}"[:400])
