import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:          # tests/known_answers.py (closed forms shared by the CPU and GPU tests)
    sys.path.insert(1, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _tolerance_report(path):
    """KP_TOL_REPORT=<file>: every numpy assert_allclose of the run appends `test file:line  atol  rtol  measured max |error|  max |desired|`,
    so that tolerances can be set from what the kernels actually deliver (tools/micro/tolerance_report.py prints the slack)."""
    import traceback
    import numpy as np
    orig = np.testing.assert_allclose

    def recording(actual, desired, rtol=1e-7, atol=0, *a, **k):
        try:
            x, y = np.asarray(actual, np.float64), np.asarray(desired, np.float64)
            err = float(np.abs(x - y).max()) if x.size else 0.0
            mag = float(np.abs(y).max()) if y.size else 0.0
            fr = [f for f in traceback.extract_stack() if os.sep + "tests" + os.sep in f.filename and "conftest" not in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "?"
            with open(path, "a") as fh:
                fh.write(f"{where}\t{atol:g}\t{rtol:g}\t{err:.3e}\t{mag:.3e}\n")
        except Exception:
            pass
        return orig(actual, desired, rtol, atol, *a, **k)
    np.testing.assert_allclose = recording


if os.environ.get("KP_TOL_REPORT"):
    _tolerance_report(os.environ["KP_TOL_REPORT"])


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a GPU skips the gpu-marked tests instead of failing them on the missing device.
    `-m gpu` (what the driver runs on the MI355X) never skips: there a missing device or extension must fail loudly."""
    if "gpu" in (config.getoption("-m") or ""):
        return
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible (run with -m gpu on an MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load
