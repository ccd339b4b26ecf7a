import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _fresh_ctc_step_choice():
    """The CTC step remembers, per workspace ADDRESS, whether the last lane-exponent step had to repair many utterances
    (and then runs the log-domain step for a while).  Tests hand recycled addresses to unrelated data: every test starts
    from scratch, so that what it compares bit for bit ran on the path it means."""
    try:
        from gtn_applications_amd import _native as N

        N.lib.wfl_ctc_adaptive_reset()
    except Exception:  # (collection on a machine without the library: the tests themselves will say so)
        pass
    yield
