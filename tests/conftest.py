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
    """The CTC step remembers, per (device, stream, shape), whether the last lane-exponent step had to repair many
    utterances (and then runs the log-domain step for a while) -- in two pinned words the CALLER of the C ABI owns
    (engine.ctc_host_state / the C++ operator's pool; wfl_ctc_call in include/wfl.h).  Tests run unrelated data through
    the same shapes: every test starts from scratch, so that what it compares bit for bit ran on the path it means."""
    try:
        from gtn_applications_amd import engine as E

        E.ctc_reset_state()
    except Exception:  # (collection on a machine without the library: the tests themselves will say so)
        pass
    yield
