import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "move2hear-active-av-separation_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _carry_tuning_knobs_into_backward_passes():
    """Tests A/B the library's engines through thread-local tuning knobs (ops.debug_set); autograd runs backward passes on its own
    thread, so the autograd Functions are told to carry the forward thread's knobs along (m2h.functional.carry_tuning)."""
    from m2h import functional as MF
    MF.carry_tuning(True)
    yield
    MF.carry_tuning(False)
