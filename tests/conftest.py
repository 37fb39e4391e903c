import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def kats():
    return load_golden("reference_kats.json")


@pytest.fixture(scope="session")
def anchors():
    return load_golden("survey_anchors.json")


@pytest.fixture(autouse=True, scope="session")
def _forced_dispatch_options():
    """EEA_TEST_OPTIONS="<option>=<value>,..." (test plumbing, read HERE and not by the library): applied through
    eea_set_option before the first test, so that a whole test file can be re-run with one implementation pinned
    (tests/test_gpu_collision_parity.py::test_both_implementations_forced)."""
    spec = os.environ.get("EEA_TEST_OPTIONS", "")
    if spec:
        from ergodic_exploration_amd import capi
        for item in spec.split(","):
            opt, val = item.split("=")
            capi.set_option(int(opt), int(val))
    yield
