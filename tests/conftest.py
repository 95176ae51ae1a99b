import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# the reference's own sample files, copied under tests/golden/ so that they travel to the GPU box (tests/golden/reference_samples/README.md)
REFERENCE_SAMPLES = os.path.join(ROOT, "tests", "golden", "reference_samples")


@pytest.fixture(scope="session")
def reference_samples():
    assert os.path.isdir(REFERENCE_SAMPLES), "tests/golden/reference_samples is part of the repository"
    return REFERENCE_SAMPLES
