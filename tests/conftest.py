import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


REFERENCE_SAMPLES = "/root/reference/samples"


@pytest.fixture(scope="session")
def reference_samples():
    if not os.path.isdir(REFERENCE_SAMPLES):
        pytest.skip("/root/reference not present (GPU box)")
    return REFERENCE_SAMPLES
