import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "survey_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def frozen():
    """Vectors frozen from the oracle (tests/golden/make_oracle_vectors.py)."""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "oracle_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def codec():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    from mrnet_amd.wave import Codec
    return Codec(max_packets=1 << 21)
