import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "path-tracer_amd", "py"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def golden_index():
    import json
    with open(os.path.join(GOLDEN, "index.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def scenes():
    import json
    with open(os.path.join(GOLDEN, "scenes.json")) as f:
        return json.load(f)
