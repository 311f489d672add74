"""pytest config: registers the `gpu` marker and puts the repo root / oracle on sys.path.
Tests marked gpu need a real MI355X (driver: `pytest -m gpu`); everything else runs on CPU."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (HIP device)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
